"""GPU busy time of each run inside a rocprofv3 --kernel-trace database:  python tools/gpu_busy.py <results.db> [gap seconds that separates runs]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
split = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = db.execute("select name, start, end from kernels order by start").fetchall()
runs, cur = [], [rows[0]]
for r in rows[1:]:
    if r[1] - max(x[2] for x in cur[-50:]) > split * 1e9:
        runs.append(cur); cur = [r]
    else:
        cur.append(r)
runs.append(cur)
for run in runs:
    if len(run) < 100:
        continue
    a, b = run[0][1], max(r[2] for r in run)
    busy, ce, gaps = 0, a, []
    for n, s, e in run:
        if s > ce:
            gaps.append((s - ce) / 1e6)
        if e > ce:
            busy += e - max(s, ce); ce = e
    gaps.sort(reverse=True)
    print("run: %5d kernels, window %.3f s, GPU busy %.3f s (%.0f %%), idle in gaps >= 10 ms: %.3f s; largest gaps (ms): %s"
          % (len(run), (b - a) / 1e9, busy / 1e9, 100.0 * busy / (b - a), sum(g for g in gaps if g >= 10) / 1e3, " ".join("%.0f" % g for g in gaps[:8])))
