"""Per-contig view of a C3R_TIMING=1 log of tools/e2e_full.py / sample_e2e.py:  python tools/timeline_view.py LOG [run index]"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
runs, cur = [], None
for l in txt:
    if l.startswith("fetch_threads") or l.startswith("run "):
        cur = [l]; runs.append(cur)
    elif cur is not None and "[timeline]" in l:
        cur.append(l)
r = runs[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
print(r[0])
ev = {}
for l in r[1:]:
    m = re.match(r"\[timeline\] (\S+)\s+(\S+)\s+([\d.]+) ->\s+([\d.]+)", l)
    ev.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(3)), float(m.group(4)))
order = sorted(ev, key=lambda c: (0, int(c[3:])) if c.startswith("chr") and c[3:].isdigit() else (1, 0))
for c in order:
    print("%-6s" % c, " ".join("%s[%.2f-%.2f]" % (k, a, b) for k, (a, b) in sorted(ev[c].items(), key=lambda kv: kv[1][0])))
