#!/bin/bash
# The whole-sample driver on 22 full-length GRCh38 contigs: wall time at several fetch-thread counts, then one kernel-traced repeat at
# 8 fetch threads for the GPU-busy fraction and the launch count per contig.   gpurun -- bash tools/e2e_trace.sh [e2e_full.py arguments, e.g. --ref_bias 4] > profiles/rN/sample_full_scale.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/e2e
rm -rf $OUT; mkdir -p $OUT
python3 $R/tools/e2e_full.py --ft 6,8,12 --reps 3 "$@" 2>&1 | grep -v "^\[" 
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/e2e_full.py --ft 8 --reps 2 "$@" > $OUT/trace.log 2>&1
grep "fetch_threads" $OUT/trace.log
python3 - <<PY
import csv, glob, re, collections
rows = []
for f in glob.glob("$OUT/trace/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# split into runs at gaps > 0.25 s
runs, cur, ce = [], [], None
for s, e, n in rows:
    if ce is not None and s - ce > 0.25e9:
        runs.append(cur); cur = []
    cur.append((s, e, n)); ce = max(ce or 0, e)
runs.append(cur)
for run in runs:
    if len(run) < 200: continue
    a, b = run[0][0], max(r[1] for r in run)
    busy, ce = 0, a
    for s, e, n in run:
        if e > ce: busy += e - max(s, ce); ce = e
    cnt = collections.Counter(re.sub(r"\(.*", "", n).replace("void ", "").replace("c3r::", "")[:28] for _s, _e, n in run)
    print("traced run: %d kernels (%.1f per contig), window %.3f s, GPU busy %.3f s = %.0f %%" % (len(run), len(run) / 22.0, (b - a) / 1e9, busy / 1e9, 100.0 * busy / (b - a)))
    print("   launches by kernel:", ", ".join("%s %d" % kv for kv in cnt.most_common(14)))
PY
