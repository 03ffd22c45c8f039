#!/bin/bash
# Mid-round look at one build on the GPU box: kernel-trace stats of a one-context bench run + the default bench line.
#   gpurun -- bash tools/quick_stats.sh <tag>      -> gpurun_out/quick/<tag>/{stats.csv, bench.json}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/quick/$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="$R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile --no_fast --no_resident --no_overlap --no_strong --no_extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/stats.csv 2>/dev/null
python3 $R/bench.py --no_cpu_baseline --no_fast > $OUT/bench.json 2> $OUT/bench.err
python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("$OUT/stats.csv")))
net = 0.0; tb = 0.0
for r in rows:
    n = r["Name"]; calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e6
    short = re.sub(r"\(.*", "", n).replace("void ", "").replace("c3r::", "")[:60]
    per_pass = float(r["TotalDurationNs"]) / 1e6 / 4           # 1 priming + 1 warm-up + 2 timed passes
    if "lstm" in n or "k_heads" in n or "k_fc4" in n: net += per_pass
    elif "rocclr" not in n: tb += per_pass
    print("%-60s calls %4d avg %9.4f ms  per pass %8.4f ms" % (short, calls, avg, per_pass))
print("network per pass %.3f ms, tensor build (all non-network kernels) per pass %.4f ms" % (net, tb))
PY
tail -c 2500 $OUT/bench.json
