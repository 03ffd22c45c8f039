#!/bin/bash
# A/B of several builds of libc3r.so on one GPU box:  bash tools/ab_bench.sh "<alt1.so> <alt2.so> ..." [bench args]
# Runs bench.py (no CPU baseline) alternately with the in-tree library and with C3R_LIB=<alt.so>, ROUNDS (default 2) times each.
ALTS=$1; shift
R=${GRAFT_REPO_ROOT:-.}
rm -rf $R/gpurun_out/ab; mkdir -p $R/gpurun_out/ab
for i in $(seq 1 ${ROUNDS:-2}); do
  python3 $R/bench.py --no_cpu_baseline "$@" > $R/gpurun_out/ab/main_$i.json 2> $R/gpurun_out/ab/main_$i.err
  for a in $ALTS; do
    t=$(basename $a .so)
    C3R_LIB=$R/$a python3 $R/bench.py --no_cpu_baseline "$@" > $R/gpurun_out/ab/${t}_$i.json 2> $R/gpurun_out/ab/${t}_$i.err
  done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$R/gpurun_out/ab/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        k=d.get("kernels_ms_per_step",{})
        print("%-22s %10.0f sites/s %7.3f ms/pass  lstm1 %s lstm2 %s" % (f.split('/')[-1], d["value"], d["ms_per_step"], k.get("k_lstm1"), k.get("k_lstm2")))
    except Exception as e:
        print(f, "ERR", e, open(f.replace('.json','.err')).read()[-500:])
PY
