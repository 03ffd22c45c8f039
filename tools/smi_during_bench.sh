#!/bin/bash
# socket power / clocks sampled while bench.py runs:  bash tools/smi_during_bench.sh
R=${GRAFT_REPO_ROOT:-.}
python3 $R/bench.py --no_cpu_baseline --no_profile --no_fast --no_strong --no_extra --steps 400 --warmup 3 > /tmp/bench_long.json 2>/dev/null &
BP=$!
for i in $(seq 1 300); do
  p=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
  if [ -n "$p" ] && [ "${p%.*}" -gt 500 ]; then break; fi
  sleep 0.5
done
sleep 2
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/=*//g' | tr -s ' \t' ' ' | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
cut -c1-200 /tmp/bench_long.json
