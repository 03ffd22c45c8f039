#!/bin/bash
# A/B of the y1 traffic probe (tools: bash tools/build_variant.sh y1probe -DC3R_PROBE_Y1=1):  gpurun -- bash tools/y1_probe.sh
for i in 1 2; do
  for v in "" gpurun_variants/libc3r_y1probe.so; do
    C3R_NO_F16_GUARD=1 C3R_LIB=$v python bench.py --steps 10 --warmup 3 --no_cpu_baseline --no_fast --no_resident --no_strong --no_extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']
print('%-12s %.0f sites/s  %.3f ms/step | k_lstm1 %.3f  k_lstm2 %.3f' % ('${v:+probe}' or 'product', d['value'], d['ms_per_step'], k['k_lstm1'], k['k_lstm2']))"
  done
done
