for a in 0 1 2 3 4 7; do
  C3R_SCAN_ABL=$a python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_fast --no_resident --no_overlap 2>/dev/null > gpurun_out/abl.json
  python -c "
import json; d=json.load(open('gpurun_out/abl.json')); k=d['kernels_ms_per_step']; print('abl $a fused', k['k_fused_tiles'], 'tokens', k['k_tokens'], 'sites', d['config']['sites_per_step_per_rank'])"
done
