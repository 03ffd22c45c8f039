for g in 2560 8192 20480 1000000; do
  for r in 1 2; do
  C3R_LIST_GRID=$g python bench.py --no_cpu_baseline --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print('$g', d['value'], d['ms_per_step'], 'scan', k['k_scan_tiles'], 'tok', k['k_tokens'], 'sel', k['k_select'], d['stage_rates']['tensor_build_ms'])"
  done
done
