for v in "" occ4 occ6 occ7 unr2; do
  if [ -z "$v" ]; then python tools/tb_kernels.py 5 2>&1 | tail -1; else C3R_LIB=gpurun_variants/libc3r_$v.so python tools/tb_kernels.py 5 2>&1 | tail -1; fi
done
