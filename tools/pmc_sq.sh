#!/bin/bash
# SQ / clock counters of the network kernels for one build:  bash tools/pmc_sq.sh <tag> [lib.so]
# (a --pmc pass with --kernel-trace only, the one combination gpurun allows: never with sys / runtime / hip / hsa / memory-copy / marker traces)
TAG=$1; LIB=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sq/$TAG
mkdir -p $OUT
[ -n "$LIB" ] && export C3R_LIB=$R/$LIB
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile --no_overlap --no_fast --no_strong --no_extra > $OUT/log.txt 2>&1
python3 $R/tools/pmc_sq_summary.py $OUT
