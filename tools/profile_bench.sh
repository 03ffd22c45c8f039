#!/bin/bash
# Round-end profiles of bench.py on the GPU box (one MI355X):  bash tools/profile_bench.sh [f16x3|f32]
# 1. rocprofv3 --kernel-trace --stats   -> gpurun_out/prof/<prec>/stats
# 2. separate --pmc passes (each with --kernel-trace only, the one combination gpurun allows): FETCH_SIZE | WRITE_SIZE | SQ matrix-pipe counters
# Summaries are written by tools/summarize_profiles.py into profiles/.
PREC=${1:-f16x3}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof/$PREC
rm -rf $OUT; mkdir -p $OUT
# --no_overlap: one context, so that a kernel's traced duration is its own run time (with two pipelined contexts a launch also
# waits for CUs the other context's persistent workgroups hold) and agrees with the avg_launch_ms bench.py measures
# k_fused_deep normally runs BESIDE k_fused_tiles on a stream of its own: a trace then shows it with the duration of the kernel it waits beside, and
# the sum of the traced durations counts that stretch twice.  The summaries are of the SERIAL order (C3R_DEEP_SERIAL=1, what bench.py's HIP-event
# profiler measures too): every kernel's duration is its own.
export C3R_DEEP_SERIAL=1
ARGS="$R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile --no_fast --no_f32 --no_resident --no_overlap --no_strong --no_extra --precision $PREC"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
# the additional configurations of the default line (configs[3] 30 channels, configs[4] 500x, the 20,000x locus): their kernels by full
# template name (k_fused_tiles<30>, k_lstm1_rs<30, ...>) next to the 18-channel ones
if [ "$PREC" = f16x3 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_extra -- python3 $R/bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_profile --no_fast --no_f32 --no_resident --no_overlap --no_strong --precision $PREC > $OUT/stats_extra.log 2>&1
fi
unset C3R_DEEP_SERIAL
python3 $R/bench.py --precision $PREC > $OUT/bench.json 2> $OUT/bench.err
ls -R $OUT | head -40
