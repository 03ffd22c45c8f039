#!/bin/bash
# Instruction counts of k_fused_tiles under the timing ablations of the diag build (C3R_SCAN_ABL), one --pmc pass per ablation:
#   gpurun -- bash tools/pmc_abl.sh <tag> "0 199 198 194 192 128 64"   -> gpurun_out/pmc_abl/<tag>.txt
# (bash tools/build_variant.sh diag -DC3R_SCAN_DIAG=1 first)
TAG=${1:-x}; ABLS=${2:-"0 199 198 194 192 128 64"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_abl
mkdir -p $OUT; : > $OUT/$TAG.txt
export C3R_LIB=$R/gpurun_variants/libc3r_diag.so
for A in $ABLS; do
  rm -rf $OUT/tmp_$A
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/tmp_$A -- python3 $R/tools/tb_kernels.py 2 $A > $OUT/tmp_$A.log 2>&1
  python3 - $OUT/tmp_$A $A >> $OUT/$TAG.txt <<PY
import csv, glob, collections, re, sys
per = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fused_tiles" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
c = {k: sorted(v)[len(v) // 2] for k, v in per.items()}
n = 23098 * 4.0
print("abl %4s  ms %.3f | per wave-span: VALU %6.0f SALU %6.0f LDS %5.0f VMEM rd %5.1f wr %5.1f | wait_any %.2f active %.2f" % (sys.argv[2], sorted(dur)[len(dur)//2] if dur else 0,
      c.get("SQ_INSTS_VALU", 0) / n, c.get("SQ_INSTS_SALU", 0) / n, c.get("SQ_INSTS_LDS", 0) / n, c.get("SQ_INSTS_VMEM_RD", 0) / n, c.get("SQ_INSTS_VMEM_WR", 0) / n,
      c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
  rm -rf $OUT/tmp_$A
done
cat $OUT/$TAG.txt
