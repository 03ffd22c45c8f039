#!/bin/bash
# SQ instruction-mix counters of the tensor-build kernels (one --pmc pass with --kernel-trace only: the combination gpurun allows):
#   gpurun -- bash tools/pmc_tb.sh <tag> [lib.so]      -> gpurun_out/pmc_tb/<tag>/summary.txt
TAG=${1:-x}; LIB=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_tb/$TAG
rm -rf $OUT; mkdir -p $OUT
[ -n "$LIB" ] && export C3R_LIB=$R/$LIB
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/tb_kernels.py 2 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/b -- python3 $R/tools/tb_kernels.py 2 > $OUT/b.log 2>&1
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("c3r::", "")
        per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(per):
    if not k.startswith("k_"): continue
    c = {n: sum(v) / len(v) for n, v in per[k].items()}
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-22s launches %3d | wave-cycles %.3g: wait_any %.2f wait_inst %.2f active %.2f | insts VALU %.3g SALU %.3g LDS %.3g VMEM rd %.3g wr %.3g | active_valu %.3g active_lds %.3g sca %.3g | lds bank conflict %.3g | busy %.3g gui %.3g" % (
        k, len(per[k].get("SQ_WAVE_CYCLES", [])), wc, c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_VMEM_WR", 0),
        c.get("SQ_ACTIVE_INST_VALU", 0), c.get("SQ_ACTIVE_INST_LDS", 0), c.get("SQ_ACTIVE_INST_SCA", 0), c.get("SQ_LDS_BANK_CONFLICT", 0), c.get("SQ_BUSY_CYCLES", 0), c.get("GRBM_GUI_ACTIVE", 0)))
PY
cat $OUT/summary.txt; tail -n 3 $OUT/a.log; tail -n 3 $OUT/b.log
