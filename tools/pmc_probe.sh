#!/bin/bash
# clock / matrix-pipe counters of a probe binary:  bash tools/pmc_probe.sh <tag> <binary> [args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_probe/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT -- $R/"$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections,re
per=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(dict)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        m=re.search(r"k_lstm\w*<[^>]*>", k) or re.search(r"k_\w+", k)
        k=m.group(0) if m else k[:40]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for k in per:
    c={n:sum(v)/len(v) for n,v in per[k].items()}
    d=sum(dur[k].values())/len(dur[k])
    clk=c["GRBM_GUI_ACTIVE"]/8/(d*1e-3)/1e9
    simd=c["GRBM_GUI_ACTIVE"]/8*256*4
    wc=c["SQ_WAVE_CYCLES"]
    print("%-40s %7.2f ms clk %.2f GHz mfma_busy %.3f wait_inst %.3f wait_any %.3f active %.3f" % (k,d,clk,c["SQ_VALU_MFMA_BUSY_CYCLES"]/simd,c["SQ_WAIT_INST_ANY"]/wc,c["SQ_WAIT_ANY"]/wc,c["SQ_ACTIVE_INST_ANY"]/wc))
PY
