#!/bin/bash
# HBM counter bytes (FETCH_SIZE, WRITE_SIZE: separate --pmc passes with --kernel-trace only) of the tensor-build kernels of a chr20 pass,
# optionally under a timing ablation of the tile kernel (C3R_SCAN_ABL; needs gpurun_variants/libc3r_diag.so = bash tools/build_variant.sh diag
# -DC3R_SCAN_DIAG=1, which tools/tb_kernels.py then loads):   gpurun -- bash tools/pmc_bytes.sh <tag> [abl]
TAG=${1:-x}; ABL=${2:-0}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_bytes/$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -- python3 $R/tools/tb_kernels.py 2 $ABL > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -- python3 $R/tools/tb_kernels.py 2 $ABL > $OUT/w.log 2>&1
[ -x $R/tools/hbm_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/hbm_calib.hip -o $R/tools/hbm_calib          # (the binary is not kept in the tree)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/cf -- $R/tools/hbm_calib > $OUT/cf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/cw -- $R/tools/hbm_calib > $OUT/cw.log 2>&1
python3 - <<PY
import csv, glob, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("c3r::", "")
        per[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0
for k in sorted(per):
    if k.startswith("calib_"):                  # the calibration launches (tools/hbm_calib.hip)
        c = {n: sum(v[-2:]) / len(v[-2:]) for n, v in per[k].items()}
        print("calibration %-60s fetch %8.1f MB  write %8.1f MB" % (k[:60], c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6))
    if not k.startswith("k_"): continue
    c = {n: sum(v[-2:]) / len(v[-2:]) for n, v in per[k].items()}          # (the last two launches: steady state)
    fb, wb = c.get("FETCH_SIZE", 0) * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6
    tot += fb + wb
    print("%-22s fetch %8.1f MB  write %8.1f MB" % (k, fb, wb))
print("abl $ABL: all tensor-build kernels %.1f MB per pass" % tot)
PY
