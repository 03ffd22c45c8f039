#!/usr/bin/env python3
"""Registers / LDS / scratch / occupancy of every kernel of libc3r.so as the compiler reports them (no GPU needed):
    python tools/kernel_resources.py [filter-regex] > profiles/rN/kernel_resources.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Wno-unused-function",
                      "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "clair3_rna_amd", "csrc", "c3r_lib.hip"), "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*(Function Name|Name): (\S+)", line)
    if m:
        cur = {"name": m.group(2)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
print("%-44s %5s %5s %5s %8s %10s %9s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "waves/SIMD", "LDS bytes"))
for r, d in zip(rows, dem):
    d = re.sub(r"^void ", "", d)
    d = re.sub(r"\(.*$", "", d).replace("c3r::", "")
    if flt and not flt.search(d):
        continue
    print("%-44s %5s %5s %5s %8s %10s %9s" % (d[:44], r.get("TotalSGPRs", "?"), r.get("VGPRs", "?"), r.get("AGPRs", "?"),
                                              r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
