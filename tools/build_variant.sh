#!/bin/bash
# A/B builds of libc3r.so with extra compiler flags:  bash tools/build_variant.sh <name> [-DFOO=1 ...]  ->  gpurun_variants/libc3r_<name>.so
# (run on the GPU box with C3R_LIB=gpurun_variants/libc3r_<name>.so; the directory travels with the snapshot and is git-ignored)
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-function "$@" $R/clair3_rna_amd/csrc/c3r_lib.hip -o $R/gpurun_variants/libc3r_$N.so
