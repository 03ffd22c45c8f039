# Tensor-build kernels of the chr20 pass (c3r_load_reads + scan of the 13 chunks), HIP-event times per kernel, for the in-tree library
# or C3R_LIB=<variant>; C3R_SCAN_DBG=1 adds the per-phase clocks of the tile kernel:   python tools/tb_kernels.py [repeats]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[2] != "0":
    os.environ["C3R_SCAN_ABL"] = sys.argv[2]          # timing / traffic ablations of the tile kernel (results are wrong then)
    if "C3R_LIB" not in os.environ:                   # (compiled in only with -DC3R_SCAN_DIAG=1: bash tools/build_variant.sh diag -DC3R_SCAN_DIAG=1)
        os.environ["C3R_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_variants", "libc3r_diag.so")
from clair3_rna_amd import capi, synth
import bench
ref, rs, info = synth.generate_contig(contig_len=synth.CHR20_LEN, seed=synth.SEED, depth=20.0)
chunks = bench.chunk_list(synth.CHR20_LEN)
rsh = capi.pinned_readset(rs)
eng = capi.Engine(0); eng.set_params(); eng.load_reads(rsh); eng.set_reference(1, ref)
for _ in range(2):
    eng.load_reads(rsh); eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng.set_profiling(True); eng.reset_kernel_stats()
for _ in range(R):
    eng.load_reads(rsh); eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
eng.set_profiling(False)
ks = eng.kernel_stats()
tot = sum(v["total_ms"] for k, v in ks.items() if k.startswith("k_")) / R          # (h2d_reads: the upload, beside the kernels)
print("%s n=%d  total %.3f ms | " % (os.environ.get("C3R_LIB", "in-tree").split("/")[-1], n, tot) + "  ".join("%s %.3f" % (k.replace("k_", ""), v["total_ms"] / R) for k, v in sorted(ks.items())))
