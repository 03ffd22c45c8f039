# Where a deep locus' tile-kernel time goes (per-phase clocks of k_fused_tiles, diag build):  python tools/deep_phases.py [depth] [max_depth]
#   bash tools/build_variant.sh diag -DC3R_SCAN_DIAG=1 first; the script picks gpurun_variants/libc3r_diag.so up by itself
import os, sys
if "C3R_LIB" not in os.environ:
    _d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_variants", "libc3r_diag.so")
    if os.path.exists(_d):
        os.environ["C3R_LIB"] = _d
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["C3R_SCAN_DBG"] = "1"
from clair3_rna_amd import capi, synth
import bench
depth = float(sys.argv[1]) if len(sys.argv) > 1 else 20000.0
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
L = 400000
ref, rs, info = synth.generate_contig(contig_len=L, seed=synth.SEED + 5, depth=depth, expressed_frac=0.01, intron_lo=100.0, intron_hi=800.0)
chunks = bench.chunk_list(L)
eng = capi.Engine(0); eng.set_params(max_depth=cap); eng.load_reads(rs); eng.set_reference(1, ref)
for _ in range(2):
    eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
eng.set_profiling(True); eng.reset_kernel_stats()
eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
ks = eng.kernel_stats()
print("depth %g cap %d reads %d n=%d  " % (depth, cap, info["n_reads"], n) + "  ".join("%s %.3f" % (k.replace("k_", ""), v["total_ms"]) for k, v in sorted(ks.items())))
