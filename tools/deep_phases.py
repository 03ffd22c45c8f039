# Where a deep locus' tile-kernel time goes (per-phase clocks of k_fused_tiles, diag build):
#   python tools/deep_phases.py [depth] [max_depth]          one 400-kb contig with loci at `depth` (default 20,000x, cap 8000)
#   python tools/deep_phases.py stress [abl bits]            bench.py's stress_500x contig (16 Mb, loci at ~500x)
#   the per-phase clocks need a diag build:  bash tools/build_variant.sh diag -DC3R_SCAN_DIAG=1, then C3R_LIB=gpurun_variants/libc3r_diag.so python tools/…
#   (the library is only ever taken from C3R_LIB: an earlier version picked a diag build up by itself, and a stale one — built before the barrier fix of
#   8c264b3 — made this tool, and only this tool, fail 7 times in 330 for two hours)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
stress = len(sys.argv) > 1 and sys.argv[1] == "stress"
if stress and len(sys.argv) > 2:
    os.environ["C3R_SCAN_ABL"] = sys.argv[2]
else:
    os.environ["C3R_SCAN_DBG"] = "1"
from clair3_rna_amd import capi, synth
import bench
if stress:
    depth, cap, L = 500.0, 8000, 16000000
    ref, rs, info = synth.generate_contig(contig_len=L, seed=synth.SEED + 4, depth=depth)
else:
    depth = float(sys.argv[1]) if len(sys.argv) > 1 else 20000.0
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
    L = 400000
    ref, rs, info = synth.generate_contig(contig_len=L, seed=synth.SEED + 5, depth=depth, expressed_frac=0.01, intron_lo=100.0, intron_hi=800.0)
chunks = bench.chunk_list(L)
eng = capi.Engine(0); eng.set_params(max_depth=cap); eng.load_reads(rs); eng.set_reference(1, ref)
for _ in range(2):
    eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
eng.set_profiling(True); eng.reset_kernel_stats()
eng.load_reads(rs)
eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
ks = eng.kernel_stats()
print("depth %g cap %d abl %s reads %d n=%d  " % (depth, cap, os.environ.get("C3R_SCAN_ABL", "0"), info["n_reads"], n) +
      "  ".join("%s %.3f" % (k.replace("k_", ""), v["total_ms"]) for k, v in sorted(ks.items())))
