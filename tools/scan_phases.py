# Per-phase wall-clock of k_scan_tiles' heavy tiles on the chr20 pass (env C3R_SCAN_DBG; the library prints the averages):
#   python tools/scan_phases.py
import os, sys
sys.path.insert(0, '.')
os.environ["C3R_SCAN_DBG"] = "1"
from clair3_rna_amd import capi, synth
import bench
ref, rs, info = synth.generate_contig(contig_len=synth.CHR20_LEN, seed=synth.SEED, depth=20.0)
chunks = bench.chunk_list(synth.CHR20_LEN)
eng = capi.Engine(0); eng.set_params(); eng.load_reads(rs); eng.set_reference(1, ref)
for _ in range(3):
    eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
eng.set_profiling(True); eng.reset_kernel_stats()
eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
ks = eng.kernel_stats()
print("n=%d  " % n + "  ".join("%s %.3f" % (k.replace("k_", ""), v["total_ms"]) for k, v in sorted(ks.items())))
