# Wall-clock of the tensor build alone (c3r_load_reads + the scan of the contig's chunks, no profiler: kernels overlap as in production), resident
# reads and host-resident reads:   python tools/step_time.py [chr20|stress|cap|nocap|real] [repeats]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clair3_rna_amd import capi, synth
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "chr20"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
params = {}
if which == "chr20":
    L, gen = synth.CHR20_LEN, dict(seed=synth.SEED, depth=20.0)
elif which == "stress":
    L, gen = 16000000, dict(seed=synth.SEED + 4, depth=500.0)
elif which == "phased":
    L, gen = synth.CHR20_LEN, dict(seed=synth.SEED + 3, depth=30.0, platform="hifi", phased=True)
    params = dict(channels=30)
elif which in ("cap", "nocap"):
    L, gen = 400000, dict(seed=synth.SEED + 5, depth=20000.0, expressed_frac=0.01, intron_lo=100.0, intron_hi=800.0)
    if which == "nocap":
        params = dict(max_depth=0)
else:
    L, gen = synth.CHR20_LEN, dict(seed=synth.SEED + 6, depth=20.0, expr_sigma=2.3, max_level=12000.0)
ref, rs, info = synth.generate_contig(contig_len=L, **gen)
chunks = bench.chunk_list(L)
rsh = rs if os.environ.get("STEP_UNPINNED") == "1" else capi.pinned_readset(rs)      # (STEP_UNPINNED=1: pageable arrays, the staging path of c3r_load_reads)
eng = capi.Engine(0); eng.set_params(**params); eng.load_reads(rsh); eng.set_reference(1, ref)
def scan():
    eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch(); return n
for _ in range(3):
    eng.load_reads(rsh); n = scan()
eng.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    n = scan()
eng.synchronize()
t1 = time.perf_counter()
for _ in range(R):
    eng.load_reads(rsh); n = scan()
eng.synchronize()
t2 = time.perf_counter()
print("%s %s: %d reads, %d sites | scan alone %.3f ms | load_reads + scan %.3f ms" % (os.environ.get("C3R_LIB", "in-tree").split("/")[-1], which, info["n_reads"], n,
                                                                                      1e3 * (t1 - t0) / R, 1e3 * (t2 - t1) / R))
