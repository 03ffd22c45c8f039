# Two contexts on two host threads over one workload (bench.py's `two_contexts`): where a step's wall-clock goes, call by call —
#   python tools/two_ctx_probe.py [stress|cap|chr20|real] [steps per context]
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clair3_rna_amd import capi, synth
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "stress"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if which == "stress":
    L, gen = 16000000, dict(seed=synth.SEED + 4, depth=500.0)
elif which == "cap":
    L, gen = 400000, dict(seed=synth.SEED + 5, depth=20000.0, expressed_frac=0.01, intron_lo=100.0, intron_hi=800.0)
elif which == "real":
    L, gen = synth.CHR20_LEN, dict(seed=synth.SEED + 6, depth=20.0, expr_sigma=2.3, max_level=12000.0)
else:
    L, gen = synth.CHR20_LEN, dict(seed=synth.SEED, depth=20.0)
ref, rs, info = synth.generate_contig(contig_len=L, **gen)
chunks = bench.chunk_list(L)
rsh = capi.pinned_readset(rs)
w = synth.random_weights(18)
engs = []
for _ in range(2):
    e = capi.Engine(0); e.set_params(); e.set_reference(1, ref); e.load_weights(w, 18); e.set_precision("f16x3")
    for _ in range(2):
        e.load_reads(rsh); e.begin_batch(); n = e.scan_regions(chunks); e.end_batch()
        if n: e.infer()
    engs.append(e)
T0 = time.perf_counter()
log = [[], []]
def drive(j):
    e = engs[j]
    for _ in range(K):
        t0 = time.perf_counter(); e.load_reads(rsh)
        t1 = time.perf_counter(); e.begin_batch(); n = e.scan_regions(chunks); e.end_batch()
        t2 = time.perf_counter()
        if n: e.infer()
        t3 = time.perf_counter()
        log[j].append((t0 - T0, t1 - t0, t2 - t1, t3 - t2))
th = [threading.Thread(target=drive, args=(j,)) for j in range(2)]
t_start = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
el = time.perf_counter() - t_start
print("%s: %d steps on two contexts in %.2f ms = %.3f ms per step (one context alone: see bench)" % (which, 2 * K, 1e3 * el, 1e3 * el / (2 * K)))
for j in range(2):
    for (ts, a, b, c) in log[j]:
        print("ctx%d  start %7.2f ms | load_reads %6.2f | scan %6.2f | infer %6.2f" % (j, 1e3 * ts, 1e3 * a, 1e3 * b, 1e3 * c))
