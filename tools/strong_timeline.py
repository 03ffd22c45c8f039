# Per-contig timeline of the strong-scaling configuration on one GPU (BASELINE.json configs[2] at a quarter of every contig's length):
# what one context spends per contig in each call, nothing overlapped, and the rate with 1 / 2 contexts.   python tools/strong_timeline.py [scale]
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clair3_rna_amd import capi, shard, synth
import bench
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
lens = [max(200000, int(l * scale)) for _n, l in shard.GRCH38]
w = synth.random_weights(18)
data = []
for ci, L in enumerate(lens):
    ref, rs, info = synth.generate_contig(contig_len=L, seed=synth.SEED + 1000 + ci, depth=30.0)
    data.append((ci, ref, capi.pinned_readset(rs), bench.chunk_list(L), info))
e = capi.Engine(0); e.set_params(); e.load_weights(w, 18)
def one(e, ref, rs, chunks, T=None):
    t = [time.perf_counter()]
    e.set_reference(1, ref); e.synchronize(); t.append(time.perf_counter())
    e.load_reads(rs); e.synchronize(); t.append(time.perf_counter())
    e.begin_batch(); n = e.scan_regions(chunks); e.end_batch(); e.synchronize(); t.append(time.perf_counter())
    if n: e.infer(fetch=False)
    e.synchronize(); t.append(time.perf_counter())
    if n: e.fetch_probs(n)
    t.append(time.perf_counter())
    if T is not None: T.append([1e3 * (b - a) for a, b in zip(t, t[1:])] + [n])
    return n
for d in data: one(e, d[1], d[2], d[3])          # sizes
T = []
t0 = time.perf_counter(); tot = sum(one(e, d[1], d[2], d[3], T) for d in data); el = time.perf_counter() - t0
T = np.array(T)
print("one context, every call synchronised: %.1f ms for %d sites = %.2f M sites/s" % (1e3 * el, tot, tot / el / 1e6))
print("per contig (ms): set_reference %.2f  load_reads %.2f  scan %.2f  network %.2f  fetch %.2f | sums: %s" % (*T[:, :5].mean(0), np.round(T[:, :5].sum(0), 1)))
print("ideal network-only rate: %.2f M sites/s; sites per contig min/mean/max %d/%d/%d" % (tot / T[:, 3].sum() / 1e3, T[:, 5].min(), T[:, 5].mean(), T[:, 5].max()))
ref_mb = sum(len(d[1]) for d in data) / 1e6
print("reference bytes per pass %.0f MB (%.1f GB/s in set_reference)" % (ref_mb, ref_mb / T[:, 0].sum()))
e.close()
