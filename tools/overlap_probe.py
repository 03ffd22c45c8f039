# Tensor build of one contig BESIDE the network pass of another (two contexts, two streams), as call_sample's pipeline runs them:
# how much longer does a chr20 load_reads + scan take while the other context's BiLSTM is on the GPU?
#   python tools/overlap_probe.py [passes] [precision]        (gpurun; prints a table for profiles/rN/prep_beside_network.txt)
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clair3_rna_amd import capi, synth
import bench

R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
ref, rs, info = synth.generate_contig(contig_len=synth.CHR20_LEN, seed=synth.SEED, depth=20.0)
chunks = bench.chunk_list(synth.CHR20_LEN)
rsh = rs if os.environ.get("PAGEABLE") else capi.pinned_readset(rs)          # PAGEABLE=1: the records in ordinary host memory (the runtime stages the copies)
weights = synth.random_weights(18)


def engine():
    e = capi.Engine(0)              # (default: one default-priority stream per context; C3R_TWO_STREAMS=1: the build on a high-priority stream, the network on a low-priority one)
    e.set_params(); e.load_reads(rsh); e.set_reference(1, ref); e.load_weights(weights, 18); e.set_precision(prec)
    return e


def build(e):
    e.load_reads(rsh); e.begin_batch(); n = e.scan_regions(chunks); e.end_batch()
    return n


net, tb = engine(), engine()
n = build(net)
for _ in range(2):
    build(tb); net.infer(fetch=False); net.synchronize()


def timed_builds(label):
    tb.set_profiling(True); tb.reset_kernel_stats()
    t0 = time.perf_counter()
    for _ in range(R):
        build(tb)
    tb.synchronize()
    wall = (time.perf_counter() - t0) / R * 1e3
    tb.set_profiling(False)
    ks = tb.kernel_stats()
    tot = sum(v["total_ms"] for k, v in ks.items() if k.startswith("k_")) / R          # (h2d_reads: the upload, beside the kernels)
    print("%-28s kernels %.3f ms per pass, wall %.3f ms per pass | " % (label, tot, wall) + "  ".join("%s %.3f" % (k.replace("k_", ""), v["total_ms"] / R) for k, v in sorted(ks.items())))
    return tot, wall


t0 = time.perf_counter()
for _ in range(3):
    net.infer(fetch=False)
net.synchronize()
net_ms = (time.perf_counter() - t0) / 3 * 1e3
print("network pass alone (%s, %d sites): %.2f ms" % (prec, n, net_ms))
alone = timed_builds("tensor build alone")

stop = False
passes = [0]


def network_loop():
    while not stop:
        net.infer(fetch=False); net.synchronize(); passes[0] += 1


th = threading.Thread(target=network_loop); th.start()
time.sleep(0.05)
passes[0] = 0
t0 = time.perf_counter()
beside = timed_builds("tensor build beside network")
dt = time.perf_counter() - t0
p = passes[0]
stop = True; th.join()
print("network passes meanwhile: %d in %.1f ms = %.2f ms each (alone %.2f ms)" % (p, dt * 1e3, dt * 1e3 / max(p, 1), net_ms))
print("kernel time beside / alone = %.2f; wall beside / alone = %.2f" % (beside[0] / alone[0], beside[1] / alone[1]))
