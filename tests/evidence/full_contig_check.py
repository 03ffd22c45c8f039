# One-off evidence run on the GPU box: the WHOLE synthetic chr20 (BASELINE.json configs[1]) through the HIP path and through
# the oracle, chunk by chunk — every line (position, ref33, 594 ints, ordered alt_info) and every rescaled tensor identical,
# probabilities within 1e-4.  ~3 min of host time (the oracle's text stages are single-threaded).  python tests/evidence/full_contig_check.py
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench
from clair3_rna_amd import capi, synth, altinfo
from oracle import oracle as orc
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--config3", action="store_true", help="BASELINE.json configs[3] flavour: MAS-Seq reads, depth 30, HP tags, 30 channels")
ap.add_argument("--contig_len", type=int, default=synth.CHR20_LEN)
ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3", "f16+f8", "auto"], help="network arithmetic (c3r_set_precision)")
opt = ap.parse_args()
CH = 30 if opt.config3 else 18
ref, rs, info = synth.generate_contig(contig_len=opt.contig_len, depth=30.0 if opt.config3 else 20.0, platform="hifi" if opt.config3 else "ont", phased=opt.config3)
L = len(ref)
refs = ref.decode()
chunks = bench.chunk_list(L)
print('config3' if opt.config3 else 'config1', 'contig', L, 'reads', len(rs), 'channels', CH, flush=True)
eng = capi.Engine(0); eng.set_params(channels=CH); eng.load_reads(rs); eng.set_reference(1, ref)
w = synth.random_weights(CH); eng.load_weights(w, CH); eng.set_precision(opt.precision)
print("precision", opt.precision, "->", eng.precision(), flush=True)
t0 = time.time()
n_tot, worst = 0, 0.0
for ci, (a, b) in enumerate(chunks):
    n = eng.scan(a, b)
    raw, X = eng.tensors(rescaled=False), eng.tensors(rescaled=True)
    sites, toks = eng.sites(), eng.tokens()
    rstart = max(1, a - 1000)
    refslice = refs[rstart - 1:b + 1000]
    lines = altinfo.format_lines("chr20", sites, raw, toks, rs, refslice.upper(), rstart) if n else []
    es, ee = max(1, a - 33), b + 33
    rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, "chr20", es, ee, with_hp=opt.config3)
    exp = orc.create_tensor(rows, "chr20", refslice.upper(), rstart, orc.make_params(phased=opt.config3))
    assert lines == exp, (ci, len(lines), len(exp))
    Xo, _ = orc.batch_from_lines(exp, CH)
    assert np.array_equal(X, Xo)
    if n:
        p = eng.infer(); po = orc.forward(w, Xo)
        worst = max(worst, float(np.abs(p - po).max()))
    n_tot += n
    print("chunk %2d/%d: %6d sites identical, max |dP| so far %.2e  (%.0f s)" % (ci + 1, len(chunks), n, worst, time.time() - t0), flush=True)
assert worst < 1e-4
print("FULL CONTIG OK (%s): %d sites, %d reads, all lines and tensors identical, max |dP| = %.2e" % (opt.precision, n_tot, len(rs), worst))
