import sys
sys.path.insert(0, '.')
import numpy as np
from clair3_rna_amd import capi, synth
L = 3000000
ref, rs, _ = synth.generate_contig(contig_len=L, seed=5, depth=20.0, expressed_frac=0.05)
eng = capi.Engine(0); eng.set_params(); eng.load_reads(rs); eng.set_reference(1, ref)
w = synth.random_weights(18); eng.load_weights(w, 18)
n = eng.scan(1, L)
X = eng.tensors()
p1 = eng.infer().copy(); p2 = eng.infer().copy()
print("n", n, "same-run determinism:", np.array_equal(p1, p2))
# shifted batch: drop the first k sites -> every site changes group/position
for k in (1, 32, 64, 100):
    q = eng.infer(tensors=X[k:]).copy()
    d = np.abs(q - p1[k:]).max(axis=1)
    bad = np.nonzero(d > 0)[0]
    print("shift", k, "n_diff", len(bad), "max", d.max(), "first bad idx", bad[:10], "(idx+k)%128:", ((bad[:10] + k) % 128), "idx%128", bad[:10] % 128)
eng.set_precision("f32")
p3 = eng.infer().copy(); q3 = eng.infer(tensors=X[64:]).copy()
print("f32 shift 64 equal:", np.array_equal(q3, p3[64:]))
