# quick look at precision 2 ("f16+f8") against the oracle and against f16x3 on the GPU precision test's inputs
import sys, numpy as np
sys.path.insert(0, '.')
from clair3_rna_amd import capi, synth
from oracle import oracle as orc
eng = capi.Engine(0)
rng = np.random.RandomState(11)
for C, wseed in ((18, 1234),):
    w = synth.random_weights(C, seed=wseed)
    X = np.concatenate([rng.randint(-216, 217, size=(40, 33, C)), rng.randint(-20, 21, size=(60, 33, C)), np.zeros((3, 33, C), int)]).astype(np.int32)
    po = orc.forward(w, X)
    eng.load_weights(w, C)
    res = {}
    for mode in ("f16x3", "f16+f8"):
        eng.set_precision(mode)
        res[mode] = eng.infer(tensors=X)
        d = np.abs(res[mode] - po)
        print(C, mode, "max|dP| %.3e mean %.3e" % (d.max(), d.mean()))
    d = np.abs(res["f16+f8"] - po).max(1)
    print("per-site max, +-216 group:", np.sort(d[:40])[-5:], " +-20 group:", np.sort(d[40:100])[-5:], " zero group:", d[100:])
    print("sites 0..63 vs 64..: ", d[:64].max(), d[64:].max())
    # pileup-like windows (tools/precision_probe.py's generator)
    r7 = np.random.RandomState(7); n = 512
    Xp = np.zeros((n, 33, C), np.int32)
    for s in range(n):
        depth = int(r7.choice([6, 12, 20, 40, 90, 216]))
        for t in range(33):
            k = r7.randint(0, 4); fwd = r7.binomial(depth, 0.5)
            Xp[s, t, k] = -fwd; Xp[s, t, 9 + k] = -(depth - fwd)
            for _ in range(r7.randint(0, 3)): Xp[s, t, r7.randint(0, C)] += r7.randint(1, max(2, depth // 3))
    pop = orc.forward(w, Xp)
    for mode in ("f16x3", "f16+f8"):
        eng.set_precision(mode)
        d = np.abs(eng.infer(tensors=Xp) - pop)
        print("pileup-like 512 sites", mode, "max|dP| %.3e mean %.3e" % (d.max(), d.mean()))
