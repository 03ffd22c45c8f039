import sys; sys.path.insert(0, '.')
import numpy as np
from clair3_rna_amd import capi, synth
from oracle import oracle as orc
eng = capi.Engine(0)
from tests.test_gpu_parity import _pileup_like
for C in (18, 30):
    w = synth.random_weights(C, seed=1234)
    X = _pileup_like(300, C, 7)
    for k in (1.0, 2.0, 3.0, 4.0, 6.0):
        wk = (k * w).astype(np.float32)
        eng.set_precision("f16x3"); eng.load_weights(wk, C)
        g = eng.precision_guard(); mode = eng.precision()[0]
        po = orc.forward(wk, X)
        e_run = float(np.abs(eng.infer(tensors=X) - po).max())
        eng.set_precision("f32"); e32 = float(np.abs(eng.infer(tensors=X) - po).max())
        print("C=%d norm x%.0f: calib f16x3-vs-f32 %.2e -> %s; vs oracle: running %.2e, f32 %.2e" % (C, k, g["f16_err"], mode, e_run, e32))
