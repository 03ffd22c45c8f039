#!/usr/bin/env python3
"""Differential run (build container only, nothing stored): N random (probabilities, ref33, alt_info) cases through the
reference's batch_output and through clair3_rna_amd/decode.py; prints the first mismatches.  Cases that ever disagree get
added to make_golden_decode.py so that the committed fixture pins them.
    python tests/golden/diff_decode.py [N] [seed]"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden_decode import GT21, rand_alt_info  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import refharness as rh
    cv = rh.load_call_variants()
    orig_q = cv.quality_score_from
    cv.quality_score_from = lambda p: orig_q(float(p))
    from clair3_rna_amd import decode
    rng = random.Random(seed)
    nrng = np.random.RandomState(seed)
    cfg = cv.OutputConfig(is_show_reference=True, is_debug=False, is_haploid_precise_mode_enabled=False,
                          is_haploid_sensitive_mode_enabled=False, is_output_for_ensemble=False, quality_score_for_pass=2,
                          tensor_fn='PIPE', input_probabilities=False, add_indel_length=False, gvcf=False, pileup=True,
                          enable_long_indel=False, maximum_variant_length_that_need_infer=50, keep_iupac_bases=False)
    bad = 0
    for i in range(n):
        ref33 = "".join(rng.choice("ACGT") for _ in range(33))
        if rng.random() < 0.04:
            ref33 = ref33[:16] + rng.choice("NRYMKSWD") + ref33[17:]
        if rng.random() < 0.05:
            k = rng.randrange(33)
            ref33 = ref33[:k] + rng.choice("NRY") + ref33[k + 1:]
        ref_base = ref33[16] if ref33[16] in "ACGT" else "A"
        depth = rng.choice([1, 2, 4, 8, 12, 20, 20, 35, 80, 300, 5000])
        kinds = rng.choice([["X"], ["X"], ["I"], ["D"], ["X", "I"], ["X", "D"], ["I", "D"], ["X", "I", "D"], []])
        alt = rand_alt_info(rng, ref_base, ref33[17:] + "ACGT" * 20, depth, kinds)
        mode = rng.randrange(6)
        g = nrng.dirichlet(np.ones(21) * rng.choice([0.05, 0.3, 1.0, 5.0]))
        if mode == 0:
            g[rng.randrange(21)] += rng.choice([0.2, 1.0, 3.0, 10.0])
        elif mode == 1:
            g[GT21.index(ref_base + ref_base)] += 5.0
        elif mode == 2:
            g = np.round(g, 1)
        elif mode == 3:
            a, b = rng.sample(range(21), 2); g[a] = g[b] = g.max() + 0.1        # exact tie at the top
        g = g / max(g.sum(), 1e-9)
        z = nrng.dirichlet(np.ones(3) * rng.choice([0.1, 0.5, 3.0]))
        if mode == 4:
            z = np.round(z, 1)
        z = z / max(z.sum(), 1e-9)
        Y = np.concatenate([g, z]).astype(np.float32)
        out = []
        util = cv.OutputUtilities(None, out.append, None, None, None)
        try:
            cv.batch_output(["chr20:%d:%s" % (1000 + i, ref33)], [alt], Y[None, :], cfg, util)
        except Exception as e:         # the reference itself fails on this input: not a parity case
            continue
        mine = decode.vcf_rows("chr20", [1000 + i], [ref33], [alt], Y[None, :])
        if [r.rstrip("\n") for r in out] != mine:
            bad += 1
            if bad <= 5:
                print("MISMATCH case", i, "\n  ref33", ref33, "\n  alt", alt.strip(), "\n  Y", [round(float(v), 4) for v in Y],
                      "\n  ref :", out, "\n  mine:", mine)
    print("%d cases, %d mismatches" % (n, bad))


if __name__ == "__main__":
    main()
