#!/usr/bin/env python3
"""Golden G4: (24 probabilities, ref33, alt_info) -> VCF row, through the reference's own
batch_output -> output_with -> output_from (clair3_rna/call_variants.py:1077-1392, :684-1020).

Run only in the build container:  python tests/golden/make_golden.py g4

numpy caveat: the reference targets numpy < 1.24 where `1.0 - np.float32(p)` is float64; under
this container's numpy 2 it would stay float32 and QUAL could move in the 2nd decimal.  The harness therefore
wraps quality_score_from so that it receives float(p) — the build's semantics are "f32 products, f64 Phred",
which is what the reference computes in its own environment.
"""
import json
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

GT21 = ['AA', 'AC', 'AG', 'AT', 'CC', 'CG', 'CT', 'GG', 'GT', 'TT', 'DelDel', 'ADel', 'CDel', 'GDel', 'TDel', 'InsIns',
        'AIns', 'CIns', 'GIns', 'TIns', 'InsDel']


def rand_alt_info(rng, ref_base, ref_after, depth, kinds):
    """Ordered alt_info string like the tensor builder emits."""
    items = []
    pool = list(kinds)
    rng.shuffle(pool)
    for k in pool:
        if k == "X":
            for b in rng.sample([x for x in "ACGT" if x != ref_base], rng.randint(1, 3)):
                items.append(("X" + b, rng.randint(1, max(1, depth // 2))))
        elif k == "I":
            for _ in range(rng.randint(1, 3)):
                n = rng.choice([1, 1, 2, 3, 5, 12, 55])
                items.append(("I" + ref_base + "".join(rng.choice("ACGT") for _ in range(n)), rng.randint(1, max(1, depth // 2))))
        elif k == "D":
            for n in rng.sample([1, 2, 3, 4, 7, 20, 60], rng.randint(1, 3)):
                items.append(("D" + ref_after[:n], rng.randint(1, max(1, depth // 2))))
    if rng.random() < 0.85:
        items.append(("R" + ref_base, rng.randint(0, depth)))
    # de-duplicate keys, keep first
    seen, out = set(), []
    for k, v in items:
        if k not in seen:
            seen.add(k)
            out.append((k, v))
    if rng.random() < 0.5:
        rng.shuffle(out)
    return "%d-%s\n" % (depth, " ".join("%s %d" % kv for kv in out))


def gen_g4():
    import refharness as rh
    cv = rh.load_call_variants()
    orig_q = cv.quality_score_from
    cv.quality_score_from = lambda p: orig_q(float(p))
    rng = random.Random(20240422 + 4)
    nrng = np.random.RandomState(44)
    cfg = cv.OutputConfig(is_show_reference=True, is_debug=False, is_haploid_precise_mode_enabled=False,
                          is_haploid_sensitive_mode_enabled=False, is_output_for_ensemble=False, quality_score_for_pass=2,
                          tensor_fn='PIPE', input_probabilities=False, add_indel_length=False, gvcf=False, pileup=True,
                          enable_long_indel=False, maximum_variant_length_that_need_infer=50, keep_iupac_bases=False)
    cases = []
    for i in range(700):
        ref33 = "".join(rng.choice("ACGT") for _ in range(33))
        if i % 37 == 0:
            ref33 = ref33[:16] + rng.choice("NRYM") + ref33[17:]
        if i % 41 == 0:
            ref33 = ref33[:20] + "R" + ref33[21:]
        ref_base = ref33[16] if ref33[16] in "ACGT" else "A"
        depth = rng.choice([4, 8, 12, 20, 20, 35, 80, 300])
        kinds = rng.choice([["X"], ["X"], ["I"], ["D"], ["X", "I"], ["X", "D"], ["I", "D"], ["X", "I", "D"], []])
        alt = rand_alt_info(rng, ref_base, ref33[17:] + "ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT", depth, kinds)
        # probabilities: boost one gt21 class and one zygosity
        g = nrng.dirichlet(np.ones(21) * 0.3)
        boost = rng.randrange(21)
        g[boost] += rng.choice([0.2, 1.0, 3.0, 10.0])
        if i % 5 == 0:   # make the reference genotype likely
            g[GT21.index(ref_base + ref_base)] += 5.0
        g /= g.sum()
        z = nrng.dirichlet(np.ones(3) * 0.5)
        z[rng.randrange(3)] += rng.choice([0.1, 1.0, 5.0])
        z /= z.sum()
        if i % 11 == 0:   # exact ties between classes
            g = np.round(g, 2)
            g /= max(g.sum(), 1e-9)
        Y = np.concatenate([g, z]).astype(np.float32)
        out = []
        util = cv.OutputUtilities(None, out.append, None, None, None)
        cv.batch_output(["chr20:%d:%s" % (1000 + i, ref33)], [alt], Y[None, :], cfg, util)
        cases.append(dict(pos=1000 + i, ref33=ref33, alt_info=alt, Y=[float(np.float32(v)) for v in Y], rows=out))
    kinds = {}
    for c in cases:
        for r in c["rows"]:
            f = r.split("\t")
            key = (f[6], f[9].split(":")[0], "," in f[4], len(f[3]) > 1, len(f[4].split(",")[0]) > 1)
            kinds[key] = kinds.get(key, 0) + 1
    with open(os.path.join(HERE, "g4_decode.json"), "w") as f:
        json.dump(dict(cases=cases, note="batch_output with showRef, qual=2, add_indel_length=False; quality_score_from(float(p))"),
                  f, separators=(",", ":"))
    print("g4: %d cases, %d rows, %d distinct (filter, GT, multi, del, ins) kinds" %
          (len(cases), sum(len(c["rows"]) for c in cases), len(kinds)))
    for k, v in sorted(kinds.items()):
        print("   ", k, v)


if __name__ == "__main__":
    import sys
    sys.path.insert(0, HERE)
    gen_g4()
