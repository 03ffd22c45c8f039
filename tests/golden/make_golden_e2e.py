#!/usr/bin/env python3
"""Golden G2b: synthetic READS -> (oracle A1 mpileup text) -> the REFERENCE's CreateTensorPileup driver ->
candidate lines.  Pins the whole tensor-build contract on realistic, mutually consistent columns: the
committed fixture holds the read records (inputs) and the reference's output lines (expected).

Run only in the build container:  python tests/golden/make_golden_e2e.py
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refharness as rh  # noqa: E402
from clair3_rna_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    cases = []
    for name, kw, argv, phased in [
        ("ont20", dict(seed=31, ref_len=14000, n_genes=4, depth=20), [], False),
        ("ont20_headtail", dict(seed=33, ref_len=9000, n_genes=3, depth=14), ["--enable_variant_calling_at_sequence_head_and_tail", "True"], False),
        ("hifi_phased", dict(seed=35, ref_len=12000, n_genes=3, depth=16, platform="hifi", phased=True), ["--add_phasing_feature", "True"], True),
        ("ont_phased", dict(seed=39, ref_len=9000, n_genes=3, depth=18, phased=True), ["--add_phasing_feature", "True"], True),
        ("ont250", dict(seed=37, ref_len=5000, n_genes=1, depth=250, mean_len=500, err_del=0.12, err_ins=0.06, err_mismatch=0.15), [], False),
    ]:
        ref, rs, _ = synth.small_case(**kw)
        L = len(ref)
        rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, "chr20", 1, L + 33, with_hp=phased)
        lines, _ = rh.run_create_tensor(rows, ref, "chr20", ["--ctgStart", "1", "--ctgEnd", str(L), "--minCoverage", "4"] + argv)
        cases.append(dict(name=name, ref=ref, argv=argv, phased=phased,
                          reads=[[int(x) for x in r] for r in rs.reads.tolist()],
                          cigar=rs.cigar.tolist(), seq=rs.seq.tolist(), lines=lines))
        print("  g2b %-16s reads=%4d rows=%6d lines=%4d" % (name, len(rs), len(rows), len(lines)))
    with gzip.open(os.path.join(HERE, "g2b_e2e.json.gz"), "wt") as f:
        json.dump(dict(cases=cases), f, separators=(",", ":"))


if __name__ == "__main__":
    main()
