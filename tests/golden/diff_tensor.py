#!/usr/bin/env python3
"""Differential run (build container only, nothing stored): random read sets (tests/test_gpu_fuzz.py's generator) ->
oracle A1 mpileup rows -> BOTH the reference's CreateTensorPileup driver and the oracle's create_tensor; the emitted lines
must be identical.  Pins the oracle's A2/A3 on inputs far outside the committed goldens (odd CIGARs, dense indels, IUPAC,
head/tail, splice padding, phased).  Cases on which the reference itself raises are skipped (and counted).
    python tests/golden/diff_tensor.py [N] [first_seed]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refharness as rh  # noqa: E402
from clair3_rna_amd.reads import ReadSet  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.test_gpu_fuzz import _case  # noqa: E402

OPTS = [
    ([], dict(), False),
    (["--samtools_1_11_rows"], dict(), False),          # (not a reference flag: the rows are printed the samtools >= 1.11 way, `+<ins>-<del>`)
    (["--samtools_1_11_rows", "--add_phasing_feature", "True"], dict(), True),
    (["--enable_variant_calling_at_sequence_head_and_tail", "True"], dict(head_tail=True), False),
    (["--enable_padding_in_splice_junction_regions", "True"], dict(splice_padding=True), False),
    (["--enable_padding_in_splice_junction_regions", "True", "--enable_variant_calling_at_sequence_head_and_tail", "True"],
     dict(splice_padding=True, head_tail=True), False),
    (["--add_phasing_feature", "True"], dict(), True),
    (["--snp_min_af", "0.0"], dict(snp_af=0.0), False),
]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = skipped = total = 0
    for seed in range(s0, s0 + n):
        for oi, (argv, okw, phased) in enumerate(OPTS):
            ref, recs = _case(300000 + 7 * seed + oi, phased=phased)
            ref = ref.upper()
            rs = ReadSet.from_records(recs)
            L = len(ref)
            compat = 1 if "--samtools_1_11_rows" in argv else 0
            argv = [a for a in argv if a != "--samtools_1_11_rows"]
            rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, "chr20", 1, L, with_hp=phased, compat=compat)
            try:
                want, _ = rh.run_create_tensor(rows, ref, "chr20", ["--ctgStart", "1", "--ctgEnd", str(L - 33), "--minCoverage", "2"] + argv)
            except Exception as e:
                skipped += 1
                continue
            got = orc.create_tensor(rows, "chr20", ref, 1, orc.make_params(min_coverage=2, phased=phased, **okw))
            total += 1
            if got != want:
                bad += 1
                if bad <= 4:
                    k = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
                    print("MISMATCH seed", seed, "opts", argv, "lines", len(got), len(want), "first diff at", k)
                    if k < len(got) and k < len(want):
                        fa, fb = got[k].split("\t"), want[k].split("\t")
                        for j in range(5):
                            if fa[j] != fb[j]:
                                print("   field", j, "\n     oracle   :", fa[j][:300], "\n     reference:", fb[j][:300])
    print("%d cases compared, %d mismatches, %d skipped (reference raised)" % (total, bad, skipped))


if __name__ == "__main__":
    main()
