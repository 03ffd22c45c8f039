"""Known-answer tests for oracle A1 (reads -> `samtools mpileup` text).

samtools/htslib are absent from the image and un-vendored in the reference (parity unpinned for this
stage, see oracle/c3r_oracle.c).  The expected strings below are hand-derived from the documented
mpileup column grammar as the reference invokes it (src/create_tensor_pileup.py:436-451:
`--reverse-del --min-MQ 5 --min-BQ 0 --excl-flags 2316`, no -f): '^'+chr(MAPQ+33) at a read's first
column, literal bases (case = strand), '*'/'#' inside deletions, '>'/'<' inside N ops, '+<n><seq>' /
'-<n><N..>' on the column BEFORE the indel, '$' at the last column.
"""
import numpy as np

from clair3_rna_amd.reads import ReadSet
from oracle import oracle as orc


def pile(records, beg=1, end=60, **kw):
    rs = ReadSet.from_records(records)
    rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, "c", beg, end, **kw)
    return {int(r.split("\t")[1]): r.split("\t") for r in rows}


def bases(records, **kw):
    return {p: f[4] for p, f in pile(records, **kw).items()}


def R(pos, cigar, seq, flag=0, mapq=60, hp=0):
    return dict(pos=pos, cigar=cigar, seq=seq, flag=flag, mapq=mapq, hp=hp)


def test_match_only_forward_and_reverse():
    assert bases([R(9, "3M", "ACG")]) == {10: "^]A", 11: "C", 12: "G$"}
    assert bases([R(9, "3M", "ACG", flag=16)]) == {10: "^]a", 11: "c", 12: "g$"}


def test_insertion_attached_to_previous_column():
    assert bases([R(9, "2M2I2M", "ACGTTA")]) == {10: "^]A", 11: "C+2GT", 12: "T", 13: "A$"}
    assert bases([R(9, "2M2I2M", "ACGTTA", flag=16)]) == {10: "^]a", 11: "c+2gt", 12: "t", 13: "a$"}


def test_deletion_forward_star_reverse_hash():
    assert bases([R(9, "2M1D2M", "ACGT")]) == {10: "^]A", 11: "C-1N", 12: "*", 13: "G", 14: "T$"}
    assert bases([R(9, "2M2D2M", "ACGT", flag=16)]) == {10: "^]a", 11: "c-2nn", 12: "#", 13: "#", 14: "g", 15: "t$"}


def test_ref_skip():
    assert bases([R(9, "2M3N2M", "ACGT")]) == {10: "^]A", 11: "C", 12: ">", 13: ">", 14: ">", 15: "G", 16: "T$"}
    assert bases([R(9, "2M2N2M", "ACGT", flag=16)]) == {10: "^]a", 11: "c", 12: "<", 13: "<", 14: "g", 15: "t$"}


def test_clips_consume_no_reference():
    assert bases([R(9, "2H2S3M1S", "TTACGT")]) == {10: "^]A", 11: "C", 12: "G$"}


def test_leading_insertion_dropped_trailing_reported():
    assert bases([R(9, "2I3M", "TTACG")]) == {10: "^]A", 11: "C", 12: "G$"}
    assert bases([R(9, "2S2I3M", "GGTTACG")]) == {10: "^]A", 11: "C", 12: "G$"}
    assert bases([R(9, "3M2I", "ACGTT")]) == {10: "^]A", 11: "C", 12: "G+2TT$"}
    assert bases([R(9, "3M2I4S", "ACGTTCCCC")]) == {10: "^]A", 11: "C", 12: "G+2TT$"}


def test_insertion_after_deletion_and_deletion_after_insertion():
    assert bases([R(9, "2M1D2I2M", "ACTTGA")]) == {10: "^]A", 11: "C-1N", 12: "*+2TT", 13: "G", 14: "A$"}
    assert bases([R(9, "2M2I1D2M", "ACTTGA")]) == {10: "^]A", 11: "C+2TT", 12: "*", 13: "G", 14: "A$"}


def test_adjacent_ops_are_merged():
    assert bases([R(9, "2M1D2D2M", "ACGT")]) == {10: "^]A", 11: "C-3NNN", 12: "*", 13: "*", 14: "*", 15: "G", 16: "T$"}
    assert bases([R(9, "2M1I2I2M", "ACTTTGA")]) == {10: "^]A", 11: "C+3TTT", 12: "G", 13: "A$"}
    assert bases([R(9, "2M1I1P1I2M", "ACTTGA")]) == {10: "^]A", 11: "C+2TT", 12: "G", 13: "A$"}
    assert bases([R(9, "1M1=1X", "ACG")]) == {10: "^]A", 11: "C", 12: "G$"}


def test_insertion_after_ref_skip():
    assert bases([R(9, "2M2N1I2M", "ACTGA")]) == {10: "^]A", 11: "C", 12: ">", 13: ">+1T", 14: "G", 15: "A$"}


def test_filters_flags_and_mapq():
    recs = [R(9, "2M", "AC", flag=256), R(9, "2M", "AC", flag=2048), R(9, "2M", "AC", flag=4), R(9, "2M", "AC", flag=8),
            R(9, "2M", "GG", flag=1024), R(9, "2M", "TT", flag=512), R(9, "2M", "AC", mapq=4), R(9, "2M", "CA", mapq=5)]
    assert bases(recs) == {10: "^]G^]T^&C", 11: "G$T$A$"}
    assert bases([R(9, "2M", "AC", mapq=200)]) == {10: "^~A", 11: "C$"}
    # --min-MQ is a parameter
    assert bases([R(9, "2M", "AC", mapq=20), R(9, "2M", "GG", mapq=40)], min_mq=30) == {10: "^IG", 11: "G$"}
    # no -A on the reference's command line: anomalous pairs (paired without the proper-pair bit) are skipped, proper pairs
    # and single-end reads are kept, whatever their mate bits say
    recs = [R(9, "2M", "AC", flag=1), R(9, "2M", "GG", flag=3), R(9, "2M", "TT", flag=1 | 64 | 16), R(9, "2M", "CA", flag=3 | 128 | 16),
            R(9, "2M", "AT", flag=2)]
    assert bases(recs) == {10: "^]G^]c^]A", 11: "G$a$T$"}


def test_bam_order_and_depth_column():
    recs = [R(9, "4M", "AAAA"), R(10, "2M", "CC", flag=16), R(10, "3M", "GGG")]
    p = pile(recs)
    assert p[11][4] == "A^]c^]G" and p[11][3] == "3"
    assert p[12][4] == "Ac$G"
    assert p[13][4] == "A$G$"
    assert sorted(p) == [10, 11, 12, 13]


def test_n_and_iupac_read_bases():
    assert bases([R(9, "4M", "ANRC")]) == {10: "^]A", 11: "N", 12: "R", 13: "C$"}
    assert bases([R(9, "3M", "A=C")]) == {10: "^]A", 11: ".", 12: "C$"}


def test_region_limits_and_mid_read_start():
    recs = [R(0, "5M2D5M", "ACGTACGTAC"), R(20, "3M", "TTT")]
    assert sorted(bases(recs, beg=4, end=8)) == [4, 5, 6, 7, 8]
    b = bases(recs, beg=4, end=8)
    assert b[4] == "T" and b[5] == "A-2NN" and b[6] == "*" and b[7] == "*" and b[8] == "C"
    assert bases(recs, beg=13, end=20) == {}
    assert bases(recs, beg=21, end=30) == {21: "^]T", 22: "T", 23: "T$"}


def test_hp_column_and_bed_filter():
    recs = [R(9, "3M", "ACG", hp=1), R(9, "3M", "ACG", hp=2, flag=16), R(10, "2M", "TT")]
    p = pile(recs, with_hp=True)
    assert p[10][6] == "1,2" and p[11][6] == "1,2,*"
    p = pile(recs, bed=[(10, 11)])
    assert sorted(p) == [11]
    assert p[11][4] == "Cc^]T"       # the read cursor keeps advancing outside the bed


def test_query_shorter_than_cigar_prints_N():
    assert bases([R(9, "4M", "AC")]) == {10: "^]A", 11: "C", 12: "N", 13: "N$"}


def test_read_without_reference_span_is_ignored():
    assert bases([R(9, "4S", "ACGT"), R(9, "2M", "GG")]) == {10: "^]G", 11: "G$"}


def test_the_two_samtools_printers_differ_only_behind_an_insertion():
    """compat = 0 restates samtools <= 1.10 (pileup_seq: `+<n>` and the next n query bases, pads skipped, nothing for a deletion behind
    the insertion); compat = 1 restates samtools >= 1.11 (htslib bam_plp_insertion: the run of I / P ops as ONE insertion with the pads
    as '*', and the length of a D that ends the run).  The reference's parser (src/create_tensor_pileup.py:151-163) reads `C+2TT-1N` as
    three tokens: a base, an insertion, a deletion."""
    # an I immediately followed by a D
    assert bases([R(9, "2M2I1D2M", "ACTTGA")], compat=0) == {10: "^]A", 11: "C+2TT", 12: "*", 13: "G", 14: "A$"}
    assert bases([R(9, "2M2I1D2M", "ACTTGA")], compat=1) == {10: "^]A", 11: "C+2TT-1N", 12: "*", 13: "G", 14: "A$"}
    assert bases([R(9, "2M2I3D2M", "ACTTGA", flag=16)], compat=1) == {10: "^]a", 11: "c+2tt-3nnn", 12: "#", 13: "#", 14: "#", 15: "g", 16: "a$"}
    # ... on a deleted column, and on the last column of a ref-skip
    assert bases([R(9, "2M1D2I1D2M", "ACTTGA")], compat=1) == {10: "^]A", 11: "C-1N", 12: "*+2TT-1N", 13: "*", 14: "G", 15: "A$"}
    assert bases([R(9, "2M2N1I2D2M", "ACTGA")], compat=1) == {10: "^]A", 11: "C", 12: ">", 13: ">+1T-2NN", 14: "*", 15: "*", 16: "G", 17: "A$"}
    # pads inside an insertion
    assert bases([R(9, "2M1I1P1I2M", "ACTTGA")], compat=0) == {10: "^]A", 11: "C+2TT", 12: "G", 13: "A$"}
    assert bases([R(9, "2M1I1P1I2M", "ACTTGA")], compat=1) == {10: "^]A", 11: "C+3T*T", 12: "G", 13: "A$"}
    assert bases([R(9, "2M1P2I2M", "ACTTGA")], compat=1) == {10: "^]A", 11: "C+3*TT", 12: "G", 13: "A$"}
    assert bases([R(9, "2M2I1P1D2M", "ACTTGA")], compat=1) == {10: "^]A", 11: "C+3TT*-1N", 12: "*", 13: "G", 14: "A$"}
    # a reverse-strand read prints its pads as '#' (pileup_seq: `pad = rev_del ? '#' : '*'`; the reference always passes --reverse-del)
    assert bases([R(9, "2M1I1P1I2M", "ACTTGA", flag=16)], compat=1) == {10: "^]a", 11: "c+3t#t", 12: "g", 13: "a$"}
    assert bases([R(9, "2M1P2I2M", "ACTTGA", flag=16)], compat=1) == {10: "^]a", 11: "c+3#tt", 12: "g", 13: "a$"}
    # everything else is printed alike
    for cigar, seq in (("2M1D2M", "ACGT"), ("2M2I2M", "ACGTTA"), ("2M3N2M", "ACGT"), ("2M1D2I2M", "ACTTGA"), ("3M2I", "ACGTT"), ("2M1P1D2M", "ACGT")):
        assert bases([R(9, cigar, seq)], compat=0) == bases([R(9, cigar, seq)], compat=1), cigar


def test_new_printer_rows_through_the_oracle_parser():
    """`C+2TT-1N` is a base, an insertion and a deletion for generate_tensor (src/create_tensor_pileup.py:151-163): the column counts a
    D (and D1) that the <= 1.10 text does not have."""
    recs = [R(9, "4M2I1D4M", "ACGTTTACGT") for _ in range(3)] + [R(9, "9M", "ACGTAACGT") for _ in range(3)]
    old = orc.generate_tensor(bases(recs, compat=0)[13], "T", 13, "N" * 9 + "ACGTAACGT" + "N" * 20, 1)
    new = orc.generate_tensor(bases(recs, compat=1)[13], "T", 13, "N" * 9 + "ACGTAACGT" + "N" * 20, 1)
    I, I1, D, D1 = 4, 5, 6, 7
    assert old["tensor"][I] == new["tensor"][I] == 3 and old["tensor"][I1] == new["tensor"][I1] == 3
    assert old["tensor"][D] == 0 and new["tensor"][D] == 3 and new["tensor"][D1] == 3
    assert [k for k, _n in new["alt"] if k.startswith("D")] == ["DA"] and not [k for k, _n in old["alt"] if k.startswith("D")]


def test_depth_cap_follows_the_pileup_engines_node_pool():
    """mpileup -d, hand-derived from htslib's bam_plp_push: a read is discarded iff the engine already stands on its start position (it
    is not the first read pushed there) and the node pool — the read list plus the list's empty tail node — holds more than max_depth
    nodes.  Reads that all start on one position therefore pile up to exactly max_depth (the 8000 samtools users see at amplicons)."""
    cap = 3
    recs = [R(9, "4M", "ACGT") for _ in range(6)]
    p = pile(recs, max_depth=cap)
    assert p[10][3] == "3" and p[13][3] == "3"                     # first read kept unseen, then list + 1 <= 3 twice more
    # staggered starts: the first read of a position is always kept, whatever the list holds
    recs = [R(9, "6M", "ACGTAC")] * 5 + [R(10, "4M", "CGTA")] * 2 + [R(12, "2M", "TA")]
    p = pile(recs, max_depth=cap)
    assert p[10][3] == "3"                                          # position 10: three of the five
    assert p[11][3] == "4"                                          # + the first read of position 11; its twin meets a pool of 5 > 3
    assert p[13][3] == "5" and p[13][4].count("^") == 1             # + the only read of position 13 (first pushed there)
    # a read is retired while the column AT its exclusive end is processed — after the reads that start on that column were pushed: three
    # reads whose last base sits right before the newcomers still fill the list (1 kept: the first), one position further they are gone
    recs = [R(9, "2M", "AC")] * 3 + [R(11, "2M", "GT")] * 4
    p = pile(recs, max_depth=cap)
    assert p[10][3] == "3" and p[11][3] == "3" and p[12][3] == "1"
    recs = [R(9, "2M", "AC")] * 3 + [R(12, "2M", "TA")] * 4
    p = pile(recs, max_depth=cap)
    assert p[11][3] == "3" and p[13][3] == "3"
    # no cap
    assert pile([R(9, "4M", "ACGT") for _ in range(6)], max_depth=0)[10][3] == "6"
