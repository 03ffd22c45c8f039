"""Pin the CPU oracle (oracle/c3r_oracle.c) against golden vectors captured from the reference's own
Python functions (tests/golden/make_golden.py; generated in the build container, committed as data).

G1  generate_tensor            src/create_tensor_pileup.py:85-302
G2  CreateTensorPileup driver  src/create_tensor_pileup.py:333-657 (+ chunk arithmetic :379-422)
G3  tensor_generator_from      clair3_rna/utils.py:64-138
"""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def g1():
    return json.load(open(os.path.join(G, "g1_columns.json")))


@pytest.fixture(scope="module")
def g2():
    return json.load(gzip.open(os.path.join(G, "g2_streams.json.gz"), "rt"))


def test_g1_columns(g1):
    bad = []
    for i, c in enumerate(g1["cases"]):
        got = orc.generate_tensor(c["bases"], c["ref_base"], c["pos"], g1["ref_seq"], g1["ref_start"], hp=c["hp"],
                                  snp_af=c["snp_af"], indel_af=c["indel_af"])
        exp = c["out"]
        for k in ("tensor", "alt", "depth", "pass_af", "pileup_list", "max_del_length", "max_skip_count"):
            if got[k] != exp[k]:
                bad.append((i, k, c["bases"][:60], got[k], exp[k]))
        assert abs(got["af"] - exp["af"]) < 1e-12
    assert not bad, bad[:5]
    assert len(g1["cases"]) > 300


def _argval(argv, name, default=None, cast=str):
    return cast(argv[argv.index(name) + 1]) if name in argv else default


def case_setup(case, g2):
    """Derive (region dict, oracle params, reference slice) for one G2 case from its CLI argv."""
    argv = case["argv"]
    ctg, seq = g2["ctg"], g2["contig_seq"]
    chunk_id = _argval(argv, "--chunk_id", 0, int)
    chunk_num = _argval(argv, "--chunk_num", 1, int)
    bed, sites = None, None
    if "vcf_sites" in case:
        s = sorted({p for c, p in case["vcf_sites"] if c == ctg})
        size = len(s) // chunk_num if len(s) % chunk_num == 0 else len(s) // chunk_num + 1
        sites = s[(chunk_id - 1) * size:(chunk_id - 1) * size + size]
        reg = orc.chunk_region(ctg_start=min(sites), ctg_end=max(sites))
    elif "bed" in case:
        ext = [(s, e) for c, s, e in case["extend_bed"] if c == ctg]
        bs, be = min(s for s, e in ext), max(e for s, e in ext)
        reg = orc.chunk_region(chunk_id=chunk_id, chunk_num=chunk_num, bed_start=bs, bed_end=be)
        bed = []
        for c, s, e in case["bed"]:
            if c != ctg or e < reg["extend_start"] or s > reg["extend_end"]:
                continue
            bed.append((s, e + 1 if s == e else e))
    elif chunk_id:
        reg = orc.chunk_region(contig_len=case["fai_len"] or len(seq), chunk_id=chunk_id, chunk_num=chunk_num)
    else:
        reg = orc.chunk_region(ctg_start=_argval(argv, "--ctgStart", cast=int), ctg_end=_argval(argv, "--ctgEnd", cast=int))
    P = orc.make_params(
        snp_af=_argval(argv, "--snp_min_af", 0.08, float), indel_af=_argval(argv, "--indel_min_af", 0.15, float),
        min_coverage=_argval(argv, "--minCoverage", 2, int),
        head_tail="--enable_variant_calling_at_sequence_head_and_tail" in argv,
        splice_padding="--enable_padding_in_splice_junction_regions" in argv,
        phased="--add_phasing_feature" in argv, bed=bed, sites=sites)
    ref = seq[reg["reference_start"] - 1:reg["reference_end"]].upper()
    return reg, P, ref


def test_g2_streams(g2):
    names = []
    for case in g2["cases"]:
        reg, P, ref = case_setup(case, g2)
        if case["mpileup_cmd"]:
            r = case["mpileup_cmd"][case["mpileup_cmd"].index("-r") + 1]
            assert r == "%s:%d-%d" % (g2["ctg"], reg["extend_start"], reg["extend_end"]), (case["name"], r, reg)
        got = orc.create_tensor(case["rows"], g2["ctg"], ref, reg["reference_start"], P)
        exp = case["lines"]
        assert len(got) == len(exp), (case["name"], len(got), len(exp))
        for a, b in zip(got, exp):
            assert a == b, (case["name"], a[:200], b[:200])
        names.append(case["name"])
    assert len(names) >= 20


def test_g3_batches():
    g3 = json.load(open(os.path.join(G, "g3_batches.json")))
    for case in g3["cases"]:
        X, depth = orc.batch_from_lines(case["lines"], case["C"])
        exp = np.concatenate([np.asarray(b["X"], dtype=np.int64).reshape(b["shape"]) for b in case["batches"]])
        assert X.shape == exp.shape
        assert np.array_equal(X, exp), case["name"]
        assert all(b["dtype"] == "int32" for b in case["batches"])


def load_g2b():
    from clair3_rna_amd.reads import ReadSet, READ_DTYPE
    g = json.load(gzip.open(os.path.join(G, "g2b_e2e.json.gz"), "rt"))
    for c in g["cases"]:
        reads = np.array([tuple(r) for r in c["reads"]], dtype=READ_DTYPE)
        c["rs"] = ReadSet(reads, np.asarray(c["cigar"], np.uint32), np.asarray(c["seq"], np.uint8))
    return g["cases"]


def test_g2b_reads_to_lines_through_reference_driver():
    """reads -> oracle mpileup text -> oracle driver must equal reads -> oracle mpileup text -> REFERENCE driver."""
    for c in load_g2b():
        rs, ref = c["rs"], c["ref"]
        rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, "chr20", 1, len(ref) + 33, with_hp=c["phased"])
        P = orc.make_params(min_coverage=4, phased=c["phased"],
                            head_tail="--enable_variant_calling_at_sequence_head_and_tail" in c["argv"])
        got = orc.create_tensor(rows, "chr20", ref, 1, P)
        assert got == c["lines"], c["name"]
