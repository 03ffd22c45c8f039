"""Randomised parity: read sets with arbitrary (legal and odd) CIGARs through the HIP path vs the oracle, column by column
and line by line.  Seeds are fixed; every case is small, so a failure message carries the whole input."""
import random
import re

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from clair3_rna_amd import capi
    e = capi.Engine(0)
    yield e
    e.close()


def _rand_cigar(rng, want_q):
    """Random op sequence: M/=/X/I/D/N/S/H/P incl. zero-length ops, leading/trailing I or D, runs of D D, I I, N next to
    I or D, pads between insertions.  Returns (cigar string, query length)."""
    ops = []
    if rng.random() < 0.15:
        ops.append((rng.randint(1, 5), "H"))
    if rng.random() < 0.25:
        ops.append((rng.randint(1, 6), "S"))
    n_core = rng.randint(1, 9)
    for k in range(n_core):
        r = rng.random()
        if r < 0.45:
            ops.append((rng.randint(1, 25), rng.choice("MMMM=X")))
        elif r < 0.58:
            ops.append((rng.randint(1, 4) if rng.random() < 0.9 else rng.randint(17, 22), "I"))
        elif r < 0.72:
            ops.append((rng.randint(1, 5), "D"))
        elif r < 0.84:
            ops.append((rng.randint(1, 40), "N"))
        elif r < 0.90:
            ops.append((rng.randint(1, 3), "P"))
        elif r < 0.95:
            ops.append((0, rng.choice("MID")))              # zero-length op
        else:
            ops.append((rng.randint(1, 3), "D")); ops.append((rng.randint(1, 3), "D"))   # split deletion
    if rng.random() < 0.25:
        ops.append((rng.randint(1, 6), "S"))
    if rng.random() < 0.1:
        ops.append((rng.randint(1, 5), "H"))
    if not any(o in "M=X" and l > 0 for l, o in ops):
        ops.insert(len(ops) // 2, (rng.randint(2, 12), "M"))
    qlen = sum(l for l, o in ops if o in "MIS=X")
    return "".join("%d%s" % lo for lo in ops), qlen


def _case(seed, phased):
    rng = random.Random(seed)
    L = rng.choice([300, 420, 777])
    ref = "".join(rng.choice("ACGT") for _ in range(L))
    if rng.random() < 0.3:          # some IUPAC / N / lower-case reference letters
        ref = list(ref)
        for _ in range(6):
            ref[rng.randrange(L)] = rng.choice("NRYacgtn")
        ref = "".join(ref)
    recs = []
    n_reads = rng.randint(25, 90)
    hot = rng.randint(30, L - 120)
    for _ in range(n_reads):
        pos = max(1, int(rng.gauss(hot, 40)))
        cg, qlen = _rand_cigar(rng, 0)
        while phased and re.search(r"N((\d+[PH])|(0[MID]))*[1-9]\d*[ID]", cg):    # documented phased-mode deviation
            cg, qlen = _rand_cigar(rng, 0)
        if rng.random() < 0.1:
            qlen = max(1, qlen - rng.randint(1, 3))          # query shorter than the CIGAR claims
        seq = "".join(rng.choice("ACGTACGTACGTACGTN" + ("" if phased else "=RY")) for _ in range(qlen))
        flag = (16 if rng.random() < 0.5 else 0) | rng.choice([0] * 14 + [256, 2048, 4, 1024, 512, 8])
        mapq = rng.choice([60] * 8 + [0, 3, 5, 4, 20, 255])
        hp = rng.choice([0, 1, 2, 1, 2]) if phased else 0
        recs.append(dict(pos=pos, cigar=cg, seq=seq, flag=flag, mapq=mapq, hp=hp))
    recs.sort(key=lambda r: r["pos"])
    # alignments must lie inside the contig (the reference indexes the reference string with every covered position)
    end = max(r["pos"] + sum(int(n) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", r["cigar"]) if o in "MDN=X") for r in recs)
    if end + 40 > len(ref):
        ref = ref + "".join(rng.choice("ACGT") for _ in range(end + 40 - len(ref)))
    return ref, recs


@pytest.mark.parametrize("kw", [dict(), dict(head_tail=1), dict(splice_padding=1), dict(splice_padding=1, head_tail=1),
                                dict(channels=30), dict(channels=30, head_tail=1), dict(snp_min_af=0.0), dict(min_mq=0, min_coverage=1)])
def test_random_cigars_match_the_oracle(eng, kw):
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    from oracle import oracle as orc
    channels = kw.get("channels", 18)
    okw = dict(kw)
    okw.pop("channels", None)
    for k in ("head_tail", "splice_padding"):
        if k in okw:
            okw[k] = bool(okw[k])
    if "snp_min_af" in okw:
        okw["snp_af"] = okw.pop("snp_min_af")
    n_cases, n_lines = 0, 0
    for seed in range(120):
        ref, recs = _case(1000 * len(kw) + seed, phased=(channels == 30))
        rs = ReadSet.from_records(recs)
        eng.params = capi.default_params()
        eng.set_bed(0, None); eng.set_bed(1, None)
        eng.set_params(min_coverage=kw.get("min_coverage", 2), **{k: v for k, v in kw.items() if k != "min_coverage"})
        got = H.engine_chunk(eng, rs, ref, 1, 1, len(ref))
        exp = H.oracle_chunk(rs, ref, 1, 1, len(ref), channels=channels, min_coverage=kw.get("min_coverage", 2),
                             **{k: v for k, v in okw.items() if k != "min_coverage"})
        assert got["lines"] == exp["lines"], (seed, recs, H.first_diff(got["lines"], exp["lines"]))
        if channels == 18 and not kw.get("splice_padding"):
            col = eng.columns()
            rows = exp["rows"]
            assert len(rows) == int((col["flags"] & 1).sum()), (seed, recs)
            for row in rows:
                f = row.split("\t")
                pos = int(f[1])
                o = orc.generate_tensor(f[4], ref[pos - 1].upper(), pos, ref.upper(), 1, snp_af=okw.get("snp_af", 0.08))
                i = pos - col["region_start"]
                assert col["cols"][i].tolist() == o["tensor"], (seed, pos, f[4], recs)
                assert col["depth"][i] == o["depth"], (seed, pos)
        n_cases += 1
        n_lines += len(exp["lines"])
    assert n_cases == 120 and n_lines > 300, n_lines
    eng.params = capi.default_params()
    eng.set_params()
