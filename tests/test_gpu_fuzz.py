"""Randomised parity: read sets with arbitrary (legal and odd) CIGARs through the HIP path vs the oracle, column by column
and line by line.  Seeds are fixed; every case is small, so a failure message carries the whole input."""
import os
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


def _seeds(n):
    """CI runs seeds 0..n-1; a soak run sets C3R_FUZZ_BASE / C3R_FUZZ_SCALE to walk further seeds (tests/evidence/README.md)."""
    base, scale = int(os.environ.get("C3R_FUZZ_BASE", "0")), int(os.environ.get("C3R_FUZZ_SCALE", "1"))
    return range(base, base + n * scale)


def _rand_cigar(rng, want_q, pads=True):
    """Random op sequence: M/=/X/I/D/N/S/H/P incl. zero-length ops, leading/trailing I or D, runs of D D, I I, N next to
    I or D, pads between insertions (pads = False: a short M in their place).  Returns (cigar string, query length)."""
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
            ops.append((rng.randint(1, 3), "P" if pads else "M"))
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


def _case(seed, phased, pads=True):
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
        cg, qlen = _rand_cigar(rng, 0, pads)
        if rng.random() < 0.1:
            qlen = max(1, qlen - rng.randint(1, 3))          # query shorter than the CIGAR claims
        seq = "".join(rng.choice("ACGTACGTACGTACGTN=RY") for _ in range(qlen))
        flag = (16 if rng.random() < 0.5 else 0) | rng.choice([0] * 14 + [256, 2048, 4, 1024, 512, 8, 1, 3, 65, 131])
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
    for seed in _seeds(120):
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
    assert n_cases == len(_seeds(120)) and n_lines > 300, n_lines
    eng.params = capi.default_params()
    eng.set_params()


@pytest.mark.parametrize("kw", [dict(), dict(channels=30), dict(head_tail=1), dict(splice_padding=1)])
def test_random_cigars_with_the_samtools_1_11_printer(eng, kw):
    """c3r_params_t.mpileup_compat = 1: an I immediately followed by a D shows both on the insertion's column (`C+2TT-1N`: one more D / d,
    D1 / d1 count, a deletion token behind the insertion token).  The generator deals I next to D often (leading, after N, after D)
    and pads next to insertions: samtools >= 1.11 prints those inside the insertion as '*' / '#' (`+3T*T`; c3r_padins_t)."""
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    channels = kw.get("channels", 18)
    okw = {k: bool(v) for k, v in kw.items() if k != "channels"}
    n_lines, n_both, n_padded, n_refused = 0, 0, 0, 0
    eng.load_reads(ReadSet.from_records([]))
    for seed in _seeds(60):
        ref, recs = _case(70000 + 100 * len(kw) + seed, phased=(channels == 30), pads=True)
        rs = ReadSet.from_records(recs)
        eng.params = capi.default_params()
        eng.set_bed(0, None); eng.set_bed(1, None)
        eng.set_params(min_coverage=2, mpileup_compat=1, **kw)
        try:
            got = H.engine_chunk(eng, rs, ref, 1, 1, len(ref))
        except capi.C3RError as e:
            # the documented limit of the pad table (c3r_padins_t: a 64-bit mask per run of I and P ops); the generator reaches it on a few seeds
            assert "more than 64 characters" in str(e), (seed, e)
            n_refused += 1
            continue
        exp = H.oracle_chunk(rs, ref, 1, 1, len(ref), channels=channels, min_coverage=2, mpileup_compat=1, **okw)
        assert got["lines"] == exp["lines"], (seed, recs, H.first_diff(got["lines"], exp["lines"]))
        n_lines += len(exp["lines"])
        n_both += sum(1 for r in exp["rows"] if re.search(r"[+][0-9]+[ACGTNacgtn=RYry*#]+-[0-9]+[Nn]", r.split("\t")[4]))
        n_padded += sum(1 for l in exp["lines"] if re.search(r" I[ACGT][A-Z=]*[*#]", l.split("\t")[4]))
        if seed % 20 == 0:                      # the same reads with the <= 1.10 text: the records are rebuilt when the parameter changes
            eng.set_params(min_coverage=2, mpileup_compat=0, **kw)
            n0 = eng.scan(1, len(ref))
            old = H.oracle_chunk(rs, ref, 1, 1, len(ref), channels=channels, min_coverage=2, **okw)
            assert n0 == len(old["lines"])
    assert n_lines > 150 and n_both > 40 and n_padded > 3 and n_refused <= 0.05 * len(_seeds(60)) + 1, (n_lines, n_both, n_padded, n_refused)
    eng.params = capi.default_params()
    eng.set_params()


def test_samtools_1_11_printer_known_answers_and_pads(eng):
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    ref = "ACGT" * 30
    recs = [dict(pos=10, cigar="6M2I1D6M", seq="GTACGTTTTACGTA") for _ in range(4)] + [dict(pos=10, cigar="13M", seq="GTACGTACGTACG", flag=16) for _ in range(4)]
    rs = ReadSet.from_records(recs)
    out = {}
    eng.load_reads(ReadSet.from_records([]))
    for compat in (0, 1):
        eng.params = capi.default_params()
        eng.set_params(min_coverage=2, mpileup_compat=compat, head_tail=1)        # (head/tail calling: the reads are shorter than a window)
        got = H.engine_chunk(eng, rs, ref, 1, 1, len(ref))
        exp = H.oracle_chunk(rs, ref, 1, 1, len(ref), min_coverage=2, mpileup_compat=compat, head_tail=True)
        assert got["lines"] == exp["lines"] and len(got["lines"]) > 0, H.first_diff(got["lines"], exp["lines"])
        line = [l for l in got["lines"] if l.split("\t")[1] == "16"][0]                # the insertion sits on the sixth aligned base
        out[compat] = line.split("\t")[4]
        tk = got["tokens"][got["tokens"]["indel"] > 0]
        assert (tk["del_after"] == (1 if compat else 0)).all() and len(tk) >= 4
    assert out == {0: "8-ITTT 4 RT 4", 1: "8-ITTT 4 DA 4"}, out      # position 16 gains the deletion
    # samtools >= 1.11 shows the pads of a run of I ops inside the insertion: '*' for a forward read, '#' for a reverse one (--reverse-del).
    # Four forward reads `4M1I1P1I4M` and three reverse ones make two alleles of the same bases; a leading pad decides the channel
    # (key[1] in "ACGTN*": I for the forward reads, i for the reverse ones) and `I P D` shows the deletion behind the padded insertion
    padded = ([dict(pos=10, cigar="6M", seq="GTACGT")] + [dict(pos=12, cigar="4M1I1P1I4M", seq="ACGTTTACGT") for _ in range(4)] +
              [dict(pos=12, cigar="4M1I1P1I4M", seq="ACGTTTACGT", flag=16) for _ in range(3)] + [dict(pos=12, cigar="4M1P2I4M", seq="ACGTTTACGT") for _ in range(2)] +
              [dict(pos=12, cigar="4M1P2I4M", seq="ACGTTTACGT", flag=16)] + [dict(pos=12, cigar="4M2I1P1D3M", seq="ACGTTTCGT") for _ in range(2)])
    rs = ReadSet.from_records(padded)
    eng.params = capi.default_params()
    eng.set_params(min_coverage=2, mpileup_compat=1, head_tail=1)
    got = H.engine_chunk(eng, rs, ref, 1, 1, len(ref))
    exp = H.oracle_chunk(rs, ref, 1, 1, len(ref), min_coverage=2, mpileup_compat=1, head_tail=True)
    assert got["lines"] == exp["lines"] and len(got["lines"]) > 0, H.first_diff(got["lines"], exp["lines"])
    pi = eng.pad_insertions()
    assert len(pi) == 12 and set(pi["total"].tolist()) == {3} and set(pi["pad_mask"].tolist()) == {1, 2, 4} and (pi["n_bases"] == 2).all()
    alt16 = [l for l in got["lines"] if l.split("\t")[1] == "16"][0].split("\t")[4]
    assert alt16 == "13-ITT*T 4 ITT#T 3 IT*TT 2 IT#TT 1 ITTT* 2 DA 2", alt16
    row16 = [r for r in exp["rows"] if r.split("\t")[1] == "16"][0].split("\t")[4]
    assert row16.count("+3T*T") == 4 and row16.count("+3t#t") == 3 and row16.count("+3*TT") == 2 and row16.count("+3#tt") == 1 and row16.count("+3TT*-1N") == 2
    col = eng.columns()
    c16 = col["cols"][16 - col["region_start"]]
    assert (c16[4], c16[5], c16[13], c16[14]) == (8, 4, 4, 3), c16      # I = 4 + 2 + 2, I1 = 4; i = 3 + 1, i1 = 3
    # the C++ decoder builds the same allele text from the packed tokens
    w = synth_weights()
    eng.load_weights(w, 18)
    probs = eng.infer()
    from clair3_rna_amd import decode
    f = [l.split("\t") for l in exp["lines"]]
    assert eng.call_rows("chr20") == decode.vcf_rows("chr20", [int(x[1]) for x in f], [x[2] for x in f], [x[4] for x in f], probs)
    # a run of more than 64 characters is refused (the table describes the pads by a 64-bit mask); the <= 1.10 text takes it
    bad = ReadSet.from_records([dict(pos=10, cigar="6M", seq="GTACGT"), dict(pos=12, cigar="4M40I1P30I4M", seq="ACGT" + "T" * 70 + "ACGT")])
    with pytest.raises(capi.C3RError, match=r"read 1: an insertion with pads \(P ops\) of more than 64 characters"):
        eng.load_reads(bad)
    eng.params = capi.default_params()
    eng.set_params()
    eng.load_reads(bad)
    eng.set_reference(1, ref)
    eng.scan(1, len(ref))
    assert len(eng.pad_insertions()) == 0


def synth_weights():
    from clair3_rna_amd import synth
    w = synth.random_weights(18, seed=4242)
    w[-24 * 129:] *= 6.0
    return w


@pytest.mark.parametrize("mode", ["lbed", "cbed", "both_beds", "sites", "subregion", "deep"])
def test_random_cigars_with_filters_and_regions(eng, mode):
    """The same random read sets through the -l BED, the confident BED, a genotyping site list, a sub-region with a shifted
    reference slice, and at depths that cross the 216 rescale threshold."""
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    n_lines = 0
    for seed in _seeds(60):
        rng = random.Random(7000 + seed)
        ref, recs = _case(50000 + seed, phased=False)
        if mode == "deep":              # replicate the reads: depth 150-400 with identical alleles (I1/D1 multiplicities, rescale)
            rep = rng.randint(6, 9)
            recs = [dict(r) for r in recs for _ in range(rep)]
            recs.sort(key=lambda r: r["pos"])
        rs = ReadSet.from_records(recs)
        L = len(ref)

        def intervals(k):
            out = []
            for _ in range(k):
                a = rng.randint(0, L - 2)
                out.append((a, min(L, a + rng.choice([1, 2, 5, 17, 33, 60, 150]))))
            return out
        lbed = intervals(rng.randint(1, 6)) if mode in ("lbed", "both_beds") else None
        cbed = intervals(rng.randint(1, 6)) if mode in ("cbed", "both_beds") else None
        sites = sorted(set(rng.randint(1, L) for _ in range(rng.randint(1, 25)))) if mode == "sites" else None
        ref_start, a, b = 1, 1, L
        if mode == "subregion":
            a = rng.randint(2, L // 2); b = rng.randint(a, L)
            ref_start = rng.randint(1, max(1, a - 49))   # the slice starts before the region's halo and its windows (the
                                                         # reference fetches ctg_start - 1000: every row and flank is covered)
        if mode == "sites":
            a, b = min(sites), max(sites)
        eng.params = capi.default_params()
        eng.set_bed(0, lbed); eng.set_bed(1, cbed)
        if sites is not None:
            eng.set_sites(sites)
        eng.set_params(min_coverage=2, genotyping_mode=int(sites is not None), head_tail=seed % 2)
        refslice = ref[ref_start - 1:]
        exp = H.oracle_chunk(rs, refslice, ref_start, a, b, lbed=lbed, bed=cbed, sites=sites, min_coverage=2, head_tail=bool(seed % 2))
        got = H.engine_chunk(eng, rs, refslice, ref_start, a, b)
        assert got["lines"] == exp["lines"], (mode, seed, lbed, cbed, sites, (ref_start, a, b), H.first_diff(got["lines"], exp["lines"]))
        assert np.array_equal(got["X"], exp["X"]), (mode, seed)
        n_lines += len(exp["lines"])
    assert n_lines > (20 if mode in ("sites", "lbed", "both_beds") else 200), (mode, n_lines)
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params()


@pytest.mark.parametrize("compat", [0, 1])
def test_random_cigars_decode_rows_cpp_equals_python_and_regions(eng, compat):
    """On the random read sets: (1) c3r_call_rows (C++: tokens -> ordered alt_info -> decode -> row text) equals the Python
    path fed with the ORACLE's alt_info strings; (2) a multi-region scan over random chunk boundaries equals successive
    scans.  Random weights make every genotype class and the decoder's retry loop show up.  compat = 1: the samtools >= 1.11
    text (a deletion token behind an insertion token travels in the packed token stream as del_after)."""
    from clair3_rna_amd import capi, decode, synth
    from clair3_rna_amd.reads import ReadSet
    from oracle import oracle as orc
    w = synth.random_weights(18, seed=4242)
    w[-24 * 129:] *= 6.0                       # sharper output layers: not everything decodes to RefCall
    eng.load_weights(w, 18)
    eng.set_precision("f16x3")
    n_rows, kinds = 0, set()
    for seed in _seeds(60):
        rng = random.Random(9000 + seed)
        ref, recs = _case(80000 + seed, phased=False, pads=True)
        rs = ReadSet.from_records(recs)
        L = len(ref)
        eng.params = capi.default_params()
        eng.set_bed(0, None); eng.set_bed(1, None)
        if seed == _seeds(60)[0]:
            eng.load_reads(ReadSet.from_records([]))
        eng.set_params(min_coverage=2, mpileup_compat=compat)
        got = H.engine_chunk(eng, rs, ref, 1, 1, L)
        exp = H.oracle_chunk(rs, ref, 1, 1, L, min_coverage=2, mpileup_compat=compat)
        assert got["lines"] == exp["lines"]
        if exp["lines"]:
            probs = eng.infer()
            po = orc.forward(w, exp["X"])
            assert np.abs(probs - po).max() < 1e-4
            f = [l.split("\t") for l in exp["lines"]]
            py = decode.vcf_rows("chr20", [int(x[1]) for x in f], [x[2] for x in f], [x[4] for x in f], probs)
            cpp = eng.call_rows("chr20")
            assert cpp == py, (seed, [a for a, b in zip(cpp, py) if a != b][:2], [b for a, b in zip(cpp, py) if a != b][:2])
            n_rows += len(py)
            kinds.update(r.split("\t")[9].split(":")[0] for r in py)
        # random chunking of the same contig
        cuts = sorted(set([1, L] + [rng.randint(2, L - 1) for _ in range(rng.randint(1, 5))]))
        chunks = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
        eng.begin_batch()
        for a, b in chunks:
            eng.scan(a, b)
        eng.end_batch()
        X1, S1, T1 = eng.tensors(), eng.sites(), eng.tokens()
        eng.begin_batch(); eng.scan_regions(chunks); eng.end_batch()
        assert np.array_equal(X1, eng.tensors()) and S1.tobytes() == eng.sites().tobytes() and T1.tobytes() == eng.tokens().tobytes(), (seed, chunks)
    assert n_rows > 1500 and {"0/0", "0/1", "1/1"} <= kinds, (n_rows, kinds)
    eng.params = capi.default_params()
    eng.set_params()


@pytest.mark.parametrize("channels", [18, 30])
def test_mpileup_depth_cap(eng, channels):
    """samtools mpileup -d (default 8000, in force in the reference): htslib discards a read that is not the first pushed
    for its start position while more than max_depth reads are live.  Small caps on replicated random read sets make the
    rule bite; the discarded reads must vanish from counts, coverage, tokens, haplotype channels and skip counts alike, and
    per region (a read may survive in one chunk's scan and not in its neighbour's)."""
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    n_dropped_cases = 0
    for seed in _seeds(40):
        rng = random.Random(4000 + seed)
        ref, recs = _case(60000 + seed, phased=(channels == 30))
        rep = rng.randint(3, 7)
        recs = [dict(r) for r in recs for _ in range(rep)]
        recs.sort(key=lambda r: r["pos"])
        rs = ReadSet.from_records(recs)
        L = len(ref)
        cap = rng.choice([8, 20, 60, 150])
        kw = dict(min_coverage=2, max_depth=cap, splice_padding=seed % 2, head_tail=(seed // 2) % 2)
        eng.params = capi.default_params()
        eng.set_bed(0, None); eng.set_bed(1, None)
        eng.set_params(channels=channels, **kw)
        a = rng.randint(1, L // 3); b = rng.randint(2 * L // 3, L)
        got = H.engine_chunk(eng, rs, ref, 1, a, b)
        exp = H.oracle_chunk(rs, ref, 1, a, b, channels=channels, min_coverage=2, max_depth=cap, splice_padding=bool(seed % 2),
                             head_tail=bool((seed // 2) % 2))
        nocap = H.oracle_chunk(rs, ref, 1, a, b, channels=channels, min_coverage=2, max_depth=0, splice_padding=bool(seed % 2),
                               head_tail=bool((seed // 2) % 2))
        n_dropped_cases += int(exp["lines"] != nocap["lines"])
        assert got["lines"] == exp["lines"], (seed, cap, H.first_diff(got["lines"], exp["lines"]))
        assert np.array_equal(got["X"], exp["X"])
        # the same through a two-region scan (masks are per region)
        mid = (a + b) // 2
        eng.begin_batch(); eng.scan(a, mid); eng.scan(mid, b); eng.end_batch()
        X1, S1 = eng.tensors(), eng.sites()
        eng.begin_batch(); eng.scan_regions([(a, mid), (mid, b)]); eng.end_batch()
        assert np.array_equal(X1, eng.tensors()) and S1.tobytes() == eng.sites().tobytes()
        e1 = H.oracle_chunk(rs, ref, 1, a, mid, channels=channels, min_coverage=2, max_depth=cap, splice_padding=bool(seed % 2), head_tail=bool((seed // 2) % 2))
        e2 = H.oracle_chunk(rs, ref, 1, mid, b, channels=channels, min_coverage=2, max_depth=cap, splice_padding=bool(seed % 2), head_tail=bool((seed // 2) % 2))
        assert [int(l.split("\t")[1]) for l in e1["lines"] + e2["lines"]] == S1["pos"].tolist()
        assert np.array_equal(X1, np.concatenate([e1["X"], e2["X"]])) if len(S1) else True
    assert n_dropped_cases > 25, n_dropped_cases          # the cap changed the output in most cases
    eng.params = capi.default_params()
    eng.set_params()
