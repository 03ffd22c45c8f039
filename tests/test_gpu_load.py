"""c3r_load_reads on the device (csrc/reads_kernels.hpp): validation errors, filter changes, pinned and pageable callers, and the
host copies the rare paths fetch on demand.  The CIGAR semantics themselves are covered by the parity and fuzz suites, which all
enter through this call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from clair3_rna_amd import capi
    e = capi.Engine(0)
    yield e
    e.close()


def _lines(e, rs, ref, **prm):
    from clair3_rna_amd import altinfo
    e.params = __import__("clair3_rna_amd.capi", fromlist=["x"]).default_params()
    e.set_params(**prm)
    e.set_bed(0, None); e.set_bed(1, None)
    e.load_reads(rs); e.set_reference(1, ref)
    e.scan(1, len(ref))
    return altinfo.format_lines("chr20", e.sites(), e.tensors(rescaled=False), e.tokens(), rs, ref, 1)


def test_validation_errors_name_the_first_bad_read(eng):
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    recs = [dict(pos=100 + 10 * k, cigar="50M", seq="ACGT" * 12 + "AC") for k in range(700)]

    def rs_with(i, **kw):
        rs = ReadSet.from_records(recs)
        for k, v in kw.items():
            rs.reads[k][i] = v
        return rs
    eng.set_params()
    with pytest.raises(capi.C3RError, match=r"sorted by pos \(read 301\)"):
        eng.load_reads(rs_with(301, pos=5))
    with pytest.raises(capi.C3RError, match=r"cigar range of read 655 out of bounds"):
        eng.load_reads(rs_with(655, n_cigar=99))
    with pytest.raises(capi.C3RError, match=r"seq range of read 12 out of bounds"):
        eng.load_reads(rs_with(12, l_seq=10 ** 6))
    bad = ReadSet.from_records(recs)
    bad.cigar[40] = (50 << 4) | 11                               # op code 11 does not exist
    bad.cigar[400] = (50 << 4) | 12
    with pytest.raises(capi.C3RError, match=r"bad cigar op in read 40"):      # the smallest failing read is reported
        eng.load_reads(bad)
    far = ReadSet.from_records([dict(pos=2 ** 31 - 40, cigar="50M", seq="A" * 50)])
    with pytest.raises(capi.C3RError, match="beyond 2\\^31"):
        eng.load_reads(far)
    long_ops = ReadSet.from_records([dict(pos=10, cigar="1M1I" * 33000 + "1M", seq="A" * 66001)])
    with pytest.raises(capi.C3RError, match="65535"):
        eng.load_reads(long_ops)
    merged = ReadSet.from_records([dict(pos=10, cigar="200000000N" + "100000000N", seq="")])
    with pytest.raises(capi.C3RError, match="too long"):
        eng.load_reads(merged)
    # a failed load leaves an empty, usable context
    eng.set_reference(1, "ACGT" * 100)
    assert eng.scan(1, 300) == 0
    eng.load_reads(ReadSet.from_records(recs))
    eng.set_reference(1, "ACGT" * 2000)
    eng.scan(1, 7000)


def test_pinned_and_pageable_records_give_the_same_tables(eng):
    from clair3_rna_amd import capi, synth
    ref, rs, _ = synth.small_case(seed=11, ref_len=60000, n_genes=10, depth=25)
    a = _lines(eng, rs, ref)
    b = _lines(eng, capi.pinned_readset(rs), ref)
    assert a == b and len(a) > 50


def test_new_filters_rebuild_the_tables_on_the_device(eng):
    """c3r_set_params with another --minMQ / --excl-flags re-derives pass flags, prefix maxima and bucket index without a reload."""
    from clair3_rna_amd import capi, synth
    ref, rs, _ = synth.small_case(seed=12, ref_len=50000, n_genes=8, depth=30)
    rs.reads["mapq"][::3] = 17
    rs.reads["flag"][1::5] |= 1024
    want = {}
    for prm in (dict(min_mq=5), dict(min_mq=20), dict(min_mq=5, excl_flags=2316 | 1024)):
        e2 = capi.Engine(0)
        want[tuple(sorted(prm.items()))] = _lines(e2, rs, ref, **prm)
        e2.close()
    assert len(set(map(tuple, want.values()))) == 3
    eng.params = capi.default_params()
    eng.set_params(); eng.set_bed(0, None); eng.set_bed(1, None)
    eng.load_reads(rs); eng.set_reference(1, ref)
    from clair3_rna_amd import altinfo
    for prm in (dict(min_mq=20), dict(min_mq=5, excl_flags=2316 | 1024), dict(min_mq=5, excl_flags=2316)):
        eng.set_params(**prm)
        eng.scan(1, len(ref))
        got = altinfo.format_lines("chr20", eng.sites(), eng.tensors(rescaled=False), eng.tokens(), rs, ref, 1)
        key = dict(min_mq=prm["min_mq"])
        if prm.get("excl_flags", 2316) != 2316:
            key["excl_flags"] = prm["excl_flags"]
        assert got == want[tuple(sorted(key.items()))], prm


def test_empty_and_filtered_out_inputs(eng):
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    eng.params = capi.default_params()
    eng.set_params()
    eng.load_reads(ReadSet.from_records([]))
    eng.set_reference(1, "ACGT" * 500)
    assert eng.scan(1, 1500) == 0
    # reads that hold no aligned base at all (soft clips, inserts, ref-skips only) and unmapped records with pos -1
    rs = ReadSet.from_records([dict(pos=-1, cigar="", seq="ACGT", flag=4), dict(pos=50, cigar="20S", seq="A" * 20), dict(pos=60, cigar="5I", seq="AAAAA"),
                               dict(pos=70, cigar="100N", seq=""), dict(pos=80, cigar="10S100N5I", seq="A" * 15)])
    eng.load_reads(rs)
    assert eng.scan(1, 1500) == 0
    cols = eng.columns()
    assert not cols["depth"].any()          # (the insertion after the ref-skip still lands on the intron's last column: channel i only)


def test_row_snapshots_survive_the_next_contigs(eng):
    """c3r_rows_begin detaches a batch's decode inputs from the context: the engine loads other reads and another reference (three
    times over: the reference buffers rotate under the snapshots' user counts), and the snapshots — decoded afterwards, on other
    threads — still give the rows c3r_call_rows gave for their contig."""
    import threading
    from clair3_rna_amd import capi, synth
    w = synth.random_weights(18, seed=77)
    cases = [synth.small_case(seed=300 + k, ref_len=30000 + 4000 * k, n_genes=5 + k, depth=18) for k in range(3)]      # (a context keeps three reference buffers: at most three undecoded snapshots)
    eng.params = capi.default_params()
    eng.set_params(); eng.set_bed(0, None); eng.set_bed(1, None)
    eng.load_weights(w, 18); eng.set_precision("f16x3")
    want, snaps = [], []
    for ref, rs, _ in cases:                                   # the one-call form, contig after contig
        eng.load_reads(rs); eng.set_reference(1, ref)
        assert eng.scan(1, len(ref)) > 20
        eng.infer(fetch=False)
        want.append(eng.call_rows_text("chr20", qual=2, show_ref=True))
    for ref, rs, _ in cases:                                   # snapshots taken, engine moves on, nothing decoded yet
        eng.load_reads(rs); eng.set_reference(1, ref)
        eng.scan(1, len(ref))
        eng.infer(fetch=False)
        snaps.append(eng.rows_begin())
    got = [None] * len(snaps)

    def work(k):
        got[k] = snaps[k].decode("chr20", qual=2, show_ref=True)
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(snaps))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert got == want and all(n > 20 for _t, n in got)
    # every snapshot released its reference buffer: the context takes new references without waiting
    for ref, rs, _ in cases[:3]:
        eng.set_reference(1, ref)


def test_reference_view_equals_the_copied_reference(eng):
    """c3r_set_reference_view (the caller's upper-cased array used in place, no host copy) gives the sites, tensors and rows of
    c3r_set_reference; more snapshots than a context has reference buffers may be outstanding, each keeping its own array alive
    after the caller has dropped it."""
    import gc
    import numpy as np
    from clair3_rna_amd import capi, synth
    w = synth.random_weights(18, seed=78)
    cases = [synth.small_case(seed=400 + k, ref_len=26000 + 3000 * k, n_genes=5 + k % 3, depth=16) for k in range(5)]
    eng.params = capi.default_params()
    eng.set_params(); eng.set_bed(0, None); eng.set_bed(1, None)
    eng.load_weights(w, 18); eng.set_precision("f16x3")
    want, snaps = [], []
    for ref, rs, _ in cases:
        eng.load_reads(rs); eng.set_reference(1, ref.lower())          # (the copying entry point upper-cases)
        assert eng.scan(1, len(ref)) > 20
        tens = eng.tensors().copy()
        eng.infer(fetch=False)
        want.append((eng.sites().tobytes(), tens.tobytes(), eng.call_rows_text("chr20", qual=2, show_ref=True)))
    for k, (ref, rs, _) in enumerate(cases):
        arr = np.frombuffer(ref.upper().encode(), dtype=np.uint8).copy()
        eng.load_reads(rs); eng.set_reference(1, arr, upper_view=True)
        del arr
        gc.collect()
        assert eng.scan(1, len(ref)) > 20
        assert (eng.sites().tobytes(), eng.tensors().tobytes()) == want[k][:2]
        eng.infer(fetch=False)
        snaps.append(eng.rows_begin())
    eng.set_reference(1, cases[0][0])                                 # the engine's own hold on the last array is gone too
    gc.collect()
    junk = [np.full(40000, 78, np.uint8) for _ in range(64)]           # (freed arrays would be reused by these)
    got = [sn.decode("chr20", qual=2, show_ref=True) for sn in snaps]
    assert got == [x[2] for x in want] and junk


def test_reserve_sizes_the_network_before_the_first_infer():
    """c3r_reserve needs the weights (the buffer shapes follow the precision in use), is idempotent, and leaves c3r_infer's results
    untouched; a second context of the process takes over the first one's released layer-1 block."""
    import numpy as np
    from clair3_rna_amd import capi, synth
    ref, rs, _ = synth.small_case(seed=515, ref_len=30000, n_genes=6, depth=18)
    w = synth.random_weights(18, seed=79)
    probs = []
    for k in range(2):
        e = capi.Engine(0)
        try:
            if k == 0:
                with pytest.raises(capi.C3RError):
                    e.reserve(1000)
            e.load_weights(w, 18); e.set_precision("f16x3")
            if k == 1:
                e.reserve(300000); e.reserve(300000); e.reserve(0)
            e.load_reads(rs); e.set_reference(1, ref)
            assert e.scan(1, len(ref)) > 20
            probs.append(np.array(e.infer(), copy=True))
        finally:
            e.close()
    assert np.array_equal(probs[0], probs[1])


def test_large_pageable_arrays_through_the_staging_buffers(eng):
    """Copies of 1 MB and more from / to ordinary host memory go through the context's two page-locked 8-MB buffers (c3r_lib.hip, big_h2d /
    big_d2h): the full synthetic chr20 — 14 MB of CIGARs and 19.7 MB of bases, three chunks each way round — loaded from ordinary numpy
    arrays and from c3r_host_alloc memory (the direct path) gives the same sites, tokens and tensors, and the row snapshot (sites, packed
    tokens, probabilities, read headers and bases copied DOWN through the same buffers) decodes to the same rows as the in-context decode."""
    import bench
    from clair3_rna_amd import capi, synth
    ref, rs, _info = synth.generate_contig()
    assert rs.seq.nbytes > 2 * (8 << 20) and rs.cigar.nbytes > (8 << 20)
    chunks = bench.chunk_list(len(ref))[5:8]
    out = []
    for records in (rs, capi.pinned_readset(rs)):
        eng.params = capi.default_params()
        eng.set_bed(0, None); eng.set_bed(1, None)
        eng.set_params()
        eng.load_reads(records)
        eng.set_reference(1, ref)
        eng.begin_batch()
        n = eng.scan_regions(chunks)
        eng.end_batch()
        out.append((n, eng.sites().tobytes(), eng.tokens().tobytes(), eng.tensors().tobytes()))
    assert out[0][0] == out[1][0] > 30000
    assert out[0] == out[1]
    eng.load_weights(synth.random_weights(18), 18)
    eng.set_precision("f16x3")
    eng.infer(fetch=False)
    rows_ctx = eng.call_rows_text("chr20")[0]
    snap = eng.rows_begin()
    rows_snap = snap.decode("chr20")[0]
    snap.free()
    assert bytes(rows_ctx) == bytes(rows_snap) and len(rows_ctx) > 1 << 20


def test_a_two_stream_context_computes_the_same(eng, monkeypatch):
    """C3R_TWO_STREAMS=1: the context's own stream at the device's highest priority, the network's kernels on a
    second stream chained by events inside c3r_infer.  Same tensors, same probabilities, same rows as the one-stream context, also when
    passes follow each other without a host wait in between (the next batch's tensors must not overtake the running network)."""
    from clair3_rna_amd import capi, synth
    ref, rs, _ = synth.small_case(seed=23, ref_len=80000, n_genes=12, depth=30)
    w = synth.random_weights(18)
    monkeypatch.setenv("C3R_TWO_STREAMS", "1")
    two = capi.Engine(0)
    monkeypatch.delenv("C3R_TWO_STREAMS")
    try:
        res = []
        for e in (eng, two):
            e.params = capi.default_params()
            e.set_bed(0, None); e.set_bed(1, None)
            e.set_params()
            e.load_weights(w, 18); e.set_precision("f16x3")
            got = []
            for rep in range(3):
                e.load_reads(rs); e.set_reference(1, ref)
                n = e.scan(1, len(ref))
                e.infer(fetch=False)
                if rep == 2:
                    got = [n, e.fetch_probs(n).tobytes(), e.tensors().tobytes(), bytes(e.call_rows_text("chr20")[0])]
            res.append(got)
        assert res[0][0] > 100 and res[0] == res[1]
    finally:
        two.close()


def test_long_reads_far_apart_take_the_recount_path_of_the_second_pass(eng):
    """k_prep<false> hands each workgroup's bin table to k_prep<true> through a 4-KB slab (510 bins); sixteen long reads that share no bin
    need more (16 x 41 bins), so the second pass counts theirs again.  The profiler's statistics say that the path ran; lines and tensors
    equal the oracle's."""
    import random
    from clair3_rna_amd import capi
    from clair3_rna_amd.reads import ReadSet
    from tests import helpers as H
    rng = random.Random(11)
    pitch, rlen, n_loci, n_lone = 4000, 1300, 12, 48
    L = (n_loci + n_lone) * pitch + 3000
    ref = "".join(rng.choice("ACGT") for _ in range(L))
    recs = []
    for i in range(n_loci):                          # loci of five reads with the same mismatches: candidates above the coverage gate
        for rep in range(5):
            p0 = 500 + i * pitch + rep * 3
            seq = list(ref[p0:p0 + rlen])
            for q in range(60 - rep * 3, rlen, 97):
                seq[q] = "A" if ref[p0 + q] != "A" else "C"
            recs.append(dict(pos=p0, cigar="%dM" % rlen, seq="".join(seq), flag=16 * (rep % 2)))
    for i in range(n_lone):                          # ... and lone long reads, one per 4 kb: sixteen of them make a workgroup of 16 x 41 bins
        p0 = 500 + (n_loci + i) * pitch
        recs.append(dict(pos=p0, cigar="%dM" % rlen, seq=ref[p0:p0 + rlen], flag=0))
    recs.sort(key=lambda r: r["pos"])
    rs = ReadSet.from_records(recs)
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params(min_coverage=2)
    eng.set_profiling(True); eng.reset_kernel_stats()
    got = H.engine_chunk(eng, rs, ref, 1, 1, len(ref))
    ks = eng.kernel_stats()
    eng.set_profiling(False)
    assert ks.get("k_prep_recount_workgroups", {}).get("launches", 0) >= 2, sorted(ks)
    exp = H.oracle_chunk(rs, ref, 1, 1, len(ref), min_coverage=2)
    assert len(exp["lines"]) > 100
    assert got["lines"] == exp["lines"], H.first_diff(got["lines"], exp["lines"])
    assert np.array_equal(got["X"], exp["X"])
    eng.params = capi.default_params()
    eng.set_params()
