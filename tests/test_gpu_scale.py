"""GPU tests at BASELINE.json sizes.  A 3 Mb slice at chr20 read density is compared bit-exactly with the oracle;
the full synthetic chr20 (configs[1]) is checked through size-independent properties: idempotence, agreement of the
13 chunked scans with one whole-contig scan (checksum of per-site checksums), agreement of the +-33 bp chunk overlaps,
probabilities normalised."""
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


def _site_hash(X):
    """order-independent-per-site 64-bit checksum of each [33][C] tensor"""
    w = (np.arange(X.shape[1] * X.shape[2], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1))
    return (X.reshape(len(X), -1).astype(np.int64).astype(np.uint64) * w).sum(axis=1, dtype=np.uint64)


def test_three_megabase_slice_bit_exact_vs_oracle(eng):
    from clair3_rna_amd import capi, synth
    L = 3000000
    ref, rs, info = synth.generate_contig(contig_len=L, seed=synth.SEED, depth=20.0)
    ref = ref.decode()
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params()
    got = H.engine_chunk(eng, rs, ref, 1, 1, L)
    exp = H.oracle_chunk(rs, ref, 1, 1, L)
    assert len(exp["lines"]) > 3000
    assert got["lines"] == exp["lines"], H.first_diff(got["lines"], exp["lines"])
    assert np.array_equal(got["X"], exp["X"])


def test_full_chr20_properties(eng):
    import bench
    from clair3_rna_amd import capi, synth
    ref, rs, info = synth.generate_contig()          # chr20, ~20x, seed 20240422: BASELINE.json configs[1]
    L = len(ref)
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params()
    eng.load_reads(rs)
    eng.set_reference(1, ref)
    w = synth.random_weights(18)
    eng.load_weights(w, 18)
    chunks = bench.chunk_list(L)
    per_chunk, probs_all = [], []
    for (a, b) in chunks:
        n = eng.scan(a, b)
        s = eng.sites()
        X = eng.tensors()
        p = eng.infer()
        per_chunk.append((s["pos"].copy(), _site_hash(X), s["depth"].copy()))
        probs_all.append(p)
        assert np.all(np.diff(s["pos"]) > 0)                      # sortedness within a chunk
    # idempotence: scanning a chunk again gives the same bytes
    a, b = chunks[3]
    eng.scan(a, b)
    again = _site_hash(eng.tensors())
    assert np.array_equal(again, per_chunk[3][1])
    # overlaps: sites in the +-33 bp margins are emitted by both neighbours with identical tensors
    n_overlap = 0
    for i in range(len(chunks) - 1):
        p0, h0, _ = per_chunk[i]
        p1, h1, _ = per_chunk[i + 1]
        common, i0, i1 = np.intersect1d(p0, p1, return_indices=True)
        n_overlap += len(common)
        assert np.array_equal(h0[i0], h1[i1])
    # chunked == whole-contig scan (checksum of checksums over the de-duplicated site set)
    n_all = eng.scan(1, L)
    s_all, h_all = eng.sites(), _site_hash(eng.tensors())
    pos_cat = np.concatenate([c[0] for c in per_chunk])
    h_cat = np.concatenate([c[1] for c in per_chunk])
    upos, first = np.unique(pos_cat, return_index=True)
    assert len(upos) == n_all and np.array_equal(upos, s_all["pos"])
    assert np.array_equal(h_cat[first], h_all)
    assert int(np.bitwise_xor.reduce(h_cat[first])) == int(np.bitwise_xor.reduce(h_all))
    assert len(pos_cat) - len(upos) == n_overlap
    # batch mode: all chunks appended, one network launch per layer == per-chunk results, bit for bit
    eng.begin_batch()
    for (a, b) in chunks:
        eng.scan(a, b)
    Pb = eng.infer()
    sb = eng.sites()
    eng.end_batch()
    assert np.array_equal(sb["pos"], pos_cat)
    assert np.array_equal(Pb, np.concatenate(probs_all))
    tk = eng.tokens()
    assert len(tk) == int(sb["n_tok"].sum()) and int(sb["tok_off"][-1]) + int(sb["n_tok"][-1]) == len(tk)
    # probabilities: two softmaxes per site
    P = np.concatenate(probs_all)
    assert len(P) == len(pos_cat) > 150000
    assert np.allclose(P[:, :21].sum(1), 1, atol=1e-5) and np.allclose(P[:, 21:].sum(1), 1, atol=1e-5)
    assert np.isfinite(P).all() and P.min() >= 0


@pytest.mark.parametrize("channels", [18, 30])
def test_fused_tile_kernel_equals_the_column_store_path(eng, channels, monkeypatch):
    """The plain mode's two implementations — k_fused_tiles (windows straight from LDS, written where they arrive, indexed into
    position order) and the column store with its selection / compaction / gather kernels (C3R_NO_FUSE=1) — give the same sites,
    tensors, raw tensors, tokens and probabilities, byte for byte, also when the scans of a batch are appended."""
    from clair3_rna_amd import capi, synth
    L = 900000
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=501 + channels, depth=40.0, expressed_frac=0.06, intron_lo=80.0, intron_hi=6000.0,
                                       phased=(channels == 30), platform="hifi" if channels == 30 else "ont")
    ref = ref.decode()
    w = synth.random_weights(channels, seed=9)
    chunks = [(1, 300000), (300000, 600000), (600000, L)]

    def run():
        eng.params = capi.default_params()
        eng.set_bed(0, None); eng.set_bed(1, None)
        eng.set_params(channels=channels)
        eng.load_reads(rs); eng.set_reference(1, ref); eng.load_weights(w, channels); eng.set_precision("f16x3")
        eng.scan(1, L)
        one = (eng.tensors().copy(), eng.tensors(rescaled=False).copy(), eng.sites().tobytes(), eng.tokens().tobytes(), eng.infer().copy())
        eng.begin_batch()
        for a, b in chunks:
            eng.scan(a, b)
        eng.end_batch()
        many = (eng.tensors().copy(), eng.sites().tobytes(), eng.tokens().tobytes(), eng.infer().copy())
        return one, many
    fused = run()
    monkeypatch.setenv("C3R_NO_FUSE", "1")
    store = run()
    monkeypatch.delenv("C3R_NO_FUSE")
    assert len(fused[0][0]) > (50 if channels == 30 else 500) and len(fused[1][0]) >= len(fused[0][0])
    for x, y in zip(fused[0] + fused[1], store[0] + store[1]):
        assert np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y


def test_one_full_chunk_of_configs1_is_bit_exact_against_the_oracle(eng):
    """BASELINE.json configs[1] at full size, one of its thirteen 5-Mb chunks end to end: every create_tensor line (position, ref33,
    594 ints, ordered alt_info) and every rescaled tensor of the chunk identical to the oracle's, probabilities within 1e-4.  (The
    whole contig, all 201,945 sites, is the same loop in tests/evidence/full_contig_check.py: ~3 min of oracle time.)"""
    import bench
    from clair3_rna_amd import altinfo, capi, synth
    from oracle import oracle as orc
    ref, rs, _info = synth.generate_contig()
    refs = ref.decode()
    a, b = bench.chunk_list(len(ref))[6]
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params()
    eng.load_reads(capi.pinned_readset(rs))
    eng.set_reference(1, ref)
    w = synth.random_weights(18)
    eng.load_weights(w, 18)
    eng.set_precision("f16x3")
    n = eng.scan(a, b)
    raw, X = eng.tensors(rescaled=False), eng.tensors(rescaled=True)
    sites, toks = eng.sites(), eng.tokens()
    rstart = max(1, a - 1000)
    refslice = refs[rstart - 1:b + 1000].upper()
    lines = altinfo.format_lines("chr20", sites, raw, toks, rs, refslice, rstart)
    rows = orc.mpileup(rs.reads, rs.cigar, rs.seq, "chr20", max(1, a - 33), b + 33)
    exp = orc.create_tensor(rows, "chr20", refslice, rstart, orc.make_params())
    assert n == len(exp) > 10000
    assert lines == exp
    Xo, _ = orc.batch_from_lines(exp, 18)
    assert np.array_equal(X, Xo)
    p = eng.infer()
    assert np.abs(p - orc.forward(w, Xo)).max() < 1e-4


@pytest.mark.parametrize("kw", [dict(), dict(head_tail=1), dict(splice_padding=1), dict(channels=30)])
def test_scan_regions_equals_successive_chunk_scans(eng, kw):
    """c3r_pileup_scan_regions: all chunks of a contig in one set of launches == the chunk-by-chunk scans in batch mode
    (same candidates in the same order, same tensors, sites and tokens), including each chunk's own +-33 bp halo,
    head/tail flush and the in-place splice padding; chunk sizes chosen so that some regions end exactly on a tile edge."""
    from clair3_rna_amd import capi, synth
    L = 1536 * 256 + 7            # chunk size 256*k: region lengths hit multiples of the 256-position tile
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=91, depth=25.0, expressed_frac=0.08, intron_lo=100.0, intron_hi=4000.0,
                                       phased=(kw.get("channels") == 30))
    ref = ref.decode()
    size = 256 * 190 + 33 * 2
    chunks = [(a, min(a + size, L)) for a in range(1, L, size)]
    assert len(chunks) >= 8
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params(**kw)
    eng.load_reads(rs)
    eng.set_reference(1, ref)
    eng.begin_batch()
    n_seq = sum(eng.scan(a, b) for a, b in chunks)
    eng.end_batch()
    X1, S1, T1 = eng.tensors(), eng.sites(), eng.tokens()
    eng.begin_batch()
    n_one = eng.scan_regions(chunks)
    eng.end_batch()
    X2, S2, T2 = eng.tensors(), eng.sites(), eng.tokens()
    assert n_seq == n_one > 500
    assert np.array_equal(X1, X2)
    assert S1.tobytes() == S2.tobytes() and T1.tobytes() == T2.tobytes()
    # region order, then position order inside a region (the halos overlap, so positions may repeat across regions)
    pos = S2["pos"].astype(np.int64)
    assert (np.diff(pos) < 0).sum() <= len(chunks) - 1
    eng.params = capi.default_params()
    eng.set_params()


def test_scan_regions_with_heavily_overlapping_regions(eng):
    """Regions may overlap arbitrarily: the same region several times, and chunks shorter than their +-33 bp halos (three and more
    halos over one indel).  The indel-event scratch is sized from the reads inside every region (it used to assume pairwise
    overlap) — the result must equal the chunk-by-chunk scans."""
    from clair3_rna_amd import capi, synth
    ref, rs, _ = synth.small_case(seed=77, ref_len=40000, n_genes=10, depth=60, err_ins=0.05, err_del=0.06)
    L = len(ref)
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params()
    eng.load_reads(rs)
    eng.set_reference(1, ref)
    whole = (1, L)
    assert eng.scan(1, L) > 50
    mid = int(np.median(eng.sites()["pos"]))                               # where the candidates (and the reads' indels) are
    tiny = [(a, min(a + 19, L)) for a in range(max(1, mid - 2000), min(L, mid + 2000), 20)]   # 200 chunks of 20 bp: ~4 halos over every position
    for regions in ([whole] * 5, tiny, tiny + [whole, whole, whole]):
        eng.begin_batch()
        n_seq = sum(eng.scan(a, b) for a, b in regions)
        eng.end_batch()
        X1, S1, T1 = eng.tensors(), eng.sites(), eng.tokens()
        eng.begin_batch()
        n_one = eng.scan_regions(regions)
        eng.end_batch()
        X2, S2, T2 = eng.tensors(), eng.sites(), eng.tokens()
        assert n_seq == n_one > 20, (n_seq, n_one, len(regions))
        assert np.array_equal(X1, X2)
        assert S1.tobytes() == S2.tobytes() and T1.tobytes() == T2.tobytes()


def test_two_contexts_on_two_streams_like_the_bench(eng):
    """bench.py pipelines passes on two contexts (two HIP streams, one GPU): context B builds tensors while context A's
    persistent LSTM workgroups run.  Interleaved asynchronous use must give what one context gives synchronously."""
    from clair3_rna_amd import capi, synth
    L = 1500000
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=123, depth=20.0, expressed_frac=0.05)
    ref = ref.decode()
    w = synth.random_weights(18, seed=9)
    chunks = [(a, min(a + 400000, L)) for a in range(1, L, 400000)]
    eng.params = capi.default_params()
    eng.set_bed(0, None); eng.set_bed(1, None)
    eng.set_params()
    eng.load_reads(rs); eng.set_reference(1, ref); eng.load_weights(w, 18); eng.set_precision("f16x3")
    eng.begin_batch(); n = eng.scan_regions(chunks); eng.end_batch()
    want = eng.infer().copy()
    assert n > 1000
    e2 = capi.Engine(0)
    try:
        e2.set_params(); e2.load_reads(rs); e2.set_reference(1, ref); e2.load_weights(w, 18); e2.set_precision("f16x3")
        engs, pending, got = [eng, e2], [None, None], []
        for i in range(6):
            e = engs[i & 1]
            if pending[i & 1] is not None:
                got.append(e.fetch_probs(pending[i & 1]).copy())
            e.begin_batch(); m = e.scan_regions(chunks); e.end_batch()
            assert m == n
            e.infer(fetch=False)
            pending[i & 1] = m
        for k in (0, 1):
            got.append(engs[k].fetch_probs(pending[k]).copy())
        assert len(got) == 6
        for g in got:
            assert np.array_equal(g, want)          # same kernels, same data: bitwise equal
    finally:
        e2.close()


def test_network_slices_a_batch_larger_than_one_slice(eng):
    """net_forward runs the network over at most NET_SLICE (2^18) sites at a time; a batch above that must give, site by site,
    what the same windows give in small batches (the kernels are position-independent), in both precisions."""
    from clair3_rna_amd import capi, synth
    rng = np.random.default_rng(5)
    base = rng.integers(-40, 60, size=(4096, 33, 18), dtype=np.int32)
    n = 262144 + 70000 + 37                        # two slices, the second ragged
    X = base[rng.integers(0, len(base), size=n)]
    w = synth.random_weights(18, seed=11)
    eng.load_weights(w, 18)
    for mode in ("f16x3", "f32"):
        eng.set_precision(mode)
        p_small = eng.infer(base)
        idx = rng.integers(0, n, size=3000)
        idx[:6] = [0, 262143, 262144, 262145, n - 1, n - 38]
        # map back: which base window each sampled site is
        p_big = eng.infer(X)
        for i in idx[:50]:
            k = int(np.nonzero((base == X[i]).all(axis=(1, 2)))[0][0])
            assert np.array_equal(p_big[i], p_small[k]), (mode, int(i))
        assert np.isfinite(p_big).all() and np.allclose(p_big.sum(axis=1)[:, None] > 0, True)
    eng.set_precision("f16x3")


def test_bench_two_ranks_control_flow_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one rank per GPU), with the test hook that lets both
    ranks share the box's single GPU and rendezvous over gloo: rank-seeded shards, barrier, max-over-ranks time, summed sites,
    ONE JSON line from rank 0."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, C3R_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--contig_len", "4000000"]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak" and j["value"] > 0
    assert j["unit"] == "sites/s" and j["higher_is_better"] is True and j["vs_baseline"] is None
    per_rank = j["config"]["sites_per_step_per_rank"]
    assert abs(j["value"] - 2 * per_rank / (j["ms_per_step"] / 1e3)) / j["value"] < 1e-3        # whole-job aggregate over both ranks
    assert j["roofline"]["kernel"] == "k_lstm2" and j.get("cpu_baseline") is None      # the CPU baseline is an N = 1 leg


def test_depth_cap_between_separate_deep_loci(eng):
    """mpileup's depth cap on a contig whose expressed loci lie far apart: the rule runs zone by zone (only where more reads than the
    cap can be live at all), each zone entered with the reads that reach into it — the lines and tensors of every chunk, and of chunks
    that cut through a locus, equal the oracle's, which walks every read of the region."""
    from clair3_rna_amd import capi, synth
    L = 400000
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=8117, depth=260.0, expressed_frac=0.03, intron_lo=100.0, intron_hi=4000.0)
    ref = ref.decode()
    n_changed = 0
    for cap in (40, 150):
        eng.params = capi.default_params()
        eng.set_bed(0, None); eng.set_bed(1, None)
        eng.set_params(min_coverage=2, max_depth=cap)
        for a, b in [(1, L), (1, 130000), (130000, 131000), (131000, 270000), (270000, L)]:
            got = H.engine_chunk(eng, rs, ref, 1, a, b)
            exp = H.oracle_chunk(rs, ref, 1, a, b, min_coverage=2, max_depth=cap)
            assert got["lines"] == exp["lines"], (cap, a, b, H.first_diff(got["lines"], exp["lines"]))
            assert np.array_equal(got["X"], exp["X"])
            if (a, b) == (1, L):
                nocap = H.oracle_chunk(rs, ref, 1, a, b, min_coverage=2, max_depth=0)
                n_changed += int(nocap["lines"] != exp["lines"])
                assert len(exp["lines"]) > 50
    assert n_changed == 2                                  # the cap bit at both settings
    eng.params = capi.default_params()
    eng.set_params()


def test_bench_eight_ranks_strong_scaling_on_one_gpu():
    """`bench.py --gpus 8 --scaling strong` as the driver would launch it on an 8-GPU node, with the one-GPU test hook (eight gloo ranks share
    the box's device): the 24 contigs are dealt to eight ranks without loss or overlap, the summed sites equal the N = 1 run's over the
    same contigs, and eight ranks' resident contexts together stay far below one device's memory (each rank of a real node has a device
    to itself).  Readiness for the 8-GPU curve nobody has been able to measure (run_clair3_rna:441-449,681-706)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(n):
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ, C3R_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        tail = ["bench.py", "--gpus", str(n), "--scaling", "strong", "--genome_scale", "0.02", "--steps", "1", "--warmup", "0"]
        cmd = ([sys.executable] if n == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                                                "--master-port", str(port)]) + tail
        r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.split("\n") if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])
    j8, j1 = run(8), run(1)
    c8, c1 = j8["config"], j1["config"]
    assert j8["n_gpus"] == 8 and j8["scaling"] == "strong" and j1["n_gpus"] == 1
    assert sum(c8["contigs_per_rank"]) == 24 and min(c8["contigs_per_rank"]) >= 1 and len(c8["contigs_per_rank"]) == 8
    assert sum(c8["bp_per_rank"]) == c8["genome_bp"] == c1["genome_bp"]
    assert c8["sites_per_step"] == c1["sites_per_step"] > 0                   # the same contigs, whoever holds them
    assert c8["lpt_imbalance"] < 1.35 and c8["read_imbalance"] < 1.5
    assert 0 < c8["hbm_in_use_bytes_max_rank"] < 64e9                         # eight ranks' contexts on ONE device
    assert c8["pinned_input_bytes_max_rank"] > 0
