"""Host logic of the whole-sample driver (clair3_rna_amd/call_sample.py): contig / chunk planning of run_clair3_rna:310-449,
split_extend_bed (:268-296) and the in-memory merge against golden G6 (outputs of the reference's own sort_vcf).  CPU-only."""
import gzip
import json
import os

import pytest

from clair3_rna_amd import call_sample, io, sort_vcf

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(gzip.open(os.path.join(HERE, "golden", "g6_sortvcf.json.gz"), "rt"))


@pytest.mark.parametrize("native", [True, False], ids=["c3r_vcf_merge", "python"])
@pytest.mark.parametrize("ci", range(len(CASES) - 1))
def test_sample_merger_matches_reference_sort_vcf(ci, native, tmp_path):
    """Rows handed over per contig (chunk files concatenated in listing order, as the driver's decode stage returns them)."""
    case = CASES[ci]
    a = case["args"]
    table = None
    if a.get("tag"):
        table = {}
        if case["rediportal"] is not None:
            redi = tmp_path / "redi.txt.gz"
            with gzip.open(str(redi), "wt") as f:
                f.write(case["rediportal"])
            tags = set(a["filter_tag"].split(":")) if a.get("filter_tag") else None
            table = sort_vcf.load_rediportal(str(redi), case["contigs"], tags)
    names = [n for n in case["listing"] if n.startswith("pileup") and n.endswith(".vcf")]
    header = []
    per = {}
    for contig in sort_vcf._contig_order(case["contigs"], case["contigs"]):
        rows = []
        for n in (n for n in names if contig in n):
            for row in case["files"][n].splitlines(True):
                if row[0] == "#":
                    if row not in header:
                        header.append(row)
                    continue
                if row.split(None, 1)[0] != contig:
                    break
                rows.append(row)
        per[contig] = "".join(rows)
    out, out_nt = str(tmp_path / "o.vcf"), str(tmp_path / "o_nt.vcf")
    m = sort_vcf.SampleMerger(out, "".join(header), a["qual"], a["show_ref"], table, out_nt, native=native)
    for contig in sort_vcf._contig_order(case["contigs"], case["contigs"]):
        m.add_contig(contig, per[contig].encode())
    m.close(log=lambda *_: None)
    assert open(out).read() == case["out"]
    if case["out_no_tagging"] is not None:
        assert open(out_nt).read() == case["out_no_tagging"]


@pytest.fixture
def ref_fa(tmp_path):
    fa = str(tmp_path / "ref.fa")
    names = ["chr2", "chr1", "chrX", "chrUn_1", "20", "chrM", "HLA-A"]
    io.write_fasta(fa, [(n, "ACGT" * (250 * (i + 1))) for i, n in enumerate(names)])     # 1000, 2000, ... bp
    return fa


def test_plan_default_is_major_contigs_in_reference_order(ref_fa):
    contigs, chunks = call_sample.plan_chunks(ref_fa, chunk_size=900)
    assert contigs == ["chr1", "chr2", "chrX", "20"]                       # chr1..22,X,Y first, then 1..22,X,Y
    assert chunks == {"chr2": 2, "chr1": 3, "chrX": 4, "20": 6}            # ceil(len / chunk_size)
    assert call_sample.plan_chunks(ref_fa, chunk_size=1000)[1]["chr2"] == 1
    assert call_sample.plan_chunks(ref_fa, chunk_size=900, chunk_num=7)[1] == {"chr2": 7, "chr1": 7, "chrX": 7, "20": 7}


def test_plan_include_all_ctg_name_bed_and_vcf(ref_fa, tmp_path):
    contigs, _ = call_sample.plan_chunks(ref_fa, include_all_ctgs=True)
    assert contigs[:4] == ["chr1", "chr2", "chrX", "20"] and sorted(contigs[4:]) == ["HLA-A", "chrM", "chrUn_1"]
    assert call_sample.plan_chunks(ref_fa, ctg_name="chrM,chr2,nope")[0] == ["chr2", "chrM"]
    bed = str(tmp_path / "a.bed")
    open(bed, "w").write("#h\nchrUn_1\t10\t500\nchr1\t0\t40\nchr1\t300\t300\nabsent\t1\t2\n")
    assert call_sample.plan_chunks(ref_fa, bed_fn=bed)[0] == ["chr1", "chrUn_1"]            # BED contigs, major or not
    assert call_sample.plan_chunks(ref_fa, bed_fn=bed, ctg_name="chr1,chr2")[0] == ["chr1"]  # --ctg_name intersects the BED
    vcf = str(tmp_path / "k.vcf")
    open(vcf, "w").write("##x\n#CHROM\tPOS\nchrX\t5\t.\tA\tC\nchrM\t9\t.\tA\tC\n")
    assert call_sample.plan_chunks(ref_fa, vcf_fn=vcf)[0] == ["chrX", "chrM"]
    assert call_sample.plan_chunks(ref_fa, vcf_fn=vcf, bed_fn=bed)[0] == ["chr1", "chrUn_1"]  # union; the BED filters the .fai rows
    d = str(tmp_path / "split")
    call_sample.split_extend_bed(bed, d, {"chr1", "chrUn_1"})
    assert sorted(os.listdir(d)) == ["chr1", "chrUn_1"]
    assert open(os.path.join(d, "chr1")).read() == "chr1 0 73\nchr1 267 333"
    assert io.read_bed(os.path.join(d, "chr1"), "chr1") == ([(0, 73), (267, 333)], 0, 333)
    bad = str(tmp_path / "bad.bed")
    open(bad, "w").write("chr1\t50\t10\n")
    with pytest.raises(SystemExit):
        call_sample.split_extend_bed(bad, d, None)


def _fake_sample(tmp):
    """FASTA + BAM (+ .bai) + weights for the stand-in engine runs."""
    import numpy as np
    from clair3_rna_amd import bam, bamio, synth
    spec = [("chr1", 40000, 3), ("chr2", 26000, 5), ("chr3", 30000, 6), ("chrX", 22000, 7), ("chr9", 9000, 11)]
    contigs, reads = [], {}
    for name, L, seed in spec:
        ref, rs, _ = synth.small_case(seed=seed, ref_len=L, n_genes=max(3, L // 5000), depth=8)
        contigs.append((name, ref))
        if name != "chr9":
            reads[name] = rs
    fa, bm, wfn = os.path.join(tmp, "ref.fa"), os.path.join(tmp, "in.bam"), os.path.join(tmp, "model")
    io.write_fasta(fa, contigs)
    bam.write_bam(bm, [(n, len(r)) for n, r in contigs], reads)
    bamio.index_build(bm)
    np.save(wfn + ".c3rw.npy", np.ones(8, dtype=np.float32))
    return fa, bm, wfn


def _launch(tmp, out, fa, bm, wfn, extra, world=1, env_extra=None, expect_fail=False, compress=False):
    import socket
    import subprocess
    import sys
    launcher = os.path.join(HERE, "support", "fake_engine_sample.py")
    argv = [sys.executable, launcher, "--bam_fn", bm, "--ref_fn", fa, "--output_dir", out, "--pileup_model_path", wfn, "--chunk_size", "9000",
            "--gpu_id", "0"] + ([] if compress else ["--no_compress"]) + extra
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ)
        if world > 1:
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        else:
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
        env.update(env_extra or {})
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    if expect_fail:
        return [p.returncode for p in procs], outs
    assert all(p.returncode == 0 for p in procs), ([p.returncode for p in procs], outs)
    if compress:
        assert os.path.exists(os.path.join(out, "output.vcf.gz.tbi")) and not os.path.exists(os.path.join(out, "output.vcf"))
        text = gzip.open(os.path.join(out, "output.vcf.gz"), "rt").read()
    else:
        text = open(os.path.join(out, "output.vcf")).read()
    return [r for r in text.split("\n") if not r.startswith("##cmdline=")]


def test_driver_orchestration_single_and_two_ranks_gloo(tmp_path):
    """call_sample end to end with a stand-in engine (tests/support/fake_engine_sample.py): planning, threads, merge order and
    seam duplicates in one process; then the same job as two ranks over gloo (world_size 2, on CPU) — contigs dealt by LPT, parts
    under tmp/parts, rank 0 assembling — must write the same file.  Also three contexts and one fetch thread."""
    tmp = str(tmp_path)
    fa, bm, wfn = _fake_sample(tmp)
    one = _launch(tmp, os.path.join(tmp, "one"), fa, bm, wfn, ["--print_ref_calls"])
    rec = [r.split("\t") for r in one if r and r[0] != "#"]
    assert len(rec) > 100 and [c for c in dict.fromkeys(r[0] for r in rec)] == ["chr1", "chr2", "chr3", "chrX"]      # chr9: no reads
    for ctg in ("chr1", "chr2", "chr3", "chrX"):
        pos = [int(r[1]) for r in rec if r[0] == ctg]
        assert pos == sorted(set(pos))                                  # ordered, seam duplicates resolved
    # at a seam the later chunk's record survives: its sample field carries the chunk number
    seams = [r for r in rec if r[0] == "chr1" and int(r[1]) in (8000, 16000, 24000, 32000)]
    assert seams and all(r[9].endswith(":%d" % (int(r[1]) // 8000)) for r in seams)
    assert open(os.path.join(tmp, "one", "tmp", "CONTIGS")).read().split("\n") == ["chr1", "chr2", "chr3", "chrX"]
    two = _launch(tmp, os.path.join(tmp, "two"), fa, bm, wfn, ["--print_ref_calls"], world=2)
    assert two == one
    parts = sorted(n for n in os.listdir(os.path.join(tmp, "two", "tmp", "parts")) if n.endswith(".json"))
    called = [json.load(open(os.path.join(tmp, "two", "tmp", "parts", n)))["called"] for n in parts]
    assert parts == ["rank0.json", "rank1.json"] and called[0] and called[1] and not set(called[0]) & set(called[1])
    assert sorted(called[0] + called[1]) == ["chr1", "chr2", "chr3", "chrX"]
    three_ctx = _launch(tmp, os.path.join(tmp, "ctx3"), fa, bm, wfn, ["--print_ref_calls", "--contexts", "3", "--fetch_threads", "1"])
    assert three_ctx == one
    # the detached path of the real engine (row snapshots decoded, merged and — one process — compressed on the worker pool)
    snap = {"C3R_FAKE_SNAPSHOTS": "1"}
    assert _launch(tmp, os.path.join(tmp, "snap1"), fa, bm, wfn, ["--print_ref_calls"], env_extra=snap) == one
    assert _launch(tmp, os.path.join(tmp, "snap2"), fa, bm, wfn, ["--print_ref_calls"], world=2, env_extra=snap) == one
    assert _launch(tmp, os.path.join(tmp, "snap1z"), fa, bm, wfn, ["--print_ref_calls"], env_extra=snap, compress=True) == one
    assert _launch(tmp, os.path.join(tmp, "snap2z"), fa, bm, wfn, ["--print_ref_calls"], world=2, env_extra=snap, compress=True) == one
    # compressed output: streamed pieces in one process; under two ranks the parts compressed on a pool at rank 0 and appended in order
    assert _launch(tmp, os.path.join(tmp, "gz1"), fa, bm, wfn, ["--print_ref_calls"], compress=True) == one
    assert _launch(tmp, os.path.join(tmp, "gz2"), fa, bm, wfn, ["--print_ref_calls"], world=2, compress=True) == one
    no_ref = _launch(tmp, os.path.join(tmp, "noref"), fa, bm, wfn, ["--qual", "10"], world=2)
    kept = [r.split("\t") for r in no_ref if r and r[0] != "#"]
    assert kept and all(r[4] != "." for r in kept) and all((r[6] == "LowQual") == (float(r[5]) <= 10) for r in kept)


def test_two_ranks_unindexed_bam_and_a_failing_rank(tmp_path):
    """(1) An unindexed BAM under two ranks: rank 0 alone builds tmp/input.bam(.bai) (renamed into place), the other rank opens it
    after the rendezvous, and the result equals the indexed single-process run.  (2) A contig whose device stage raises on one
    rank: that rank still reaches the rendezvous, and BOTH ranks exit non-zero promptly instead of one of them waiting in a
    barrier for the gloo timeout."""
    import time
    tmp = str(tmp_path)
    fa, bm, wfn = _fake_sample(tmp)
    one = _launch(tmp, os.path.join(tmp, "one"), fa, bm, wfn, ["--print_ref_calls"])
    os.remove(bm + ".bai")
    two = _launch(tmp, os.path.join(tmp, "two"), fa, bm, wfn, ["--print_ref_calls"], world=2)
    assert two == one
    made = sorted(os.listdir(os.path.join(tmp, "two", "tmp")))
    assert "input.bam" in made and "input.bam.bai" in made and not [n for n in made if ".tmp" in n]
    t0 = time.time()
    codes, outs = _launch(tmp, os.path.join(tmp, "bad"), fa, bm, wfn, ["--print_ref_calls"], world=2,
                          env_extra={"C3R_FAKE_FAIL_LEN": "30000"}, expect_fail=True)          # chr3
    assert all(c != 0 for c in codes), (codes, outs)
    assert time.time() - t0 < 120
    assert any("injected failure" in o for o in outs) and any("another rank failed" in o for o in outs)
    codes, outs = _launch(tmp, os.path.join(tmp, "bad1"), fa, bm, wfn, ["--print_ref_calls"], env_extra={"C3R_FAKE_FAIL_LEN": "30000"},
                          expect_fail=True)
    assert codes == [1] and "injected failure" in outs[0]


def test_fetch_reference_slices_like_faidx(tmp_path):
    """io.fetch_reference: 1-based inclusive slices across line ends, clamped to the contig, upper-cased like the reference's
    reference_sequence_from (shared/utils.py:168-194) — or raw bytes (case kept) for c3r_set_reference, which upper-cases itself."""
    fa = str(tmp_path / "r.fa")
    seq1 = ("ACGTacgtNNry" * 37)[:431]
    seq2 = "G" * 60 + "t" * 61
    io.write_fasta(fa, [("c1", seq1), ("c2", seq2)], width=60)
    for a, b in ((1, 431), (1, 1), (60, 61), (59, 121), (430, 431), (100, 99), (-5, 10), (400, 9999)):
        lo, hi = max(1, a), min(431, b)
        want = seq1[lo - 1:hi] if hi >= lo else ""
        assert io.fetch_reference(fa, "c1", a, b) == want.upper()
        assert io.fetch_reference(fa, "c1", a, b, raw=True) == want.encode()
    assert io.fetch_reference(fa, "c2", 55, 70) == "GGGGGGTTTTTTTTTT"
    assert io.fetch_reference(fa, "c2", 1, 10 ** 9, raw=True) == seq2.encode()
    with pytest.raises(KeyError):
        io.fetch_reference(fa, "nope", 1, 2)


def test_a_contig_fetched_as_position_ranges_equals_the_whole_fetch(tmp_path, monkeypatch):
    """call_sample's fetcher splits a long contig into position ranges fetched on several threads and joins them (reads that start
    before a range belong to the range before; offsets moved): the joined ReadSet must be the whole-contig fetch — with long
    ref-skips crossing the range edges, ranges without a read, and every number of ranges."""
    import numpy as np
    from clair3_rna_amd import bam, bamio, call_sample, synth
    L = 900000
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=21, depth=30.0, expressed_frac=0.05, intron_hi=120000.0)
    bm = str(tmp_path / "x.bam")
    bam.write_bam(bm, [("chr1", L), ("chr2", 1000)], {"chr1": rs})
    bamio.index_build(bm)
    fa = str(tmp_path / "r.fa")
    io.write_fasta(fa, [("chr1", ref.decode()), ("chr2", "A" * 1000)])
    f = call_sample._Fetcher(bm, fa)
    whole = f.part("chr1", 0, None)
    assert len(whole.reads) > 500 and np.array_equal(whole.reads["pos"], rs.reads["pos"])
    end = whole.reads["pos"].astype(np.int64) + 1
    for n_parts, part_bp in ((2, 400000), (5, 100000), (8, 30000), (64, 1000)):
        monkeypatch.setenv("C3R_FETCH_PART_BP", str(part_bp))
        ranges = f.plan(L, n_parts)
        assert len(ranges) == min(n_parts, L // part_bp) and ranges[0][0] == 0 and ranges[-1][1] is None
        assert all(ranges[k][1] == ranges[k + 1][0] for k in range(len(ranges) - 1))
        parts = [f.part("chr1", b, e) for b, e in ranges]
        assert sum(len(p_.reads) for p_ in parts) == len(whole.reads)           # every read in exactly one range
        got = f.join(parts)
        for name in ("pos", "n_cigar", "l_seq", "flag", "mapq", "hp", "cigar_off", "seq_off"):
            assert np.array_equal(got.reads[name], whole.reads[name]), (n_parts, name)
        assert np.array_equal(got.cigar, whole.cigar) and np.array_equal(got.seq, whole.seq)
    assert any(len(p_.reads) == 0 for p_ in parts)                               # (5 % of the contig is expressed: empty ranges exist)
    monkeypatch.setenv("C3R_FETCH_PART_BP", "100")
    assert f.plan(1000, 8) == [(0, 125), (125, 250), (250, 375), (375, 500), (500, 625), (625, 750), (750, 875), (875, None)]
    assert len(f.join([f.part("chr2", b, e) for b, e in f.plan(1000, 8)]).reads) == 0
    rs2, ref2, _dt = f("chr1", L)
    assert np.array_equal(rs2.cigar, whole.cigar) and ref2.tobytes().decode() == ref.decode().upper()
    f.close()


def test_a_failing_fetch_ends_the_run_with_an_error_not_a_hang(tmp_path):
    """A BAM whose blocks are damaged inside the second contig: that contig's fetch raises on a fetch thread, the error reaches the
    main loop through the joined future, the workers are stopped and the driver exits 1 with the message."""
    import time
    tmp = str(tmp_path)
    fa, bm, wfn = _fake_sample(tmp)
    raw = bytearray(open(bm, "rb").read())
    # find the BGZF block that holds the middle of the file and damage its deflate stream
    off, blocks = 0, []
    while off + 18 <= len(raw):
        bs = int.from_bytes(raw[off + 16:off + 18], "little") + 1
        blocks.append((off, bs))
        off += bs
    o, bs = blocks[len(blocks) // 2]
    for k in range(o + 30, o + min(bs - 10, 400)):
        raw[k] ^= 0x5a
    open(bm, "wb").write(bytes(raw))
    t0 = time.time()
    codes, outs = _launch(tmp, os.path.join(tmp, "bad"), fa, bm, wfn, ["--print_ref_calls"], expect_fail=True)
    assert codes == [1] and "[ERROR] call_sample" in outs[0] and "c3r_bam_fetch" in outs[0], outs
    assert time.time() - t0 < 60
    # nothing half-written stays behind: no output.vcf.gz without its EOF block, no index (a stale .tbi of an earlier run would sit
    # beside it)
    left = sorted(f for f in os.listdir(os.path.join(tmp, "bad")) if f.startswith("output"))
    assert left == [], left
