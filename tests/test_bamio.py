"""libc3r_io.so (csrc/bamio.cpp, include/c3r_io.h): BAM/BGZF/BAI -> flat read records.  CPU-only.
The pure-Python reader/writer clair3_rna_amd/bam.py is the independent checker (SAM spec restated twice)."""
import os
import re
import struct
import zlib

import numpy as np
import pytest

from clair3_rna_amd import bam, bamio, synth
from clair3_rna_amd.reads import ReadSet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    assert len(a) == len(b)
    for f in ("pos", "n_cigar", "l_seq", "flag", "mapq", "hp"):
        assert np.array_equal(a.reads[f], b.reads[f]), f
    assert np.array_equal(a.cigar, b.cigar) and np.array_equal(a.seq, b.seq)
    assert np.array_equal(a.reads["cigar_off"], b.reads["cigar_off"]) and np.array_equal(a.reads["seq_off"], b.reads["seq_off"])


def _subset(rs, keep):
    """Reads of `rs` selected by boolean mask, re-packed (offsets recomputed) — what a region fetch must return."""
    reads = rs.reads[keep].copy()
    cig, seq, co, so = [], [], 0, 0
    for r in reads:
        c = rs.cigar[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        nb = (int(r["l_seq"]) + 1) // 2
        s = rs.seq[int(r["seq_off"]):int(r["seq_off"]) + nb]
        r["cigar_off"], r["seq_off"] = co, so
        cig.append(c); seq.append(s); co += len(c); so += nb
    return ReadSet(reads, np.concatenate(cig) if cig else np.zeros(0, np.uint32), np.concatenate(seq) if seq else np.zeros(0, np.uint8))


def _ref_end(rs):
    end = np.zeros(len(rs), np.int64)
    for i, r in enumerate(rs.reads):
        c = rs.cigar[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
        op, ln = c & 15, c >> 4
        end[i] = int(r["pos"]) + max(1, int(ln[np.isin(op, (0, 2, 3, 7, 8))].sum()))
    return end


@pytest.fixture(scope="module")
def bam_case(tmp_path_factory):
    d = tmp_path_factory.mktemp("bamio")
    L = 600000
    ref, rs, _ = synth.generate_contig(contig_len=L, seed=9, depth=25.0, expressed_frac=0.08, intron_hi=20000.0, phased=True)
    ref2, rs2, _ = synth.generate_contig(contig_len=150000, seed=10, depth=15.0, expressed_frac=0.1)
    p = str(d / "x.bam")
    bam.write_bam(p, [("chr20", L), ("chr21", 150000), ("chrEmpty", 5000)], {"chr20": rs, "chr21": rs2})
    return dict(path=p, rs=rs, rs2=rs2, L=L)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "c3r_io.h")).read()
    declared = sorted(set(re.findall(r"\b(c3r_(?:bam|vcf)_[a-z_0-9]+)\s*\(", hdr)))
    lib = bamio.load_library()
    assert declared and not [s for s in declared if not hasattr(lib, s)]
    assert sorted(bamio.EXPORTS) == declared


def test_full_contig_equals_python_reader_without_index(bam_case):
    with bamio.BamFile(bam_case["path"], threads=3) as bf:
        assert not bf.has_index
        assert bf.contigs() == [("chr20", bam_case["L"]), ("chr21", 150000), ("chrEmpty", 5000)]
        _same(bf.fetch("chr20"), bam.read_contig(bam_case["path"], "chr20"))
        _same(bf.fetch("chr20"), bam_case["rs"])
        _same(bf.fetch("chr21"), bam_case["rs2"])
        assert len(bf.fetch("chrEmpty")) == 0 and len(bf.fetch("chrNope")) == 0


@pytest.mark.parametrize("batch", ["1", "3"])
def test_records_spanning_inflate_rounds(bam_case, batch, monkeypatch):
    monkeypatch.setenv("C3R_IO_BATCH", batch)
    with bamio.BamFile(bam_case["path"], threads=2) as bf:
        _same(bf.fetch("chr20"), bam_case["rs"])
        _same(bf.fetch("chr21"), bam_case["rs2"])
    bai = bamio.index_build(bam_case["path"], bam_case["path"] + ".tmp.bai")
    monkeypatch.delenv("C3R_IO_BATCH")
    ref_bai = bamio.index_build(bam_case["path"], bam_case["path"] + ".tmp2.bai")
    assert open(bai, "rb").read() == open(ref_bai, "rb").read()      # virtual offsets do not depend on the batching
    os.remove(bai); os.remove(ref_bai)


def test_region_fetch_with_and_without_index(bam_case):
    rs = bam_case["rs"]
    end = _ref_end(rs)
    pos = rs.reads["pos"].astype(np.int64)
    regions = [(0, 1000), (100000, 100001), (123456, 234567), (16383, 16385), (590000, 600000), (0, 600000), (300000, 300000 + (1 << 14))]
    with bamio.BamFile(bam_case["path"]) as bf:
        plain = [bf.fetch("chr20", a, b) for a, b in regions]
    bai = bamio.index_build(bam_case["path"])
    assert open(bai, "rb").read(4) == b"BAI\x01"
    try:
        with bamio.BamFile(bam_case["path"]) as bf:
            assert bf.has_index
            for (a, b), pl in zip(regions, plain):
                exp = _subset(rs, (pos < b) & (end > a))
                got = bf.fetch("chr20", a, b)
                _same(got, exp)
                _same(pl, exp)
            _same(bf.fetch("chr21"), bam_case["rs2"])          # whole contig through the index
            _same(bf.fetch("chr21", 70000, 90000), _subset(bam_case["rs2"], (bam_case["rs2"].reads["pos"] < 90000) & (_ref_end(bam_case["rs2"]) > 70000)))
            assert len(bf.fetch("chrEmpty", 0, 100)) == 0
    finally:
        os.remove(bai)


def test_index_touches_only_the_region_blocks(bam_case, tmp_path):
    """With an index a small region must not decode the whole file: corrupt a late block and fetch an early region."""
    src = open(bam_case["path"], "rb").read()
    p = str(tmp_path / "y.bam")
    open(p, "wb").write(src)
    bamio.index_build(p)
    # find the block table and damage the deflate payload of a block in the last quarter of the file
    off, offs = 0, []
    while off < len(src):
        bs = struct.unpack_from("<H", src, off + 16)[0] + 1
        offs.append(off); off += bs
    victim = offs[(3 * len(offs)) // 4]
    dam = bytearray(src)
    for k in range(30, 60):
        dam[victim + k] ^= 0xff
    open(p, "wb").write(bytes(dam))
    with bamio.BamFile(p) as bf:
        got = bf.fetch("chr20", 1000, 50000)
        exp = _subset(bam_case["rs"], (bam_case["rs"].reads["pos"] < 50000) & (_ref_end(bam_case["rs"]) > 1000))
        _same(got, exp)
    os.remove(p + ".bai")
    with bamio.BamFile(p) as bf, pytest.raises(IOError):      # the unindexed path has to inflate it and must fail loudly
        bf.fetch("chr21")


def _raw_bam(path, records, contigs=(("c1", 100000),)):
    text = "@HD\tVN:1.6\tSO:coordinate\n"
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs)))
    for name, ln in contigs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<i", ln)
    for (tid, pos, mapq, flag, cig, seq_codes, aux) in records:
        qn = b"q\x00"
        l_seq = len(seq_codes)
        packed = bytearray((l_seq + 1) // 2)
        for i, c in enumerate(seq_codes):
            packed[i >> 1] |= c << (0 if i & 1 else 4)
        body = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), mapq, 0, len(cig), flag, l_seq, -1, -1, 0) + qn + \
            np.asarray(cig, "<u4").tobytes() + bytes(packed) + b"\xff" * l_seq + aux
        out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        bam._bgzf_write(f, bytes(out))
        f.write(bam._BGZF_EOF)


def test_aux_tags_long_cigar_and_odd_records(tmp_path):
    M, I, D, N, S = 0, 1, 2, 3, 4
    op = lambda ln, o: (ln << 4) | o
    seq10 = [1, 2, 4, 8, 15, 1, 2, 4, 8, 1]
    real = [op(3, M), op(1, I), op(2, M), op(50, N), op(4, M)]
    recs = [
        (0, 100, 60, 0, [op(10, M)], seq10, b"HPC\x01"),
        (0, 120, 60, 16, [op(10, M)], seq10, b"NMi" + struct.pack("<i", 3) + b"HPc\x02" + b"RGZgrp\x00"),
        (0, 130, 60, 0, [op(10, M)], seq10, b"HPS" + struct.pack("<H", 2) + b"XAAx"),
        (0, 140, 60, 0, [op(10, M)], seq10, b"HPi" + struct.pack("<i", -1)),                    # non-positive HP -> untagged
        (0, 150, 60, 0, [op(10, S), op(57, N)], seq10, b"CGBI" + struct.pack("<I", len(real)) + np.asarray(real, "<u4").tobytes() + b"HPC\x01"),
        (0, 160, 60, 0, [op(10, M)], seq10, b"ZZBs" + struct.pack("<I", 3) + struct.pack("<hhh", 1, 2, 3) + b"HPC\x02"),
        (0, 170, 0, 4, [], seq10, b""),                                                           # no CIGAR: mpileup never sees it
        (0, 180, 60, 0, [op(10, M)], [], b""),                                                    # l_seq = 0 ('*')
        (-1, -1, 0, 4, [], seq10, b""),                                                           # unplaced
    ]
    p = str(tmp_path / "t.bam")
    _raw_bam(p, recs)
    with bamio.BamFile(p) as bf:
        got = bf.fetch("c1")
    exp = bam.read_contig(p, "c1")
    keep = exp.reads["n_cigar"] > 0
    _same(got, _subset(exp, keep))
    assert list(got.reads["hp"]) == [1, 2, 2, 0, 1, 2, 0]
    assert list(got.reads["pos"]) == [100, 120, 130, 140, 150, 160, 180]
    assert list(got.cigar[int(got.reads["cigar_off"][4]):][:5]) == real and got.reads["n_cigar"][4] == 5
    bamio.index_build(p)
    with bamio.BamFile(p) as bf:
        _same(bf.fetch("c1", 0, 100000), got)
        assert list(bf.fetch("c1", 205, 206).reads["pos"]) == [150]      # inside the long-CIGAR read's intron
        assert list(bf.fetch("c1", 109, 121).reads["pos"]) == [100, 120]


def test_unsorted_file_is_refused_by_the_indexer(tmp_path):
    op = lambda ln, o: (ln << 4) | o
    p = str(tmp_path / "u.bam")
    _raw_bam(p, [(0, 500, 60, 0, [op(5, 0)], [1] * 5, b""), (0, 100, 60, 0, [op(5, 0)], [1] * 5, b"")])
    with pytest.raises(IOError):
        bamio.index_build(p)
    with pytest.raises(IOError):
        bamio.BamFile(str(tmp_path / "missing.bam"))
    open(str(tmp_path / "junk.bam"), "wb").write(b"not a bam at all")
    with pytest.raises(IOError):
        bamio.BamFile(str(tmp_path / "junk.bam"))


def test_malformed_records_are_errors_not_overreads(tmp_path):
    """A CG:B,I array (or any B array) that claims more elements than its record holds, and a record whose CIGAR / sequence
    lengths run past its block, make the fetch fail with a message; nothing is read outside the record."""
    M, N, S = 0, 3, 4
    op = lambda ln, o: (ln << 4) | o
    seq10 = [1, 2, 4, 8, 15, 1, 2, 4, 8, 1]
    good = (0, 100, 60, 0, [op(10, M)], seq10, b"HPC\x01")
    bad_cg = (0, 150, 60, 0, [op(10, S), op(57, N)], seq10, b"CGBI" + struct.pack("<I", 1 << 20) + b"\x00" * 8)
    bad_b = (0, 150, 60, 0, [op(10, M)], seq10, b"ZZBs" + struct.pack("<I", 5000) + b"\x00" * 4)
    for k, bad in enumerate((bad_cg, bad_b)):
        p = str(tmp_path / ("m%d.bam" % k))
        _raw_bam(p, [good, bad])
        with bamio.BamFile(p) as bf:
            with pytest.raises(IOError, match="malformed alignment record"):
                bf.fetch("c1")
            assert len(bf.fetch("c1", 0, 120).reads) == 1            # the record before it is still readable
    # l_seq far beyond the record
    p = str(tmp_path / "m2.bam")
    _raw_bam(p, [good, (0, 150, 60, 0, [op(10, M)], seq10, b"")])
    raw = bytearray(b"".join(bam._bgzf_blocks(p)))
    off = raw.rfind(struct.pack("<ii", 0, 150)) + 16
    raw[off:off + 4] = struct.pack("<i", 1 << 24)
    with open(p, "wb") as f:
        bam._bgzf_write(f, bytes(raw)); f.write(bam._BGZF_EOF)
    with bamio.BamFile(p) as bf:
        with pytest.raises(IOError, match="malformed alignment record"):
            bf.fetch("c1")
    # header with a negative reference-name length
    hdr = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", -5) + b"\x00" * 16
    p = str(tmp_path / "h.bam")
    with open(p, "wb") as f:
        bam._bgzf_write(f, hdr); f.write(bam._BGZF_EOF)
    with pytest.raises(IOError, match="reference name length"):
        bamio.BamFile(p)


@pytest.mark.parametrize("margin,batch", [("1", "3"), ("70000", "512"), ("4194304", "1")])
def test_long_index_chunks_inflate_in_parallel(bam_case, margin, batch, monkeypatch):
    """Indexed fetches whose chunk is long (a whole contig) inflate their blocks on threads (scan_blocks) instead of through the
    one-block cursor.  Forced here on a small file: every chunk takes the parallel path, with a margin so short that records
    run past the listed blocks (the fetch must start over with more), and with batches of 1-3 blocks (records across rounds)."""
    rs, rs2 = bam_case["rs"], bam_case["rs2"]
    end = _ref_end(rs)
    pos = rs.reads["pos"].astype(np.int64)
    bai = bamio.index_build(bam_case["path"])
    try:
        monkeypatch.setenv("C3R_IO_PAR_MIN", str(1 << 40))            # cursor path: the reference result
        with bamio.BamFile(bam_case["path"], threads=4) as bf:
            want = [bf.fetch("chr20"), bf.fetch("chr21"), bf.fetch("chr20", 123456, 234567), bf.fetch("chr20", 590000, 600000)]
        monkeypatch.setenv("C3R_IO_PAR_MIN", "0")
        monkeypatch.setenv("C3R_IO_MARGIN", margin)
        monkeypatch.setenv("C3R_IO_BATCH", batch)
        with bamio.BamFile(bam_case["path"], threads=4) as bf:
            got = [bf.fetch("chr20"), bf.fetch("chr21"), bf.fetch("chr20", 123456, 234567), bf.fetch("chr20", 590000, 600000)]
            assert len(bf.fetch("chrEmpty")) == 0
        for g, w in zip(got, want):
            _same(g, w)
        _same(got[0], rs); _same(got[1], rs2)
        _same(got[2], _subset(rs, (pos < 234567) & (end > 123456)))
    finally:
        os.remove(bai)
