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
    declared = sorted(set(re.findall(r"\b(c3r_(?:bam|vcfz?|fasta|io)_[a-z_0-9]+)\s*\(", hdr)))
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
    # the same through the index, blocks inflated and records parsed on threads: the error of the FIRST bad record, nothing kept
    p = str(tmp_path / "m_par.bam")
    many = [(0, 100 + 3 * i, 60, 0, [op(10, M)], seq10, b"HPC\x01") for i in range(40)]
    bad_mid = (0, 100 + 3 * 40, 60, 0, [op(10, M)], seq10, b"ZZBs" + struct.pack("<I", 5000) + b"\x00" * 4)
    _raw_bam(p, many + [bad_mid] + [(0, 300 + 3 * i, 60, 0, [op(10, M)], seq10, b"") for i in range(40)])
    bai = bamio.index_build(p)
    try:
        import pytest as _pt
        mp = _pt.MonkeyPatch()
        mp.setenv("C3R_IO_PAR_MIN", "0"); mp.setenv("C3R_IO_PARSE_MIN", "3")
        try:
            with bamio.BamFile(p, threads=4) as bf:
                with pytest.raises(IOError, match="malformed alignment record at position %d" % (100 + 3 * 40 + 1)):
                    bf.fetch("c1")
                assert len(bf.fetch("c1", 0, 150).reads) == 17           # the records before it are still readable
        finally:
            mp.undo()
    finally:
        os.remove(bai)
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


@pytest.mark.parametrize("margin,batch,parse_min", [("1", "3", "1024"), ("70000", "512", "7"), ("4194304", "1", "1"), ("1", "3", "2")])
def test_long_index_chunks_inflate_in_parallel(bam_case, margin, batch, parse_min, monkeypatch):
    """Indexed fetches whose chunk is long (a whole contig) inflate their blocks on threads (scan_blocks) instead of through the
    one-block cursor, and parse the records of every batch on threads (take_records: per-thread arrays appended in file order).
    Forced here on a small file: every chunk takes the parallel path, with a margin so short that records run past the listed
    blocks (the fetch must start over with more), with batches of 1-3 blocks (records across rounds), and with threads started for
    runs of 1, 2 or 7 records."""
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
        monkeypatch.setenv("C3R_IO_PARSE_MIN", parse_min)
        with bamio.BamFile(bam_case["path"], threads=4) as bf:
            got = [bf.fetch("chr20"), bf.fetch("chr21"), bf.fetch("chr20", 123456, 234567), bf.fetch("chr20", 590000, 600000)]
            assert len(bf.fetch("chrEmpty")) == 0
        for g, w in zip(got, want):
            _same(g, w)
        _same(got[0], rs); _same(got[1], rs2)
        _same(got[2], _subset(rs, (pos < 234567) & (end > 123456)))
    finally:
        os.remove(bai)


# ---- an independently written BAM + BAI (SAM spec sections 4.2, 5.1.1, 5.2; nothing below uses clair3_rna_amd/bam.py or the C++ indexer):
# BGZF blocks cut at random byte offsets (records span blocks, many records per block, empty blocks in between), reads placed on the
# 2^14 / 2^17 / 2^20 bin boundaries, a 70,000-op CIGAR behind a CG:B,I tag, and a .bai computed from the block table.
def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


def _independent_bam_and_bai(path, recs, contig_len, seed):
    import random
    rng = random.Random(seed)
    text = "@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:c1\tLN:%d\n" % contig_len
    stream = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", 1))
    stream += struct.pack("<i", 3) + b"c1\x00" + struct.pack("<i", contig_len)
    starts = []
    for k, (pos, cig, codes, flag, mapq, hp) in enumerate(recs):
        starts.append(len(stream))
        qn = ("read%05d" % k).encode() + b"\x00"
        l_seq = len(codes)
        packed = bytearray((l_seq + 1) // 2)
        for i, c in enumerate(codes):
            packed[i >> 1] |= c << (0 if i & 1 else 4)
        ref_len = sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8))
        aux = b""
        in_rec = list(cig)
        if len(cig) > 65535:                                    # SAM spec 4.2.2: real CIGAR in CG:B,I, placeholder <l_seq>S<ref_len>N
            aux += b"CGBI" + struct.pack("<I", len(cig)) + np.asarray(cig, "<u4").tobytes()
            in_rec = [(l_seq << 4) | 4, (ref_len << 4) | 3]
        if hp:
            aux += b"HPC" + struct.pack("<B", hp)
        aux += b"NMi" + struct.pack("<i", k)                    # an unrelated tag after the ones that matter
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(qn), mapq, _reg2bin(pos, pos + max(1, ref_len)), len(in_rec), flag, l_seq, -1, -1, 0) + qn + \
            np.asarray(in_rec, "<u4").tobytes() + bytes(packed) + bytes([30] * l_seq) + aux
        stream += struct.pack("<i", len(body)) + body
    starts.append(len(stream))
    # BGZF: random cuts, some empty blocks, one maximal block
    cuts, p = [0], 0
    while p < len(stream):
        r = rng.random()
        step = 0 if r < 0.05 else (0xff00 if r < 0.08 else rng.randint(1, 3000))
        p = min(len(stream), p + step)
        cuts.append(p)
    blocks, f = [], bytearray()                                 # (uncompressed start, compressed offset)
    for a, b in zip(cuts[:-1], cuts[1:]):
        chunk = bytes(stream[a:b])
        co = zlib.compressobj(rng.choice([0, 1, 9]), zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        blocks.append((a, len(f), b - a))
        f += b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(cdata) + 25) + cdata + \
            struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    eof_at = len(f)
    f += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    open(path, "wb").write(bytes(f))

    def voff(u):                                               # virtual offset of uncompressed offset u: the LAST block that starts at or before it and holds it
        for (a, c, n) in reversed(blocks):
            if a <= u < a + n:
                return (c << 16) | (u - a)
        return eof_at << 16                                     # one past the data: the EOF block
    bins, lin = {}, {}
    for k, (pos, cig, _codes, _flag, _mapq, _hp) in enumerate(recs):
        ref_len = max(1, sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3, 7, 8)))
        vb, ve = voff(starts[k]), voff(starts[k + 1])
        ch = bins.setdefault(_reg2bin(pos, pos + ref_len), [])
        if ch and ch[-1][1] == vb:
            ch[-1][1] = ve
        else:
            ch.append([vb, ve])
        for w in range(pos >> 14, ((pos + ref_len - 1) >> 14) + 1):
            lin[w] = min(lin.get(w, vb), vb)
    out = bytearray(b"BAI\x01" + struct.pack("<i", 1) + struct.pack("<i", len(bins)))
    for b in sorted(bins):
        out += struct.pack("<Ii", b, len(bins[b]))
        for vb, ve in bins[b]:
            out += struct.pack("<QQ", vb, ve)
    n_intv = (max(lin) + 1) if lin else 0
    out += struct.pack("<i", n_intv)
    last = 0
    for w in range(n_intv):                                     # empty windows carry the previous offset forward (samtools' convention) or 0
        last = lin.get(w, last)
        out += struct.pack("<Q", last)
    open(path + ".bai", "wb").write(bytes(out))
    return dict(starts=starts, voff=voff, blocks=blocks)


def _parse_bai(path):
    d = open(path, "rb").read()
    assert d[:4] == b"BAI\x01"
    n_ref, p = struct.unpack_from("<i", d, 4)[0], 8
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", d, p)[0]; p += 4
        bins = {}
        for _b in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", d, p); p += 8
            bins[b] = [struct.unpack_from("<QQ", d, p + 16 * i) for i in range(n_chunk)]; p += 16 * n_chunk
        n_intv = struct.unpack_from("<i", d, p)[0]; p += 4
        lin = list(struct.unpack_from("<%dQ" % n_intv, d, p)); p += 8 * n_intv
        refs.append((bins, lin))
    return refs


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_independent_writer_bin_boundaries_long_cigar_and_index(tmp_path, seed):
    import random
    rng = random.Random(100 + seed)
    M, I, D, N, S = 0, 1, 2, 3, 4
    op = lambda ln, o: (ln << 4) | o
    L = 3 << 20
    spots = [0, 5, 16383, 16384, 16385, (1 << 17) - 3, 1 << 17, (1 << 17) + 1, (1 << 20) - 40, 1 << 20, (2 << 20) - 1, L - 300]
    recs = []
    for s_ in spots:
        for _ in range(rng.randint(1, 4)):
            ln = rng.choice([1, 7, 30, 64, 200])
            cig = [op(ln, M)] if rng.random() < 0.5 else [op(3, S), op(ln, M), op(rng.choice([1, 20000, 140000]), N), op(5, M), op(2, I), op(4, M)]
            if s_ + sum(c >> 4 for c in cig if (c & 15) in (0, 2, 3)) >= L:
                cig = [op(ln, M)]
            l_seq = sum(c >> 4 for c in cig if (c & 15) in (0, 1, 4))
            recs.append((s_, cig, [rng.choice([1, 2, 4, 8, 15]) for _ in range(l_seq)], rng.choice([0, 16, 1024, 256]), rng.choice([60, 3, 0]), rng.choice([0, 1, 2])))
    for _ in range(150):                                        # filler: many small records per block
        pos = rng.randrange(0, L - 5000)
        ln = rng.randint(20, 400)
        recs.append((pos, [op(ln, M)], [rng.choice([1, 2, 4, 8]) for _ in range(ln)], 0, 60, 0))
    big = []                                                    # 70,001 ops: 1M1I... (more than the 16-bit n_cigar_op field holds)
    for _ in range(35000):
        big += [op(1, M), op(1, I)]
    big.append(op(1, M))
    recs.append((40000, big, [rng.choice([1, 2, 4, 8]) for _ in range(70001)], 0, 60, 1))
    recs.sort(key=lambda r: r[0])
    p = str(tmp_path / "ind.bam")
    info = _independent_bam_and_bai(p, recs, L, seed)
    pos = np.array([r[0] for r in recs], np.int64)
    end = np.array([r[0] + max(1, sum(c >> 4 for c in r[1] if (c & 15) in (0, 2, 3, 7, 8))) for r in recs], np.int64)
    regions = [(0, 1), (16383, 16384), (16384, 16385), (16000, 17000), ((1 << 17) - 1, (1 << 17) + 1), (1 << 17, (1 << 17) + 1), ((1 << 20) - 1, 1 << 20),
               (1 << 20, (1 << 20) + 1), (39999, 40001), (75000, 75001), (0, L), (L - 1, L), (2 << 20, (2 << 20) + 5), (200000, 900000)]

    def check(bf):
        for a, b in regions:
            want = np.nonzero((pos < b) & (end > a))[0]
            got = bf.fetch("c1", a, b)
            assert len(got) == len(want), (a, b, len(got), len(want))
            for g, k in zip(range(len(got)), want):
                r = got.reads[g]
                pos_k, cig_k, codes_k, flag_k, mapq_k, hp_k = recs[k]
                assert (int(r["pos"]), int(r["flag"]), int(r["mapq"]), int(r["hp"]), int(r["l_seq"]), int(r["n_cigar"])) == (pos_k, flag_k, mapq_k, hp_k, len(codes_k), len(cig_k))
                c = got.cigar[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
                assert np.array_equal(c, np.asarray(cig_k, np.uint32))
                if len(codes_k) <= 400:
                    sq = got.seq[int(r["seq_off"]):int(r["seq_off"]) + (len(codes_k) + 1) // 2]
                    assert [(int(sq[i >> 1]) >> (0 if i & 1 else 4)) & 15 for i in range(len(codes_k))] == codes_k
    with bamio.BamFile(p) as bf:                                # 1. the C++ reader through the INDEPENDENT index
        assert bf.has_index
        check(bf)
    mine = _parse_bai(p + ".bai")
    os.remove(p + ".bai")
    with bamio.BamFile(p) as bf:                                # 2. no index: linear scan
        assert not bf.has_index
        check(bf)
    bamio.index_build(p)                                        # 3. the C++ indexer's own index, read back by independent code
    with bamio.BamFile(p) as bf:
        check(bf)
    (bins, lin), (mbins, mlin) = _parse_bai(p + ".bai")[0], mine[0]
    real = {b: c for b, c in bins.items() if b != 37450}        # (37450: samtools' optional metadata pseudo-bin)
    assert set(real) == set(mbins)                              # the same bins are populated
    for k, (pos_k, cig_k, *_rest) in enumerate(recs):
        b = _reg2bin(pos_k, int(end[k]))
        v = info["voff"](info["starts"][k])
        assert any(cb <= v < ce for cb, ce in real[b]), (k, b)   # every record starts inside a chunk of its bin
        for w in range(pos_k >> 14, ((int(end[k]) - 1) >> 14) + 1):
            assert w < len(lin) and 0 < lin[w] <= v              # the linear index never points past an overlapping record
    assert len(lin) == len(mlin)


def test_index_metadata_pseudo_bin_gives_the_contigs_work(bam_case, tmp_path):
    """The index builder writes the metadata pseudo-bin (37450: file range, mapped / unmapped reads) as `samtools index` does; the reader
    hands it out as the weight a contig gets when a sample's contigs are dealt to the GPUs (by read count, from the index).
    An index without the pseudo-bin still yields the compressed bytes a contig's records span; no index: nothing (-1)."""
    import shutil
    from clair3_rna_amd import shard
    p = str(tmp_path / "w.bam")
    shutil.copy(bam_case["path"], p)
    with bamio.BamFile(p) as bf:
        assert not bf.has_index and bf.contig_weights()["chr20"] == (-1, -1)
    bamio.index_build(p)
    n20, n21 = len(bam_case["rs"]), len(bam_case["rs2"])
    with bamio.BamFile(p) as bf:
        w = bf.contig_weights()
    assert w["chr20"][0] == n20 and w["chr21"][0] == n21 and w["chrEmpty"] == (0, 0)      # (no bin at all: a contig without reads, not "unknown")
    assert w["chr20"][1] > w["chr21"][1] > 0                                        # compressed bytes follow the read counts here
    # independent parse of the pseudo-bin: two 16-byte "chunks" — (first, last virtual offset), (mapped, unmapped)
    raw = open(p + ".bai", "rb").read()
    o = 8
    n_bin = struct.unpack_from("<i", raw, o)[0]; o += 4
    meta = None
    for _ in range(n_bin):
        b, nch = struct.unpack_from("<Ii", raw, o); o += 8
        if b == 37450:
            meta = struct.unpack_from("<QQQQ", raw, o)
        o += 16 * nch
    assert meta is not None and meta[2] == n20 and meta[3] == 0 and meta[1] > meta[0]
    costs, basis = shard.contig_costs(p, ["chr20", "chr21"], {"chr20": 600000, "chr21": 150000})
    assert basis.startswith("mapped reads") and costs == [n20, n21]
    # a read-less contig (samtools index writes no pseudo-bin for it) and one the BAM header does not even have: the deal stays on read counts
    costs, basis = shard.contig_costs(p, ["chr20", "chr21", "chrEmpty", "chrNotInBam"], {"chr20": 600000, "chr21": 150000, "chrEmpty": 5000, "chrNotInBam": 9000})
    assert basis.startswith("mapped reads") and costs[:2] == [n20, n21] and costs[2] == costs[3] >= 1 and costs[2] < n21
    # the same index with the pseudo-bins cut out (an index from a tool that writes none)
    out = bytearray(raw[:8])
    o = 8
    for _ref in range(struct.unpack_from("<i", raw, 4)[0]):
        n_bin = struct.unpack_from("<i", raw, o)[0]; o += 4
        keep = bytearray()
        kept = 0
        for _ in range(n_bin):
            b, nch = struct.unpack_from("<Ii", raw, o)
            if b != 37450:
                keep += raw[o:o + 8 + 16 * nch]; kept += 1
            o += 8 + 16 * nch
        n_intv = struct.unpack_from("<i", raw, o)[0]
        out += struct.pack("<i", kept) + keep + raw[o:o + 4 + 8 * n_intv]
        o += 4 + 8 * n_intv
    open(p + ".bai", "wb").write(bytes(out))
    with bamio.BamFile(p) as bf:
        w2 = bf.contig_weights()
    assert w2["chr20"][0] == -1 and w2["chr20"][1] > w2["chr21"][1] > 0
    costs, basis = shard.contig_costs(p, ["chr20", "chr21"], {"chr20": 600000, "chr21": 150000})
    assert basis.startswith("compressed bytes")
    costs, basis = shard.contig_costs(str(tmp_path / "reads.cram"), ["chr20", "chr21"], {"chr20": 600000, "chr21": 150000})
    assert basis == "length" and costs == [600000, 150000]
