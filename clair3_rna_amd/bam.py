"""Minimal BGZF/BAM reader + writer (SAM spec v1 §4): BAM alignments -> flat read records (c3r_read_t).

Replaces the input side of `samtools mpileup <bam> -r ctg:...` (src/create_tensor_pileup.py:446-451): core.pos,
flag, MAPQ, CIGAR (incl. the CG:B,I long-CIGAR convention), 4-bit SEQ and the HP aux tag are all the tensor
builder needs (base qualities are not used: --min-BQ 0).  Sequential scan of the file, no .bai needed; records of
other contigs are skipped.  The writer exists so that tests and the demo can produce real BAM files.
"""
import struct
import zlib

import numpy as np

from .reads import READ_DTYPE, ReadSet

_BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_blocks(path):
    with open(path, "rb") as f:
        data = f.read()
    off, n = 0, len(data)
    while off < n:
        if data[off:off + 4] != b"\x1f\x8b\x08\x04":
            raise ValueError("%s: not a BGZF file (offset %d)" % (path, off))
        xlen = struct.unpack_from("<H", data, off + 10)[0]
        p, bsize = off + 12, None
        while p < off + 12 + xlen:
            si1, si2, slen = data[p], data[p + 1], struct.unpack_from("<H", data, p + 2)[0]
            if si1 == 66 and si2 == 67:
                bsize = struct.unpack_from("<H", data, p + 4)[0] + 1
            p += 4 + slen
        if bsize is None:
            raise ValueError("%s: BGZF block without BC field" % path)
        cdata = data[off + 12 + xlen:off + bsize - 8]
        yield zlib.decompress(cdata, -15)
        off += bsize


_AUX_SIZE = {b"A": 1, b"c": 1, b"C": 1, b"s": 2, b"S": 2, b"i": 4, b"I": 4, b"f": 4}
_AUX_FMT = {b"c": "<b", b"C": "<B", b"s": "<h", b"S": "<H", b"i": "<i", b"I": "<I"}


def _scan_aux(buf, p, end):
    """-> (hp, long_cigar or None)"""
    hp, cg = 0, None
    while p + 3 <= end:
        tag, typ = buf[p:p + 2], buf[p + 2:p + 3]
        p += 3
        if typ in _AUX_SIZE:
            if tag == b"HP" and typ in _AUX_FMT:
                v = struct.unpack_from(_AUX_FMT[typ], buf, p)[0]
                hp = v if 0 < v < 256 else 0
            p += _AUX_SIZE[typ]
        elif typ in (b"Z", b"H"):
            q = buf.index(b"\x00", p)
            p = q + 1
        elif typ == b"B":
            sub = buf[p:p + 1]
            cnt = struct.unpack_from("<I", buf, p + 1)[0]
            esz = _AUX_SIZE[sub]
            if tag == b"CG" and sub == b"I":
                cg = np.frombuffer(buf, dtype="<u4", count=cnt, offset=p + 5).copy()
            p += 5 + cnt * esz
        else:
            raise ValueError("bad aux type %r" % typ)
    return hp, cg


def read_header(path):
    gen = _bgzf_blocks(path)
    buf = b""
    for blk in gen:
        buf += blk
        if len(buf) >= 12:
            l_text = struct.unpack_from("<i", buf, 4)[0]
            if len(buf) >= 12 + l_text:
                break
    if buf[:4] != b"BAM\x01":
        raise ValueError("%s: bad BAM magic" % path)
    return buf, gen


def read_contig(path, contig):
    """All alignments of `contig`, in file order -> ReadSet."""
    blocks = list(_bgzf_blocks(path))
    buf = b"".join(blocks)
    if buf[:4] != b"BAM\x01":
        raise ValueError("%s: bad BAM magic" % path)
    l_text = struct.unpack_from("<i", buf, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", buf, p)[0]
    p += 4
    names = []
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", buf, p)[0]
        names.append(buf[p + 4:p + 4 + l_name - 1].decode())
        p += 4 + l_name + 4
    if contig not in names:
        return ReadSet(np.zeros(0, READ_DTYPE), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    tid = names.index(contig)
    recs, cigs, seqs = [], [], []
    coff = soff = 0
    n = len(buf)
    while p + 4 <= n:
        block_size = struct.unpack_from("<i", buf, p)[0]
        q = p + 4
        ref_id, pos, l_read_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", buf, q)
        end = q + block_size
        if ref_id == tid:
            c0 = q + 32 + l_read_name
            cigar = np.frombuffer(buf, dtype="<u4", count=n_cig, offset=c0).copy()
            s0 = c0 + 4 * n_cig
            nb = (l_seq + 1) // 2
            seq = np.frombuffer(buf, dtype=np.uint8, count=nb, offset=s0)
            hp, cg = _scan_aux(buf, s0 + nb + l_seq, end)
            if cg is not None and n_cig == 2 and (cigar[0] & 15) == 4 and (cigar[0] >> 4) == l_seq and (cigar[1] & 15) == 3:
                cigar = cg
            recs.append((pos, coff, len(cigar), l_seq, soff, flag, mapq, hp, 0))
            cigs.append(cigar)
            seqs.append(seq)
            coff += len(cigar)
            soff += nb
        p = end
    reads = np.array(recs, dtype=READ_DTYPE) if recs else np.zeros(0, READ_DTYPE)
    return ReadSet(reads, np.concatenate(cigs) if cigs else np.zeros(0, np.uint32),
                   np.concatenate(seqs) if seqs else np.zeros(0, np.uint8))


def _bgzf_write(f, payload):
    for i in range(0, len(payload), 0xff00):
        chunk = payload[i:i + 0xff00]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cdata = co.compress(chunk) + co.flush()
        bsize = len(cdata) + 25
        f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize))
        f.write(cdata)
        f.write(struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))


def write_bam(path, contigs, contig_reads, sample="SAMPLE"):
    """contigs: [(name, length)]; contig_reads: {name: ReadSet}.  Coordinate-sorted within each contig."""
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join("@SQ\tSN:%s\tLN:%d\n" % c for c in contigs) + "@RG\tID:1\tSM:%s\n" % sample
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs)))
    for name, length in contigs:
        out += struct.pack("<i", len(name) + 1) + name.encode() + b"\x00" + struct.pack("<i", length)
    k = 0
    for tid, (name, _l) in enumerate(contigs):
        rs = contig_reads.get(name)
        if rs is None:
            continue
        for r in rs.reads:
            k += 1
            qname = ("r%d" % k).encode() + b"\x00"
            cig = rs.cigar[int(r["cigar_off"]):int(r["cigar_off"]) + int(r["n_cigar"])]
            nb = (int(r["l_seq"]) + 1) // 2
            seq = rs.seq[int(r["seq_off"]):int(r["seq_off"]) + nb].tobytes()
            aux = b""
            if r["hp"]:
                aux = b"HPC" + struct.pack("<B", int(r["hp"]))
            body = struct.pack("<iiBBHHHiiii", tid, int(r["pos"]), len(qname), int(r["mapq"]), 4680, len(cig), int(r["flag"]),
                               int(r["l_seq"]), -1, -1, 0) + qname + cig.astype("<u4").tobytes() + seq + b"\xff" * int(r["l_seq"]) + aux
            out += struct.pack("<i", len(body)) + body
    with open(path, "wb") as f:
        _bgzf_write(f, bytes(out))
        f.write(_BGZF_EOF)
