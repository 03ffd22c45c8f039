"""Flat read records: the host-side container handed to libc3r (include/c3r_types.h: c3r_read_t).

One ReadSet = the alignments of one contig in BAM order, i.e. what `samtools mpileup <bam> -r ctg:...`
would stream in the reference (src/create_tensor_pileup.py:446-451).
"""
import numpy as np

READ_DTYPE = np.dtype([("pos", "<i4"), ("cigar_off", "<u4"), ("n_cigar", "<u4"), ("l_seq", "<u4"),
                       ("seq_off", "<u8"), ("flag", "<u2"), ("mapq", "u1"), ("hp", "u1"), ("reserved", "<u4")],
                      align=True)
assert READ_DTYPE.itemsize == 32

CIGAR_OPS = "MIDNSHP=X"
NT16 = "=ACMGRSVTWYHKDBN"
_NT16_CODE = {c: i for i, c in enumerate(NT16)}
_NT16_CODE.update({c.lower(): i for i, c in enumerate(NT16)})


def parse_cigar(s):
    """'10M2I5N3M' -> uint32 array of BAM-encoded ops (len << 4 | op)."""
    out, num = [], 0
    for ch in s:
        if ch.isdigit():
            num = num * 10 + ord(ch) - 48
        else:
            out.append((num << 4) | CIGAR_OPS.index(ch))
            num = 0
    return np.asarray(out, dtype=np.uint32)


def pack_seq(seq):
    """ASCII bases -> BAM 4-bit packed bytes (first base in the high nibble)."""
    codes = np.fromiter((_NT16_CODE.get(c, 15) for c in seq), dtype=np.uint8, count=len(seq))
    if len(codes) & 1:
        codes = np.append(codes, np.uint8(0))
    return ((codes[0::2] << 4) | codes[1::2]).astype(np.uint8)


class ReadSet(object):
    """reads (READ_DTYPE[n]), cigar (uint32[]), seq (uint8[] 4-bit packed), sorted by pos."""

    def __init__(self, reads, cigar, seq):
        self.reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
        self.cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
        self.seq = np.ascontiguousarray(seq, dtype=np.uint8)

    def __len__(self):
        return len(self.reads)

    @classmethod
    def from_records(cls, records):
        """records: iterable of dicts/tuples (pos0, cigar_str, seq_str, flag, mapq, hp); sorted by pos (stable)."""
        recs = [r if isinstance(r, dict) else dict(zip(("pos", "cigar", "seq", "flag", "mapq", "hp"), r)) for r in records]
        recs.sort(key=lambda r: r["pos"])
        reads = np.zeros(len(recs), dtype=READ_DTYPE)
        cig, seqs = [], []
        coff = soff = 0
        for i, r in enumerate(recs):
            c = parse_cigar(r["cigar"]) if isinstance(r["cigar"], str) else np.asarray(r["cigar"], dtype=np.uint32)
            s = pack_seq(r["seq"])
            reads[i] = (r["pos"], coff, len(c), len(r["seq"]), soff, r.get("flag", 0), r.get("mapq", 60), r.get("hp", 0), 0)
            cig.append(c)
            seqs.append(s)
            coff += len(c)
            soff += len(s)
        cigar = np.concatenate(cig) if cig else np.zeros(0, np.uint32)
        seq = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
        return cls(reads, cigar, seq)

    def read_bases(self, i, q0, n):
        """Decode n bases of read i starting at query offset q0 (upper-case ASCII)."""
        r = self.reads[i]
        out = []
        for q in range(q0, q0 + n):
            if q >= r["l_seq"]:
                out.append("N")
                continue
            b = int(self.seq[int(r["seq_off"]) + (q >> 1)])
            out.append(NT16[(b & 15) if (q & 1) else (b >> 4)])
        return "".join(out)
