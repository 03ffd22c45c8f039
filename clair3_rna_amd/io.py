"""Host-side file inputs of the per-chunk driver: indexed FASTA slices, BED intervals, candidate-site VCFs,
flat read archives and network weights.  Mirrors the reference helpers it replaces:
    shared/utils.py:168-194   reference_sequence_from (samtools faidx ctg:a-b, upper-cased)
    shared/interval_tree.py:8-74  bed_tree_from (gzip -fdc, '#' comments, start==end widened by one)
    shared/utils.py:196-216   vcf_candidates_from
"""
import gzip
import os

import numpy as np

from .reads import READ_DTYPE, ReadSet


def _open_text(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rt") if magic == b"\x1f\x8b" else open(path, "r")


# ----------------------------------------------------------------------------- FASTA
def read_fai(ref_fn):
    fai = ref_fn + ".fai"
    if not os.path.exists(fai):
        alt = ".".join(ref_fn.split(".")[:-1]) + ".fai"     # fn.fa.fai -> fn.fai (shared/utils.py:84-88)
        if os.path.exists(alt):
            fai = alt
        else:
            raise FileNotFoundError("[ERROR] file %s not found" % fai)
    out = []
    with open(fai) as f:
        for row in f:
            c = row.rstrip("\n").split("\t")
            out.append((c[0], int(c[1]), int(c[2]), int(c[3]), int(c[4])))
    return out


def fetch_reference(ref_fn, ctg, start1, end1, raw=False):
    """Upper-cased bases of ctg:start1-end1 (1-based inclusive, clamped to the contig), like `samtools faidx`.
    raw=True: the bytes as they stand in the file (line ends removed, case kept — c3r_set_reference upper-cases on its own),
    which spares a 60 MB contig two passes in Python."""
    for name, length, offset, linebases, linewidth in read_fai(ref_fn):
        if name != ctg:
            continue
        start1 = max(1, start1)
        end1 = min(length, end1)
        if end1 < start1:
            return b"" if raw else ""
        b0, b1 = start1 - 1, end1           # 0-based half-open
        fo = offset + (b0 // linebases) * linewidth + b0 % linebases
        lo = offset + ((b1 - 1) // linebases) * linewidth + (b1 - 1) % linebases + 1
        with open(ref_fn, "rb") as f:
            f.seek(fo)
            data = f.read(lo - fo)
        data = data.replace(b"\n", b"").replace(b"\r", b"")
        return data if raw else data.decode().upper()
    raise KeyError("contig %s not in %s.fai" % (ctg, ref_fn))


def write_fasta(path, contigs, width=60):
    """contigs: list of (name, sequence).  Also writes the .fai."""
    fai = []
    with open(path, "w") as f:
        for name, seq in contigs:
            hdr = ">%s\n" % name
            f.write(hdr)
            off = f.tell()
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + "\n")
            fai.append((name, len(seq), off, width, width + 1))
    with open(path + ".fai", "w") as f:
        for r in fai:
            f.write("%s\t%d\t%d\t%d\t%d\n" % r)


# ----------------------------------------------------------------------------- BED / VCF sites
def read_bed(bed_fn, contig, keep_start=None, keep_end=None):
    """-> (intervals [(start,end)...] 0-based half-open, bed_start, bed_end) for one contig.
    keep_start/keep_end reproduce bed_ctg_start/bed_ctg_end filtering (shared/interval_tree.py:62-64)."""
    iv = []
    bed_start, bed_end = float("inf"), 0
    with _open_text(bed_fn) as f:
        for row in f:
            if not row.strip() or row[0] == "#":
                continue
            c = row.strip().split()
            if c[0] != contig:
                continue
            s, e = int(c[1]), int(c[2])
            if e < s or s < 0 or e < 0:
                raise ValueError("[ERROR] Invalid bed input %s %d %d" % (c[0], s, e))
            if keep_start and keep_end and (e < keep_start or s > keep_end):
                continue
            bed_start, bed_end = min(s, bed_start), max(e, bed_end)
            if s == e:
                e += 1
            iv.append((s, e))
    return iv, (None if not iv else bed_start), (None if not iv else bed_end)


def read_vcf_sites(vcf_fn, contig):
    sites = set()
    with _open_text(vcf_fn) as f:
        for row in f:
            if row[0] == "#":
                continue
            c = row.strip().split(None, 3)
            if c[0] != contig:
                continue
            sites.add(int(c[1]))
    return sorted(sites)


# ----------------------------------------------------------------------------- flat read archive
def save_reads(path, contig_reads):
    """contig_reads: dict name -> ReadSet.  One .npz holding the flat records of include/c3r_types.h."""
    arrs = {}
    for name, rs in contig_reads.items():
        arrs["reads__" + name] = rs.reads
        arrs["cigar__" + name] = rs.cigar
        arrs["seq__" + name] = rs.seq
    np.savez(path, **arrs)


def load_reads(path, contig, beg0=None, end0=None):
    """Read source of the per-chunk driver.  *.npz = flat read archive; *.bam = BGZF/BAM through libc3r_io.so
    (csrc/bamio.cpp): with a .bai only the BGZF blocks holding alignments that overlap the 0-based half-open region
    [beg0, end0) are inflated — the `-r ctg:beg-end` of the reference's mpileup call (create_tensor_pileup.py:446-451)."""
    if path.endswith(".npz"):
        z = np.load(path)
        key = "reads__" + contig
        if key not in z.files:
            return ReadSet(np.zeros(0, READ_DTYPE), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
        return ReadSet(z[key], z["cigar__" + contig], z["seq__" + contig])
    from . import bamio
    with bamio.BamFile(path) as bf:
        return bf.fetch(contig, beg0 or 0, end0)


# ----------------------------------------------------------------------------- weights
def load_weights(chkpnt_fn, channels):
    """Flat fp32 blob in the layout of include/c3r.h.  Looked for in this order: `<prefix>.c3rw.npy` (converted once where
    TensorFlow exists, INTEGRATION.md section 3 — the verified route); `<prefix>.index` + `.data-*`, the TensorFlow checkpoint
    bundle itself, through the TF-free reader tfckpt.py (written from the format descriptions; no TF-written file was available
    to verify it against — DESIGN.md section 7, F5)."""
    for cand in (chkpnt_fn, chkpnt_fn + ".c3rw.npy", chkpnt_fn + ".npy"):
        if os.path.isfile(cand) and cand.endswith(".npy"):
            w = np.load(cand).astype(np.float32).reshape(-1)
            return w
    if os.path.isfile(chkpnt_fn + ".index"):
        from . import tfckpt
        return tfckpt.weights_from_bundle(chkpnt_fn, channels)
    raise FileNotFoundError("no weights found for --chkpnt_fn %s (expected %s.c3rw.npy or the checkpoint bundle %s.index)"
                            % (chkpnt_fn, chkpnt_fn, chkpnt_fn))
