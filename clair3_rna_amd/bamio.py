"""ctypes binding of libc3r_io.so (include/c3r_io.h): BAM + .bai -> ReadSet.

The C++ reader replaces the BAM side of `samtools mpileup <bam> -r ctg:beg-end` (reference
src/create_tensor_pileup.py:436-451).  `bam.py` (pure Python) stays as the independent checker and the test writer."""
import ctypes as C
import os

import numpy as np

from .reads import READ_DTYPE, ReadSet

EXPORTS = ["c3r_bam_open", "c3r_bam_close", "c3r_bam_last_error", "c3r_bam_n_contigs", "c3r_bam_contig", "c3r_bam_has_index", "c3r_bam_contig_weight",
           "c3r_bam_fetch", "c3r_bam_copy", "c3r_bam_index_build", "c3r_vcf_merge", "c3r_vcf_compress", "c3r_vcfz_open", "c3r_vcfz_write",
           "c3r_vcfz_close", "c3r_vcfz_piece_make", "c3r_vcfz_append", "c3r_vcfz_piece_free", "c3r_fasta_fetch", "c3r_io_alloc", "c3r_io_free"]
_LIB = None


def load_library():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libc3r_io.so")
        if not os.path.exists(path):
            raise RuntimeError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        L = C.CDLL(path)
        L.c3r_bam_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.c3r_bam_close.argtypes = [C.c_void_p]
        L.c3r_bam_close.restype = None
        L.c3r_bam_last_error.argtypes = [C.c_void_p]
        L.c3r_bam_last_error.restype = C.c_char_p
        L.c3r_bam_n_contigs.argtypes = [C.c_void_p]
        L.c3r_bam_contig.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]
        L.c3r_bam_has_index.argtypes = [C.c_void_p]
        L.c3r_bam_contig_weight.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.c3r_bam_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64] + [C.POINTER(C.c_int64)] * 3
        L.c3r_bam_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.c3r_bam_index_build.argtypes = [C.c_char_p, C.c_char_p]
        L.c3r_vcf_merge.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int64,
                                    C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.c3r_vcf_compress.argtypes = [C.c_char_p, C.c_int]
        L.c3r_io_alloc.argtypes = [C.c_size_t]
        L.c3r_io_alloc.restype = C.c_void_p
        L.c3r_io_free.argtypes = [C.c_void_p]
        L.c3r_io_free.restype = None
        _LIB = L
    return _LIB


def huge_empty(n, dtype=np.uint8):
    """np.empty(n, dtype) in memory from c3r_io_alloc (2-MB aligned, advised huge) for arrays of 4 MB and more; freed with the
    last array that views it."""
    dtype = np.dtype(dtype)
    nbytes = int(n) * dtype.itemsize
    if nbytes < (4 << 20):
        return np.empty(int(n), dtype)
    import weakref
    L = load_library()
    p = L.c3r_io_alloc(nbytes)
    if not p:
        raise MemoryError("c3r_io_alloc(%d) failed" % nbytes)
    flat = np.frombuffer((C.c_char * nbytes).from_address(p), dtype=np.uint8)      # every later view keeps `flat` alive as its base
    weakref.finalize(flat, L.c3r_io_free, p)
    return flat.view(dtype)


class BamFile:
    """Open BAM; `fetch(contig, beg0, end0)` returns the alignments overlapping the 0-based half-open region as a
    ReadSet in file order (whole contig when end0 is None).  Uses `<bam>.bai` / `<stem>.bai` when present."""

    def __init__(self, path, threads=0):
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.c3r_bam_open(os.fsencode(path), threads, C.byref(self.h))
        if rc != 0:
            msg = self.L.c3r_bam_last_error(self.h).decode() if self.h else "cannot open %s" % path
            self.close()
            raise IOError("c3r_bam_open: %s" % msg)

    def close(self):
        if self.h:
            self.L.c3r_bam_close(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def has_index(self):
        return bool(self.L.c3r_bam_has_index(self.h))

    def contigs(self):
        out = []
        for i in range(self.L.c3r_bam_n_contigs(self.h)):
            name, ln = C.c_char_p(), C.c_int64()
            self.L.c3r_bam_contig(self.h, i, C.byref(name), C.byref(ln))
            out.append((name.value.decode(), ln.value))
        return out

    def contig_weights(self):
        """{contig: (mapped reads, compressed bytes)} from the index; -1 where it does not say (no index / no metadata pseudo-bin)."""
        out = {}
        for i, (name, _ln) in enumerate(self.contigs()):
            nm, fb = C.c_int64(-1), C.c_int64(-1)
            self.L.c3r_bam_contig_weight(self.h, i, C.byref(nm), C.byref(fb))
            out[name] = (nm.value, fb.value)
        return out

    def fetch(self, contig, beg0=0, end0=None):
        n, nc, ns = C.c_int64(), C.c_int64(), C.c_int64()
        rc = self.L.c3r_bam_fetch(self.h, contig.encode(), int(beg0), int(end0) if end0 else 0, C.byref(n), C.byref(nc), C.byref(ns))
        if rc != 0:
            raise IOError("c3r_bam_fetch: %s" % self.L.c3r_bam_last_error(self.h).decode())
        reads = huge_empty(n.value, READ_DTYPE)
        cigar = huge_empty(nc.value, np.uint32)
        seq = huge_empty(ns.value, np.uint8)
        self.L.c3r_bam_copy(self.h, reads.ctypes.data, cigar.ctypes.data, seq.ctypes.data)
        return ReadSet(reads, cigar, seq)


def index_build(bam_path, bai_path=None):
    """`samtools index` equivalent: writes <bam>.bai."""
    bai_path = bai_path or bam_path + ".bai"
    rc = load_library().c3r_bam_index_build(os.fsencode(bam_path), os.fsencode(bai_path))
    if rc != 0:
        raise IOError("c3r_bam_index_build(%s) failed with %d (is the file coordinate-sorted?)" % (bam_path, rc))
    return bai_path


def _text_ptr(x):
    """(address, length, keep-alive) of bytes or a uint8 array."""
    if isinstance(x, np.ndarray):
        a = np.ascontiguousarray(x, dtype=np.uint8)
        return a.ctypes.data, a.size, a
    return C.cast(C.c_char_p(x), C.c_void_p).value, len(x), x


def vcf_merge(rows, qual=2, show_ref=False, edits=None, want_no_tagging=False):
    """c3r_vcf_merge on the records of one contig (bytes, or a uint8 array — then the results are arrays too: no copies are made
    on this side).  edits: [(pos, ref, alt), ...] REDIportal entries of the contig or
    None.  -> (merged bytes, merged-without-tagging bytes or None, (n_read, n_kept, n_tagged))."""
    L = load_library()
    n_edit = len(edits) if edits else 0
    if n_edit:
        edits = sorted(edits)
        epos = np.ascontiguousarray([e[0] for e in edits], dtype=np.int32)
        eref = (C.c_char_p * n_edit)(*[e[1].encode() for e in edits])
        ealt = (C.c_char_p * n_edit)(*[e[2].encode() for e in edits])
        epos_p = epos.ctypes.data
    else:
        eref = ealt = None
        epos_p = None
    n, n_nt, counts = C.c_int64(), C.c_int64(), (C.c_int64 * 3)()
    nt_len = C.byref(n_nt) if want_no_tagging else None
    as_array = isinstance(rows, np.ndarray)
    rows_p, rows_n, _keep = _text_ptr(rows)
    cap = rows_n + 16 * (n_edit + 1) + 64                                    # a relabel adds at most 7 bytes to a record
    out = huge_empty(cap, np.uint8)                                          # (not zero-filled: a large contig's records are ~100 MB)
    out_nt = huge_empty(cap, np.uint8) if want_no_tagging else None
    rc = L.c3r_vcf_merge(rows_p, rows_n, int(qual or 0), int(bool(show_ref)), epos_p, eref, ealt, n_edit, out.ctypes.data, cap, C.byref(n),
                         out_nt.ctypes.data if want_no_tagging else None, cap if want_no_tagging else 0, nt_len, counts)
    if rc == -6:                                                             # C3R_EOVERFLOW: sizes are known now
        out = np.empty(n.value + 1, np.uint8)
        out_nt = np.empty(n_nt.value + 1, np.uint8) if want_no_tagging else None
        rc = L.c3r_vcf_merge(rows_p, rows_n, int(qual or 0), int(bool(show_ref)), epos_p, eref, ealt, n_edit, out.ctypes.data, out.size, C.byref(n),
                             out_nt.ctypes.data if want_no_tagging else None, out_nt.size if want_no_tagging else 0, nt_len, counts)
    if rc != 0:
        raise IOError("c3r_vcf_merge failed with %d (malformed VCF record?)" % rc)
    if as_array:
        return out[:n.value], (out_nt[:n_nt.value] if want_no_tagging else None), tuple(counts)
    return out[:n.value].tobytes(), (out_nt[:n_nt.value].tobytes() if want_no_tagging else None), tuple(counts)


def vcf_compress(path, threads=0):
    """bgzip + tabix: <path> -> <path>.gz + <path>.gz.tbi, <path> removed."""
    rc = load_library().c3r_vcf_compress(os.fsencode(path), threads)
    if rc != 0:
        raise IOError("c3r_vcf_compress(%s) failed with %d" % (path, rc))
    return path + ".gz"


def fasta_fetch(ref_fn, fai_row, beg0=0, end0=None, upper=True, threads=0):
    """Bases [beg0, end0) of one contig of an uncompressed faidx-indexed FASTA as a uint8 array (c3r_fasta_fetch: parallel pread,
    line ends dropped, upper-cased) — `samtools faidx` without the subprocess or a pass in Python.  fai_row: (name, length, offset,
    linebases, linewidth) as io.read_fai yields them."""
    _name, length, offset, linebases, linewidth = fai_row
    beg0 = max(0, int(beg0))
    end0 = length if end0 is None else min(length, int(end0))
    out = huge_empty(max(0, end0 - beg0), np.uint8)
    if out.size:
        L = load_library()
        L.c3r_fasta_fetch.argtypes = [C.c_char_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        rc = L.c3r_fasta_fetch(os.fsencode(ref_fn), offset, linebases, linewidth, beg0, end0, int(upper), threads, out.ctypes.data)
        if rc != 0:
            raise IOError("c3r_fasta_fetch(%s, %s): the file does not match its .fai" % (ref_fn, _name))
    return out


class VcfPiece(object):
    """A run of whole lines compressed and indexed on its own (c3r_vcfz_piece_make; any thread) for VcfGzWriter.append."""

    def __init__(self, text, threads=0):
        L = load_library()
        L.c3r_vcfz_piece_make.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        L.c3r_vcfz_piece_free.argtypes = [C.c_void_p]
        L.c3r_vcfz_piece_free.restype = None
        ptr, n, _keep = _text_ptr(text.encode() if isinstance(text, str) else text)
        self.L, self.h, self.n = L, C.c_void_p(), n
        rc = L.c3r_vcfz_piece_make(ptr, n, threads, C.byref(self.h))
        if rc != 0:
            raise IOError("c3r_vcfz_piece_make failed with %d (text must end in a newline)" % rc)

    def free(self):
        if self.h:
            self.L.c3r_vcfz_piece_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class VcfGzWriter(object):
    """Streaming bgzip + tabix (c3r_vcfz_*): write() takes newline-terminated text in file order; close() leaves <path> (BGZF) and
    <path>.tbi — the bytes vcf_compress makes of the concatenated text; discard() removes what was written."""

    def __init__(self, gz_path, threads=0):
        L = load_library()
        L.c3r_vcfz_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.c3r_vcfz_write.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.c3r_vcfz_close.argtypes = [C.c_void_p, C.c_int]
        self.L, self.path, self.h = L, gz_path, C.c_void_p()
        rc = L.c3r_vcfz_open(os.fsencode(gz_path), threads, C.byref(self.h))
        if rc != 0:
            raise IOError("c3r_vcfz_open(%s) failed with %d" % (gz_path, rc))

    def write(self, text):
        ptr, n, _keep = _text_ptr(text.encode() if isinstance(text, str) else text)
        if not n:
            return
        rc = self.L.c3r_vcfz_write(self.h, ptr, n)
        if rc != 0:
            raise IOError("c3r_vcfz_write(%s) failed with %d (text must end in a newline)" % (self.path, rc))

    def append(self, piece):
        """A VcfPiece, in file order; the piece is consumed."""
        self.L.c3r_vcfz_append.argtypes = [C.c_void_p, C.c_void_p]
        try:
            rc = self.L.c3r_vcfz_append(self.h, piece.h)
        finally:
            piece.free()
        if rc != 0:
            raise IOError("c3r_vcfz_append(%s) failed with %d" % (self.path, rc))

    def _end(self, keep):
        if self.h:
            h, self.h = self.h, None
            rc = self.L.c3r_vcfz_close(h, int(keep))
            if rc != 0:
                raise IOError("c3r_vcfz_close(%s) failed with %d" % (self.path, rc))

    def close(self):
        self._end(True)

    def discard(self):
        self._end(False)
