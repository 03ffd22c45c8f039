"""ctypes binding of libc3r_io.so (include/c3r_io.h): BAM + .bai -> ReadSet.

The C++ reader replaces the BAM side of `samtools mpileup <bam> -r ctg:beg-end` (reference
src/create_tensor_pileup.py:436-451).  `bam.py` (pure Python) stays as the independent checker and the test writer."""
import ctypes as C
import os

import numpy as np

from .reads import READ_DTYPE, ReadSet

EXPORTS = ["c3r_bam_open", "c3r_bam_close", "c3r_bam_last_error", "c3r_bam_n_contigs", "c3r_bam_contig", "c3r_bam_has_index",
           "c3r_bam_fetch", "c3r_bam_copy", "c3r_bam_index_build"]
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
        L.c3r_bam_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64] + [C.POINTER(C.c_int64)] * 3
        L.c3r_bam_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.c3r_bam_index_build.argtypes = [C.c_char_p, C.c_char_p]
        _LIB = L
    return _LIB


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

    def fetch(self, contig, beg0=0, end0=None):
        n, nc, ns = C.c_int64(), C.c_int64(), C.c_int64()
        rc = self.L.c3r_bam_fetch(self.h, contig.encode(), int(beg0), int(end0) if end0 else 0, C.byref(n), C.byref(nc), C.byref(ns))
        if rc != 0:
            raise IOError("c3r_bam_fetch: %s" % self.L.c3r_bam_last_error(self.h).decode())
        reads = np.zeros(n.value, READ_DTYPE)
        cigar = np.zeros(nc.value, np.uint32)
        seq = np.zeros(ns.value, np.uint8)
        self.L.c3r_bam_copy(self.h, reads.ctypes.data, cigar.ctypes.data, seq.ctypes.data)
        return ReadSet(reads, cigar, seq)


def index_build(bam_path, bai_path=None):
    """`samtools index` equivalent: writes <bam>.bai."""
    bai_path = bai_path or bam_path + ".bai"
    rc = load_library().c3r_bam_index_build(os.fsencode(bam_path), os.fsencode(bai_path))
    if rc != 0:
        raise IOError("c3r_bam_index_build(%s) failed with %d (is the file coordinate-sorted?)" % (bam_path, rc))
    return bai_path
