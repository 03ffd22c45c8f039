"""ctypes binding of the CPU test oracle (oracle/c3r_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (clair3_rna_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libc3r_oracle.so")

READ_DTYPE = np.dtype([("pos", "<i4"), ("cigar_off", "<u4"), ("n_cigar", "<u4"), ("l_seq", "<u4"),
                       ("seq_off", "<u8"), ("flag", "<u2"), ("mapq", "u1"), ("hp", "u1"), ("reserved", "<u4")],
                      align=True)
assert READ_DTYPE.itemsize == 32


def build(force=False):
    src = os.path.join(HERE, "c3r_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB


class CtParams(C.Structure):
    _fields_ = [("snp_af", C.c_double), ("indel_af", C.c_double), ("min_coverage", C.c_int32),
                ("head_tail", C.c_int32), ("splice_padding", C.c_int32), ("phased", C.c_int32),
                ("has_bed", C.c_int32), ("n_bed", C.c_int32), ("bed", C.POINTER(C.c_int32)),
                ("has_sites", C.c_int32), ("n_sites", C.c_int32), ("sites", C.POINTER(C.c_int32)),
                ("platform_hifi", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_mpileup.restype = C.c_void_p
        L.orc_mpileup.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.c_int64,
                                  C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.orc_mpileup_d.restype = C.c_void_p
        L.orc_mpileup_d.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.c_int64,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.orc_mpileup_c.restype = C.c_void_p
        L.orc_mpileup_c.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.c_int64,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]
        L.orc_generate_tensor.restype = C.c_void_p
        L.orc_generate_tensor.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_char,
                                          C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.orc_chunk_region.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        L.orc_create_tensor.restype = C.c_void_p
        L.orc_create_tensor.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, C.POINTER(CtParams), C.POINTER(C.c_int64)]
        L.orc_batch_from_lines.restype = C.c_int64
        L.orc_batch_from_lines.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_weight_count.restype = C.c_int64
        L.orc_weight_count.argtypes = [C.c_int]
        L.orc_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _take_str(ptr):
    s = C.string_at(ptr).decode()
    lib().orc_free(ptr)
    return s


def _arr_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def mpileup(reads, cigar, seq, ctg, beg1, end1, min_mq=5, excl_flags=2316, bed=None, with_hp=False, max_depth=8000, compat=0):
    """A1: reads -> mpileup text rows (list of str).  compat: 0 = samtools <= 1.10 printer, 1 = samtools >= 1.11 (`+<ins>-<del>`)."""
    reads = np.ascontiguousarray(reads, dtype=READ_DTYPE)
    cigar = np.ascontiguousarray(cigar, dtype=np.uint32)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    bedarr = None
    if bed is not None:
        bedarr = np.ascontiguousarray(np.asarray(sorted(bed), dtype=np.int32).reshape(-1, 2))
    n = C.c_int64(0)
    p = lib().orc_mpileup_c(_arr_ptr(reads), len(reads), _arr_ptr(cigar), _arr_ptr(seq), ctg.encode(), beg1, end1,
                            min_mq, excl_flags, _arr_ptr(bedarr), 0 if bedarr is None else len(bedarr), int(with_hp), int(max_depth), int(compat), C.byref(n))
    text = _take_str(p)
    rows = text.split("\n")
    if rows and rows[-1] == "":
        rows.pop()
    return rows


def generate_tensor(bases, ref_base, pos, ref_seq, ref_start, hp=None, snp_af=0.08, indel_af=0.15):
    """A2: one column.  Returns dict like the G1 'out' record."""
    t = np.zeros(30, dtype=np.int32)
    depth, pass_af, mdl, msk = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    af = C.c_double()
    hp_csv = None if hp is None else ",".join(hp).encode()
    p = lib().orc_generate_tensor(bases.encode(), hp_csv, pos, ref_seq.encode(), ref_start, ref_base.encode(),
                                  snp_af, indel_af, _arr_ptr(t), C.byref(depth), C.byref(pass_af), C.byref(mdl),
                                  C.byref(msk), C.byref(af))
    alt_s, pl_s = _take_str(p).split("\n")[:2]

    def kv(s):
        it = s.split()
        return [[it[i], int(it[i + 1])] for i in range(0, len(it), 2)]

    n = 30 if hp is not None else 18
    return dict(tensor=t[:n].tolist(), alt=kv(alt_s), af=af.value, depth=depth.value, pass_af=bool(pass_af.value),
                pileup_list=kv(pl_s), max_del_length=mdl.value, max_skip_count=msk.value)


def chunk_region(contig_len=0, chunk_id=0, chunk_num=1, bed_start=None, bed_end=None, ctg_start=0, ctg_end=0):
    io = np.array([ctg_start, ctg_end, 0, 0, 0, 0], dtype=np.int64)
    mode = 0 if bed_start is None else 1
    lib().orc_chunk_region(mode, contig_len, bed_start or 0, bed_end or 0, chunk_id, chunk_num, _arr_ptr(io))
    return dict(ctg_start=int(io[0]), ctg_end=int(io[1]), extend_start=int(io[2]), extend_end=int(io[3]),
                reference_start=int(io[4]), reference_end=int(io[5]))


def make_params(snp_af=0.08, indel_af=0.15, min_coverage=4, head_tail=False, splice_padding=False, phased=False,
                bed=None, sites=None, platform="ont"):
    P = CtParams()
    P.snp_af, P.indel_af, P.min_coverage = snp_af, indel_af, min_coverage
    P.head_tail, P.splice_padding, P.phased = int(head_tail), int(splice_padding), int(phased)
    keep = []
    if bed is not None:
        b = np.ascontiguousarray(np.asarray(bed, dtype=np.int32).reshape(-1, 2))
        keep.append(b)
        P.has_bed, P.n_bed, P.bed = 1, len(b), b.ctypes.data_as(C.POINTER(C.c_int32))
    if sites is not None:
        s = np.ascontiguousarray(np.asarray(sites, dtype=np.int32))
        keep.append(s)
        P.has_sites, P.n_sites, P.sites = 1, len(s), s.ctypes.data_as(C.POINTER(C.c_int32))
    P.platform_hifi = int(platform == "hifi")
    P._keep = keep
    return P


def create_tensor(rows, ctg, ref_seq, reference_start, params):
    """A3: mpileup rows -> candidate lines (list of str)."""
    text = "".join(r + "\n" for r in rows)
    n = C.c_int64(0)
    p = lib().orc_create_tensor(text.encode(), ctg.encode(), ref_seq.encode(), reference_start, C.byref(params), C.byref(n))
    out = _take_str(p).split("\n")
    if out and out[-1] == "":
        out.pop()
    return out


def batch_from_lines(lines, channels):
    """A5: candidate lines -> (int32 [n,33,C], depth[n])."""
    text = "".join(l + "\n" for l in lines)
    out = np.zeros((max(1, len(lines)), 33, channels), dtype=np.int32)
    depth = np.zeros(max(1, len(lines)), dtype=np.int32)
    n = lib().orc_batch_from_lines(text.encode(), channels, _arr_ptr(out), _arr_ptr(depth))
    return out[:n], depth[:n]


def weight_count(channels):
    return int(lib().orc_weight_count(channels))


def forward(weights, X, return_hidden=False):
    """A6: fp32 network forward.  weights: flat float32 blob (layout in c3r_oracle.c), X int32 [n,33,C]."""
    X = np.ascontiguousarray(X, dtype=np.int32)
    n, W, Cc = X.shape
    w = np.ascontiguousarray(weights, dtype=np.float32)
    assert w.size == weight_count(Cc), (w.size, weight_count(Cc))
    probs = np.zeros((n, 24), dtype=np.float32)
    y1 = np.zeros((n, 33, 256), dtype=np.float32) if return_hidden else None
    y2 = np.zeros((n, 33, 320), dtype=np.float32) if return_hidden else None
    lib().orc_forward(_arr_ptr(w), Cc, _arr_ptr(X), n, _arr_ptr(probs), _arr_ptr(y1), _arr_ptr(y2))
    return (probs, y1, y2) if return_hidden else probs
