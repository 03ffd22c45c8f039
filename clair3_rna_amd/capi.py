"""ctypes binding of libc3r.so (include/c3r.h).  There is no CPU fallback: if the HIP library is
missing or no MI355X is visible, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

from .reads import READ_DTYPE, ReadSet

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C3R_LIB") or os.path.join(HERE, "libc3r.so")        # C3R_LIB: A/B builds of the kernels (development)

SITE_DTYPE = np.dtype([("pos", "<i4"), ("depth", "<i4"), ("ref33", "S36"), ("n_tok", "<i4"), ("tok_off", "<u4")], align=True)
TOKEN_DTYPE = np.dtype([("read_idx", "<u4"), ("indel", "<i4"), ("qpos", "<u4"), ("base", "u1"), ("rev", "u1"), ("del_after", "<u2")],
                       align=True)
PADINS_DTYPE = np.dtype([("read_idx", "<u4"), ("qpos", "<u4"), ("n_bases", "<u4"), ("total", "<u4"), ("pad_mask", "<u8")], align=True)
assert SITE_DTYPE.itemsize == 52 and TOKEN_DTYPE.itemsize == 16 and PADINS_DTYPE.itemsize == 24

C3R_ERRORS = {-1: "EINVAL", -2: "ENODEVICE", -3: "EHIP", -4: "ENOMEM", -5: "EUNSUPPORTED", -6: "EOVERFLOW"}


class Params(C.Structure):
    _fields_ = [("channels", C.c_int32), ("min_mq", C.c_int32), ("excl_flags", C.c_int32), ("min_coverage", C.c_int32),
                ("snp_min_af", C.c_double), ("indel_min_af", C.c_double), ("head_tail", C.c_int32),
                ("splice_padding", C.c_int32), ("genotyping_mode", C.c_int32), ("max_depth_rescale", C.c_int32),
                ("max_depth", C.c_int32), ("mpileup_compat", C.c_int32)]


class C3RError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libc3r: %s (%s)" % (msg, C3R_ERRORS.get(code, code)))
        self.code = code


EXPORTS = ["c3r_version", "c3r_create", "c3r_destroy", "c3r_trim", "c3r_last_error", "c3r_synchronize", "c3r_stream",
           "c3r_default_params", "c3r_set_params", "c3r_load_reads", "c3r_host_alloc", "c3r_host_free", "c3r_set_reference", "c3r_set_reference_view", "c3r_set_bed", "c3r_set_sites",
           "c3r_pileup_scan", "c3r_pileup_scan_regions", "c3r_batch_begin", "c3r_batch_end", "c3r_batch_count", "c3r_get_tensors", "c3r_get_sites", "c3r_token_count", "c3r_get_tokens", "c3r_get_pad_insertions", "c3r_get_columns",
           "c3r_weight_count", "c3r_load_weights", "c3r_set_precision", "c3r_get_precision", "c3r_get_precision_guard", "c3r_reserve", "c3r_infer", "c3r_get_probs", "c3r_call_rows", "c3r_get_rows", "c3r_rows_begin", "c3r_rows_begin_ex", "c3r_rows_decode", "c3r_rows_get", "c3r_rows_free", "c3r_decode_text", "c3r_set_profiling", "c3r_reset_kernel_stats",
           "c3r_get_kernel_stats"]

_lib = None


def load_library():
    """dlopen libc3r.so and declare prototypes.  Raises ImportError when the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libc3r.so not found at %s — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    L.c3r_version.restype = C.c_char_p
    L.c3r_create.argtypes = [i32, vp, C.POINTER(vp)]
    L.c3r_destroy.argtypes = [vp]
    L.c3r_destroy.restype = None
    L.c3r_trim.argtypes = []
    L.c3r_trim.restype = C.c_int64
    L.c3r_last_error.argtypes = [vp]
    L.c3r_last_error.restype = C.c_char_p
    L.c3r_synchronize.argtypes = [vp]
    L.c3r_stream.argtypes = [vp]
    L.c3r_stream.restype = vp
    L.c3r_default_params.argtypes = [C.POINTER(Params)]
    L.c3r_default_params.restype = None
    L.c3r_set_params.argtypes = [vp, C.POINTER(Params)]
    L.c3r_load_reads.argtypes = [vp, vp, i64, vp, i64, vp, i64]
    L.c3r_host_alloc.argtypes = [C.c_size_t]
    L.c3r_host_alloc.restype = vp
    L.c3r_host_free.argtypes = [vp]
    L.c3r_host_free.restype = None
    L.c3r_set_reference.argtypes = [vp, i64, vp, i64]
    L.c3r_set_reference_view.argtypes = [vp, i64, vp, i64]
    L.c3r_set_bed.argtypes = [vp, i32, vp, i64]
    L.c3r_set_sites.argtypes = [vp, vp, i64]
    L.c3r_pileup_scan.argtypes = [vp, i64, i64, C.POINTER(i64)]
    L.c3r_pileup_scan_regions.argtypes = [vp, C.c_int32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.c3r_batch_begin.argtypes = [vp]
    L.c3r_batch_end.argtypes = [vp]
    L.c3r_batch_count.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.c3r_get_tensors.argtypes = [vp, i32, vp, i64]
    L.c3r_get_sites.argtypes = [vp, vp, i64]
    L.c3r_token_count.argtypes = [vp, C.POINTER(i64)]
    L.c3r_get_tokens.argtypes = [vp, vp, i64]
    L.c3r_get_pad_insertions.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.c3r_get_columns.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), vp, vp, vp, i64]
    L.c3r_weight_count.argtypes = [i32]
    L.c3r_weight_count.restype = i64
    L.c3r_load_weights.argtypes = [vp, vp, i64, i32]
    L.c3r_set_precision.argtypes = [vp, i32]
    L.c3r_get_precision.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_double)]
    L.c3r_get_precision_guard.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(i32)]
    L.c3r_infer.argtypes = [vp, vp, i64, vp]
    L.c3r_reserve.argtypes = [vp, i64]
    L.c3r_get_probs.argtypes = [vp, vp, i64]
    L.c3r_call_rows.argtypes = [vp, C.c_char_p, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.c3r_get_rows.argtypes = [vp, vp, i64]
    L.c3r_rows_begin.argtypes = [vp, C.POINTER(vp)]
    L.c3r_rows_begin_ex.argtypes = [vp, i32, vp, vp, C.POINTER(vp)]
    L.c3r_rows_decode.argtypes = [vp, C.c_char_p, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    L.c3r_rows_get.argtypes = [vp, vp, i64]
    L.c3r_rows_free.argtypes = [vp]
    L.c3r_rows_free.restype = None
    L.c3r_decode_text.argtypes = [C.c_char_p, i64, vp, vp, i32, C.POINTER(C.c_char_p), vp, i32, i32, vp, i64, C.POINTER(i64)]
    L.c3r_set_profiling.argtypes = [vp, i32]
    L.c3r_reset_kernel_stats.argtypes = [vp]
    L.c3r_get_kernel_stats.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(i64), i32, C.POINTER(i32)]
    _lib = L
    return L


PRECISIONS = {"f32": 0, "f16x3": 1, "f16+f8": 2, "auto": 3}          # c3r_set_precision modes


def env_precision(default="f16x3"):
    """The C3R_PRECISION environment default of the drivers' --gpu_precision flag, validated like the flag itself (argparse does
    not run `choices` on defaults)."""
    v = os.environ.get("C3R_PRECISION", default)
    if v not in PRECISIONS:
        raise SystemExit("C3R_PRECISION=%r: must be one of %s" % (v, ", ".join(sorted(PRECISIONS))))
    return v


def default_params():
    p = Params()
    load_library().c3r_default_params(C.byref(p))
    return p


def pinned_copy(a):
    """A copy of numpy array `a` in page-locked host memory (c3r_host_alloc): uploads from it are asynchronous DMA transfers.
    The block is freed when the last array that views it is collected."""
    import weakref
    a = np.ascontiguousarray(a)
    L = load_library()
    nbytes = max(1, a.nbytes)
    p = L.c3r_host_alloc(nbytes)
    if not p:
        raise MemoryError("c3r_host_alloc(%d) failed" % nbytes)
    flat = np.frombuffer((C.c_char * nbytes).from_address(p), dtype=np.uint8)      # every later view keeps `flat` alive as its base
    weakref.finalize(flat, L.c3r_host_free, p)
    out = flat[:a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def pinned_readset(rs):
    """ReadSet with its three arrays in page-locked memory."""
    out = ReadSet.__new__(ReadSet)
    out.reads, out.cigar, out.seq = pinned_copy(rs.reads), pinned_copy(rs.cigar), pinned_copy(rs.seq)
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def trim():
    """Give the device blocks libc3r.so keeps between contexts back to the driver (c3r_trim); returns the bytes released."""
    return int(load_library().c3r_trim())


class Engine(object):
    """One GPU context: tensor build (A1-A5) + network forward (A6/A7) on one MI355X."""

    def __init__(self, device=0, stream=None):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.c3r_create(device, stream, C.byref(h))
        if rc != 0:
            raise C3RError(rc, "c3r_create(device=%d) failed: no usable MI355X/HIP device (no CPU fallback)" % device)
        self.h = h
        self.params = default_params()
        self.n_candidates = 0
        self._keep = []
        import weakref
        self._snaps = weakref.WeakSet()          # row snapshots must be released before the context goes

    def close(self):
        if getattr(self, "h", None):
            for sn in list(getattr(self, "_snaps", ())):
                sn.free()
            self.L.c3r_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise C3RError(rc, self.L.c3r_last_error(self.h).decode())

    # ---- configuration / inputs
    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise KeyError(k)
            setattr(self.params, k, v)
        self._chk(self.L.c3r_set_params(self.h, C.byref(self.params)))

    def load_reads(self, rs):
        assert isinstance(rs, ReadSet)
        self.readset = rs
        self._chk(self.L.c3r_load_reads(self.h, _ptr(rs.reads), len(rs.reads), _ptr(rs.cigar), len(rs.cigar), _ptr(rs.seq), len(rs.seq)))

    def set_reference(self, ref_start, seq, upper_view=False):
        """seq: str, bytes, or a uint8 array (bamio.fasta_fetch) — an array is handed over by pointer, no copy on this side.
        upper_view: the array is upper-cased already and is used in place (c3r_set_reference_view); this object and the row
        snapshots taken from it keep the array alive for as long as the library reads it."""
        if isinstance(seq, np.ndarray):
            b = np.ascontiguousarray(seq, dtype=np.uint8)
            self.ref_start, self._ref_bytes, self._ref_upper = ref_start, b, None
            self._chk((self.L.c3r_set_reference_view if upper_view else self.L.c3r_set_reference)(self.h, ref_start, b.ctypes.data, b.size))
            return
        b = seq.encode() if isinstance(seq, str) else (seq if isinstance(seq, bytes) else bytes(seq))
        self.ref_start, self._ref_bytes, self._ref_upper = ref_start, b, None
        self._chk(self.L.c3r_set_reference(self.h, ref_start, C.cast(C.c_char_p(b), C.c_void_p), len(b)))

    @property
    def ref_seq(self):
        """Upper-cased reference slice as str (debug dumps only: 40 ms per 64 MB contig, so built on demand)."""
        if self._ref_upper is None:
            self._ref_upper = bytes(self._ref_bytes).decode().upper()
        return self._ref_upper

    def set_bed(self, which, intervals):
        a = np.ascontiguousarray(np.asarray(intervals if intervals is not None else [], dtype=np.int32).reshape(-1, 2))
        self._chk(self.L.c3r_set_bed(self.h, which, _ptr(a), len(a)))

    def set_sites(self, sites):
        a = np.ascontiguousarray(np.asarray(sites, dtype=np.int32))
        self._chk(self.L.c3r_set_sites(self.h, _ptr(a), len(a)))

    # ---- tensor build
    def scan(self, ctg_start, ctg_end):
        n = C.c_int64(0)
        self._chk(self.L.c3r_pileup_scan(self.h, ctg_start, ctg_end, C.byref(n)))
        tot = C.c_int64(0)
        self._chk(self.L.c3r_batch_count(self.h, C.byref(tot), None))
        self.n_candidates = tot.value          # resident candidates (== n outside batch mode)
        return n.value

    def scan_regions(self, regions):
        """All (ctg_start, ctg_end) regions — e.g. the chunks of a contig — in one set of launches; same candidates, in
        the same order, as successive scan() calls in batch mode."""
        starts = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
        ends = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        n = C.c_int64(0)
        self._chk(self.L.c3r_pileup_scan_regions(self.h, len(starts), starts.ctypes.data_as(C.POINTER(C.c_int64)),
                                                 ends.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n)))
        tot = C.c_int64(0)
        self._chk(self.L.c3r_batch_count(self.h, C.byref(tot), None))
        self.n_candidates = tot.value
        return n.value

    def begin_batch(self):
        """Scans append to the device-resident batch until end_batch(); infer() then covers all of them."""
        self._chk(self.L.c3r_batch_begin(self.h))
        self.n_candidates = 0

    def end_batch(self):
        self._chk(self.L.c3r_batch_end(self.h))

    def tensors(self, rescaled=True):
        n, Cc = self.n_candidates, self.params.channels
        out = np.zeros((n, 33, Cc), dtype=np.int32)
        if n:
            self._chk(self.L.c3r_get_tensors(self.h, int(rescaled), _ptr(out), n))
        return out

    def sites(self):
        out = np.zeros(self.n_candidates, dtype=SITE_DTYPE)
        if self.n_candidates:
            self._chk(self.L.c3r_get_sites(self.h, _ptr(out), len(out)))
        return out

    def tokens(self):
        n = C.c_int64(0)
        self._chk(self.L.c3r_token_count(self.h, C.byref(n)))
        out = np.zeros(n.value, dtype=TOKEN_DTYPE)
        if n.value:
            self._chk(self.L.c3r_get_tokens(self.h, _ptr(out), len(out)))
        return out

    def pad_insertions(self):
        """mpileup_compat = 1: the loaded reads' insertions that hold pads (PADINS_DTYPE; empty for aligner-made CIGARs) —
        altinfo.format_lines needs them beside the tokens to print `+3T*T` alleles."""
        n = C.c_int64(0)
        self._chk(self.L.c3r_get_pad_insertions(self.h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=PADINS_DTYPE)
        if n.value:
            self._chk(self.L.c3r_get_pad_insertions(self.h, _ptr(out), n.value, C.byref(n)))
        return out

    def columns(self):
        rs, n = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.c3r_get_columns(self.h, C.byref(rs), C.byref(n), None, None, None, 0))
        Cc = self.params.channels
        cols = np.zeros((n.value, Cc), dtype=np.int32)
        depth = np.zeros(n.value, dtype=np.int32)
        flags = np.zeros(n.value, dtype=np.uint8)
        self._chk(self.L.c3r_get_columns(self.h, C.byref(rs), C.byref(n), _ptr(cols), _ptr(depth), _ptr(flags), n.value))
        return dict(region_start=rs.value, cols=cols, depth=depth, flags=flags)

    # ---- network
    def load_weights(self, blob, channels=None):
        channels = channels or self.params.channels
        w = np.ascontiguousarray(blob, dtype=np.float32)
        self._chk(self.L.c3r_load_weights(self.h, _ptr(w), w.size, channels))

    PRECISIONS = PRECISIONS

    def set_precision(self, mode):
        """'f32' (fp32 MFMA), 'f16x3' (split-f16, fp32-equivalent; default), 'f16+f8' (f16 main term + fp8 corrections, opt-in) or
        'auto' ('f16+f8' where a calibration run through the loaded weights agrees with 'f16x3' to 4e-5, else 'f16x3')."""
        self._chk(self.L.c3r_set_precision(self.h, self.PRECISIONS[mode]))

    def precision(self):
        """(mode in use, calibration max |dP| or -1.0)."""
        m, e = C.c_int32(0), C.c_double(-1.0)
        self._chk(self.L.c3r_get_precision(self.h, C.byref(m), C.byref(e)))
        return {v: k for k, v in self.PRECISIONS.items()}[m.value], e.value

    def precision_guard(self):
        """dict(f16_err = max |dP| of split-f16 against the fp32 MFMA path on the calibration windows (-1 before weights are loaded),
        scale_log2 = the power-of-two scales of LSTM 1 / LSTM 2 / L4, fell_back = the guard sent the request to the fp32 path)."""
        e, sc, fb = C.c_double(-1.0), (C.c_int32 * 3)(), C.c_int32(0)
        self._chk(self.L.c3r_get_precision_guard(self.h, C.byref(e), sc, C.byref(fb)))
        return dict(f16_err=e.value, scale_log2=[int(v) for v in sc], fell_back=bool(fb.value))

    def infer(self, tensors=None, n=None, fetch=True):
        if tensors is None:
            n = self.n_candidates if n is None else n
            x = None
        else:
            x = np.ascontiguousarray(tensors, dtype=np.int32)
            n = x.shape[0]
        probs = np.zeros((n, 24), dtype=np.float32) if fetch else None
        self._chk(self.L.c3r_infer(self.h, _ptr(x), n, _ptr(probs)))
        return probs

    def fetch_probs(self, n):
        """Probabilities of the last infer(fetch=False): waits for the context's stream, then copies [n][24]."""
        probs = np.zeros((n, 24), dtype=np.float32)
        self._chk(self.L.c3r_get_probs(self.h, _ptr(probs), n))
        return probs

    def call_rows_text(self, ctg, qual=2, show_ref=True):
        """A8 on host threads (C++): the VCF rows of the resident candidates as ONE bytes object (newline-terminated rows),
        ready to be appended to the chunk's VCF; returns (text, n_rows).  After infer()."""
        n, nr = C.c_int64(0), C.c_int64(0)
        self._chk(self.L.c3r_call_rows(self.h, ctg.encode(), -1 if qual is None else int(qual), int(show_ref), C.byref(n), C.byref(nr)))
        buf = C.create_string_buffer(n.value + 1)
        self._chk(self.L.c3r_get_rows(self.h, buf, n.value + 1))
        return buf.raw[:n.value], nr.value

    def reserve(self, n_sites):
        """Size the network's device buffers for batches of up to n_sites candidates now rather than in the first infer()."""
        self._chk(self.L.c3r_reserve(self.h, int(n_sites)))

    def rows_begin(self, drop_ref_calls=False, host_reads=True):
        """Detach the decode inputs of the resident batch (sites, tokens, probabilities, read bases; after infer()) into a host
        snapshot: the engine is free for the next contig, RowSnapshot.decode() may run on any thread.
        drop_ref_calls: only the sites that can print a row WITHOUT show_ref leave the device (the decoder's early RefCall exit is applied
        there; decode such a snapshot with show_ref=False only).  host_reads: the decoder reads inserted bases from the ReadSet handed to
        load_reads (kept alive by the snapshot) instead of from a copy fetched back from the device."""
        h = C.c_void_p()
        rs = getattr(self, "readset", None) if host_reads else None
        if rs is not None and not (rs.reads.flags.c_contiguous and rs.seq.flags.c_contiguous and rs.reads.dtype == READ_DTYPE and len(rs.reads)):
            rs = None
        self._chk(self.L.c3r_rows_begin_ex(self.h, int(bool(drop_ref_calls)), _ptr(rs.reads) if rs is not None else None,
                                           _ptr(rs.seq) if rs is not None else None, C.byref(h)))
        snap = RowSnapshot(self.L, h)
        snap._ref_keep = getattr(self, "_ref_bytes", None)       # (c3r_set_reference_view: the decoder reads this array)
        snap._rs_keep = rs                                       # (c3r_rows_begin_ex: and these)
        snap.dropped_ref_calls = bool(drop_ref_calls)
        self._snaps.add(snap)
        return snap

    def call_rows(self, ctg, qual=2, show_ref=True):
        """The same as a list of row strings (tests; 200 k Python strings cost more than producing the rows)."""
        text, _ = self.call_rows_text(ctg, qual, show_ref)
        return text.decode().split("\n")[:-1] if text else []

    # ---- measurement
    def synchronize(self):
        self._chk(self.L.c3r_synchronize(self.h))

    def stream(self):
        return self.L.c3r_stream(self.h)

    def set_profiling(self, on):
        self._chk(self.L.c3r_set_profiling(self.h, int(on)))

    def reset_kernel_stats(self):
        self._chk(self.L.c3r_reset_kernel_stats(self.h))

    def kernel_stats(self):
        cap = 64
        names = (C.c_char_p * cap)()
        ms = (C.c_double * cap)()
        cnt = (C.c_int64 * cap)()
        n = C.c_int(0)
        self._chk(self.L.c3r_get_kernel_stats(self.h, names, ms, cnt, cap, C.byref(n)))
        return {names[i].decode(): dict(total_ms=ms[i], launches=cnt[i]) for i in range(min(n.value, cap))}


class RowSnapshot(object):
    """Host-side decode inputs of one batch (c3r_rows_begin); decode() needs no GPU and no engine."""

    def __init__(self, L, h):
        self.L, self.h, self._ref_keep, self._rs_keep, self.dropped_ref_calls = L, h, None, None, False

    def decode(self, ctg, qual=2, show_ref=True, as_array=False):
        """-> (bytes of newline-terminated VCF rows, number of rows); releases the snapshot.  as_array: the rows as a uint8 array
        instead (a large contig's rows are ~100 MB: turning them into a bytes object is a copy made with the GIL held, which
        stalls every other thread of a whole-sample run)."""
        if show_ref and self.dropped_ref_calls:
            raise ValueError("this snapshot was taken without the RefCall sites (rows_begin(drop_ref_calls=True)): decode it with show_ref=False")
        try:
            n, nr = C.c_int64(0), C.c_int64(0)
            rc = self.L.c3r_rows_decode(self.h, ctg.encode(), -1 if qual is None else int(qual), int(show_ref), C.byref(n), C.byref(nr))
            if rc != 0:
                raise C3RError(rc, "c3r_rows_decode failed")
            from . import bamio
            buf = bamio.huge_empty(n.value + 1, np.uint8)
            rc = self.L.c3r_rows_get(self.h, buf.ctypes.data, n.value + 1)
            if rc != 0:
                raise C3RError(rc, "c3r_rows_get failed")
            return (buf[:n.value] if as_array else buf[:n.value].tobytes()), nr.value
        finally:
            self.free()

    def free(self):
        if self.h:
            self.L.c3r_rows_free(self.h)
            self.h = None
        self._ref_keep = None
        self._rs_keep = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def decode_text(ctg, positions, ref33_list, alt_info_list, probs, qual=2, show_ref=True):
    """The C++ decoder on caller-supplied text (no GPU needed): list of VCF rows."""
    L = load_library()
    n = len(positions)
    pos = np.ascontiguousarray(positions, dtype=np.int32)
    r33 = np.zeros((n, 36), dtype="S1")
    refs = np.frombuffer(b"".join(r.encode().ljust(36, b"\0") for r in ref33_list), dtype=np.uint8).copy() if n else np.zeros(0, np.uint8)
    alts = (C.c_char_p * max(n, 1))(*[a.encode() for a in alt_info_list])
    pr = np.ascontiguousarray(probs, dtype=np.float32)
    need = C.c_int64(0)
    rc = L.c3r_decode_text(ctg.encode(), n, _ptr(pos), _ptr(refs), 36, alts, _ptr(pr), -1 if qual is None else int(qual), int(show_ref), None, 0,
                           C.byref(need))
    buf = C.create_string_buffer(need.value + 1)
    rc = L.c3r_decode_text(ctg.encode(), n, _ptr(pos), _ptr(refs), 36, alts, _ptr(pr), -1 if qual is None else int(qual), int(show_ref), buf,
                           need.value + 1, C.byref(need))
    if rc != 0:
        raise C3RError(rc, "c3r_decode_text failed")
    text = buf.value.decode()
    return text.split("\n")[:-1] if text else []
