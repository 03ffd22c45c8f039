// c3r_lib.hip — host side of libc3r.so: the C-ABI of include/c3r.h over the gfx950 kernels.
// No CPU fallback: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <queue>
#include <chrono>

#include <algorithm>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/c3r.h"
#include "decode.hpp"
#include "net_kernels.hpp"
#include "pileup_kernels.hpp"
#include "reads_kernels.hpp"
#include <sched.h>

using namespace c3r;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct KStat { double ms = 0; int64_t n = 0; };

}  // namespace

struct c3r_rows;
// host copies of a loaded contig's read headers and packed bases (the decoder reads inserted bases from them): made by the first row
// snapshot after c3r_load_reads and shared by every later one of the same contig — genotyping mode decodes a contig chunk by chunk
// (c3r_call_rows ~50 times for chr1) and used to copy 100+ MB each time
struct HostReads { std::vector<DevRead> reads; std::vector<uint8_t> seq; };
typedef std::vector<c3r_padins_t> PadInsTab;
struct c3r_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    // With C3R_TWO_STREAMS=1 (opt-in; by default everything runs on ONE stream) a context that owns its streams keeps TWO: `stream` (the
    // device's highest priority) for everything but the network, and `net_stream` (lowest priority) for A6's kernels, chained by events
    // inside c3r_infer.  A layer-2 workgroup owns its CU (8 waves x
    // 256 VGPRs, 148 KB of LDS: nothing co-resides), and with one priority for all queues the dispatcher did not start another
    // queue's kernel before the running LSTM kernel had handed out its last workgroup: the ~10 short kernels of another context's
    // load_reads + scan each waited out most of a 7-14 ms kernel (profiles/r4/prep_beside_network.txt: 75-110 ms beside a network
    // pass against 2 ms alone).  At high priority they take the CUs as LSTM workgroups retire (one lives ~1.2 ms).
    hipStream_t net_stream = nullptr;
    hipEvent_t ev_x = nullptr, ev_net = nullptr;
    std::string err;
    c3r_params_t prm;
    bool profiling = false;
    std::map<std::string, KStat> kstats;
    std::vector<std::string> kstat_names;   // stable storage for c3r_get_kernel_stats
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // k_fused_deep runs BESIDE k_fused_tiles on its own stream (created at the first fused scan), forked and joined by two events
    hipStream_t deep_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_up = nullptr;            // c3r_load_reads: "my uploads have landed" (the per-device upload gate)

    // ---- inputs: the device holds the read tables; host copies are fetched on demand (depth cap, decode)
    int32_t n_reads = 0;                   // of the loaded contig
    int64_t n_indel_ops = 0;               // I + D ops of the passing reads: bounds the indel-event scratch of a scan
    int64_t n_seq_bytes = 0, n_cigar_ops = 0;
    DevBuf d_rawreads, d_rawcig;           // the caller's records as they arrived (c3r_read_t, BAM-encoded ops)
    // the pile table (reads_kernels.hpp): bin counters {cnt | sc | ec | pc}, their prefix sums, the records, per-read notes of k_prep
    DevBuf d_bincnt, d_binoff, d_rtab, d_recs, d_serial, d_nind, d_lbk, d_stats;
    BinGeo bins{0, 0};
    int32_t first_pos = 0;                 // pos of the first read (the bins start there)
    int64_t last_pos = 0;                  // pos of the last read (first guess of where the bins end)
    bool bins_dirty = true;                // the counters are not all zero (failed load)
    size_t bins_zeroed = 0;                // bytes of d_bincnt known to be zero when !bins_dirty
    // legacy tables for token_at (30-channel mode only), built on demand: normalised CIGARs, aligned segments in read order
    DevBuf d_lcnt;
    bool legacy_valid = false;
    bool legacy_wanted = false;           // a fused 30-channel scan of this context has met a column whose haplotype channels depend on the reads' order
    LoadStats *h_stats = nullptr;          // pinned
    int32_t *h_lstm = nullptr;             // pinned: this context's copy of the layer-2 time-out word (queue_lstm_status)
    void *h_pack = nullptr;                // pinned: the indel-record count of k_pack_tokens
    // large copies from / to ORDINARY host memory go through two page-locked 8-MB buffers the context owns (big_h2d / big_d2h): the
    // runtime would pin the caller's pages for every copy and unpin them afterwards, which takes the process's memory-map lock — the
    // lock the BAM fetchers' page faults take as well (call_sample: 8-120 ms for a contig's records where the transfer is 3 ms)
    char *pin_buf[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    DevBuf d_tokb, d_tokrec, d_recoff;     // packed tokens of a row snapshot (c3r_rows_begin)
    DevBuf d_keep, d_sites_c, d_probs_c;   // c3r_rows_begin_ex(drop_ref_calls): keep flags / row numbers, the kept sites' records and probabilities
    std::vector<DevRead> h_reads;          // lazily: ensure_host_reads
    std::vector<int32_t> h_prefmax;        // lazily: host copy of the prefix max of read ends (passing reads)
    std::vector<uint32_t> h_zone_sc, h_zone_ec;   // depth_cap_mask: reads per 256-position bin by start / by exclusive end
    bool host_reads_valid = false;
    std::vector<uint8_t> h_seq;            // lazily: ensure_host_seq (decode reads inserted bases)
    bool host_seq_valid = false;
    DevBuf d_reads, d_cigar, d_seq, d_prefmax;
    DevBuf d_wgtab;                        // k_prep's per-workgroup bin tables between its two passes
    // mpileup_compat = 1: the insertions of the loaded reads that hold pads (c3r_padins_t), on the device for ev_equal and on the host for
    // the decoder (row snapshots share the vector); empty for every CIGAR an aligner emits
    DevBuf d_padins;
    std::shared_ptr<PadInsTab> padins;
    DevBuf d_aftab;                        // the AF gates as integer thresholds per depth (ScanArgs::af_tab), rebuilt when the AFs change
    double aftab_snp = -1.0, aftab_indel = -1.0;
    DevBuf d_dbg;                          // C3R_SCAN_DBG: phase timers of k_scan_tiles
    DevBuf d_tile_cand;                    // [n_tiles] {first candidate, count} of the most recent scan (k_compact_write -> k_tile_tokens)
    DevBuf d_tile_cols, d_tile_rng, d_tile_list, d_tile_list2, d_rsegs, d_rseg_first;
    ScanArgs last_scan;                    // arguments of the most recent scan (c3r_get_columns completes the pruned tiles with them)
    bool last_scan_pruned = false;
    // the fused path (k_fused_tiles): look-back words and counters, region bounds, what the last scan covered
    DevBuf d_lb, d_regb, d_span, d_spanbase, d_meta, d_spanrec, d_deep;       // (d_deep: list positions of the deep spans, k_fused_deep)
    DevBuf d_evwg;                         // k_fused_deep: an arrival-order event buffer per workgroup (DEEP_EVG_CAP records each), allocated once a scan has met a deep span
    bool seen_deep = false;
    DevBuf d_giant, d_giant_ev, d_giant_tab;            // k_deep_walk: the giant spans' accumulator slots (+ the slice list) and event buffers, allocated once a scan has met a giant span
    bool seen_giant = false;
    // the resident windows are int16 unless a scan of this read set has met a position that 32,768 reads or more cover (possible only above mpileup's
    // default cap, max_depth = 0 or > 32,767): that scan is repeated with int32 windows and the context keeps them from then on (a context that
    // has met such depth will meet it again: every pass over the same sample would otherwise scan twice)
    bool win32 = false;
    int n_cu = 0;                          // compute units of the device (k_fused_deep: one workgroup each)
    DevBuf d_winidx;                       // [resident candidates] row of the i-th site's window in d_tensors (the fused path writes windows as they arrive)
    DevBuf d_rawidx, d_export;             // c3r_get_tensors: index of a raw re-run, windows gathered into position order
    DevBuf d_tokexp, d_tokoff;             // c3r_get_tokens: tokens gathered into site order, their offsets there
    int32_t *h_scan = nullptr;             // pinned: what a fused scan reads back (totals, overflow flags)
    bool last_fused = false;
    std::vector<int64_t> last_starts, last_ends;
    // page-locked buffers for the upper-cased reference slice (upload source, decoder's view).  Three of them: a row snapshot
    // (c3r_rows_begin) keeps the buffer of ITS contig alive while the context already works on the next one
    struct RefBuf { char *p = nullptr; size_t cap = 0; std::atomic<int> users{0}; hipEvent_t ev = nullptr; };
    RefBuf refbuf[3];
    int ref_cur = -1;                                             // the slot c3r_set_reference filled last
    char *h_ref = nullptr; size_t ref_len = 0;                    // = refbuf[ref_cur].p
    int64_t ref_start1 = 1;
    std::mutex pool_mu;                                           // guards stage_pool (snapshots are released from other threads)
    std::vector<std::pair<void *, size_t>> stage_pool;            // staging blocks of released row snapshots (stage_pinned())
    c3r_rows *rows_snap = nullptr;                                // c3r_call_rows keeps its snapshot here for c3r_get_rows
    std::shared_ptr<HostReads> host_cache;                        // of the loaded contig (null until a snapshot needs it)
    DevBuf d_ref;
    std::vector<int32_t> h_bed[2];
    DevBuf d_bed[2];
    bool has_bed[2] = {false, false};
    std::vector<int32_t> h_sites;
    DevBuf d_sites;

    // ---- scan state
    int32_t reg_beg0 = 0, reg_end0 = 0;   // first region of the most recent scan (c3r_get_columns)
    int64_t n_pos = 0;                    // position slots of the most recent scan (all regions, tile-padded)
    int32_t max_cover = 0;                // upper bound of htslib's read list at any read's start (LoadStats): below max_depth the cap cannot bite
    std::vector<uint32_t> h_drop;         // depth cap of the most recent scan: [n_regions][drop_words] bit per read
    DevBuf d_drop;
    std::vector<TileGeo> h_geo;           // tile geometry of the most recent scan; re-uploaded only when it changes
    std::vector<int64_t> geo_key;         // the (start, end) list h_geo was built for
    DevBuf d_geo, d_lastrow;
    int32_t n_regions = 0;
    DevBuf d_cols, d_depth, d_ncov, d_flags, d_skipmax, d_ev, d_small /* cursor,last_row,totals */, d_blockcnt, d_scan_tops;
    int64_t n_cand = 0, n_tok = 0;        // totals resident on the device (all scans of the current batch)
    // rows of d_tensors / slots of d_tok taken by the batch so far: more than n_cand / n_tok when a large fused scan handed them out through
    // several sub-allocators (pileup_kernels.hpp, ALLOC_SHARDS) — sites reach their window and tokens through win_idx / tok_off
    int64_t n_rows = 0, n_tokspace = 0;
    int32_t hint_rows = 4096, hint_toks = 131072;      // per-shard sizes the last sharded scan needed (+ 25 %)
    int32_t last_shards = 1, last_shard_rows = 0;      // the most recent fused scan's row space (raw re-run of c3r_get_tensors)
    int64_t last_cand = 0, last_base = 0; // candidates of the most recent scan and their offset in the batch
    bool batching = false;
    DevBuf d_cand, d_tensors, d_raw, d_sites_out, d_tokcnt, d_tok;
    bool tokens_ready = false;

    // ---- network
    NetState net;
    int precision_req = 1;                 // what c3r_set_precision asked for (3 = auto); net.precision is what runs
    double mx_calib_err = -1.0;            // max |dP| of precision 2 against precision 1 on the calibration windows (-1: not measured)
    double f16_calib_err = -1.0;           // max |dP| of split-f16 against the fp32 MFMA path on the same windows, measured at c3r_load_weights
    bool f16_fell_back = false;            // the guard sent a split-f16 request to the fp32 path

    // ---- host decode (A8)
};

namespace {

// CPUs this process may run on (its affinity mask: one process per GPU is pinned to its share of the node, shard.host_budget), not the
// machine's: what default thread counts are taken from
static inline unsigned usable_cpus() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return (unsigned)n; }
    return std::max(1u, std::thread::hardware_concurrency());
}

int fail(c3r_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail(ctx, C3R_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// The layer-2 kernels meet their wavefronts through LDS counters with bounded waits (net_kernels.hpp, lds_wait): a wait that ever gives
// up raises the context's time-out word (NetState::d_tmo) instead of hanging the GPU.  Whoever hands probabilities to the host reads the
// word first — queued behind the network on the same stream, so it costs no extra synchronisation — and fails the call rather than pass
// on numbers computed from a half-written h_t.  The word belongs to the CONTEXT and c3r_infer clears it before it queues a network pass:
// the faulty batch fails (every fetch of its probabilities, until the next c3r_infer), a healthy context beside it does not, and a host
// application that embeds the library goes on with the next batch or a fresh context (round 4 kept one word per process and never cleared it).
int queue_lstm_status(c3r_ctx *ctx, int32_t **slot) {
    if (!ctx->h_lstm) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_lstm, 64, hipHostMallocDefault));
    *ctx->h_lstm = 0;
    if (ctx->net.d_tmo) HIPCHK(ctx, hipMemcpyAsync(ctx->h_lstm, ctx->net.d_tmo, 4, hipMemcpyDeviceToHost, ctx->stream));
    *slot = ctx->h_lstm;
    return C3R_OK;
}
int check_lstm_status(c3r_ctx *ctx, const int32_t *slot) {
    if (slot && *slot)
        return fail(ctx, C3R_EHIP, "internal: a layer-2 wavefront rendezvous timed out in this batch — its probabilities are not valid (the flag is cleared by the next c3r_infer of this context)");
    return C3R_OK;
}

// Debug aid (tests/test_gpu_poison.py): C3R_POISON=<byte> fills every fresh device allocation with that byte, so that a
// kernel reading memory nobody wrote shows up as a parity failure instead of depending on what the allocator handed out.
inline int poison_byte() {
    static const int v = [] { const char *e = getenv("C3R_POISON"); return e && *e ? atoi(e) & 0xff : -1; }();
    return v;
}

// Staging blocks of row snapshots (sites | tokens | probabilities of one batch, ~0.3 GB for a large contig): ordinary memory by
// default — the runtime stages the three copies (~30 ms for a large contig), but nothing has to be pinned on the first contigs and
// unpinned when the context goes (0.1 ms per MB each way: 22 full-length contigs ran 3.4 s against 3.6-3.8 s with page-locked
// blocks).  C3R_SNAPSHOT_PINNED=1 takes page-locked blocks.
inline bool stage_pinned() {
    static const bool v = [] { const char *e = getenv("C3R_SNAPSHOT_PINNED"); return e && *e == '1'; }();
    return v;
}
inline void stage_free(void *p) { if (!p) return; if (stage_pinned()) (void)hipHostFree(p); else free(p); }
// 2-MB aligned and advised huge (C3R_IO_HUGE=0: ordinary pages): a fresh 0.3-GB block is touched for the first time by the copies into it
// and handed back page by page when the context goes — 512 times fewer pages where transparent huge pages are available.  free() releases it.
inline void *huge_alloc(size_t bytes) {
    void *p = nullptr;
    const size_t cap = (std::max<size_t>(bytes, 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (posix_memalign(&p, (size_t)2 << 20, cap) != 0) return nullptr;
    static const bool huge = [] { const char *e = getenv("C3R_IO_HUGE"); return !(e && *e == '0'); }();
    if (huge) (void)madvise(p, cap, MADV_HUGEPAGE);
    return p;
}

int ensure(c3r_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return C3R_OK;
    size_t want = std::max(bytes, (size_t)256);
    want = want + want / 4;   // grow-only with slack so steady-state steps never reallocate
    const bool timing = getenv("C3R_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (b.p) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const auto t1 = std::chrono::steady_clock::now();
    if (hipMalloc(&b.p, want) != hipSuccess) {
        (void)hipGetLastError();
        b.p = nullptr;
        // (which buffer: its offset inside the context says so — a size in the hundreds of GB is a bad count read back from the device, not a full GPU)
        return fail(ctx, C3R_ENOMEM, "hipMalloc of %zu bytes (asked for: %zu) failed for the device buffer at context offset %zu", want, bytes, (size_t)((char *)&b - (char *)ctx));
    }
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        const double a = std::chrono::duration<double, std::milli>(t1 - t0).count(), m = std::chrono::duration<double, std::milli>(t2 - t1).count();
        if (a + m > 1.0) fprintf(stderr, "[ensure %p] %.1f MB: sync+free %.1f ms, malloc %.1f ms\n", (void *)ctx, want / 1e6, a, m);
    }
    if (poison_byte() >= 0) { HIPCHK(ctx, hipMemsetAsync(b.p, poison_byte(), want, ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }
    b.cap = want;
    return C3R_OK;
}

// grow a buffer but keep its first `used` bytes (batch mode appends scan after scan)
int ensure_keep(c3r_ctx *ctx, DevBuf &b, size_t bytes, size_t used) {
    if (bytes <= b.cap && b.p) return C3R_OK;
    if (!b.p || used == 0) return ensure(ctx, b, bytes);
    size_t want = bytes + bytes / 2;
    void *np_ = nullptr;
    HIPCHK(ctx, hipMalloc(&np_, want));
    if (poison_byte() >= 0) HIPCHK(ctx, hipMemsetAsync(np_, poison_byte(), want, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(np_, b.p, used, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipFree(b.p));
    b.p = np_; b.cap = want;
    return C3R_OK;
}

int big_h2d_fwd(c3r_ctx *ctx, void *dst, const void *src, size_t bytes);
template <typename T>
int upload(c3r_ctx *ctx, DevBuf &b, const T *src, size_t n) {
    int rc = ensure(ctx, b, std::max<size_t>(n * sizeof(T), 16));
    if (rc) return rc;
    if (n) return big_h2d_fwd(ctx, b.p, src, n * sizeof(T));
    return C3R_OK;
}

// ---- large copies between ordinary (pageable) host memory and the device, staged through the context's two page-locked buffers
constexpr size_t PIN_CHUNK = (size_t)8 << 20, PIN_MIN = (size_t)1 << 20;
bool pin_ring_on() {
    static const bool on = [] { const char *e = getenv("C3R_PIN_RING"); return !(e && *e == '0'); }();
    return on;
}
bool host_is_page_locked(const void *p) {
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
int pin_ring_init(c3r_ctx *ctx) {
    for (int k = 0; k < 2; ++k) {
        if (!ctx->pin_buf[k]) HIPCHK(ctx, hipHostMalloc((void **)&ctx->pin_buf[k], PIN_CHUNK, hipHostMallocDefault));
        if (!ctx->pin_ev[k]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->pin_ev[k], hipEventDisableTiming));
    }
    return C3R_OK;
}
// host -> device; on return the caller's bytes have been read (the last chunks may still be on their way to the device, in stream order)
int big_h2d(c3r_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return C3R_OK;
    if (bytes < PIN_MIN || !pin_ring_on() || host_is_page_locked(src)) {
        HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return C3R_OK;
    }
    int rc = pin_ring_init(ctx);
    if (rc) return rc;
    size_t off = 0;
    for (int i = 0; off < bytes; ++i) {
        const int k = i & 1;
        const size_t n = std::min(PIN_CHUNK, bytes - off);
        if (ctx->pin_busy[k]) HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[k]));          // the transfer out of this buffer is through
        memcpy(ctx->pin_buf[k], (const char *)src + off, n);
        HIPCHK(ctx, hipMemcpyAsync((char *)dst + off, ctx->pin_buf[k], n, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->pin_ev[k], ctx->stream));
        ctx->pin_busy[k] = true;
        off += n;
    }
    return C3R_OK;
}
// device -> host, complete on return (the transfer of chunk i runs beside the host copy of chunk i - 1)
int big_d2h(c3r_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return C3R_OK;
    if (bytes < PIN_MIN || !pin_ring_on() || host_is_page_locked(dst)) {
        HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return C3R_OK;
    }
    int rc = pin_ring_init(ctx);
    if (rc) return rc;
    size_t off = 0, prev_off = 0, prev_n = 0;
    int i = 0;
    for (; off < bytes; ++i) {
        const int k = i & 1;
        const size_t n = std::min(PIN_CHUNK, bytes - off);
        if (ctx->pin_busy[k]) HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[k]));          // (an upload that used the buffer before)
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin_buf[k], (const char *)src + off, n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->pin_ev[k], ctx->stream));
        ctx->pin_busy[k] = true;
        if (i > 0) {
            const int pk = k ^ 1;
            HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[pk]));
            memcpy((char *)dst + prev_off, ctx->pin_buf[pk], prev_n);
            ctx->pin_busy[pk] = false;
        }
        prev_off = off; prev_n = n; off += n;
    }
    const int lk = (i - 1) & 1;
    HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[lk]));
    memcpy((char *)dst + prev_off, ctx->pin_buf[lk], prev_n);
    ctx->pin_busy[lk] = false;
    return C3R_OK;
}

int big_h2d_fwd(c3r_ctx *ctx, void *dst, const void *src, size_t bytes) { return big_h2d(ctx, dst, src, bytes); }

// Launch helper: optional per-kernel HIP-event timing on the context's stream.
struct Launch {
    c3r_ctx *ctx;
    const char *name;
    Launch(c3r_ctx *c, const char *n) : ctx(c), name(n) {
        if (ctx->profiling) (void)hipEventRecord(ctx->ev0, ctx->stream);
    }
    ~Launch() {
        if (ctx->profiling) {
            (void)hipEventRecord(ctx->ev1, ctx->stream);
            (void)hipEventSynchronize(ctx->ev1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
            KStat &k = ctx->kstats[name];
            k.ms += ms; k.n += 1;
        }
    }
};

}  // namespace
static int device_excl_scan(c3r_ctx *ctx, int32_t *d, int n, int32_t *d_total);
namespace {

// ---- the device tables of the loaded reads (reads_kernels.hpp): headers, prefix maxima, the pile table.  Everything depends on the
// filters (records exist only for reads that pass them), so c3r_set_params with new --min-MQ / --excl-flags runs this again on the raw
// records the device still holds.  Four kernels, one host wait (sizes and validation errors).
// mpileup_compat = 1, reads with pads inside a run of I ops (k_prep counted them): the table of those runs, from the raw records the device
// holds (the caller's arrays may be gone when c3r_set_params switches the printer).  Same run rule as walk_serial's first look: I and P ops
// in a row, zero-length ops and hard clips skipped; the entry's key is the query offset of the run's first inserted base.
int build_padins(c3r_ctx *ctx, int n) {
    std::vector<c3r_read_t> rd((size_t)n);
    std::vector<uint32_t> cg((size_t)std::max<int64_t>(ctx->n_cigar_ops, 1));
    HIPCHK(ctx, hipMemcpyAsync(rd.data(), ctx->d_rawreads.p, (size_t)n * sizeof(c3r_read_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(cg.data(), ctx->d_rawcig.p, (size_t)ctx->n_cigar_ops * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    auto tab = std::make_shared<PadInsTab>();
    for (int i = 0; i < n; ++i) {
        const c3r_read_t &r = rd[(size_t)i];
        if ((int64_t)r.cigar_off + r.n_cigar > ctx->n_cigar_ops) continue;
        uint32_t y = 0;
        c3r_padins_t e;
        memset(&e, 0, sizeof e);
        bool open = false;
        auto close = [&]() { if (open && e.n_bases && e.pad_mask && e.total <= 64) tab->push_back(e); open = false; };
        for (uint32_t k = 0; k < r.n_cigar; ++k) {
            const uint32_t c = cg[(size_t)r.cigar_off + k], op = c & 15u, len = c >> 4;
            if (len == 0 || op == C3R_CIG_H) continue;
            if (op == C3R_CIG_I || op == C3R_CIG_P) {
                if (!open) { memset(&e, 0, sizeof e); e.read_idx = (uint32_t)i; e.qpos = y; open = true; }
                for (uint32_t j = 0; j < len && e.total <= 64; ++j, ++e.total) if (op == C3R_CIG_P && e.total < 64) e.pad_mask |= 1ull << e.total;
                if (op == C3R_CIG_I) { e.n_bases += len; y += len; }
            } else {
                close();
                if (op == C3R_CIG_M || op == C3R_CIG_EQ || op == C3R_CIG_X || op == C3R_CIG_S) y += len;
            }
        }
        close();
    }
    ctx->padins = tab;
    if (tab->empty()) return C3R_OK;
    int rc = upload(ctx, ctx->d_padins, tab->data(), tab->size());
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return C3R_OK;
}

// Uploads of a contig's records take the device's whole PCIe link for milliseconds.  When several contexts of one process share a device
// (two pipelined contexts: one uploads while the other computes), two uploads at once halve each other's rate and BOTH contexts then compute
// at the same time — the link idles while the kernels run and the kernels wait while the link is busy.  One upload at a time per device keeps
// the contexts out of phase: the gate is taken before the first copy of c3r_load_reads and released at that call's first host wait (the
// read-back of the table sizes, by which the copies have landed).
static std::mutex &upload_gate(int device) {
    static std::mutex gates[64];
    return gates[(unsigned)device % 64u];
}
static std::atomic<int> &upload_waiters(int device) {          // contexts parked at the gate: whoever holds it lets go as soon as its copies have landed
    static std::atomic<int> w[64];
    return w[(unsigned)device % 64u];
}

int prepare_tables(c3r_ctx *ctx, int n, int64_t last_pos, bool timing, std::unique_lock<std::mutex> *gate = nullptr) {
    int rc;
    ctx->host_reads_valid = false; ctx->legacy_valid = false;
    if (!ctx->h_stats) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_stats, sizeof(LoadStats), hipHostMallocDefault));
    if ((rc = ensure(ctx, ctx->d_stats, sizeof(LoadStats))) || (rc = ensure(ctx, ctx->d_reads, (size_t)n * sizeof(DevRead))) ||
        (rc = ensure(ctx, ctx->d_serial, (size_t)n + 16)) || (rc = ensure(ctx, ctx->d_nind, (size_t)n * 4 + 16)) || (rc = ensure(ctx, ctx->d_prefmax, (size_t)n * 4 + 16)))
        return rc;
    const auto t_begin = std::chrono::steady_clock::now();
    // bins from the first read's position (the records are sorted) to a guess of the largest end: the last read's position plus 2 Mb —
    // when a read reaches further (the device knows after the first pass) the table grows and the pass runs again
    int64_t want_end = std::max<int64_t>(last_pos, ctx->first_pos) + ((int64_t)1 << 21);
    LoadStats hs;
    for (int attempt = 0;; ++attempt) {
        const int32_t base = ctx->first_pos >> BIN_SHIFT;
        want_end = std::min<int64_t>(want_end, (int64_t)INT32_MAX);
        const int32_t nb = (int32_t)((want_end >> BIN_SHIFT) - base + 2), nbc = (nb >> CBIN_SHIFT) + 2;
        const int32_t nb_pad = (nb + 1023) & ~1023;                           // (the bin counters are laid out in whole blocks of 1024: cnt_at)
        const size_t cnt_bytes = ((size_t)nb_pad + 3 * (size_t)nbc) * 4;      // records per bin | reads started | ended | prefix-max histogram per coarse bin
        if (ctx->d_bincnt.cap < cnt_bytes || !ctx->d_bincnt.p) ctx->bins_zeroed = 0;                                    // (a fresh block)
        if ((rc = ensure(ctx, ctx->d_bincnt, cnt_bytes)) || (rc = ensure(ctx, ctx->d_binoff, (size_t)(nb + 1) * 4 + 16)) || (rc = ensure(ctx, ctx->d_rtab, (size_t)(nbc + 1) * sizeof(int4)))) return rc;
        ctx->bins = BinGeo{base, nb, nbc, 0};
        // the counter arrays lie back to back and every pass leaves them all zero (k_bin_scan, k_prep<true>): nothing to clear as long as
        // they stay inside what has been cleared once
        if (ctx->bins_dirty || cnt_bytes > ctx->bins_zeroed) {
            HIPCHK(ctx, hipMemsetAsync(ctx->d_bincnt.p, 0, std::max(cnt_bytes, ctx->bins_zeroed), ctx->stream));
            ctx->bins_zeroed = std::max(cnt_bytes, ctx->bins_zeroed);
        }
        ctx->bins_dirty = true;                               // until this pass has come through
        const int nb_pm = (n + PM_BLK - 1) / PM_BLK, nb_bs = (nb + BS_BLK - 1) / BS_BLK, nb_bc = (nbc + BS_BLK - 1) / BS_BLK;
        const size_t lbk_bytes = 64 + (size_t)(nb_pm + nb_bs + 2 * nb_bc) * 8;
        if ((rc = ensure(ctx, ctx->d_lbk, lbk_bytes))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_lbk.p, 0, lbk_bytes, ctx->stream));
        LoadStats init;
        memset(&init, 0, sizeof init);
        init.err = ~0ull;
        *ctx->h_stats = init;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_stats.p, ctx->h_stats, sizeof init, hipMemcpyHostToDevice, ctx->stream));
        PrepArgs a;
        memset(&a, 0, sizeof a);
        a.reads = (const c3r_read_t *)ctx->d_rawreads.p; a.n_reads = n; a.cigars = (const uint32_t *)ctx->d_rawcig.p;
        a.n_cigar_ops = ctx->n_cigar_ops; a.n_seq_bytes = ctx->n_seq_bytes; a.min_mq = ctx->prm.min_mq; a.excl_flags = ctx->prm.excl_flags; a.compat = ctx->prm.mpileup_compat;
        a.geo = ctx->bins;
        uint32_t *cnt = (uint32_t *)ctx->d_bincnt.p;
        a.cnt = cnt; a.sc = cnt + nb_pad; a.ec = a.sc + nbc; uint32_t *pc = a.ec + nbc;
        a.rec_off = (const uint32_t *)ctx->d_binoff.p; a.out = (DevRead *)ctx->d_reads.p; a.serial = (uint8_t *)ctx->d_serial.p; a.nind = (int32_t *)ctx->d_nind.p;
        a.st = (LoadStats *)ctx->d_stats.p;
        char *lbk = (char *)ctx->d_lbk.p;
        const unsigned grid = (unsigned)((n + PREP_READS - 1) / PREP_READS);
        if ((rc = ensure(ctx, ctx->d_wgtab, (size_t)grid * WG_TAB_WORDS * 4))) return rc;    // what the workgroups counted, kept for the second pass
        a.wg_tab = (uint32_t *)ctx->d_wgtab.p;
        {
            Launch L(ctx, "k_prep_count");
            hipLaunchKernelGGL(k_prep<false>, dim3(grid), dim3(PREP_THREADS), 0, ctx->stream, a);
        }
        {
            Launch L(ctx, "k_prefmax_bins");
            hipLaunchKernelGGL(k_prefmax_bins, dim3(nb_pm), dim3(1024), 0, ctx->stream, (const DevRead *)ctx->d_reads.p, n, ctx->prm.min_mq, ctx->prm.excl_flags, ctx->bins,
                               (const int32_t *)ctx->d_nind.p, (int32_t *)ctx->d_prefmax.p, pc, a.st, (int32_t *)lbk, (unsigned long long *)(lbk + 64));
        }
        {
            Launch L(ctx, "k_bin_scan");
            hipLaunchKernelGGL(k_bin_scan, dim3(nb_bs + nb_bc), dim3(1024), 0, ctx->stream, (const uint32_t *)a.cnt, a.sc, pc, a.ec, ctx->bins, nb_bs, (uint32_t *)ctx->d_binoff.p,
                               (int4 *)ctx->d_rtab.p, a.st, (int32_t *)(lbk + 4), (unsigned long long *)(lbk + 64 + (size_t)nb_pm * 8),
                               (unsigned long long *)(lbk + 64 + (size_t)(nb_pm + nb_bs) * 8), (unsigned long long *)(lbk + 64 + (size_t)(nb_pm + nb_bs + nb_bc) * 8));
        }
        // ---- the one synchronisation: sizes, errors
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_stats, ctx->d_stats.p, sizeof(LoadStats), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (gate && gate->owns_lock()) gate->unlock();          // (the uploads have landed: the next context's may start)
        HIPCHK(ctx, hipGetLastError());
        hs = *ctx->h_stats;
        if (hs.err != ~0ull) {
            const long long i = (long long)(hs.err >> 8);
            switch ((int)(hs.err & 0xff)) {
                case LD_UNSORTED: return fail(ctx, C3R_EINVAL, "reads must be sorted by pos (read %lld)", i);
                case LD_CIGAR_RANGE: return fail(ctx, C3R_EINVAL, "cigar range of read %lld out of bounds", i);
                case LD_SEQ_RANGE: return fail(ctx, C3R_EINVAL, "seq range of read %lld out of bounds", i);
                case LD_BAD_OP: return fail(ctx, C3R_EINVAL, "bad cigar op in read %lld", i);
                case LD_OP_LONG: return fail(ctx, C3R_EINVAL, "cigar op too long in read %lld", i);
                case LD_END_2G: return fail(ctx, C3R_EINVAL, "read %lld ends beyond 2^31", i);
                case LD_SEG_OPS: return fail(ctx, C3R_EINVAL, "read %lld: more than 65535 CIGAR ops between two N ops", i);
                case LD_PAD_INS: return fail(ctx, C3R_EINVAL, "read %lld: an insertion with pads (P ops) of more than 64 characters is not supported with mpileup_compat = 1", i);
                default: return fail(ctx, C3R_EINVAL, "invalid read %lld", i);
            }
        }
        if ((int64_t)hs.max_end <= (((int64_t)base + nb) << BIN_SHIFT)) break;
        if (attempt >= 1) return fail(ctx, C3R_EINVAL, "internal: the position bins did not grow to the reads' ends");
        want_end = (int64_t)hs.max_end + 64;                  // a read reaches beyond the table: the counts of its far bins were clamped
    }
    if (hs.n_rec < 0) return fail(ctx, C3R_EINVAL, "too many CIGAR ops");
    ctx->padins.reset();
    if (hs.n_padreads > 0 && (rc = build_padins(ctx, n))) return rc;          // (hand-made CIGARs only: off the measured path)
    const auto t_sync = std::chrono::steady_clock::now();
    // ---- second pass (nothing below waits for the device): every record into its bin
    if ((rc = ensure(ctx, ctx->d_recs, (size_t)hs.n_rec * sizeof(PileRec) + 64))) return rc;
    {
        PrepArgs a;
        memset(&a, 0, sizeof a);
        a.n_reads = n; a.cigars = (const uint32_t *)ctx->d_rawcig.p; a.min_mq = ctx->prm.min_mq; a.excl_flags = ctx->prm.excl_flags; a.compat = ctx->prm.mpileup_compat; a.geo = ctx->bins;
        a.cnt = (uint32_t *)ctx->d_bincnt.p; a.rec_off = (const uint32_t *)ctx->d_binoff.p; a.out = (DevRead *)ctx->d_reads.p; a.serial = (uint8_t *)ctx->d_serial.p;
        a.recs = (PileRec *)ctx->d_recs.p;
        a.wg_tab = (uint32_t *)ctx->d_wgtab.p;
        Launch L(ctx, "k_prep_write");
        hipLaunchKernelGGL(k_prep<true>, dim3((unsigned)((n + PREP_READS - 1) / PREP_READS)), dim3(PREP_THREADS), 0, ctx->stream, a);
    }
    HIPCHK(ctx, hipGetLastError());
    ctx->bins_dirty = false;                                  // (the record counters are back at zero once k_prep<true> is through)
    ctx->n_reads = n; ctx->n_indel_ops = hs.n_indel; ctx->max_cover = hs.max_cover;
    // (under the profiler: how many workgroups of the second pass counted their records again — no kernel, `launches` is the count)
    if (ctx->profiling && hs.n_recount > 0) ctx->kstats["k_prep_recount_workgroups"].n += hs.n_recount;
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[c3r_load_reads] %d reads, %d records in %d bins: first pass until the sync %.2f ms, second pass queued in %.2f ms\n", n, hs.n_rec, ctx->bins.nb,
                ms(t_begin, t_sync), ms(t_sync, std::chrono::steady_clock::now()));
    }
    return C3R_OK;
}

// New filters for the reads already loaded (c3r_set_params changed --min-MQ / --excl-flags): the tables are rebuilt from the raw records.
int refilter(c3r_ctx *ctx) {
    const int n = ctx->n_reads;
    if (n == 0) return C3R_OK;
    ctx->n_reads = 0;
    return prepare_tables(ctx, n, ctx->last_pos, false);
}

// The legacy tables (reads_kernels.hpp): only token_at reads them — the ordered haplotype recompute of flagged columns, 30 channels.
int ensure_legacy_tables(c3r_ctx *ctx) {
    if (ctx->legacy_valid || ctx->n_reads == 0) return C3R_OK;
    const int n = ctx->n_reads;
    int rc;
    if ((rc = ensure(ctx, ctx->d_lcnt, (size_t)(n + 2) * sizeof(int2)))) return rc;
    // sizes without asking the device: normalisation only drops or merges ops, and a read with k ref-skips has k + 1 segments out of at least
    // 2 k + 1 ops — (ops + reads) / 2 segments at most.  (Round 4 read the two totals back: a host wait in every 30-channel pass.)
    const size_t max_ops = (size_t)ctx->n_cigar_ops, max_segs = ((size_t)ctx->n_cigar_ops + (size_t)n) / 2 + 1;
    if ((rc = ensure(ctx, ctx->d_cigar, max_ops * 4 + 16)) || (rc = ensure(ctx, ctx->d_rsegs, max_segs * sizeof(DevSeg) + 16)) ||
        (rc = ensure(ctx, ctx->d_rseg_first, (size_t)(n + 1) * 4)))
        return rc;
    Launch L(ctx, "k_legacy_tables");
    hipLaunchKernelGGL(k_legacy_count, dim3((unsigned)(n / 256 + 1)), dim3(256), 0, ctx->stream, (const DevRead *)ctx->d_reads.p, n, (const uint32_t *)ctx->d_rawcig.p, (int2 *)ctx->d_lcnt.p);
    hipLaunchKernelGGL(k_legacy_scan, dim3(1), dim3(1024), 0, ctx->stream, (int2 *)ctx->d_lcnt.p, n + 1, (int2 *)ctx->d_lcnt.p + (n + 1));
    hipLaunchKernelGGL(k_legacy_write, dim3((unsigned)(n / 256 + 1)), dim3(256), 0, ctx->stream, (const DevRead *)ctx->d_reads.p, n, (const uint32_t *)ctx->d_rawcig.p,
                       (const int2 *)ctx->d_lcnt.p, (uint32_t *)ctx->d_cigar.p, (DevSeg *)ctx->d_rsegs.p, (uint32_t *)ctx->d_rseg_first.p);
    HIPCHK(ctx, hipGetLastError());
    ctx->legacy_valid = true;
    return C3R_OK;
}

// Host copies, fetched only by the paths that walk reads on the host: mpileup's depth cap (sequential by nature) and the decoder
// (inserted bases of the alt alleles).
int ensure_host_reads(c3r_ctx *ctx) {
    if (ctx->host_reads_valid) return C3R_OK;
    ctx->h_reads.resize((size_t)ctx->n_reads); ctx->h_prefmax.resize((size_t)ctx->n_reads);
    if (ctx->n_reads) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_reads.data(), ctx->d_reads.p, (size_t)ctx->n_reads * sizeof(DevRead), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_prefmax.data(), ctx->d_prefmax.p, (size_t)ctx->n_reads * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->host_reads_valid = true;
    return C3R_OK;
}
int ensure_host_seq(c3r_ctx *ctx) {
    if (ctx->host_seq_valid) return C3R_OK;
    ctx->h_seq.resize((size_t)ctx->n_seq_bytes + 16);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_seq.data(), ctx->d_seq.p, ctx->h_seq.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->host_seq_valid = true;
    return C3R_OK;
}

// sort + merge intervals into a disjoint union (the overlap test against the union equals the
// reference's any-interval overlap, shared/interval_tree.py:77-87)
void merge_intervals(std::vector<int32_t> &iv) {
    std::vector<std::pair<int32_t, int32_t>> v;
    for (size_t i = 0; i + 1 < iv.size(); i += 2) v.push_back({iv[i], iv[i + 1]});
    std::sort(v.begin(), v.end());
    iv.clear();
    for (auto &x : v) {
        if (x.second <= x.first) continue;
        if (!iv.empty() && x.first <= iv.back()) iv.back() = std::max(iv.back(), x.second);
        else { iv.push_back(x.first); iv.push_back(x.second); }
    }
}

}  // namespace

static int list_grid() { static const int v = [] { const char *e = getenv("C3R_LIST_GRID"); return e && atoi(e) > 0 ? atoi(e) : LIST_GRID; }(); return v; }

extern "C" {

const char *c3r_version(void) { return "c3r 0.1 (gfx950, HIP)"; }

void c3r_default_params(c3r_params_t *p) {
    memset(p, 0, sizeof *p);
    p->max_depth = 8000;
    p->channels = C3R_CH;
    p->min_mq = 5;
    p->excl_flags = 2316;
    p->min_coverage = 4;
    p->snp_min_af = 0.08;
    p->indel_min_af = 0.15;
    p->max_depth_rescale = 144;
}

int c3r_create(int device_id, void *stream, c3r_ctx **out) {
    if (!out) return C3R_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return C3R_ENODEVICE;
    if (device_id < 0 || device_id >= n) return C3R_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return C3R_ENODEVICE;
    c3r_ctx *ctx = new c3r_ctx();
    ctx->device = device_id;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess) ctx->n_cu = cu; }
    c3r_default_params(&ctx->prm);
    if (stream) ctx->stream = (hipStream_t)stream;
    else {
        const char *two = getenv("C3R_TWO_STREAMS");
        if (!(two && *two && *two != '0')) {
            if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return C3R_EHIP; }
        } else {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest) != hipSuccess) { delete ctx; return C3R_EHIP; }
            ctx->owns_stream = true;
            if (hipStreamCreateWithPriority(&ctx->net_stream, hipStreamNonBlocking, least) != hipSuccess ||
                hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_net, hipEventDisableTiming) != hipSuccess) { c3r_destroy(ctx); return C3R_EHIP; }
        }
        ctx->owns_stream = true;
    }
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { delete ctx; return C3R_EHIP; }
    *out = ctx;
    return C3R_OK;
}

void c3r_destroy(c3r_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->net_stream) (void)hipStreamSynchronize(ctx->net_stream);
    if (ctx->deep_stream) (void)hipStreamSynchronize(ctx->deep_stream);
    (void)hipStreamSynchronize(ctx->stream);
    const bool timing = getenv("C3R_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    DevBuf *bufs[] = {&ctx->d_wgtab, &ctx->d_rawreads, &ctx->d_rawcig, &ctx->d_bincnt, &ctx->d_binoff, &ctx->d_rtab, &ctx->d_recs, &ctx->d_serial, &ctx->d_nind, &ctx->d_lbk, &ctx->d_lcnt, &ctx->d_tokexp, &ctx->d_tokoff,
                      &ctx->d_stats, &ctx->d_lb, &ctx->d_regb, &ctx->d_span, &ctx->d_spanbase, &ctx->d_meta, &ctx->d_spanrec, &ctx->d_deep, &ctx->d_evwg, &ctx->d_giant, &ctx->d_giant_ev, &ctx->d_giant_tab, &ctx->d_winidx, &ctx->d_rawidx, &ctx->d_export, &ctx->d_dbg, &ctx->d_tile_cand, &ctx->d_reads, &ctx->d_cigar, &ctx->d_seq, &ctx->d_prefmax, &ctx->d_tile_cols, &ctx->d_tile_rng, &ctx->d_tile_list, &ctx->d_tile_list2, &ctx->d_rsegs, &ctx->d_rseg_first, &ctx->d_ref, &ctx->d_bed[0], &ctx->d_bed[1],
                      &ctx->d_sites, &ctx->d_cols, &ctx->d_depth, &ctx->d_ncov, &ctx->d_flags, &ctx->d_skipmax, &ctx->d_geo, &ctx->d_lastrow, &ctx->d_drop, &ctx->d_ev, &ctx->d_small,
                      &ctx->d_blockcnt, &ctx->d_scan_tops, &ctx->d_cand, &ctx->d_tensors, &ctx->d_raw, &ctx->d_sites_out, &ctx->d_tokcnt, &ctx->d_tok, &ctx->d_tokb, &ctx->d_tokrec, &ctx->d_recoff, &ctx->d_padins, &ctx->d_aftab, &ctx->d_keep, &ctx->d_sites_c, &ctx->d_probs_c};
    int n_dev = 0; size_t b_dev = 0, b_pin = 0;
    for (DevBuf *b : bufs) if (b->p) { (void)hipFree(b->p); ++n_dev; b_dev += b->cap; }
    const auto t1 = std::chrono::steady_clock::now();
    net_free(ctx->net);
    const auto t2 = std::chrono::steady_clock::now();
    if (ctx->rows_snap) c3r_rows_free(ctx->rows_snap);
    for (auto &sp : ctx->stage_pool) { stage_free(sp.first); b_pin += sp.second; }
    if (ctx->h_stats) (void)hipHostFree(ctx->h_stats);
    if (ctx->h_pack) (void)hipHostFree(ctx->h_pack);
    for (int k = 0; k < 2; ++k) { if (ctx->pin_buf[k]) (void)hipHostFree(ctx->pin_buf[k]); if (ctx->pin_ev[k]) (void)hipEventDestroy(ctx->pin_ev[k]); }
    if (ctx->h_lstm) (void)hipHostFree(ctx->h_lstm);
    if (ctx->h_scan) (void)hipHostFree(ctx->h_scan);
    for (auto &rb : ctx->refbuf) { if (rb.p) { (void)hipHostFree(rb.p); b_pin += rb.cap; } if (rb.ev) (void)hipEventDestroy(rb.ev); }
    const auto t3 = std::chrono::steady_clock::now();
    if (ctx->deep_stream) (void)hipStreamDestroy(ctx->deep_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_up) (void)hipEventDestroy(ctx->ev_up);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->net_stream) (void)hipStreamDestroy(ctx->net_stream);
    if (ctx->ev_x) (void)hipEventDestroy(ctx->ev_x);
    if (ctx->ev_net) (void)hipEventDestroy(ctx->ev_net);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[c3r_destroy %p] %d device buffers (%.0f MB) %.1f ms, network buffers %.1f ms, host blocks (%.0f MB) %.1f ms, events + stream %.1f ms\n",
                (void *)ctx, n_dev, b_dev / 1e6, ms(t0, t1), ms(t1, t2), b_pin / 1e6, ms(t2, t3), ms(t3, std::chrono::steady_clock::now()));
    }
    delete ctx;
}

int64_t c3r_trim(void) { return (int64_t)big_trim(); }

const char *c3r_last_error(const c3r_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int c3r_synchronize(c3r_ctx *ctx) {
    if (!ctx) return C3R_EINVAL;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return C3R_OK;
}

void *c3r_stream(c3r_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int c3r_set_params(c3r_ctx *ctx, const c3r_params_t *p) {
    if (!ctx || !p) return C3R_EINVAL;
    if (p->channels != C3R_CH && p->channels != C3R_CH_PHASED) return fail(ctx, C3R_EINVAL, "channels must be 18 or 30");
    if (p->mpileup_compat != 0 && p->mpileup_compat != 1) return fail(ctx, C3R_EINVAL, "mpileup_compat must be 0 (samtools <= 1.10) or 1 (samtools >= 1.11)");
    // (the pile table holds the records of the reads that pass the filters, with the indel look-ahead of the chosen samtools)
    const bool refilter = p->min_mq != ctx->prm.min_mq || p->excl_flags != ctx->prm.excl_flags || p->mpileup_compat != ctx->prm.mpileup_compat;
    ctx->prm = *p;
    if (ctx->prm.max_depth_rescale <= 0) ctx->prm.max_depth_rescale = 144;
    ctx->last_scan_pruned = false;
    if (refilter) return ::refilter(ctx);
    return C3R_OK;
}

int c3r_load_reads(c3r_ctx *ctx, const c3r_read_t *reads, int64_t n_reads, const uint32_t *cigars, int64_t n_cigar_ops,
                   const uint8_t *seq4, int64_t n_seq_bytes) {
    if (!ctx || n_reads < 0 || n_cigar_ops < 0 || n_seq_bytes < 0 || (n_reads && (!reads || !cigars || !seq4))) return C3R_EINVAL;
    if (n_reads >= INT32_MAX) return fail(ctx, C3R_EINVAL, "too many reads");
    if (n_cigar_ops >= INT32_MAX) return fail(ctx, C3R_EINVAL, "too many CIGAR ops");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool timing = getenv("C3R_TIMING") != nullptr;
    // whatever happens below, the previous contig's tables are gone
    ctx->n_reads = 0; ctx->n_indel_ops = 0; ctx->n_seq_bytes = 0; ctx->n_cigar_ops = 0; ctx->max_cover = 0;
    ctx->host_reads_valid = false; ctx->host_seq_valid = false; ctx->last_scan_pruned = false; ctx->legacy_valid = false;
    ctx->host_cache.reset();              // (snapshots of the previous contig keep their copy alive)
    ctx->padins.reset();
    const int n = (int)n_reads;
    int rc;
    // ---- the caller's records go up as they are (three copies; truly asynchronous when the caller's arrays are pinned, see
    // c3r_host_alloc) and every table the tile kernels need is derived from them on the device
    if ((rc = ensure(ctx, ctx->d_rawreads, std::max<size_t>((size_t)n * sizeof(c3r_read_t), 16))) || (rc = ensure(ctx, ctx->d_rawcig, std::max<size_t>((size_t)n_cigar_ops * 4, 16))) ||
        (rc = ensure(ctx, ctx->d_seq, (size_t)n_seq_bytes + 16))) return rc;
    upload_waiters(ctx->device).fetch_add(1);
    std::unique_lock<std::mutex> gate(upload_gate(ctx->device));
    upload_waiters(ctx->device).fetch_sub(1);
    {
        Launch l(ctx, "h2d_reads");           // (profiling: the three uploads as one entry of the kernel statistics — PCIe time, not a kernel)
        if ((rc = upload(ctx, ctx->d_rawreads, reads, (size_t)n)) || (rc = upload(ctx, ctx->d_rawcig, cigars, (size_t)n_cigar_ops))) return rc;
        if (n_seq_bytes && (rc = big_h2d(ctx, ctx->d_seq.p, seq4, (size_t)n_seq_bytes))) return rc;
    }
    HIPCHK(ctx, hipMemsetAsync((char *)ctx->d_seq.p + n_seq_bytes, 0, 16, ctx->stream));        // (the walk reads the packed bases 16 bytes at a time)
    if (upload_waiters(ctx->device).load() > 0) {
        // another context is waiting to upload: the link is handed over when THESE copies are through, not a first table pass later
        if (!ctx->ev_up) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_up, hipEventDisableTiming));
        HIPCHK(ctx, hipEventRecord(ctx->ev_up, ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(ctx->ev_up));
        gate.unlock();
    }
    ctx->n_seq_bytes = n_seq_bytes; ctx->n_cigar_ops = n_cigar_ops;
    if (n == 0) return C3R_OK;
    ctx->first_pos = std::max(reads[0].pos, 0);
    // the last read's position bounds the first guess of the bins (the records are sorted; an unsorted set fails in the first pass)
    const int64_t last_pos = std::max(reads[n - 1].pos, reads[0].pos);
    ctx->last_pos = last_pos;
    return prepare_tables(ctx, n, last_pos, timing, &gate);
}

void *c3r_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void c3r_host_free(void *p) { if (p) (void)hipHostFree(p); }

int c3r_set_reference(c3r_ctx *ctx, int64_t ref_start, const char *ref, int64_t len) {
    if (!ctx || !ref || len < 0 || ref_start < 1) return C3R_EINVAL;
    ctx->last_scan_pruned = false;       // (c3r_get_columns completes a pruned scan with the arguments of that scan: stale now)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // The slice is upper-cased straight into a page-locked buffer the context keeps (one pass over the caller's bytes, on threads for a
    // whole chromosome), which is both the source of an asynchronous DMA upload — nothing here waits for the device — and the decoder's
    // view of the reference.  The previous upload must have left the buffer before it is overwritten.
    ctx->ref_len = 0;
    // a buffer no snapshot holds: the current one if it is free, else another (waiting for a decode to finish if all three are held)
    int slot = -1;
    for (int waited = 0; slot < 0; ++waited) {
        if (ctx->ref_cur >= 0 && ctx->refbuf[ctx->ref_cur].users.load() == 0) slot = ctx->ref_cur;
        for (int k = 0; k < 3 && slot < 0; ++k) if (k != ctx->ref_cur && ctx->refbuf[k].users.load() == 0) slot = k;
        if (slot >= 0) break;
        if (waited > 600000) return fail(ctx, C3R_EINVAL, "all three reference buffers are held by row snapshots that were never released (c3r_rows_free)");
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    c3r_ctx::RefBuf &rb = ctx->refbuf[slot];
    if (!rb.ev) HIPCHK(ctx, hipEventCreateWithFlags(&rb.ev, hipEventDisableTiming));
    else HIPCHK(ctx, hipEventSynchronize(rb.ev));                 // (its previous upload has left the buffer)
    if ((size_t)len + 16 > rb.cap) {
        if (rb.p) (void)hipHostFree(rb.p);
        rb.p = nullptr; rb.cap = 0;
        const size_t cap = (size_t)len + (size_t)len / 8 + 4096;
        HIPCHK(ctx, hipHostMalloc((void **)&rb.p, cap, hipHostMallocDefault));
        rb.cap = cap;
    }
    ctx->ref_cur = slot; ctx->h_ref = rb.p;
    {
        char *dst = ctx->h_ref;
        auto upper = [&](int64_t a, int64_t e) {
            for (int64_t i = a; i < e; ++i) {        // branch-free, so the loop vectorises
                const unsigned char c = (unsigned char)ref[i];
                dst[i] = (char)(c - (((unsigned)(c - 'a') < 26u) << 5));
            }
        };
        // a 250 MB chromosome is memory-bound work for one core: split it
        int64_t nt = std::max<int64_t>(1, std::min<int64_t>({8, (int64_t)usable_cpus(), len >> 23}));
        if (const char *e = getenv("C3R_THREADS")) nt = std::max<int64_t>(1, std::min<int64_t>(nt, atoi(e)));
        std::vector<std::thread> th;
        for (int64_t t = 1; t < nt; ++t) th.emplace_back(upper, len * t / nt, len * (t + 1) / nt);
        upper(0, len / nt);
        for (auto &x : th) x.join();
    }
    ctx->ref_start1 = ref_start;
    int rc = ensure(ctx, ctx->d_ref, std::max<size_t>((size_t)len, 16));
    if (rc) return rc;
    if (len) HIPCHK(ctx, hipMemcpyAsync(ctx->d_ref.p, ctx->h_ref, (size_t)len, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipEventRecord(rb.ev, ctx->stream));
    ctx->ref_len = (size_t)len;
    return C3R_OK;
}

int c3r_set_reference_view(c3r_ctx *ctx, int64_t ref_start, const char *ref_upper, int64_t len) {
    if (!ctx || !ref_upper || len < 0 || ref_start < 1) return C3R_EINVAL;
    ctx->last_scan_pruned = false;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // Nothing is copied on the host: the decoder reads the caller's bytes (kept alive and unchanged by the caller, see c3r.h) and the
    // upload takes them as they are — from pageable memory the runtime stages the copy itself and returns once the bytes have left.
    ctx->ref_len = 0;
    ctx->ref_cur = -1; ctx->h_ref = const_cast<char *>(ref_upper);
    ctx->ref_start1 = ref_start;
    int rc = ensure(ctx, ctx->d_ref, std::max<size_t>((size_t)len, 16));
    if (rc) return rc;
    if (len && (rc = big_h2d(ctx, ctx->d_ref.p, ref_upper, (size_t)len))) return rc;
    ctx->ref_len = (size_t)len;
    return C3R_OK;
}

int c3r_set_bed(c3r_ctx *ctx, int which, const int32_t *pairs, int64_t n) {
    if (!ctx || which < 0 || which > 1 || n < 0 || (n && !pairs)) return C3R_EINVAL;
    ctx->last_scan_pruned = false;       // (c3r_get_columns completes a pruned scan with the arguments of that scan: stale now)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->has_bed[which] = n > 0;
    ctx->h_bed[which].assign(pairs, pairs + 2 * n);
    merge_intervals(ctx->h_bed[which]);
    int rc = upload(ctx, ctx->d_bed[which], ctx->h_bed[which].data(), ctx->h_bed[which].size());
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return C3R_OK;
}

int c3r_set_sites(c3r_ctx *ctx, const int32_t *sites, int64_t n) {
    if (!ctx || n < 0 || (n && !sites)) return C3R_EINVAL;
    ctx->last_scan_pruned = false;       // (c3r_get_columns completes a pruned scan with the arguments of that scan: stale now)
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->h_sites.assign(sites, sites + n);
    std::sort(ctx->h_sites.begin(), ctx->h_sites.end());
    ctx->h_sites.erase(std::unique(ctx->h_sites.begin(), ctx->h_sites.end()), ctx->h_sites.end());
    int rc = upload(ctx, ctx->d_sites, ctx->h_sites.data(), ctx->h_sites.size());
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return C3R_OK;
}

// ------------------------------------------------------------------------------------------------
// dst16: the windows go out as int16 (the resident, rescaled tensors), else int32 (c3r_get_tensors' raw export)
static int run_gather(c3r_ctx *ctx, int rescale, void *dst, bool with_sites, bool dst16) {
    // operates on the candidates of the most recent scan; `dst` already points at their slot
    GatherArgs g;
    g.cols = (const int32_t *)ctx->d_cols.p; g.depth = (const int32_t *)ctx->d_depth.p; g.ncov = (const int32_t *)ctx->d_ncov.p;
    g.flags = (const uint8_t *)ctx->d_flags.p; g.tile_cols = (const uint8_t *)ctx->d_tile_cols.p; g.cand_idx = (const int32_t *)ctx->d_cand.p; g.n_cand = (int32_t)ctx->last_cand;
    g.n_pos = (int32_t)ctx->n_pos; g.geo = (const TileGeo *)ctx->d_geo.p;
    g.ref = (const uint8_t *)ctx->d_ref.p; g.ref_beg0 = (int32_t)(ctx->ref_start1 - 1); g.ref_len = (int32_t)ctx->ref_len;
    g.head_tail = ctx->prm.head_tail; g.last_row = (const int32_t *)ctx->d_lastrow.p;
    g.rescale = rescale; g.max_depth = ctx->prm.max_depth_rescale;
    g.tensors = dst; g.x16 = dst16 ? 1 : 0;
    g.raw = nullptr; g.skipmax = nullptr;
    g.sites = with_sites ? (c3r_site_t *)ctx->d_sites_out.p + ctx->last_base : nullptr;
    g.tok_cnt = with_sites ? (int32_t *)ctx->d_tokcnt.p : nullptr;
    const int blocks = (int)((ctx->last_cand * 64 + 255) / 256);
    if (ctx->prm.splice_padding) {
        // in-place column edits, candidate after candidate: runs once per scan and also keeps the raw windows
        g.raw = (int32_t *)ctx->d_raw.p; g.skipmax = (const int32_t *)ctx->d_skipmax.p;
        Launch L(ctx, "k_splice_gather");
        if (ctx->prm.channels == C3R_CH) hipLaunchKernelGGL(k_splice_gather<C3R_CH>, dim3(blocks), dim3(256), 0, ctx->stream, g);
        else hipLaunchKernelGGL(k_splice_gather<C3R_CH_PHASED>, dim3(blocks), dim3(256), 0, ctx->stream, g);
        return C3R_OK;
    }
    Launch L(ctx, "k_gather");
    if (ctx->prm.channels == C3R_CH) hipLaunchKernelGGL(k_gather<C3R_CH>, dim3(blocks), dim3(256), 0, ctx->stream, g);
    else hipLaunchKernelGGL(k_gather<C3R_CH_PHASED>, dim3(blocks), dim3(256), 0, ctx->stream, g);
    return C3R_OK;
}

// exclusive scan of n ints in place on the context's stream, total to *d_total: one block for short inputs, three launches for long
static int device_excl_scan(c3r_ctx *ctx, int32_t *d, int n, int32_t *d_total) {
    if (n <= 4 * SCAN_BLK) {
        Launch L(ctx, "k_excl_scan");
        hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, ctx->stream, d, n, d_total);
        return C3R_OK;
    }
    const int nb = (n + SCAN_BLK - 1) / SCAN_BLK;
    int rc = ensure(ctx, ctx->d_scan_tops, (size_t)nb * 4 + 16);
    if (rc) return rc;
    Launch L(ctx, "k_excl_scan");
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(1024), 0, ctx->stream, d, n, (int32_t *)ctx->d_scan_tops.p);
    hipLaunchKernelGGL(k_excl_scan, dim3(1), dim3(1024), 0, ctx->stream, (int32_t *)ctx->d_scan_tops.p, nb, d_total);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(1024), 0, ctx->stream, d, n, (const int32_t *)ctx->d_scan_tops.p);
    return C3R_OK;
}

// Capacity of the indel-event scratch of one scan: a tile workgroup reserves (its events rounded up to 16) records.  An I / D op
// yields at most one event per region whose rows it falls into, and regions may overlap arbitrarily (the same region twice, chunks
// shorter than their +-33 bp halos): (I / D ops of the whole contig) x (the most regions any one position falls into), plus the
// rounding slack of every tile.  An upper bound that needs no host copy of the reads.
static size_t event_capacity(const c3r_ctx *ctx, int n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, int n_tiles, int spans_per_position) {
    size_t ev_cap = 16 * (size_t)n_tiles + 16;
    std::vector<std::pair<int64_t, int>> edge;
    for (int r = 0; r < n_regions; ++r) { edge.push_back({ctg_starts[r] - C3R_WINDOW - 2, +1}); edge.push_back({ctg_ends[r] + C3R_WINDOW + 2, -1}); }
    std::sort(edge.begin(), edge.end(), [](const std::pair<int64_t, int> &x, const std::pair<int64_t, int> &y) { return x.first != y.first ? x.first < y.first : x.second > y.second; });
    int cur = 0, maxov = 0;
    for (auto &e : edge) { cur += e.second; maxov = std::max(maxov, cur); }
    // (fused path: a position lies in its own span and in the flank of at most one neighbour)
    // (a deep tile also takes 2 x 8-byte hash slots per event behind its events: 2/3 of a record each)
    return ev_cap + 2 * ((size_t)maxov * (size_t)spans_per_position * (size_t)ctx->n_indel_ops);
}

// samtools mpileup -d (default 8000), restated from htslib's bam_plp_push / bam_plp_next (third-party, absent: parity
// unpinned; the oracle restates the same rule independently): reads arrive in file order, filtered, and only those
// overlapping the region; a read is discarded iff it is not the first read pushed for its start position and the
// engine's node pool — the kept reads with exclusive end > start - 1, plus the list's empty tail node — holds more than
// max_depth nodes (`iter->mp->cnt > iter->maxcnt`): reads that all start on one position pile up to exactly max_depth.  Sequential by
// nature, so it runs here on the host, per region, and only when the data can reach the cap at all (then the reads'
// headers are fetched back from the device).  *d_drop: the per-region bit masks on the device, or null when no read is discarded.
//
// Only inside the contig's HOT ZONES: with the reads counted into 256-position bins by start and by exclusive end,
//   B(c) = reads started before the end of bin c  -  reads whose end lies before bin c
// bounds the list at the push of every read that starts in bin c (whatever was discarded earlier, whatever the region), so a read that
// starts in a bin with B(c) <= max_depth is always kept.  The sequential rule runs over the maximal runs of bins above the cap, each
// entered with the list it would hold there: the kept reads from before the zone that reach into it.  A contig with one 20,000x locus
// costs a counting pass over its read headers and a heap over that locus' reads, not a heap over every read of every region.
constexpr int PLP_POOL_EXTRA = 1;      // nodes of htslib's pool that are not reads: the list's empty tail
static int depth_cap_mask(c3r_ctx *ctx, int n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, const uint32_t **d_drop, int *drop_words_out) {
    *d_drop = nullptr;
    const int drop_words = (int)(((size_t)ctx->n_reads + 31) / 32);
    *drop_words_out = drop_words;
    // (max_cover bounds the engine's read list at every read's start, k_prep / k_bin_scan: below the cap no read can be discarded)
    if (!(ctx->prm.max_depth > 0 && (int64_t)ctx->max_cover + PLP_POOL_EXTRA > ctx->prm.max_depth)) return C3R_OK;
    int rc;
    if ((rc = ensure_host_reads(ctx))) return rc;
    const std::vector<DevRead> &R = ctx->h_reads;
    const size_t n = R.size();
    if (!n) return C3R_OK;
    auto kept_by_filters = [&](const DevRead &rd) { return !(flag_fails(rd.flag, ctx->prm.excl_flags) || rd.mapq < ctx->prm.min_mq || rd.end <= rd.pos); };
    // ---- hot zones [z0, z1) in positions, ascending
    constexpr int ZSH = 8;
    const int64_t zb0 = (int64_t)R.front().pos >> ZSH;
    int64_t top = R.front().pos;
    for (size_t i = 0; i < n; ++i) if (kept_by_filters(R[i])) top = std::max<int64_t>(top, R[i].end);
    const size_t nz = (size_t)((top >> ZSH) - zb0) + 2;
    std::vector<uint32_t> &sc = ctx->h_zone_sc, &ec = ctx->h_zone_ec;
    sc.assign(nz, 0u); ec.assign(nz, 0u);
    for (size_t i = 0; i < n; ++i) {
        if (!kept_by_filters(R[i])) continue;
        ++sc[(size_t)(((int64_t)R[i].pos >> ZSH) - zb0)];
        ++ec[(size_t)(((int64_t)R[i].end >> ZSH) - zb0)];
    }
    std::vector<std::pair<int64_t, int64_t>> zones;
    {
        int64_t started = 0, ended_before = 0;
        bool open = false;
        for (size_t c = 0; c < nz; ++c) {
            started += sc[c];
            const bool hot = started - ended_before + PLP_POOL_EXTRA > (int64_t)ctx->prm.max_depth;
            const int64_t p = ((int64_t)c + zb0) << ZSH;
            if (hot && !open) { zones.push_back({p, p}); open = true; }
            if (open) { if (hot) zones.back().second = p + (1 << ZSH); else open = false; }
            ended_before += ec[c];
        }
    }
    if (zones.empty()) return C3R_OK;
    if (const char *e = getenv("C3R_CAP_ALL")) if (*e == '1') { zones.clear(); zones.push_back({(int64_t)R.front().pos, top + 1}); }     // (A/B aid: the rule over every read, as before round 5)
    ctx->h_drop.assign((size_t)n_regions * drop_words, 0u);
    bool any = false;
    auto first_at_or_after = [&](int64_t p) {           // first read with pos >= p (reads are sorted by pos)
        size_t lo = 0, hi = n;
        while (lo < hi) { const size_t m = (lo + hi) / 2; if ((int64_t)R[m].pos < p) lo = m + 1; else hi = m; }
        return lo;
    };
    for (int r = 0; r < n_regions; ++r) {
        int64_t es = ctg_starts[r] - C3R_WINDOW, ee = ctg_ends[r] + C3R_WINDOW;
        if (es < 1) es = 1;
        const int32_t beg0 = (int32_t)(es - 1), end0 = (int32_t)ee;                // rows for [beg0, end0)
        uint32_t *drop = ctx->h_drop.data() + (size_t)r * drop_words;
        // the first read fetched for this region: the first one whose end (as a prefix maximum) passes beg0
        const size_t i_region = (size_t)(std::upper_bound(ctx->h_prefmax.begin(), ctx->h_prefmax.end(), beg0) - ctx->h_prefmax.begin());
        for (const auto &z : zones) {
            if (z.first >= end0) break;
            const size_t i0 = std::max(first_at_or_after(z.first), i_region);
            if (i0 >= n || R[i0].pos >= end0 || (int64_t)R[i0].pos >= z.second) continue;
            std::priority_queue<int32_t, std::vector<int32_t>, std::greater<int32_t>> live;
            // the list on entry: kept reads of this region from before i0 that still reach R[i0].pos - 1 ... (popped lazily below, so "reach
            // the zone" is enough)
            const int32_t reach = (int32_t)std::max<int64_t>(std::max<int64_t>(z.first - 1, beg0), INT32_MIN);
            size_t j = (size_t)(std::upper_bound(ctx->h_prefmax.begin(), ctx->h_prefmax.end(), reach) - ctx->h_prefmax.begin());
            for (j = std::max(j, i_region); j < i0; ++j) {
                const DevRead &rd = R[j];
                if (!kept_by_filters(rd) || rd.end <= beg0 || rd.end <= reach) continue;
                if (drop[j >> 5] >> (j & 31) & 1u) continue;                         // discarded in an earlier zone
                live.push(rd.end);
            }
            int32_t last_pos = INT_MIN;
            // (the read before i0 starts before the zone or was not fetched: R[i0] is the first pushed for its position either way — unless
            // i0 is the region's first read, where the rule starts afresh as well)
            for (size_t i = i0; i < n; ++i) {
                const DevRead &rd = R[i];
                if (rd.pos >= end0 || (int64_t)rd.pos >= z.second) break;
                if (!kept_by_filters(rd)) continue;
                if (rd.end <= beg0) continue;                                            // not fetched for this region
                while (!live.empty() && live.top() <= rd.pos - 1) live.pop();
                const bool first = rd.pos != last_pos;
                last_pos = rd.pos;
                if (!first && (int64_t)live.size() + PLP_POOL_EXTRA > ctx->prm.max_depth) {
                    drop[i >> 5] |= 1u << (i & 31);
                    any = true;
                    continue;
                }
                live.push(rd.end);
            }
        }
    }
    if (any) {
        if ((rc = upload(ctx, ctx->d_drop, ctx->h_drop.data(), ctx->h_drop.size()))) return rc;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));        // (h_drop is rebuilt by the next scan)
        *d_drop = (const uint32_t *)ctx->d_drop.p;
    }
    return C3R_OK;
}

// The reference's AF gates divide in float64 (float(count) / denominator >= minimum_af, src/create_tensor_pileup.py:267-285).  For every
// depth below AF_TAB the smallest count that passes is found HERE with that very division, so that the kernels compare integers: the
// quotient grows with the count, hence "count >= threshold" is the same predicate, bit for bit.  65535: no count passes (af > 1).
static int af_table(c3r_ctx *ctx) {
    if (ctx->d_aftab.p && ctx->aftab_snp == ctx->prm.snp_min_af && ctx->aftab_indel == ctx->prm.indel_min_af) return C3R_OK;
    std::vector<uint32_t> tab((size_t)AF_TAB);
    auto thr = [](int d, double af) -> uint32_t {
        const double denom = d > 0 ? (double)d : 1.0;
        const int top = std::max(d, 1);
        // an AF no count can reach (well above 65534 / depth: a user's AF > 1 at every depth of the table; NaN) is "never" at once, not after 65 k
        // divisions (the margin leaves the last ulp to the divisions below)
        if (!(af * denom <= 65536.0)) return 65535u;
        int c = (int)std::ceil(af * denom);                       // near the answer; settled by the division itself
        c = std::max(1, std::min(c, top + 1));
        while (c > 1 && (double)(c - 1) / denom >= af) --c;
        while (c <= top && !((double)c / denom >= af)) ++c;
        // (counts above the depth exist: insertions on ref-skip columns are not part of the depth)
        while (c <= 65534 && !((double)c / denom >= af)) ++c;
        return c > 65534 ? 65535u : (uint32_t)c;
    };
    for (int d = 0; d < AF_TAB; ++d) tab[(size_t)d] = thr(d, ctx->prm.snp_min_af) | (thr(d, ctx->prm.indel_min_af) << 16);
    int rc = upload(ctx, ctx->d_aftab, tab.data(), tab.size());
    if (rc) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));               // (tab is a temporary)
    ctx->aftab_snp = ctx->prm.snp_min_af; ctx->aftab_indel = ctx->prm.indel_min_af;
    return C3R_OK;
}

// the inputs every tile kernel sees (the output side is filled in by the two scan paths)
static void scan_inputs(c3r_ctx *ctx, ScanArgs &a, const uint32_t *d_drop, int drop_words, int n_tiles) {
    memset(&a, 0, sizeof a);
    a.drop = d_drop; a.drop_words = drop_words;
    a.reads = (const DevRead *)ctx->d_reads.p; a.seq = (const uint8_t *)ctx->d_seq.p; a.n_reads = ctx->n_reads; a.compat = ctx->prm.mpileup_compat;
    a.recs = (const PileRec *)ctx->d_recs.p; a.rec_off = (const uint32_t *)ctx->d_binoff.p; a.rtab = (const int4 *)ctx->d_rtab.p; a.bins = ctx->bins;
    a.n_tiles = n_tiles;
    a.ref = (const uint8_t *)ctx->d_ref.p; a.ref_beg0 = (int32_t)(ctx->ref_start1 - 1); a.ref_len = (int32_t)ctx->ref_len;
    a.geo = (const TileGeo *)ctx->d_geo.p;
    a.lbed = (const int32_t *)ctx->d_bed[0].p; a.n_lbed = (int32_t)(ctx->h_bed[0].size() / 2); a.has_lbed = ctx->has_bed[0];
    a.cbed = (const int32_t *)ctx->d_bed[1].p; a.n_cbed = (int32_t)(ctx->h_bed[1].size() / 2); a.has_cbed = ctx->has_bed[1];
    a.sites = (const int32_t *)ctx->d_sites.p; a.n_sites = (int32_t)ctx->h_sites.size(); a.genotyping = ctx->prm.genotyping_mode;
    a.min_mq = ctx->prm.min_mq; a.excl_flags = ctx->prm.excl_flags; a.min_cov = ctx->prm.min_coverage;
    a.snp_af = ctx->prm.snp_min_af; a.indel_af = ctx->prm.indel_min_af; a.af_tab = (const uint32_t *)ctx->d_aftab.p;
    a.head_tail = ctx->prm.head_tail; a.splice = ctx->prm.splice_padding;
    if (ctx->padins && !ctx->padins->empty()) { a.padins = (const c3r_padins_t *)ctx->d_padins.p; a.n_padins = (int32_t)ctx->padins->size(); }
    { const char *e = getenv("C3R_SCAN_ABL"); a.abl = e ? atoi(e) : 0; }
    { const char *e = getenv("C3R_DEEP_MIN"); a.deep_min = e ? std::max(1, atoi(e)) : DEEP_MIN_RECORDS; }
    { const char *e = getenv("C3R_SPLIT_MIN"); a.split_min = e ? std::max(1, atoi(e)) : SPLIT_MIN_RECORDS; }
    { const char *e = getenv("C3R_SPLIT_SLICE"); a.split_slice = e ? std::max(1, atoi(e)) : GIANT_SLICE; }
    { const char *e = getenv("C3R_NO_SHIFT"); a.no_shift = (e && *e == '1') ? 1 : 0; }
}

static int scan_column_store(c3r_ctx *ctx, int32_t n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, int64_t *n_candidates, bool columns_only);
static int scan_fused(c3r_ctx *ctx, int32_t n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, int64_t *n_candidates, bool raw_rerun);

int c3r_pileup_scan(c3r_ctx *ctx, int64_t ctg_start, int64_t ctg_end, int64_t *n_candidates) {
    return c3r_pileup_scan_regions(ctx, 1, &ctg_start, &ctg_end, n_candidates);
}

int c3r_pileup_scan_regions(c3r_ctx *ctx, int32_t n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, int64_t *n_candidates) {
    if (!ctx || n_regions < 1 || !ctg_starts || !ctg_ends) return C3R_EINVAL;
    if (ctx->ref_len == 0) return fail(ctx, C3R_EINVAL, "c3r_set_reference must be called before c3r_pileup_scan");
    for (int r = 0; r < n_regions; ++r) {
        if (ctg_ends[r] < ctg_starts[r]) return C3R_EINVAL;
        if (ctg_ends[r] + C3R_WINDOW > INT32_MAX - 1) return fail(ctx, C3R_EINVAL, "region beyond 2^31");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (n_candidates) *n_candidates = 0;
    { const int rc_af = af_table(ctx); if (rc_af) return rc_af; }
    ctx->last_starts.assign(ctg_starts, ctg_starts + n_regions); ctx->last_ends.assign(ctg_ends, ctg_ends + n_regions);
    ctx->last_scan_pruned = false;
    // The plain mode runs in one fused tile kernel (k_fused_tiles).  Head/tail calling (the end-of-stream rule needs the last row of the
    // whole region), splice padding (in-place column edits, candidate after candidate) and genotyping mode (candidates in intron-only
    // spans) keep the column store and its selection / compaction / gather kernels.
    const bool fused = !ctx->prm.head_tail && !ctx->prm.splice_padding && !ctx->prm.genotyping_mode && !getenv("C3R_NO_FUSE");
    ctx->last_fused = fused;
    return fused ? scan_fused(ctx, n_regions, ctg_starts, ctg_ends, n_candidates, false)
                 : scan_column_store(ctx, n_regions, ctg_starts, ctg_ends, n_candidates, false);
}

// ---- the column-store path: columns, depth and flags of every position go to HBM (k_scan_tiles), then window selection, ordered
// compaction, gather and tokens as separate kernels, with two host read-backs for sizes.  columns_only: c3r_get_columns after a
// fused scan — the columns and flags of the given region, nothing else is touched.
static int scan_column_store(c3r_ctx *ctx, int32_t n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, int64_t *n_candidates, bool columns_only) {
    const int C = ctx->prm.channels;
    // per region, rows: 1-based [max(1, ctg_start-33), ctg_end+33]  (src/create_tensor_pileup.py:411-415); slots: the regions
    // back to back, each padded to whole tiles plus one guard tile (pileup_kernels.hpp, TileGeo)
    std::vector<int64_t> key;
    key.push_back(-1);                                                 // (geometry of the column store)
    for (int r = 0; r < n_regions; ++r) { key.push_back(ctg_starts[r]); key.push_back(ctg_ends[r]); }
    bool geo_changed = key != ctx->geo_key;
    if (geo_changed) {
        ctx->h_geo.clear();
        for (int r = 0; r < n_regions; ++r) {
            int64_t es = ctg_starts[r] - C3R_WINDOW, ee = ctg_ends[r] + C3R_WINDOW;
            if (es < 1) es = 1;
            const int32_t beg0 = (int32_t)(es - 1), end0 = (int32_t)ee;
            for (int32_t p0 = beg0; p0 < end0; p0 += TILE) ctx->h_geo.push_back(TileGeo{p0, std::min(p0 + TILE, end0), r, 0});
            ctx->h_geo.push_back(TileGeo{end0, end0, r, 0});          // guard tile
            if (r == 0) { ctx->reg_beg0 = beg0; ctx->reg_end0 = end0; }
        }
        if ((int64_t)ctx->h_geo.size() * TILE > INT32_MAX - TILE) return fail(ctx, C3R_EINVAL, "regions too large for one scan (2^31 slots)");
        ctx->geo_key = key;
    }
    ctx->n_regions = n_regions;
    const int n_tiles = (int)ctx->h_geo.size();
    ctx->n_pos = (int64_t)n_tiles * TILE;
    if (!columns_only) {
        if (!ctx->batching) { ctx->n_cand = 0; ctx->n_tok = 0; ctx->n_rows = 0; ctx->n_tokspace = 0; }
        ctx->last_cand = 0; ctx->last_base = ctx->n_cand; ctx->tokens_ready = false;
    }
    const int64_t base_cand = ctx->n_cand, base_row = ctx->n_rows, base_tok = ctx->n_tokspace;
    const int64_t n_pos = ctx->n_pos;
    const int n_cblocks = (int)((n_pos + CMP_BLOCK - 1) / CMP_BLOCK);
    int rc;
    if (geo_changed && (rc = upload(ctx, ctx->d_geo, ctx->h_geo.data(), ctx->h_geo.size()))) return rc;
    if ((rc = ensure(ctx, ctx->d_lastrow, (size_t)n_regions * 4 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_cols, (size_t)n_pos * C * 4))) return rc;
    if ((rc = ensure(ctx, ctx->d_depth, (size_t)n_pos * 4))) return rc;
    if ((rc = ensure(ctx, ctx->d_ncov, (size_t)n_pos * 4))) return rc;
    if ((rc = ensure(ctx, ctx->d_flags, (size_t)n_pos))) return rc;
    const size_t ev_cap = event_capacity(ctx, n_regions, ctg_starts, ctg_ends, n_tiles, 1);
    if ((rc = ensure(ctx, ctx->d_ev, ev_cap * sizeof(EvRec)))) return rc;
    if ((rc = ensure(ctx, ctx->d_small, 64))) return rc;
    if ((rc = ensure(ctx, ctx->d_tile_cols, (size_t)n_tiles + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_tile_rng, (size_t)n_tiles * 16 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_tile_list, (size_t)n_tiles * 4 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_tile_list2, (size_t)n_tiles * 4 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_tile_cand, (size_t)n_tiles * 8 + 16))) return rc;
    if ((rc = ensure(ctx, ctx->d_blockcnt, (size_t)(n_cblocks + 1) * 4))) return rc;
    if (ctx->prm.splice_padding && (rc = ensure(ctx, ctx->d_skipmax, (size_t)n_pos * 4))) return rc;
    // d_small: [0..7] ev_cursor (u64), [8..11] event-scratch overflow flag, [12..15] n_cand, [16..19] n_tok, [20..23] n_tile_list,
    //          [24..27] n_tile_list2 (pruned intron-only tiles), [28..31] tokens written, [32..35] tokens that found no slot (k_tile_tokens)
    int32_t init[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_small.p, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_lastrow.p, 0xff, (size_t)n_regions * 4, ctx->stream));      // -1
    HIPCHK(ctx, hipMemsetAsync(ctx->d_flags.p, 0, (size_t)n_pos, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cols.p, 0, (size_t)n_tiles, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_tile_cand.p, 0, (size_t)n_tiles * 8, ctx->stream));

    const uint32_t *d_drop = nullptr;
    int drop_words = 0;
    if ((rc = depth_cap_mask(ctx, n_regions, ctg_starts, ctg_ends, &d_drop, &drop_words))) return rc;

    ScanArgs a;
    scan_inputs(ctx, a, d_drop, drop_words, n_tiles);
    a.tile_cols = (uint8_t *)ctx->d_tile_cols.p;
    a.tile_rng = (int4 *)ctx->d_tile_rng.p; a.tile_list = (int32_t *)ctx->d_tile_list.p;
    a.n_tile_list = (int32_t *)((char *)ctx->d_small.p + 20);
    // rows of intron-only tiles far from every aligned segment only matter to the end-of-stream rule (head/tail), to splice
    // padding and to genotyping sites: in the plain mode they are left out of the scan
    a.prune = (!columns_only && !ctx->prm.head_tail && !ctx->prm.splice_padding && !ctx->prm.genotyping_mode && !getenv("C3R_NO_PRUNE")) ? 1 : 0;
    a.tile_list2 = (int32_t *)ctx->d_tile_list2.p; a.n_tile_list2 = (int32_t *)((char *)ctx->d_small.p + 24);
    a.dbg = nullptr;
    if (getenv("C3R_SCAN_DBG")) {
        if ((rc = ensure(ctx, ctx->d_dbg, 32 * 8))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_dbg.p, 0, 32 * 8, ctx->stream));
        a.dbg = (unsigned long long *)ctx->d_dbg.p;
    }
    a.cols = (int32_t *)ctx->d_cols.p; a.depth = (int32_t *)ctx->d_depth.p; a.ncov = (int32_t *)ctx->d_ncov.p; a.flags = (uint8_t *)ctx->d_flags.p;
    a.ev = (EvRec *)ctx->d_ev.p; a.ev_cursor = (unsigned long long *)ctx->d_small.p; a.last_row = (int32_t *)ctx->d_lastrow.p;
    a.ev_cap = (unsigned long long)ev_cap; a.ev_overflow = (int32_t *)((char *)ctx->d_small.p + 8);
    a.skipmax = (int32_t *)ctx->d_skipmax.p;
    if (!columns_only) { ctx->last_scan = a; ctx->last_scan_pruned = a.prune && a.n_reads > 0; }
    if (a.n_reads > 0) {
        Launch L(ctx, "k_tile_ranges");
        hipLaunchKernelGGL(k_tile_ranges, dim3((n_tiles + 255) / 256), dim3(256), 0, ctx->stream, a);
    }
    if (a.n_reads > 0) {
        Launch L(ctx, "k_scan_tiles");
        if (C == C3R_CH) hipLaunchKernelGGL(k_scan_tiles<C3R_CH>, dim3(std::min(n_tiles, list_grid())), dim3(SCAN_THREADS), 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_scan_tiles<C3R_CH_PHASED>, dim3(std::min(n_tiles, list_grid())), dim3(SCAN_THREADS), 0, ctx->stream, a);
    }
    if (a.n_reads > 0 && C == C3R_CH_PHASED) {
        if ((rc = ensure_legacy_tables(ctx))) return rc;
        PhaseArgs f;
        f.tile_list = a.tile_list; f.n_tile_list = a.n_tile_list; f.tile_rng = a.tile_rng; f.geo = a.geo;
        f.reads = a.reads; f.rsegs = (const DevSeg *)ctx->d_rsegs.p; f.rseg_first = (const uint32_t *)ctx->d_rseg_first.p;
        f.cigar = (const uint32_t *)ctx->d_cigar.p; f.seq = a.seq; f.flags = a.flags; f.cols = a.cols; f.min_mq = a.min_mq; f.excl_flags = a.excl_flags;
        f.drop = a.drop; f.drop_words = a.drop_words;
        Launch L(ctx, "k_phase_recompute");
        hipLaunchKernelGGL(k_phase_recompute, dim3(std::min(n_tiles, list_grid())), dim3(TILE), 0, ctx->stream, f);
    }
    if (a.n_reads > 0 && a.splice) {
        HIPCHK(ctx, hipMemsetAsync(ctx->d_skipmax.p, 0, (size_t)n_pos * 4, ctx->stream));
        Launch L(ctx, "k_skip_counts");
        hipLaunchKernelGGL(k_skip_counts, dim3(std::min(n_tiles, list_grid())), dim3(SCAN_THREADS), 0, ctx->stream, a);
    }
    // candidates live in tiles that hold aligned bases — except in genotyping mode, where any row of the site list is one
    const uint8_t *heavy = ctx->prm.genotyping_mode ? nullptr : (const uint8_t *)ctx->d_tile_cols.p;
    {
        Launch L(ctx, "k_select");
        hipLaunchKernelGGL(k_select, dim3(std::min(n_tiles, list_grid())), dim3(TILE), 0, ctx->stream, (uint8_t *)ctx->d_flags.p,
                           (int)n_pos, (const TileGeo *)ctx->d_geo.p, ctx->prm.head_tail, (const int32_t *)ctx->d_lastrow.p, heavy,
                           (const int32_t *)ctx->d_tile_list.p, (const int32_t *)((char *)ctx->d_small.p + 20));
    }
    if (columns_only) { HIPCHK(ctx, hipGetLastError()); return C3R_OK; }
    {
        Launch L(ctx, "k_compact_count");
        hipLaunchKernelGGL(k_compact_count, dim3(n_cblocks), dim3(CMP_THREADS), 0, ctx->stream, (const uint8_t *)ctx->d_flags.p, (int)n_pos,
                           (int32_t *)ctx->d_blockcnt.p, heavy);
    }
    {
        if ((rc = device_excl_scan(ctx, (int32_t *)ctx->d_blockcnt.p, n_cblocks, (int32_t *)((char *)ctx->d_small.p + 12)))) return rc;
    }
    int32_t flag_cand[2] = {0, 0};       // event-scratch overflow flag, n_cand
    HIPCHK(ctx, hipMemcpyAsync(flag_cand, (char *)ctx->d_small.p + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    if (a.dbg) {
        unsigned long long d[16];
        HIPCHK(ctx, hipMemcpy(d, ctx->d_dbg.p, sizeof d, hipMemcpyDeviceToHost));
        const double nt = d[15] ? (double)d[15] : 1.0;
        fprintf(stderr, "[k_scan_tiles] %llu heavy tiles; per tile: segments in range %.1f, listed %.1f, ops %.1f; us per tile: zero %.2f | cover+list+walk %.2f | scans %.2f | events %.2f | gates %.2f | store %.2f | first-seen %.2f\n",
                d[15], d[13] / nt, d[12] / nt, d[14] / nt, d[0] / nt / 100, d[1] / nt / 100, d[2] / nt / 100, d[3] / nt / 100, d[4] / nt / 100, d[5] / nt / 100, d[6] / nt / 100);
    }
    if ((flag_cand[0] & 2) && !ctx->win32) {
        ctx->last_scan_pruned = false;
        if (columns_only) { /* columns are int32 anyway */ }
        else if (base_row != 0) return fail(ctx, C3R_EOVERFLOW, "a position is covered by more than 32,767 reads and the batch already holds 16-bit windows: scan this region first, or alone");
        else { ctx->win32 = true; return scan_column_store(ctx, n_regions, ctg_starts, ctg_ends, n_candidates, columns_only); }
    }
    if (flag_cand[0] & 1) { ctx->last_scan_pruned = false; return fail(ctx, C3R_EOVERFLOW, "internal: indel-event scratch too small (%zu records) — nothing was written past it", ev_cap); }
    const int32_t n_cand = flag_cand[1];
    ctx->last_cand = n_cand;
    if (n_candidates) *n_candidates = n_cand;
    if (n_cand == 0) return C3R_OK;
    const size_t tbytes = (size_t)C3R_WINDOW * C * (ctx->win32 ? sizeof(int32_t) : sizeof(int16_t));          // the resident windows are int16 (a caller sees int32: c3r_get_tensors)
    if ((rc = ensure(ctx, ctx->d_cand, (size_t)n_cand * 4))) return rc;
    if ((rc = ensure_keep(ctx, ctx->d_tensors, (size_t)(base_row + n_cand) * tbytes, (size_t)base_row * tbytes))) return rc;
    if ((rc = ensure_keep(ctx, ctx->d_sites_out, (size_t)(base_cand + n_cand) * sizeof(c3r_site_t), (size_t)base_cand * sizeof(c3r_site_t)))) return rc;
    if ((rc = ensure(ctx, ctx->d_tokcnt, (size_t)(n_cand + 1) * 4))) return rc;
    if (ctx->prm.splice_padding && (rc = ensure(ctx, ctx->d_raw, (size_t)n_cand * C3R_WINDOW * C * sizeof(int32_t)))) return rc;      // (raw windows: int32)
    {
        Launch L(ctx, "k_compact_write");
        hipLaunchKernelGGL(k_compact_write, dim3(n_cblocks), dim3(CMP_THREADS), 0, ctx->stream, (const uint8_t *)ctx->d_flags.p, (int)n_pos,
                           (const int32_t *)ctx->d_blockcnt.p, (int32_t *)ctx->d_cand.p, heavy, (int2 *)ctx->d_tile_cand.p);
    }
    if ((rc = run_gather(ctx, 1, (char *)ctx->d_tensors.p + (size_t)base_row * tbytes, true, !ctx->win32))) return rc;
    if ((rc = ensure_keep(ctx, ctx->d_winidx, (size_t)(base_cand + n_cand) * 4, (size_t)base_cand * 4))) return rc;
    hipLaunchKernelGGL(k_iota, dim3((unsigned)((n_cand + 255) / 256)), dim3(256), 0, ctx->stream, (int32_t *)ctx->d_winidx.p + base_cand, (int)n_cand, (int)base_row);      // (windows in site order)
    {
        // token SLOTS per candidate (the gates' bound of the reads with something to say, TileOut::cov) -> first slots; [n_cand] = their total
        HIPCHK(ctx, hipMemsetAsync((int32_t *)ctx->d_tokcnt.p + n_cand, 0, 4, ctx->stream));
        if ((rc = device_excl_scan(ctx, (int32_t *)ctx->d_tokcnt.p, (int)n_cand + 1, (int32_t *)((char *)ctx->d_small.p + 16)))) return rc;
    }
    int32_t n_tok = 0;                   // slots
    HIPCHK(ctx, hipMemcpyAsync(&n_tok, (char *)ctx->d_small.p + 16, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = ensure_keep(ctx, ctx->d_tok, std::max<size_t>((size_t)(base_tok + n_tok) * sizeof(c3r_token_t), 16),
                          (size_t)base_tok * sizeof(c3r_token_t)))) return rc;
    {
        TileTokArgs t;
        t.a = a;
        t.cand_idx = (const int32_t *)ctx->d_cand.p; t.tile_cand = (const int2 *)ctx->d_tile_cand.p;
        t.tok_off = (const int32_t *)ctx->d_tokcnt.p; t.sites = (c3r_site_t *)ctx->d_sites_out.p + base_cand;
        t.tok = (c3r_token_t *)ctx->d_tok.p; t.tok_base = (int32_t)base_tok;     // (sized exactly above)
        t.totals = (int32_t *)((char *)ctx->d_small.p + 28);
        Launch L(ctx, "k_tokens");
        hipLaunchKernelGGL(k_tile_tokens, dim3(std::min(n_tiles, list_grid())), dim3(SCAN_THREADS), 0, ctx->stream, t);
    }
    int32_t tok_done[2] = {0, 0};        // tokens written, tokens that found no slot
    HIPCHK(ctx, hipMemcpyAsync(tok_done, (char *)ctx->d_small.p + 28, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (tok_done[1]) return fail(ctx, C3R_EOVERFLOW, "internal: %d tokens found no slot (the gates' bound of a site's tokens was too small)", tok_done[1]);
    ctx->tokens_ready = true;
    ctx->n_cand = base_cand + n_cand;
    ctx->n_tok += tok_done[0];
    ctx->n_rows = base_row + n_cand; ctx->n_tokspace = base_tok + n_tok;
    HIPCHK(ctx, hipGetLastError());
    return C3R_OK;
}

// ---- the fused path (pileup_kernels.hpp, k_fused_tiles): memset of the look-back words, k_tile_ranges_fused, k_fused_tiles (windows
// AND tokens), k_order_spans, k_finalize_sites, and ONE read-back at the end (totals + overflow flags).  Output buffers are sized from what earlier passes needed;
// kernels never write past them, and when the totals say something did not fit the buffers grow and the scan is repeated (the first
// pass of a context; steady-state passes run once, without talking to the host in between).
// raw_rerun: c3r_get_tensors(rescaled = 0) — the same scan again, un-rescaled windows only, into d_raw.
static int scan_fused(c3r_ctx *ctx, int32_t n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, int64_t *n_candidates, bool raw_rerun) {
    const int C = ctx->prm.channels;
    // per region, rows: 1-based [max(1, ctg_start-33), ctg_end+33]  (src/create_tensor_pileup.py:411-415), cut into spans of FUSE_IN
    // positions; a span's slots are tile * TILE + (p - p0)
    std::vector<int64_t> key;
    key.push_back(-2);                                                 // (geometry of the fused path)
    for (int r = 0; r < n_regions; ++r) { key.push_back(ctg_starts[r]); key.push_back(ctg_ends[r]); }
    const bool geo_changed = key != ctx->geo_key;
    int rc;
    if (geo_changed) {
        ctx->h_geo.clear();
        std::vector<int2> regb((size_t)n_regions);
        for (int r = 0; r < n_regions; ++r) {
            int64_t es = ctg_starts[r] - C3R_WINDOW, ee = ctg_ends[r] + C3R_WINDOW;
            if (es < 1) es = 1;
            const int32_t beg0 = (int32_t)(es - 1), end0 = (int32_t)ee;
            for (int32_t p0 = beg0; p0 < end0; p0 += FUSE_IN) ctx->h_geo.push_back(TileGeo{p0, std::min(p0 + FUSE_IN, end0), r, 0});
            regb[(size_t)r] = make_int2(beg0, end0);
            if (r == 0) { ctx->reg_beg0 = beg0; ctx->reg_end0 = end0; }
        }
        if ((int64_t)ctx->h_geo.size() * TILE > INT32_MAX - TILE) { ctx->geo_key.clear(); return fail(ctx, C3R_EINVAL, "regions too large for one scan (2^31 slots)"); }
        ctx->geo_key = key;
        if ((rc = upload(ctx, ctx->d_geo, ctx->h_geo.data(), ctx->h_geo.size())) || (rc = upload(ctx, ctx->d_regb, regb.data(), regb.size()))) { ctx->geo_key.clear(); return rc; }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));               // (regb is a temporary)
    }
    ctx->n_regions = n_regions;
    const int n_tiles = (int)ctx->h_geo.size();
    ctx->n_pos = (int64_t)n_tiles * TILE;
    if (!raw_rerun) {
        if (!ctx->batching) { ctx->n_cand = 0; ctx->n_tok = 0; ctx->n_rows = 0; ctx->n_tokspace = 0; }
        ctx->last_cand = 0; ctx->last_base = ctx->n_cand; ctx->tokens_ready = false;
    }
    const int64_t base_cand = raw_rerun ? 0 : ctx->n_cand, base_row = raw_rerun ? 0 : ctx->n_rows, base_tok = raw_rerun ? 0 : ctx->n_tokspace;
    if (ctx->n_reads == 0 || n_tiles == 0) return C3R_OK;
    const size_t ev_cap = event_capacity(ctx, n_regions, ctg_starts, ctg_ends, n_tiles, 2);
    const int nblk = (n_tiles + 255) / 256;
    // d_lb: [0] ticket of k_tile_ranges_fused, [4] ticket of k_deep_alleles, [8] candidates, [12] tokens (k_order_spans), [16] overflow
    // bits, [20] listed spans, [24..31] event-scratch cursor, [32] event-scratch overflow, [36] rows handed out, [40] ticket of
    // k_order_spans, [44] deep spans listed, [48] ticket of k_fused_deep, [52] giant spans met, [56] their slices listed, [60] ticket of k_deep_walk, [64..] the TICKET_Q ticket words of k_fused_tiles, 256 bytes apart, then the ALLOC_SHARDS allocator words (tokens << 32 |
    // rows), 256 bytes apart; then look-back words: one per block of 256 spans for k_tile_ranges_fused, then the same for k_order_spans
    const size_t lb_alloc = 64 + (size_t)TICKET_Q * TICKET_STRIDE * 4;
    const size_t lb_head = lb_alloc + (size_t)ALLOC_SHARDS * ALLOC_STRIDE * 8;
    const size_t lb_bytes = lb_head + (size_t)nblk * 24;         // (look-back words: k_tile_ranges_fused' sums, k_order_spans', k_tile_ranges_fused' run starts)
    if ((rc = ensure(ctx, ctx->d_ev, ev_cap * sizeof(EvRec))) || (rc = ensure(ctx, ctx->d_lb, lb_bytes)) || (rc = ensure(ctx, ctx->d_tile_rng, (size_t)n_tiles * 16 + 16)) ||
        (rc = ensure(ctx, ctx->d_tile_list, (size_t)n_tiles * 4 + 16)) ||
        (rc = ensure(ctx, ctx->d_span, (size_t)n_tiles * sizeof(int4) + 16)) || (rc = ensure(ctx, ctx->d_spanbase, (size_t)n_tiles * 4 + 16)) ||
        (rc = ensure(ctx, ctx->d_spanrec, (size_t)n_tiles * sizeof(SpanRec) + 16)) || (rc = ensure(ctx, ctx->d_deep, (size_t)n_tiles * 4 + 16)))
        return rc;
    if (!ctx->h_scan) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_scan, 64 + (size_t)ALLOC_SHARDS * ALLOC_STRIDE * 8, hipHostMallocDefault));
    const uint32_t *d_drop = nullptr;
    int drop_words = 0;
    if ((rc = depth_cap_mask(ctx, n_regions, ctg_starts, ctg_ends, &d_drop, &drop_words))) return rc;
    char *lb = (char *)ctx->d_lb.p;
    FusedArgs f;
    memset(&f, 0, sizeof f);
    ScanArgs &a = f.a;
    scan_inputs(ctx, a, d_drop, drop_words, n_tiles);
    a.tile_rng = (int4 *)ctx->d_tile_rng.p; a.tile_list = (int32_t *)ctx->d_tile_list.p; a.n_tile_list = (int32_t *)(lb + 20);
    a.ev = (EvRec *)ctx->d_ev.p; a.ev_cursor = (unsigned long long *)(lb + 24); a.ev_cap = (unsigned long long)ev_cap; a.ev_overflow = (int32_t *)(lb + 32);
    if (getenv("C3R_SCAN_DBG")) {          // phase clocks of tile_columns (timing aid, off in production)
        if ((rc = ensure(ctx, ctx->d_dbg, 32 * 8))) return rc;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_dbg.p, 0, 32 * 8, ctx->stream));
        a.dbg = (unsigned long long *)ctx->d_dbg.p;
    }
    f.span_rec = (const SpanRec *)ctx->d_spanrec.p;
    // the deep kernel's per-workgroup event buffers (every event of a span in arrival order: a span that outgrows the LDS store is bucketed from there
    // instead of walking its records again): 300 MB, so only for a context that has met a deep span (its first such scan walks twice)
    const int deep_grid = std::min(n_tiles, ctx->n_cu > 0 ? ctx->n_cu : 256);
    const char *evwg_env = getenv("C3R_EVWG");          // (tests) 0: never, 1: from the first scan on
    if ((ctx->seen_deep || (evwg_env && *evwg_env == '1')) && !(evwg_env && *evwg_env == '0')) {
        if ((rc = ensure(ctx, ctx->d_evwg, (size_t)deep_grid * DEEP_EVG_CAP * sizeof(EvRec)))) return rc;
        a.ev_wg = (EvRec *)ctx->d_evwg.p; a.ev_wg_cap = DEEP_EVG_CAP;
    }
    // giant spans (k_deep_walk, k_deep_alleles): 64 accumulator slots + the slice list (2 MB), a pool of 12 M events (288 MB) and their allele table
    // (192 MB), for a context that has met one
    const char *giant_env = getenv("C3R_GIANT");        // (tests) 0: never, 1: from the first scan on
    const size_t giant_acc_bytes = (size_t)GIANT_SLOTS * GIANT_STRIDE * 4 + 16;          // (+ the pool's cursor: cleared with the slots)
    a.n_giant = (int32_t *)(lb + 52); a.n_help = (int32_t *)(lb + 56);
    if ((ctx->seen_giant || (giant_env && *giant_env == '1')) && !(giant_env && *giant_env == '0')) {
        if ((rc = ensure(ctx, ctx->d_giant, giant_acc_bytes + (size_t)GIANT_SLOTS * GIANT_MAX_HELP * sizeof(int4))) ||
            (rc = ensure(ctx, ctx->d_giant_ev, (size_t)GIANT_POOL_EVENTS * sizeof(EvRec)))) return rc;
        a.giant_acc = (int32_t *)ctx->d_giant.p; a.help_list = (int4 *)((char *)ctx->d_giant.p + giant_acc_bytes); a.giant_ev = (EvRec *)ctx->d_giant_ev.p;
        a.giant_pool = GIANT_POOL_EVENTS; a.n_giant_ev = (int32_t *)((char *)ctx->d_giant.p + giant_acc_bytes - 16);
        a.deep_recs = (unsigned long long *)((char *)ctx->d_giant.p + giant_acc_bytes - 8);
        { const char *e = getenv("C3R_SPLIT_CUS"); a.split_cus = e ? std::max(1, atoi(e)) : (ctx->n_cu > 0 ? ctx->n_cu : 256); }
        if (!getenv("C3R_NO_GIANT_TAB")) {
            // (the allele table is all-zero between scans: cleared here when it is new, by k_fused_deep as it reads afterwards)
            const size_t tab_bytes = (size_t)GIANT_POOL_EVENTS * 2 * sizeof(uint2);
            const bool fresh = ctx->d_giant_tab.cap < tab_bytes;
            if ((rc = ensure(ctx, ctx->d_giant_tab, tab_bytes))) return rc;
            if (fresh) HIPCHK(ctx, hipMemsetAsync(ctx->d_giant_tab.p, 0, tab_bytes, ctx->stream));
            a.giant_tab = (uint2 *)ctx->d_giant_tab.p;
        }
    }
    f.ticket = (int32_t *)(lb + 64); f.alloc = (unsigned long long *)(lb + lb_alloc); f.overflow = (int32_t *)(lb + 16);
    f.rescale = raw_rerun ? 0 : 1; f.max_depth = ctx->prm.max_depth_rescale;
    f.span_info = (int4 *)ctx->d_span.p;
    // 30 channels: the per-read op / segment tables behind the ordered recompute of a flagged column (an IUPAC read base, an indel right
    // behind a ref-skip) are built only for a context that has met such a column: aligners emit neither, and the tables cost 0.12 ms of
    // a 0.7-ms MAS-Seq chr20 pass.  Without them a flagged column raises bit 2 of the scan's flags and the scan is repeated with them.
    const bool env_eager = [] { const char *e = getenv("C3R_LEGACY_EAGER"); return e && *e == '1'; }();
    if (C == C3R_CH_PHASED && (ctx->legacy_wanted || env_eager) && (rc = ensure_legacy_tables(ctx))) return rc;
    const bool have_tables = C == C3R_CH_PHASED && ctx->legacy_valid;
    f.ph.reads = a.reads; f.ph.rsegs = have_tables ? (const DevSeg *)ctx->d_rsegs.p : nullptr; f.ph.rseg_first = have_tables ? (const uint32_t *)ctx->d_rseg_first.p : nullptr;
    f.ph.cigar = (const uint32_t *)ctx->d_cigar.p; f.ph.seq = a.seq;
    f.ph.min_mq = a.min_mq; f.ph.excl_flags = a.excl_flags; f.ph.drop = a.drop; f.ph.drop_words = a.drop_words;
    const size_t tbytes = (size_t)C3R_WINDOW * C * (ctx->win32 ? sizeof(int32_t) : sizeof(int16_t));          // the resident windows are int16 (a caller sees int32: c3r_get_tensors)
    // What this scan may write.  A small scan has one allocator and dense output: whatever the buffers can take beyond what the batch
    // already holds.  A large one hands rows and token slots out through ALLOC_SHARDS sub-allocators of equal size, sized from what the
    // previous large scan needed; when a shard runs over, the scan is repeated with what the counters say it needs.
    const int nsh = raw_rerun ? ctx->last_shards : ((n_tiles >= 4096 && !getenv("C3R_ONE_SHARD")) ? ALLOC_SHARDS : 1);
    int64_t want_c, want_t;            // per shard
    if (raw_rerun) { want_c = ctx->last_shard_rows; want_t = 0; }
    else if (nsh == 1) {
        want_c = std::min<int64_t>({(int64_t)(ctx->d_tensors.cap / tbytes) - base_row, (int64_t)(ctx->d_sites_out.cap / sizeof(c3r_site_t)) - base_cand,
                                    (int64_t)(ctx->d_winidx.cap / 4) - base_cand, (int64_t)(ctx->d_cand.cap / 4), (int64_t)(ctx->d_meta.cap / sizeof(CandMeta))});
        want_t = (int64_t)(ctx->d_tok.cap / sizeof(c3r_token_t)) - base_tok;
        if (want_c < 1024) want_c = 65536;
        if (want_t < 1024) want_t = 32 * want_c;
    } else { want_c = ctx->hint_rows; want_t = ctx->hint_toks; }
    int32_t n_cand = 0, n_tok = 0;
    for (int attempt = 0;; ++attempt) {
        FinalizeArgs z;
        memset(&z, 0, sizeof z);
        want_c = std::min<int64_t>(want_c, INT32_MAX / nsh); want_t = std::min<int64_t>(want_t, (INT32_MAX - base_tok) / nsh);
        const int64_t rows = want_c * nsh, toks = want_t * nsh;
        if ((rc = ensure(ctx, ctx->d_meta, std::max<size_t>((size_t)rows * sizeof(CandMeta), 16)))) return rc;
        if (raw_rerun) {
            if ((rc = ensure(ctx, ctx->d_raw, std::max<size_t>((size_t)rows * C3R_WINDOW * C * sizeof(int32_t), 16))) || (rc = ensure(ctx, ctx->d_rawidx, std::max<size_t>((size_t)rows * 4, 16)))) return rc;
            f.tensors = ctx->d_raw.p; f.x16 = 0;
            z.win_idx = (int32_t *)ctx->d_rawidx.p; z.row_base = 0;
        } else {
            if ((rc = ensure(ctx, ctx->d_cand, (size_t)rows * 4)) ||
                (rc = ensure_keep(ctx, ctx->d_tensors, (size_t)(base_row + rows) * tbytes, (size_t)base_row * tbytes)) ||
                (rc = ensure_keep(ctx, ctx->d_sites_out, (size_t)(base_cand + rows) * sizeof(c3r_site_t), (size_t)base_cand * sizeof(c3r_site_t))) ||
                (rc = ensure_keep(ctx, ctx->d_winidx, (size_t)(base_cand + rows) * 4, (size_t)base_cand * 4)) ||
                (rc = ensure_keep(ctx, ctx->d_tok, (size_t)(base_tok + toks) * sizeof(c3r_token_t), (size_t)base_tok * sizeof(c3r_token_t))))
                return rc;
            f.tensors = (char *)ctx->d_tensors.p + (size_t)base_row * tbytes; f.x16 = ctx->win32 ? 0 : 1;
            z.sites = (c3r_site_t *)ctx->d_sites_out.p + base_cand;
            z.cand_idx = (int32_t *)ctx->d_cand.p;
            z.win_idx = (int32_t *)ctx->d_winidx.p + base_cand; z.row_base = (int32_t)base_row; z.tok_base = (int32_t)base_tok;
            f.tok = (c3r_token_t *)ctx->d_tok.p; f.tok_base = (int32_t)base_tok; f.tok_cap = (int32_t)toks;
        }
        f.meta = (CandMeta *)ctx->d_meta.p;
        f.n_shards = nsh; f.shard_rows = (int32_t)want_c; f.shard_toks = (int32_t)want_t;
        f.cand_cap = (int32_t)rows;
        z.meta = f.meta; z.span_info = f.span_info; z.span_base = (const int32_t *)ctx->d_spanbase.p; z.alloc = f.alloc; z.n_shards = nsh; z.shard_rows = f.shard_rows; z.overflow = f.overflow;
        z.ref = a.ref; z.ref_beg0 = a.ref_beg0; z.ref_len = a.ref_len;
        HIPCHK(ctx, hipMemsetAsync(ctx->d_lb.p, 0, lb_bytes, ctx->stream));
        if (a.giant_acc) HIPCHK(ctx, hipMemsetAsync(a.giant_acc, 0, giant_acc_bytes, ctx->stream));
        {
            Launch L(ctx, "k_tile_ranges");
            hipLaunchKernelGGL(k_tile_ranges_fused, dim3(nblk), dim3(256), 0, ctx->stream, a, (int32_t *)lb, (unsigned long long *)(lb + lb_head), nblk, (const int2 *)ctx->d_regb.p,
                               (SpanRec *)ctx->d_spanrec.p, (int32_t *)ctx->d_deep.p, (int32_t *)(lb + 44), (unsigned long long *)(lb + lb_head + (size_t)nblk * 16));
        }
        // the deep spans k_fused_tiles leaves out go to k_fused_deep: one workgroup of sixteen wavefronts per CU, spans by ticket (no deep span: the
        // workgroups leave at once).  The two kernels share nothing but the allocators and run side by side, the deep one on its own stream; under
        // the profiler (one kernel at a time, each timed on the context's stream) they run one after the other.
        const bool side = !ctx->profiling && !getenv("C3R_DEEP_SERIAL");
        // the giant spans' walks and allele counts, slice by slice on every CU (two workgroups each: the allele pass is round trips), before the deep
        // kernel takes the spans
        static const int wg_mul = [] { const char *e = getenv("C3R_GIANT_WGS"); return e ? std::max(1, atoi(e)) : 2; }();
        const int help_grid = (ctx->n_cu > 0 ? ctx->n_cu : 256) * wg_mul;
        auto launch_walk = [&](hipStream_t st) {
            WalkArgs w{(const SpanRec *)ctx->d_spanrec.p, (int32_t *)(lb + 60)};
            if (C == C3R_CH) hipLaunchKernelGGL(k_deep_walk<C3R_CH>, dim3(help_grid), dim3(DEEP_THREADS), 0, st, a, w);
            else hipLaunchKernelGGL(k_deep_walk<C3R_CH_PHASED>, dim3(help_grid), dim3(DEEP_THREADS), 0, st, a, w);
        };
        auto launch_alleles = [&](hipStream_t st) {
            if (C == C3R_CH) hipLaunchKernelGGL(k_deep_alleles<C3R_CH>, dim3(help_grid), dim3(DEEP_THREADS), 0, st, a, (int32_t *)(lb + 4));
            else hipLaunchKernelGGL(k_deep_alleles<C3R_CH_PHASED>, dim3(help_grid), dim3(DEEP_THREADS), 0, st, a, (int32_t *)(lb + 4));
        };
        auto launch_deep = [&](hipStream_t st) {
            DeepArgs d{(const int32_t *)ctx->d_deep.p, (const int32_t *)(lb + 44), (int32_t *)(lb + 48)};
            if (C == C3R_CH) hipLaunchKernelGGL(k_fused_deep<C3R_CH>, dim3(deep_grid), dim3(DEEP_THREADS), 0, st, f, d);
            else hipLaunchKernelGGL(k_fused_deep<C3R_CH_PHASED>, dim3(deep_grid), dim3(DEEP_THREADS), 0, st, f, d);
        };
        if (side) {
            if (!ctx->deep_stream) {
                // (the device's highest priority: the deep spans are the long ones — their sixteen-wavefront workgroups should get their CUs before
                // k_fused_tiles' small workgroups have filled every CU's LDS, not after)
                int least = 0, greatest = 0;
                if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); greatest = 0; }
                HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->deep_stream, hipStreamNonBlocking, greatest));
                HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
                HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
            }
            HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->deep_stream, ctx->ev_fork, 0));
            if (a.giant_acc) launch_walk(ctx->deep_stream);
            if (a.giant_tab) launch_alleles(ctx->deep_stream);
            launch_deep(ctx->deep_stream);
            HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->deep_stream));
        }
        {
            Launch L(ctx, "k_fused_tiles");
            const int grid = std::min(n_tiles, 2048);
            if (C == C3R_CH) hipLaunchKernelGGL(k_fused_tiles<C3R_CH>, dim3(grid), dim3(SCAN_THREADS), 0, ctx->stream, f);
            else hipLaunchKernelGGL(k_fused_tiles<C3R_CH_PHASED>, dim3(grid), dim3(SCAN_THREADS), 0, ctx->stream, f);
        }
        if (side) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        else {
            if (a.giant_acc) { Launch L(ctx, "k_deep_walk"); launch_walk(ctx->stream); }
            if (a.giant_tab) { Launch L(ctx, "k_deep_alleles"); launch_alleles(ctx->stream); }
            Launch L(ctx, "k_fused_deep");
            launch_deep(ctx->stream);
        }
        {
            Launch L(ctx, "k_order_sites");
            hipLaunchKernelGGL(k_order_spans, dim3(nblk), dim3(256), 0, ctx->stream, (const int4 *)ctx->d_span.p, (const int32_t *)(lb + 20), (int32_t *)(lb + 40),
                               (unsigned long long *)(lb + lb_head + (size_t)nblk * 8), (int32_t *)ctx->d_spanbase.p, (int32_t *)(lb + 8));
            hipLaunchKernelGGL(k_finalize_sites, dim3((unsigned)std::min<int64_t>((rows + 15) / 16 + 1, 8192)), dim3(256), 0, ctx->stream, z);
        }
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_scan, lb + 8, 48, hipMemcpyDeviceToHost, ctx->stream));          // (.. [9]: deep spans listed)
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_scan + 16, lb + lb_alloc, (size_t)ALLOC_SHARDS * ALLOC_STRIDE * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipGetLastError());
        n_cand = ctx->h_scan[0]; n_tok = ctx->h_scan[1];
        if (ctx->h_scan[9] > 0) ctx->seen_deep = true;
        if (ctx->h_scan[9] > 0 && ctx->h_scan[11] != 0) ctx->seen_giant = true;
        int64_t need_c = 0, need_t = 0;          // the fullest shard
        for (int sh = 0; sh < nsh; ++sh) {
            const unsigned long long v = ((const unsigned long long *)(ctx->h_scan + 16))[(size_t)sh * ALLOC_STRIDE];
            need_c = std::max<int64_t>(need_c, (int64_t)(uint32_t)v); need_t = std::max<int64_t>(need_t, (int64_t)(v >> 32));
        }
        if (a.dbg && a.giant_acc) {
            // events met by the giant spans' walks, slot by slot
            std::vector<int32_t> evn(GIANT_SLOTS, 0);
            for (int g = 0; g < GIANT_SLOTS; ++g)
                HIPCHK(ctx, hipMemcpy(&evn[g], a.giant_acc + (size_t)g * GIANT_STRIDE + GIANT_META, 4, hipMemcpyDeviceToHost));
            std::sort(evn.begin(), evn.end(), std::greater<int32_t>());
            fprintf(stderr, "[giant spans] %d met; events per slot, largest first:", ctx->h_scan[11]);
            for (int g = 0; g < GIANT_SLOTS && evn[g] > 0; ++g) fprintf(stderr, " %d", evn[g]);
            fprintf(stderr, "\n");
        }
        if (a.dbg) {
            unsigned long long d[32];
            HIPCHK(ctx, hipMemcpy(d, ctx->d_dbg.p, sizeof d, hipMemcpyDeviceToHost));
            const double nt = d[15] ? (double)d[15] : 1.0;
            fprintf(stderr, "[k_fused_tiles] %llu spans; per span: reads in range %.1f, records in range %.1f, indel events %.1f; us per span: zero %.2f | cover+walk %.2f | scans %.2f | event buckets (LDS store) %.2f | events %.2f | gates %.2f | first-seen %.2f | select+store %.2f | tokens %.2f\n",
                    d[15], d[13] / nt, d[14] / nt, d[12] / nt, d[0] / nt / 100, d[1] / nt / 100, d[2] / nt / 100, d[5] / nt / 100, d[3] / nt / 100, d[4] / nt / 100, d[6] / nt / 100, d[7] / nt / 100, d[8] / nt / 100);
            fprintf(stderr, "   tail: row mask %.2f | scan %.2f | allocator atomic %.2f us\n", d[9] / nt / 100, d[10] / nt / 100, d[11] / nt / 100);
            fprintf(stderr, "   LDS event store: leader search (thread 0) %.2f | wait for the others %.2f us\n", d[16] / nt / 100, d[17] / nt / 100);
            if (d[18]) fprintf(stderr, "   k_fused_deep: the longest span took %.1f us (%llu records in range)\n", (double)(d[18] >> 24) / 100.0, (d[18] & 0xffffffull) << 4);
        }
        if ((ctx->h_scan[2] & 4) && !f.ph.rsegs) {
            // a flagged column and no tables yet: build them (this context keeps doing so from now on) and run the scan again
            ctx->legacy_wanted = true;
            if ((rc = ensure_legacy_tables(ctx))) return rc;
            f.ph.rsegs = (const DevSeg *)ctx->d_rsegs.p; f.ph.rseg_first = (const uint32_t *)ctx->d_rseg_first.p; f.ph.cigar = (const uint32_t *)ctx->d_cigar.p;
            --attempt;
            continue;
        }
        if ((ctx->h_scan[6] & 2) && !ctx->win32 && !raw_rerun) {
            if (base_row != 0) return fail(ctx, C3R_EOVERFLOW, "a position is covered by more than 32,767 reads and the batch already holds 16-bit windows: scan this region first, or alone");
            ctx->win32 = true;                 // (16-bit windows cannot hold this scan's counts: once more, with int32 windows)
            return scan_fused(ctx, n_regions, ctg_starts, ctg_ends, n_candidates, false);
        }
        if (ctx->h_scan[6] & 1) return fail(ctx, C3R_EOVERFLOW, "internal: indel-event scratch too small (%zu records) — nothing was written past it", ev_cap);
        if (ctx->h_scan[2] & 8) return fail(ctx, C3R_EOVERFLOW, "internal: tokens found no slot (the gates' bound of a site's tokens was too small)");
        if (n_cand < 0 || n_tok < 0) return fail(ctx, C3R_EOVERFLOW, "too many candidates or tokens for one scan");
        if (raw_rerun) {
            if (ctx->h_scan[2] & 1) {
                // a shard ran over: which workgroup takes a span depends on dynamic tickets and stealing across queues, so the shards' loads
                // differ from run to run and a shard that fitted in the scan itself may not fit now — grow and repeat like the scan does
                // (d_raw / d_rawidx are private to the re-run)
                if (attempt >= 2) return fail(ctx, C3R_EOVERFLOW, "internal: the raw re-run's buffers are still too small after growing (%d candidates)", n_cand);
                want_c = std::max<int64_t>(want_c, need_c + need_c / 4 + 1024);
                continue;
            }
            if (n_cand != ctx->last_cand) return fail(ctx, C3R_EINVAL, "internal: the raw re-run found %d candidates, the scan %lld", n_cand, (long long)ctx->last_cand);
            return C3R_OK;
        }
        if (need_c <= want_c && need_t <= want_t && !(ctx->h_scan[2] & 1)) {
            if (nsh > 1) { ctx->hint_rows = (int32_t)std::min<int64_t>(need_c + need_c / 4 + 64, INT32_MAX / nsh); ctx->hint_toks = (int32_t)std::min<int64_t>(need_t + need_t / 4 + 2048, INT32_MAX / nsh); }
            break;
        }
        if (attempt >= 2) return fail(ctx, C3R_EOVERFLOW, "internal: output buffers still too small after growing (%d candidates, %d tokens)", n_cand, n_tok);
        want_c = std::max<int64_t>(want_c, need_c + need_c / 4 + 1024);
        want_t = std::max<int64_t>(want_t, need_t + need_t / 4 + 1024);
    }
    ctx->last_cand = n_cand; ctx->last_shards = nsh; ctx->last_shard_rows = (int32_t)want_c;
    if (n_candidates) *n_candidates = n_cand;
    ctx->tokens_ready = true;
    ctx->n_cand = base_cand + n_cand;
    ctx->n_tok += n_tok;
    // (one allocator: dense output, the next scan of the batch appends; several: the scan's whole row / token space is taken)
    ctx->n_rows = base_row + (nsh == 1 ? (int64_t)n_cand : want_c * nsh);
    // (token SLOTS: what the allocators handed out — a site reserves the gates' bound and n_tok counts what was written)
    ctx->n_tokspace = base_tok + (nsh == 1 ? (int64_t)(((const unsigned long long *)(ctx->h_scan + 16))[0] >> 32) : want_t * nsh);
    return C3R_OK;
}

int c3r_batch_begin(c3r_ctx *ctx) {
    if (!ctx) return C3R_EINVAL;
    ctx->batching = true; ctx->n_cand = 0; ctx->n_tok = 0; ctx->n_rows = 0; ctx->n_tokspace = 0; ctx->last_cand = 0; ctx->last_base = 0;
    return C3R_OK;
}

int c3r_batch_end(c3r_ctx *ctx) {
    if (!ctx) return C3R_EINVAL;
    ctx->batching = false;
    return C3R_OK;
}

int c3r_batch_count(c3r_ctx *ctx, int64_t *n_sites, int64_t *n_tokens) {
    if (!ctx) return C3R_EINVAL;
    if (n_sites) *n_sites = ctx->n_cand;
    if (n_tokens) *n_tokens = ctx->n_tok;
    return C3R_OK;
}

int c3r_get_tensors(c3r_ctx *ctx, int rescaled, int32_t *tensors, int64_t cap_sites) {
    if (!ctx || !tensors) return C3R_EINVAL;
    if (cap_sites < ctx->n_cand) return fail(ctx, C3R_EOVERFLOW, "need room for %lld sites", (long long)ctx->n_cand);
    if (ctx->n_cand == 0) return C3R_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int row_ints = C3R_WINDOW * ctx->prm.channels;
    const size_t bytes = (size_t)ctx->n_cand * row_ints * 4;
    const void *src = ctx->d_tensors.p;
    const int32_t *idx = (const int32_t *)ctx->d_winidx.p;       // the fused path stores windows in arrival order
    if (!rescaled) {
        // raw (un-rescaled) windows are rebuilt for the most recent scan only
        if (ctx->last_base != 0 || ctx->last_cand != ctx->n_cand)
            return fail(ctx, C3R_EINVAL, "raw tensors are only available for a single (non-batched) scan");
        if (ctx->last_fused) {               // the fused path keeps no columns: the scan runs again, un-rescaled windows only
            int rc = scan_fused(ctx, (int32_t)ctx->last_starts.size(), ctx->last_starts.data(), ctx->last_ends.data(), nullptr, true);
            if (rc) return rc;
            idx = (const int32_t *)ctx->d_rawidx.p;
        } else {
            if (!ctx->prm.splice_padding) {     // (splice padding: the scan already kept the raw windows, the columns have moved on)
                int rc = ensure(ctx, ctx->d_raw, bytes);
                if (rc) return rc;
                if ((rc = run_gather(ctx, 0, ctx->d_raw.p, false, false))) return rc;
            }
            idx = nullptr;
        }
        src = ctx->d_raw.p;
    }
    if (idx) {
        int rc = ensure(ctx, ctx->d_export, bytes);
        if (rc) return rc;
        // (the resident windows are int16 rows in arrival order: gathered into position order and widened; a raw re-run's are int32)
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)std::min<int64_t>((ctx->n_cand + 3) / 4, 8192)), dim3(256), 0, ctx->stream, src, idx, (int)ctx->n_cand, row_ints,
                           (int32_t *)ctx->d_export.p, (rescaled && !ctx->win32) ? 1 : 0);
        src = ctx->d_export.p;
    }
    HIPCHK(ctx, hipMemcpyAsync(tensors, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return C3R_OK;
}

int c3r_get_sites(c3r_ctx *ctx, c3r_site_t *sites, int64_t cap_sites) {
    if (!ctx || !sites) return C3R_EINVAL;
    if (cap_sites < ctx->n_cand) return fail(ctx, C3R_EOVERFLOW, "need room for %lld sites", (long long)ctx->n_cand);
    if (ctx->n_cand == 0) return C3R_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipMemcpyAsync(sites, ctx->d_sites_out.p, (size_t)ctx->n_cand * sizeof(c3r_site_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // on the device a site's tokens lie where its span's token slots were handed out; c3r_get_tokens returns them site after site
    uint32_t off = 0;
    for (int64_t i = 0; i < ctx->n_cand; ++i) { sites[i].tok_off = off; off += (uint32_t)sites[i].n_tok; }
    return C3R_OK;
}

int c3r_token_count(c3r_ctx *ctx, int64_t *n_tokens) {
    if (!ctx || !n_tokens) return C3R_EINVAL;
    *n_tokens = ctx->n_tok;
    return C3R_OK;
}

int c3r_get_tokens(c3r_ctx *ctx, c3r_token_t *tokens, int64_t cap_tokens) {
    if (!ctx || !tokens) return C3R_EINVAL;
    if (cap_tokens < ctx->n_tok) return fail(ctx, C3R_EOVERFLOW, "need room for %lld tokens", (long long)ctx->n_tok);
    if (ctx->n_tok == 0) return C3R_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // site order (see c3r_get_sites): offsets = exclusive sums of the sites' token counts, then one wavefront per site copies its run
    const int n = (int)ctx->n_cand;
    int rc;
    if ((rc = ensure(ctx, ctx->d_tokoff, (size_t)(n + 2) * 4)) || (rc = ensure(ctx, ctx->d_tokexp, (size_t)ctx->n_tok * sizeof(c3r_token_t))) || (rc = ensure(ctx, ctx->d_small, 64))) return rc;
    hipLaunchKernelGGL(k_site_ntok, dim3((unsigned)(n / 256 + 1)), dim3(256), 0, ctx->stream, (const c3r_site_t *)ctx->d_sites_out.p, n, (int32_t *)ctx->d_tokoff.p);
    if ((rc = device_excl_scan(ctx, (int32_t *)ctx->d_tokoff.p, n + 1, (int32_t *)((char *)ctx->d_small.p + 48)))) return rc;
    hipLaunchKernelGGL(k_export_tokens, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 8192)), dim3(256), 0, ctx->stream, (const c3r_site_t *)ctx->d_sites_out.p,
                       (const int32_t *)ctx->d_tokoff.p, n, (const c3r_token_t *)ctx->d_tok.p, (c3r_token_t *)ctx->d_tokexp.p);
    HIPCHK(ctx, hipMemcpyAsync(tokens, ctx->d_tokexp.p, (size_t)ctx->n_tok * sizeof(c3r_token_t), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return C3R_OK;
}

int c3r_get_pad_insertions(c3r_ctx *ctx, c3r_padins_t *out, int64_t cap, int64_t *n) {
    if (!ctx || !n) return C3R_EINVAL;
    const int64_t have = ctx->padins ? (int64_t)ctx->padins->size() : 0;
    *n = have;
    if (!out) return C3R_OK;
    if (cap < have) return fail(ctx, C3R_EOVERFLOW, "need room for %lld entries", (long long)have);
    if (have) memcpy(out, ctx->padins->data(), (size_t)have * sizeof(c3r_padins_t));
    return C3R_OK;
}

int c3r_get_columns(c3r_ctx *ctx, int64_t *region_start, int64_t *n_pos, int32_t *cols, int32_t *depth, uint8_t *flags, int64_t cap_pos) {
    if (!ctx) return C3R_EINVAL;
    if (region_start) *region_start = (int64_t)ctx->reg_beg0 + 1;
    // the first region of the most recent scan (its slots start at 0 and are contiguous)
    const int64_t npos0 = (int64_t)ctx->reg_end0 - ctx->reg_beg0;
    if (n_pos) *n_pos = npos0;
    if (!cols && !depth && !flags) return C3R_OK;
    if (cap_pos < npos0) return fail(ctx, C3R_EOVERFLOW, "need room for %lld positions", (long long)npos0);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->last_fused) {
        // the fused path never materialises columns: build those of the scan's first region now, through the column store
        if (ctx->last_starts.empty()) return fail(ctx, C3R_EINVAL, "no scan yet");
        int rc = af_table(ctx);
        if (!rc) rc = scan_column_store(ctx, 1, ctx->last_starts.data(), ctx->last_ends.data(), nullptr, true);
        ctx->geo_key.clear();                // (the device now holds the column store's geometry)
        if (rc) return rc;
    }
    if (ctx->last_scan_pruned) {
        // the scan skipped intron-only tiles that no candidate window can reach: compute their row flags now
        ScanArgs a = ctx->last_scan;
        a.tile_list = a.tile_list2; a.n_tile_list = a.n_tile_list2;
        if (ctx->prm.channels == C3R_CH) hipLaunchKernelGGL(k_scan_tiles<C3R_CH>, dim3(std::min((int)a.n_tiles, list_grid())), dim3(SCAN_THREADS), 0, ctx->stream, a);
        else hipLaunchKernelGGL(k_scan_tiles<C3R_CH_PHASED>, dim3(std::min((int)a.n_tiles, list_grid())), dim3(SCAN_THREADS), 0, ctx->stream, a);
        ctx->last_scan_pruned = false;
        HIPCHK(ctx, hipGetLastError());
    }
    const size_t n = (size_t)npos0;
    if (cols) HIPCHK(ctx, hipMemcpyAsync(cols, ctx->d_cols.p, n * ctx->prm.channels * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (depth) HIPCHK(ctx, hipMemcpyAsync(depth, ctx->d_depth.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (flags) HIPCHK(ctx, hipMemcpyAsync(flags, ctx->d_flags.p, n, hipMemcpyDeviceToHost, ctx->stream));
    const size_t n_tiles = (n + TILE - 1) / TILE;
    std::vector<uint8_t> tc(n_tiles);
    HIPCHK(ctx, hipMemcpyAsync(tc.data(), ctx->d_tile_cols.p, n_tiles, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // tiles that hold no aligned base are not materialised on the device: their columns are all-zero by definition
    for (size_t t = 0; t < n_tiles; ++t) {
        if (tc[t]) continue;
        const size_t p0 = t * TILE, p1 = std::min(n, p0 + TILE);
        if (cols) memset(cols + p0 * ctx->prm.channels, 0, (p1 - p0) * ctx->prm.channels * 4);
        if (depth) for (size_t q = p0; q < p1; ++q) depth[q] = 0;        // (rows of such a tile show ref-skips only: depth 0)
    }
    return C3R_OK;
}

// ------------------------------------------------------------------------------------------------
int64_t c3r_weight_count(int channels) { return net_weight_count(channels); }

// Precision "auto" (3): the fp8-corrected path (precision 2) is ~10x less exact than the split-f16 path and its error grows fast with
// the weights' norm, so it is used only where it has been MEASURED: both paths run on 2048 pileup-shaped
// windows drawn from a fixed seed and the probabilities must agree to C3R_MX_GUARD, a 2.5x margin under the 1e-4 tolerance.
static constexpr double C3R_MX_GUARD = 4e-5;
// The split-f16 path itself (precision 1, the default) is fp32-equivalent only while its operands fit f16 with the layer's scale and the
// products do not cancel pathologically; nothing in clair3_rna/model.py:126-172 bounds a trained model's weights.  So it is MEASURED too:
// the same calibration windows through the fp32 MFMA path (precision 0) and through split-f16 must agree to C3R_F16_GUARD, else the context
// runs the fp32 path and says so (stderr, c3r_get_precision, c3r_get_precision_guard).  The guard is the parity tolerance itself, 1e-4, and
// not a fraction of it, because of what the two paths were measured to do as the weights' gain grows (profiles/r5/f16_guard_vs_weight_norm.txt:
// N(0, 0.05) test weights times 1 / 2 / 3 / 4 / 6): they drift apart by 2.9e-6 / 2.3e-5 / 8.4e-5 / 1.4e-4 / 2.7e-4 — and each of them drifts
// from the fp32 oracle by the SAME 2.4e-6 / 1.8e-5 / 7.9e-5 / 8.8e-5 / 3.3e-4: at high gain the network is ill-conditioned in fp32 itself
// (summation order, hardware exp2 / rcp), and the fp32 MFMA path is no closer to the oracle than split-f16 is.  A tighter guard would send
// such weights to a path three times slower for no gain in parity; what the guard is for — an operand that overflows f16, a scale that
// flushes lo halves, a NaN — shows up as a disagreement of 1e-3 .. 1.
static constexpr double C3R_F16_GUARD = 1e-4;
static void calibration_windows(int C, int n, std::vector<int32_t> &X) {
    X.assign((size_t)n * C3R_WINDOW * C, 0);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&](uint32_t m) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)((st >> 11) % m); };
    static const int depths[6] = {6, 12, 20, 40, 90, 216};
    for (int s_ = 0; s_ < n; ++s_) {
        const int depth = depths[rnd(6)];
        for (int t = 0; t < C3R_WINDOW; ++t) {
            int32_t *col = &X[((size_t)s_ * C3R_WINDOW + t) * C];
            const int k = (int)rnd(4);
            int fwd = 0;
            for (int i = 0; i < depth; ++i) fwd += (int)rnd(2);
            col[k] = -fwd; col[9 + k] = -(depth - fwd);                     // the reference-base channels carry minus the strand totals
            for (int r = (int)rnd(3); r > 0; --r) col[rnd((uint32_t)C)] += 1 + (int)rnd((uint32_t)std::max(1, depth / 3));
        }
    }
}
// max |dP| between two precisions on the calibration windows (a NaN on either side counts as 1.0)
static int calibrate_pair(c3r_ctx *ctx, int mode_a, int mode_b, double *err_out) {
    const int C = ctx->net.channels, n = 2048;
    std::vector<int32_t> X;
    calibration_windows(C, n, X);
    int32_t *d_x = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d_x, X.size() * 4));
    std::vector<float> p1((size_t)n * C3R_NPROB), p2(p1.size());
    const int keep = ctx->net.precision;
    auto run = [&](int mode, std::vector<float> &out) -> int {
        ctx->net.precision = mode;
        std::string e;
        int rc = net_forward(ctx->net, d_x, nullptr, n, ctx->stream, [](const char *, int) {}, e);
        if (rc) return fail(ctx, rc, "%s", e.c_str());
        HIPCHK(ctx, hipMemcpyAsync(out.data(), ctx->net.d_probs, out.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return C3R_OK;
    };
    int rc = C3R_OK;
    if (hipMemcpyAsync(d_x, X.data(), X.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = C3R_EHIP;
    if (!rc) rc = run(mode_a, p1);
    if (!rc) rc = run(mode_b, p2);
    (void)hipFree(d_x);
    ctx->net.precision = keep;
    if (rc) return rc;
    double worst = 0.0;
    for (size_t i = 0; i < p1.size(); ++i) {
        const double d = std::fabs((double)p1[i] - (double)p2[i]);
        worst = (d == d) ? std::max(worst, d) : 1.0;                          // (a NaN disqualifies)
    }
    *err_out = worst;
    return C3R_OK;
}
static int apply_precision(c3r_ctx *ctx) {
    const int req = ctx->precision_req;
    ctx->net.precision = req == 3 ? 1 : req;
    ctx->f16_fell_back = false;
    if (!ctx->net.loaded || req == 0) return C3R_OK;                          // decided when the weights arrive
    const bool rts = ctx->net.wlog2[0] != 12 || ctx->net.wlog2[1] != 12 || ctx->net.wlog2[2] != 12;
    int rc;
    // ---- the guard of the split-f16 arithmetic: once per set of weights
    // (C3R_NO_F16_GUARD=1: development aid for timing probes whose kernels compute garbage on purpose — tools/y1_probe.sh)
    static const bool no_guard = [] { const char *e = getenv("C3R_NO_F16_GUARD"); return e && *e == '1'; }();
    if (no_guard) ctx->f16_calib_err = 0.0;
    if (ctx->f16_calib_err < 0.0) {
        double err = 1.0;
        if ((rc = calibrate_pair(ctx, 0, 1, &err))) { ctx->net.precision = 0; return rc; }
        ctx->f16_calib_err = err;
    }
    if (ctx->f16_calib_err > C3R_F16_GUARD) {
        ctx->net.precision = 0; ctx->f16_fell_back = true;
        fprintf(stderr, "[c3r] warning: with these weights the split-f16 network differs from the fp32 one by %.3g on the calibration windows (guard %.0e): "
                        "running the fp32 MFMA path (c3r_set_precision 0), about a third of the speed\n", ctx->f16_calib_err, C3R_F16_GUARD);
        return C3R_OK;
    }
    if (req == 1) return C3R_OK;
    if (rts) {                                                                // (the fp8 fragments are built for the 2^12 scale)
        if (req == 2) { ctx->net.precision = 1; return fail(ctx, C3R_EINVAL, "precision 2 (f16 + fp8 corrections) needs weights that fit the 2^12 split-f16 scale: max |w| < 8 per layer"); }
        ctx->net.precision = 1;
        return C3R_OK;
    }
    if (req == 2) return C3R_OK;
    double err = -1.0;
    if ((rc = calibrate_pair(ctx, 1, 2, &err))) { ctx->net.precision = 1; return rc; }
    ctx->mx_calib_err = err;
    ctx->net.precision = err <= C3R_MX_GUARD ? 2 : 1;
    return C3R_OK;
}

int c3r_load_weights(c3r_ctx *ctx, const float *blob, int64_t n_floats, int channels) {
    if (!ctx || !blob) return C3R_EINVAL;
    if (channels != C3R_CH && channels != C3R_CH_PHASED) return fail(ctx, C3R_EINVAL, "channels must be 18 or 30");
    if (n_floats != net_weight_count(channels))
        return fail(ctx, C3R_EINVAL, "weight blob has %lld floats, expected %lld", (long long)n_floats, (long long)net_weight_count(channels));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string e;
    int rc = net_load(ctx->net, blob, channels, ctx->stream, e);
    if (rc) return fail(ctx, rc, "%s", e.c_str());
    ctx->mx_calib_err = -1.0; ctx->f16_calib_err = -1.0;
    return apply_precision(ctx);
}

int c3r_set_precision(c3r_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 3) return C3R_EINVAL;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->precision_req = mode;
    return apply_precision(ctx);
}

int c3r_get_precision(c3r_ctx *ctx, int *mode_in_use, double *calibration_err) {
    if (!ctx) return C3R_EINVAL;
    if (mode_in_use) *mode_in_use = ctx->net.precision;
    if (calibration_err) *calibration_err = ctx->mx_calib_err;
    return C3R_OK;
}

int c3r_get_precision_guard(c3r_ctx *ctx, double *f16_err, int32_t *scale_log2, int *fell_back) {
    if (!ctx) return C3R_EINVAL;
    if (f16_err) *f16_err = ctx->f16_calib_err;
    if (scale_log2) for (int l = 0; l < 3; ++l) scale_log2[l] = ctx->net.wlog2[l];
    if (fell_back) *fell_back = ctx->f16_fell_back ? 1 : 0;
    return C3R_OK;
}

int c3r_reserve(c3r_ctx *ctx, int64_t n_sites) {
    if (!ctx || n_sites < 0) return C3R_EINVAL;
    if (!ctx->net.loaded) return fail(ctx, C3R_EINVAL, "c3r_load_weights must be called before c3r_reserve");
    if (n_sites == 0) return C3R_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string e;
    int rc = net_reserve(ctx->net, n_sites, ctx->stream, e);
    if (rc) return fail(ctx, rc, "%s", e.c_str());
    return C3R_OK;
}

int c3r_infer(c3r_ctx *ctx, const int32_t *tensors, int64_t n, float *probs) {
    if (!ctx || n < 0) return C3R_EINVAL;
    if (!ctx->net.loaded) return fail(ctx, C3R_EINVAL, "c3r_load_weights must be called before c3r_infer");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int C = ctx->net.channels;
    const void *d_x = nullptr;
    const int32_t *d_rows = nullptr;
    const bool x16 = tensors == nullptr && !ctx->win32;  // the resident windows are int16 (unless the read set needed int32), a caller's batch int32
    if (tensors == nullptr) {
        if (C != ctx->prm.channels) return fail(ctx, C3R_EINVAL, "weights are for %d channels, scan produced %d", C, ctx->prm.channels);
        if (n != ctx->n_cand) return fail(ctx, C3R_EINVAL, "n=%lld but %lld candidates are resident", (long long)n, (long long)ctx->n_cand);
        d_x = ctx->d_tensors.p;
        d_rows = (const int32_t *)ctx->d_winidx.p;       // windows lie in arrival order (k_fused_tiles)
    } else if (n > 0) {
        int rc = ensure(ctx, ctx->d_raw, (size_t)n * C3R_WINDOW * C * 4);
        if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_raw.p, tensors, (size_t)n * C3R_WINDOW * C * 4, hipMemcpyHostToDevice, ctx->stream));
        if (!probs) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // the caller's buffer must be free to go when we return
        d_x = ctx->d_raw.p;
    }
    if (n == 0) return C3R_OK;
    {
        // layer 1 addresses a window by a 32-bit element index (k_lstm1_rs): the rows it can be asked for — the resident windows' row space, or the
        // caller's batch — must stay below 2^32 / (33 * channels) = 7.2 M rows at 18 channels (a GRCh38 chromosome holds one to two million candidates)
        const int64_t rows = tensors == nullptr ? std::max<int64_t>(ctx->n_rows, n) : n;
        if ((uint64_t)rows * (uint64_t)(C3R_WINDOW * C) >= (1ull << 32))
            return fail(ctx, C3R_EINVAL, "%lld window rows of %d channels exceed the network's 32-bit row addressing: call c3r_infer on smaller batches", (long long)rows, C);
    }
    std::string e;
    if (ctx->net.d_tmo) HIPCHK(ctx, hipMemsetAsync(ctx->net.d_tmo, 0, 4, ctx->stream));     // (a time-out fails the batch it happened in, not the ones after it)
    // the network's kernels go to the context's default-priority stream, behind everything queued on `stream` so far (the tensors, an
    // uploaded batch); whatever is queued on `stream` afterwards (probabilities, rows, the next batch's tensors) comes behind them
    hipStream_t nst = ctx->net_stream ? ctx->net_stream : ctx->stream;
    if (ctx->net_stream) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_x, ctx->stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->net_stream, ctx->ev_x, 0));
    }
    auto prof = [&](const char *name, int phase) {
        if (!ctx->profiling) return;
        if (phase == 0) (void)hipEventRecord(ctx->ev0, nst);
        else {
            (void)hipEventRecord(ctx->ev1, nst); (void)hipEventSynchronize(ctx->ev1);
            float ms = 0; (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
            KStat &k = ctx->kstats[name]; k.ms += ms; k.n += 1;
        }
    };
    int rc = net_forward(ctx->net, d_x, d_rows, n, nst, prof, e, x16);
    if (ctx->net_stream) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_net, ctx->net_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_net, 0));
    }
    if (rc) return fail(ctx, rc, "%s", e.c_str());
    if (probs) {
        int32_t *st = nullptr;
        if ((rc = queue_lstm_status(ctx, &st))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(probs, ctx->net.d_probs, (size_t)n * C3R_NPROB * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if ((rc = check_lstm_status(ctx, st))) return rc;
    }
    HIPCHK(ctx, hipGetLastError());
    return C3R_OK;
}

int c3r_get_probs(c3r_ctx *ctx, float *probs, int64_t n) {
    if (!ctx || !probs || n < 0) return C3R_EINVAL;
    if (n == 0) return C3R_OK;
    if (!ctx->net.d_probs || n > ctx->net.cap_probs) return fail(ctx, C3R_EINVAL, "no probabilities resident for %lld sites", (long long)n);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int32_t *st = nullptr;
    int rc = queue_lstm_status(ctx, &st);
    if (rc) return rc;
    HIPCHK(ctx, hipMemcpyAsync(probs, ctx->net.d_probs, (size_t)n * C3R_NPROB * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return check_lstm_status(ctx, st);
}

// ------------------------------------------------------------------------------------------------ A8 on the host
int c3r_decode_text(const char *ctg, int64_t n, const int32_t *pos, const char *ref33s, int ref33_stride, const char *const *alt_infos,
                    const float *probs, int qual, int show_ref, char *out, int64_t cap, int64_t *out_len) {
    if (!ctg || n < 0 || (n && (!pos || !ref33s || !alt_infos || !probs)) || !out_len) return C3R_EINVAL;
    std::string rows;
    for (int64_t i = 0; i < n; ++i) {
        int depth; AltDict alt;
        parse_alt_info(alt_infos[i], depth, alt);
        vcf_row(ctg, pos[i], ref33s + (size_t)i * ref33_stride, depth, alt, probs + (size_t)i * C3R_NPROB, qual, show_ref != 0, rows);
    }
    *out_len = (int64_t)rows.size();
    if (!out || cap < (int64_t)rows.size() + 1) return C3R_EOVERFLOW;
    memcpy(out, rows.c_str(), rows.size() + 1);
    return C3R_OK;
}

// ---- row snapshots: everything the decoder needs of one batch, on the host, detached from the context
}  // extern "C"
struct c3r_rows {
    c3r_ctx *ctx = nullptr;
    // one staging block: sites | token bytes | probabilities | indel-record offsets | read headers | packed bases
    void *stage = nullptr; size_t stage_cap = 0;
    int64_t n = 0, n_tok = 0;
    c3r_site_t *sites = nullptr; uint8_t *tokb = nullptr; float *probs = nullptr; uint32_t *rec_off = nullptr;
    std::shared_ptr<HostReads> hr;                        // the contig's read headers and packed bases (inserted bases of the alt alleles)
    std::shared_ptr<PadInsTab> padins;                    // mpileup_compat = 1: its insertions with pads (null / empty otherwise)
    const DevRead *reads = nullptr; const uint8_t *seq = nullptr;
    const c3r_read_t *creads = nullptr;                   // c3r_rows_begin_ex: the caller's own records and bases are read in place (no copy back)
    TokRec *recs = nullptr; int64_t n_recs = 0;           // the tokens that carry an indel (k_pack_tokens), own allocation
    int ref_slot = -1; const char *ref = nullptr; size_t ref_len = 0; int64_t ref_start1 = 1;
    std::string rows; int64_t rows_count = 0;
};
extern "C" {

int c3r_rows_begin(c3r_ctx *ctx, c3r_rows **out) { return c3r_rows_begin_ex(ctx, 0, nullptr, nullptr, out); }

int c3r_rows_begin_ex(c3r_ctx *ctx, int drop_ref_calls, const c3r_read_t *host_reads, const uint8_t *host_seq, c3r_rows **out) {
    if (!ctx || !out || ((host_reads == nullptr) != (host_seq == nullptr))) return C3R_EINVAL;
    *out = nullptr;
    int64_t n = ctx->n_cand;
    if (n > 0 && (!ctx->net.d_probs || n > ctx->net.cap_probs)) return fail(ctx, C3R_EINVAL, "c3r_infer must run before c3r_call_rows");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    c3r_rows *r = new c3r_rows();
    r->ctx = ctx; r->n = n; r->n_tok = ctx->n_tok;
    if (n == 0) { *out = r; return C3R_OK; }
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    int64_t n_tok = ctx->n_tokspace;                 // (token bytes are addressed through the sites' tok_off: the whole slot space)
    // ---- drop_ref_calls: only the sites that can print a row without --show_ref leave the device (k_row_keep / k_pack_rows)
    const int64_t n_all = n;
    const c3r_site_t *d_sites_src = (const c3r_site_t *)ctx->d_sites_out.p;
    const float *d_probs_src = ctx->net.d_probs;
    bool packed = false;
    if (drop_ref_calls) {
        int rc0;
        if ((rc0 = ensure(ctx, ctx->d_keep, (size_t)(n_all + 2) * 4)) || (rc0 = ensure(ctx, ctx->d_small, 64))) { delete r; return rc0; }
        hipLaunchKernelGGL(k_row_keep, dim3((unsigned)(n_all / 256 + 1)), dim3(256), 0, ctx->stream, d_sites_src, d_probs_src, (int)n_all, (int32_t *)ctx->d_keep.p);
        if ((rc0 = device_excl_scan(ctx, (int32_t *)ctx->d_keep.p, (int)n_all + 1, (int32_t *)((char *)ctx->d_small.p + 52)))) { delete r; return rc0; }
        int32_t n_keep = 0;
        // (the keep decision reads the probabilities: the layer-2 time-out word is checked with the same wait, before the count is acted on —
        // a timed-out batch whose garbage looks like RefCalls must fail, not come back as an empty snapshot)
        int32_t *lstm_st0 = nullptr;
        if ((rc0 = queue_lstm_status(ctx, &lstm_st0))) { delete r; return rc0; }
        if (hipMemcpyAsync(&n_keep, (char *)ctx->d_small.p + 52, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            delete r; return fail(ctx, C3R_EHIP, "reading the number of kept sites failed");
        }
        if ((rc0 = check_lstm_status(ctx, lstm_st0))) { delete r; return rc0; }
        n = n_keep; r->n = n;
        if (n == 0) { r->n_tok = 0; *out = r; return C3R_OK; }
        if ((rc0 = ensure(ctx, ctx->d_sites_c, (size_t)n * sizeof(c3r_site_t))) || (rc0 = ensure(ctx, ctx->d_probs_c, (size_t)n * C3R_NPROB * sizeof(float)))) { delete r; return rc0; }
        packed = true;
    }
    const size_t b_sites = up((size_t)n * sizeof(c3r_site_t)), b_tokb = up((size_t)std::max<int64_t>(n_tok, 1)), b_probs = up((size_t)n * C3R_NPROB * sizeof(float)),
                 b_off = up((size_t)n * 4);
    const size_t need = b_sites + b_tokb + b_probs + b_off;
    {   // a block from the pool of released snapshots
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        for (size_t k = 0; k < ctx->stage_pool.size(); ++k)
            if (ctx->stage_pool[k].second >= need) { r->stage = ctx->stage_pool[k].first; r->stage_cap = ctx->stage_pool[k].second; ctx->stage_pool.erase(ctx->stage_pool.begin() + (long)k); break; }
    }
    if (!r->stage) {
        const size_t cap = need * 5 / 4 + 4096;
        if (stage_pinned()) { if (hipHostMalloc(&r->stage, cap, hipHostMallocDefault) != hipSuccess) r->stage = nullptr; }
        else r->stage = huge_alloc(cap);
        if (!r->stage) { delete r; return fail(ctx, C3R_ENOMEM, "staging block of %zu bytes: allocation failed", cap); }
        r->stage_cap = cap;
    }
    char *sp = (char *)r->stage;
    r->sites = (c3r_site_t *)sp; sp += b_sites;
    r->tokb = (uint8_t *)sp; sp += b_tokb;
    r->probs = (float *)sp; sp += b_probs;
    r->rec_off = (uint32_t *)sp; sp += b_off;
    const bool fresh_reads = !host_reads && !ctx->host_cache;
    if (host_reads) { r->creads = host_reads; r->seq = host_seq; }
    else if (fresh_reads) {
        ctx->host_cache = std::make_shared<HostReads>();
        ctx->host_cache->reads.resize((size_t)std::max(ctx->n_reads, 1));
        ctx->host_cache->seq.resize((size_t)ctx->n_seq_bytes + 16);
    }
    if (!host_reads) { r->hr = ctx->host_cache; r->reads = r->hr->reads.data(); r->seq = r->hr->seq.data(); }
    r->padins = ctx->padins;
    auto bail = [&](int rc) { if (fresh_reads) ctx->host_cache.reset(); c3r_rows_free(r); return rc; };      // (a half-copied cache is no cache)
    const bool timing = getenv("C3R_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    if (timing) (void)hipStreamSynchronize(ctx->stream);              // (separates the wait for the network from the copies in the report)
    const auto t1 = now();
    // tokens leave the device packed (k_pack_tokens: a byte per token, 12-byte records for the tokens with an indel)
    int rc = C3R_OK;
    if ((rc = ensure(ctx, ctx->d_tokb, (size_t)std::max<int64_t>(n_tok, 1))) || (rc = ensure(ctx, ctx->d_tokrec, (size_t)std::max<int64_t>(n_tok, 1) * sizeof(TokRec))) ||
        (rc = ensure(ctx, ctx->d_recoff, (size_t)n * 4 + 32)))
        return bail(rc);
    if (!ctx->h_pack && hipHostMalloc((void **)&ctx->h_pack, 64, hipHostMallocDefault) != hipSuccess) return bail(fail(ctx, C3R_ENOMEM, "hipHostMalloc(64) failed"));
    unsigned long long *d_counter = (unsigned long long *)((char *)ctx->d_recoff.p + (((size_t)n * 4 + 7) & ~(size_t)7));
    if (hipMemsetAsync(d_counter, 0, 16, ctx->stream) != hipSuccess) return bail(fail(ctx, C3R_EHIP, "hipMemsetAsync failed"));
    if (packed) {
        Launch L(ctx, "k_pack_tokens");
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)((n_all + 3) / 4)), dim3(256), 0, ctx->stream, d_sites_src, (const c3r_token_t *)ctx->d_tok.p, d_probs_src,
                           (const int32_t *)ctx->d_keep.p, n_all, (c3r_site_t *)ctx->d_sites_c.p, (float *)ctx->d_probs_c.p, (uint8_t *)ctx->d_tokb.p, (TokRec *)ctx->d_tokrec.p,
                           (uint32_t *)ctx->d_recoff.p, d_counter);
        d_sites_src = (const c3r_site_t *)ctx->d_sites_c.p; d_probs_src = (const float *)ctx->d_probs_c.p;
    } else {
        Launch L(ctx, "k_pack_tokens");
        hipLaunchKernelGGL(k_pack_tokens, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, (const c3r_site_t *)ctx->d_sites_out.p, (const c3r_token_t *)ctx->d_tok.p, n,
                           (uint8_t *)ctx->d_tokb.p, (TokRec *)ctx->d_tokrec.p, (uint32_t *)ctx->d_recoff.p, d_counter);
    }
    // (the snapshot's block is ordinary memory: the large arrays come down through the context's page-locked buffers, big_d2h)
    auto d2h = [&](void *dst, const void *src, size_t bytes) {
        if (bytes >= PIN_MIN && pin_ring_on()) return big_d2h(ctx, dst, src, bytes) == C3R_OK;
        return bytes == 0 || hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    };
    // (only the packed snapshot sizes its token copy from the counters: the plain one keeps the single wait at the end)
    if (!d2h(ctx->h_pack, d_counter, 16) || (packed && hipStreamSynchronize(ctx->stream) != hipSuccess)) return bail(fail(ctx, C3R_EHIP, "copying the token counters to the host failed"));
    if (packed) n_tok = (int64_t)((unsigned long long *)ctx->h_pack)[1];          // (the kept sites' token bytes lie back to back)
    if (!d2h(r->sites, d_sites_src, (size_t)n * sizeof(c3r_site_t)) || !d2h(r->tokb, ctx->d_tokb.p, (size_t)n_tok) ||
        !d2h(r->rec_off, ctx->d_recoff.p, (size_t)n * 4) ||
        (fresh_reads && (!d2h(r->hr->reads.data(), ctx->d_reads.p, (size_t)ctx->n_reads * sizeof(DevRead)) || !d2h(r->hr->seq.data(), ctx->d_seq.p, (size_t)ctx->n_seq_bytes + 16))))
        return bail(fail(ctx, C3R_EHIP, "copying sites / tokens / reads to the host failed"));
    int32_t *lstm_st = nullptr;
    rc = queue_lstm_status(ctx, &lstm_st);
    if (rc) return bail(rc);
    if (!d2h(r->probs, d_probs_src, (size_t)n * C3R_NPROB * sizeof(float)) || hipStreamSynchronize(ctx->stream) != hipSuccess)
        return bail(fail(ctx, C3R_EHIP, "copying probabilities to the host failed"));
    if ((rc = check_lstm_status(ctx, lstm_st))) return bail(rc);
    const auto t2 = now();
    r->n_recs = (int64_t)*(unsigned long long *)ctx->h_pack;
    if (r->n_recs > 0) {
        r->recs = (TokRec *)huge_alloc((size_t)r->n_recs * sizeof(TokRec));
        if (!r->recs) return bail(fail(ctx, C3R_ENOMEM, "indel records of %lld tokens: allocation failed", (long long)r->n_recs));
        if (!d2h(r->recs, ctx->d_tokrec.p, (size_t)r->n_recs * sizeof(TokRec)) || hipStreamSynchronize(ctx->stream) != hipSuccess)
            return bail(fail(ctx, C3R_EHIP, "copying indel records to the host failed"));
    }
    if (timing)
        fprintf(stderr, "[rows_begin %p] %lld sites, %lld tokens (%lld with an indel): wait %.1f ms, pack + copies (%.0f MB) %.1f ms, indel records (%.0f MB) %.1f ms\n", (void *)ctx,
                (long long)n, (long long)n_tok, (long long)r->n_recs, ms(t0, t1), need / 1e6, ms(t1, t2), r->n_recs * sizeof(TokRec) / 1e6, ms(t2, now()));
    r->ref_slot = ctx->ref_cur; r->ref = ctx->h_ref; r->ref_len = ctx->ref_len; r->ref_start1 = ctx->ref_start1;
    if (r->ref_slot >= 0) ctx->refbuf[r->ref_slot].users.fetch_add(1);
    *out = r;
    return C3R_OK;
}

int c3r_rows_decode(c3r_rows *r, const char *ctg, int qual, int show_ref, int64_t *out_len, int64_t *n_rows) {
    if (!r || !ctg || !out_len) return C3R_EINVAL;
    r->rows.clear(); r->rows_count = 0;
    *out_len = 0; if (n_rows) *n_rows = 0;
    const int64_t n = r->n;
    if (n == 0) return C3R_OK;
    const c3r_site_t *sites = r->sites; const uint8_t *tokb = r->tokb; const TokRec *recs = r->recs; const uint32_t *rec_off = r->rec_off; const float *probs = r->probs;
    const uint8_t *seq = r->seq;
    const DevRead *reads = r->reads;
    const RefView refv{r->ref, r->ref_len};
    const int64_t ref_start1 = r->ref_start1;
    const c3r_read_t *creads = r->creads;
    auto get_read = [&](uint32_t k) { return creads ? ReadView{seq, creads[k].seq_off, creads[k].l_seq} : ReadView{seq, reads[k].seq_off, reads[k].l_seq}; };
    const PadView pads{r->padins && !r->padins->empty() ? r->padins->data() : nullptr, r->padins ? r->padins->size() : 0};
    // host threads: C3R_THREADS, else up to 32 (one process per GPU shares the node's cores with its peers)
    unsigned nt = std::max(1u, std::min(32u, usable_cpus()));
    if (const char *e = getenv("C3R_THREADS")) nt = (unsigned)std::max(1, atoi(e));
    if ((int64_t)nt * 64 > n) nt = 1;
    std::vector<std::string> part(nt);
    std::vector<int64_t> cnt(nt, 0);
    auto work = [&](unsigned t) {
        const int64_t a = n * t / nt, b = n * (t + 1) / nt;
        AltDict alt;
        for (int64_t i = a; i < b; ++i) {
            int depth_tok;
            const uint8_t *tb = tokb + sites[(size_t)i].tok_off;
            const TokRec *rc_ = recs ? recs + rec_off[(size_t)i] : nullptr;      // this site's indel records, in token order
            alt_from_stream(sites[(size_t)i].n_tok, [tb, rc_](int k) mutable {
                const uint8_t by = tb[k];
                if (by & 0x80) { const TokRec &q = *rc_++; return TokView{by & 31, q.indel, q.read_idx, q.qpos, q.del_after, (by & 0x20) != 0}; }
                return TokView{by & 31, 0, 0u, 0u, 0u, (by & 0x20) != 0};
            }, get_read, refv, ref_start1, sites[(size_t)i].pos, alt, depth_tok, pads, sites[(size_t)i].depth);
            if (vcf_row(ctg, sites[(size_t)i].pos, sites[(size_t)i].ref33, sites[(size_t)i].depth, alt, probs + (size_t)i * C3R_NPROB, qual,
                        show_ref != 0, part[t]))
                cnt[t]++;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    size_t total = 0;
    for (unsigned t = 0; t < nt; ++t) total += part[t].size();
    r->rows.reserve(total + 1);
    for (unsigned t = 0; t < nt; ++t) { r->rows += part[t]; r->rows_count += cnt[t]; }
    *out_len = (int64_t)r->rows.size();
    if (n_rows) *n_rows = r->rows_count;
    return C3R_OK;
}

int c3r_rows_get(c3r_rows *r, char *out, int64_t cap) {
    if (!r || !out) return C3R_EINVAL;
    if (cap < (int64_t)r->rows.size() + 1) return C3R_EOVERFLOW;
    memcpy(out, r->rows.c_str(), r->rows.size() + 1);
    return C3R_OK;
}

// give back what only the decode needed (reference buffer, staging block, read copies); the rows text stays
static void rows_release_inputs(c3r_rows *r) {
    c3r_ctx *ctx = r->ctx;
    if (r->ref_slot >= 0) { ctx->refbuf[r->ref_slot].users.fetch_sub(1); r->ref_slot = -1; r->ref = nullptr; r->ref_len = 0; }
    if (r->stage) {
        std::lock_guard<std::mutex> g(ctx->pool_mu);
        if (ctx->stage_pool.size() < 3) ctx->stage_pool.push_back({r->stage, r->stage_cap});
        else stage_free(r->stage);
        r->stage = nullptr; r->stage_cap = 0; r->sites = nullptr; r->tokb = nullptr; r->probs = nullptr; r->rec_off = nullptr; r->reads = nullptr; r->seq = nullptr; r->creads = nullptr;
    }
    if (r->recs) { free(r->recs); r->recs = nullptr; r->n_recs = 0; }
    r->n = 0;
}

void c3r_rows_free(c3r_rows *r) {
    if (!r) return;
    rows_release_inputs(r);
    delete r;
}

// the one-call form: snapshot + decode on this thread; the rows stay with the context until the next call
int c3r_call_rows(c3r_ctx *ctx, const char *ctg, int qual, int show_ref, int64_t *out_len, int64_t *n_rows) {
    if (!ctx || !ctg || !out_len) return C3R_EINVAL;
    if (ctx->rows_snap) { c3r_rows_free(ctx->rows_snap); ctx->rows_snap = nullptr; }
    *out_len = 0; if (n_rows) *n_rows = 0;
    const bool timing = getenv("C3R_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = c3r_rows_begin(ctx, &ctx->rows_snap);
    if (rc) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    rc = c3r_rows_decode(ctx->rows_snap, ctg, qual, show_ref, out_len, n_rows);
    rows_release_inputs(ctx->rows_snap);          // (only the text is kept for c3r_get_rows: no reference buffer stays pinned by it)
    if (timing) {
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[c3r_call_rows] snapshot (D2H of sites, tokens, probabilities, read bases) %.1f ms, decode %.1f ms\n", ms(t0, t1), ms(t1, std::chrono::steady_clock::now()));
    }
    return rc;
}

int c3r_get_rows(c3r_ctx *ctx, char *out, int64_t cap) {
    if (!ctx || !out) return C3R_EINVAL;
    static const std::string empty;
    const std::string &rows = ctx->rows_snap ? ctx->rows_snap->rows : empty;
    if (cap < (int64_t)rows.size() + 1) return fail(ctx, C3R_EOVERFLOW, "need %lld bytes", (long long)rows.size() + 1);
    memcpy(out, rows.c_str(), rows.size() + 1);
    return C3R_OK;
}

// ------------------------------------------------------------------------------------------------
int c3r_set_profiling(c3r_ctx *ctx, int enabled) {
    if (!ctx) return C3R_EINVAL;
    ctx->profiling = enabled != 0;
    return C3R_OK;
}
int c3r_reset_kernel_stats(c3r_ctx *ctx) {
    if (!ctx) return C3R_EINVAL;
    ctx->kstats.clear();
    return C3R_OK;
}
int c3r_get_kernel_stats(c3r_ctx *ctx, const char **names, double *total_ms, int64_t *launches, int cap, int *n) {
    if (!ctx || !n) return C3R_EINVAL;
    ctx->kstat_names.clear();
    for (auto &kv : ctx->kstats) ctx->kstat_names.push_back(kv.first);
    int i = 0;
    for (auto &kv : ctx->kstats) {
        if (i < cap) {
            if (names) names[i] = ctx->kstat_names[i].c_str();
            if (total_ms) total_ms[i] = kv.second.ms;
            if (launches) launches[i] = kv.second.n;
        }
        ++i;
    }
    *n = i;
    return C3R_OK;
}

}  // extern "C"
