// reads_kernels.hpp — gfx950 kernels that turn the caller's flat read records (include/c3r_types.h: c3r_read_t, BAM-encoded
// CIGARs, 4-bit bases) into the device tables the tile kernels walk.  Included by c3r_lib.hip only.
//
// What it replaces: the BAM side of `samtools mpileup <bam> -r ...` (src/create_tensor_pileup.py:436-451) — htslib resolves
// each read's CIGAR with a per-read cursor while it streams the file; here every read is prepared once, on the device, when
// the contig's records arrive (c3r_load_reads).  Round 2 did this on host threads (normalise, segment, std::sort, pageable
// uploads: 32 ms per chr20) outside the measured path.
//
//   k_reads_count  one lane per read: validate, count normalised ops / aligned segments / indel ops / op records, reference end
//   k_scan4_*      exclusive prefix sums of the four counts (int4 per read)
//   k_reads_pass   filters (flag, MAPQ) -> pass flags and sort keys of the read ends
//   k_cover_max    deepest coverage by passing reads (decides whether mpileup's -d cap can bite at all)
//   k_reads_write  normalised CIGARs, DevRead headers, aligned segments in read order, sort keys of the segments
//   k_seg_gather   segments into ext_start order (the permutation comes from rocPRIM's radix sort: a plain library sort)
//   k_prefmax_*    inclusive prefix maxima of the passing reads' / segments' ends (binary-searched per tile)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "pileup_kernels.hpp"

namespace c3r {

// What the host learns from the one synchronisation of c3r_load_reads.
struct LoadStats {
    unsigned long long err;   // ~0: none; else (read index << 8) | code — the smallest failing read wins
    int32_t max_end;          // largest reference end of any read (exclusive, 0-based)
    int32_t max_cover;        // most passing reads over one position
    // the totals of the int4 scan, 16-byte aligned (k_scan4_tops stores them as one int4)
    int32_t n_norm;           // normalised CIGAR ops
    int32_t n_segs;           // aligned segments
    int32_t n_indel;          // I + D ops after normalisation
    int32_t n_oprec;          // expanded op records (pileup_kernels.hpp, OpRec)
};
static_assert(sizeof(LoadStats) == 32 && offsetof(LoadStats, n_norm) == 16, "LoadStats layout");
enum { LD_OK = 0, LD_UNSORTED = 1, LD_CIGAR_RANGE, LD_SEQ_RANGE, LD_BAD_OP, LD_OP_LONG, LD_END_2G, LD_SEG_OPS };

__device__ __forceinline__ void load_fail(LoadStats *st, int read, int code) {
    atomicMin(&st->err, ((unsigned long long)(unsigned)read << 8) | (unsigned)code);
}

// CIGAR normalisation as a stream: drop H, zero-length ops and pads (a pad is kept — as a 1-long op that consumes nothing —
// exactly when the next real op is a D: htslib marks a deletion only when the D IMMEDIATELY follows the op that ends on the
// column, while insertions are found through pads), fold = / X into M, merge equal neighbours.  emit(op, len) receives every
// finalised op in order.  Returns LD_OK or the error code.
template <class Emit>
__device__ __forceinline__ int walk_norm(const uint32_t *cig, uint32_t n, Emit &&emit) {
    bool have = false;
    uint32_t cop = 0;
    unsigned long long clen = 0;
    // ops are fetched eight at a time: the loads of a batch are independent of each other, so a read's walk pays one memory round trip
    // per eight ops instead of one per op (the kernels end when the longest read — several hundred ops — is done)
    constexpr uint32_t NB = 8;
    uint32_t buf[NB];
    for (uint32_t k = 0; k < n; ++k) {
        if ((k & (NB - 1)) == 0) {
#pragma unroll
            for (uint32_t j = 0; j < NB; ++j) buf[j] = (k + j < n) ? cig[k + j] : 0u;
        }
        uint32_t c = buf[0];
#pragma unroll
        for (uint32_t j = 1; j < NB; ++j) c = ((k & (NB - 1)) == j) ? buf[j] : c;
        uint32_t op = c & 15u, len = c >> 4;
        if (op == C3R_CIG_EQ || op == C3R_CIG_X) op = C3R_CIG_M;
        if (len == 0 || op == C3R_CIG_H) continue;
        if (op == C3R_CIG_P) {
            uint32_t k2 = k + 1;
            while (k2 < n && ((cig[k2] >> 4) == 0 || (cig[k2] & 15u) == C3R_CIG_P || (cig[k2] & 15u) == C3R_CIG_H)) ++k2;
            if (k2 >= n || (cig[k2] & 15u) != C3R_CIG_D) continue;
            len = 1;
        }
        if (op > C3R_CIG_X) return LD_BAD_OP;
        if (have && cop == op) {
            clen += len;
            if (clen >= (1ull << 28)) return LD_OP_LONG;
        } else {
            if (have) emit(cop, (uint32_t)clen);
            cop = op; clen = len; have = true;
        }
    }
    if (have) emit(cop, (uint32_t)clen);
    return LD_OK;
}

// The aligned segments of a read = the runs of normalised ops between N ops, as a consumer of walk_norm's stream.
// seg(first op index, op count, pos, qstart, end_x, lead_n, lead_indel, records) is called for every segment that is kept:
// one that holds an M or D, or that starts with an I / D right after an N (that indel is attached to the last intron column).
struct SegWalk {
    long long x;          // reference cursor
    uint32_t y;           // query cursor
    uint32_t k;           // index of the next normalised op
    // the open segment
    uint32_t k0, q0, first_op;
    long long x0;
    bool open, useful, after_n;
    int nrec;             // op records of the open segment (M: one per OP_CHOP bases, I / D: one)
    int bad;              // LD_SEG_OPS when a segment holds more than 65535 ops
    __device__ __forceinline__ void begin(int32_t pos) { x = pos; y = 0; k = 0; open = false; useful = false; after_n = false; nrec = 0; bad = 0; first_op = 15; k0 = 0; q0 = 0; x0 = pos; }
    template <class Seg>
    __device__ __forceinline__ void close(Seg &&seg) {
        if (!open) return;
        const bool lead_indel = after_n && (first_op == C3R_CIG_I || first_op == C3R_CIG_D);
        if (useful || lead_indel) {
            if (k - k0 > 0xffffu) bad = LD_SEG_OPS;
            seg(k0, k - k0, x0, q0, x, after_n, lead_indel, nrec);
        }
        open = false;
    }
    template <class Seg>
    __device__ __forceinline__ void op(uint32_t o, uint32_t len, Seg &&seg) {
        if (o == C3R_CIG_N) {
            close(seg);
            x += len; after_n = true; ++k;
            return;
        }
        if (!open) { open = true; useful = false; k0 = k; q0 = y; x0 = x; first_op = o; nrec = 0; }
        if (o == C3R_CIG_M) { nrec += (int)((len + OP_CHOP - 1) / OP_CHOP); x += len; y += len; useful = true; }
        else if (o == C3R_CIG_D) { nrec += 1; x += len; useful = true; }
        else if (o == C3R_CIG_I) { nrec += 1; y += len; }
        else if (o == C3R_CIG_S) y += len;
        ++k;
    }
};

// counts: int4 {normalised ops, segments, indel ops, op records} per read (+ a zero entry at n_reads for the exclusive scan)
__global__ __launch_bounds__(256) void k_reads_count(const c3r_read_t *reads, int n_reads, const uint32_t *cigars, long long n_cigar_ops,
                                                     long long n_seq_bytes, int4 *cnt, int32_t *rend, LoadStats *st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int my_end = 0;
    if (i < n_reads) {
        const c3r_read_t r = reads[i];
        int4 c = make_int4(0, 0, 0, 0);
        int err = LD_OK;
        if (i > 0 && r.pos < reads[i - 1].pos) err = LD_UNSORTED;
        else if ((long long)r.cigar_off + r.n_cigar > n_cigar_ops) err = LD_CIGAR_RANGE;
        else if ((long long)r.seq_off + (r.l_seq + 1) / 2 > n_seq_bytes) err = LD_SEQ_RANGE;
        long long rlen = 0;
        if (!err) {
            SegWalk w;
            w.begin(r.pos);
            auto seg = [&](uint32_t, uint32_t, long long, uint32_t, long long, bool, bool, int nrec) { c.y += 1; c.w += nrec; };
            err = walk_norm(cigars + r.cigar_off, r.n_cigar, [&](uint32_t op, uint32_t len) {
                c.x += 1;
                if (op == C3R_CIG_I || op == C3R_CIG_D) c.z += 1;
                if (op == C3R_CIG_M || op == C3R_CIG_D || op == C3R_CIG_N) rlen += len;
                w.op(op, len, seg);
            });
            w.close(seg);
            if (!err && w.bad) err = w.bad;
            if (!err && (long long)r.pos + rlen > INT32_MAX) err = LD_END_2G;
        }
        if (err) { load_fail(st, i, err); c = make_int4(0, 0, 0, 0); rlen = 0; }
        cnt[i] = c;
        my_end = (int32_t)(r.pos + rlen);
        rend[i] = my_end;
    } else if (i == n_reads) cnt[i] = make_int4(0, 0, 0, 0);
    int m = my_end;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&st->max_end, m);
}

// ---- exclusive scan of int4 items, three short launches (one when the input fits one block): local / tops / add
constexpr int S4_IT = 4, S4_BLK = 1024 * S4_IT;
__device__ __forceinline__ int4 add4(const int4 a, const int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ int4 wave_incl_scan4(int4 v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int4 t;
        t.x = __shfl_up(v.x, off, 64); t.y = __shfl_up(v.y, off, 64); t.z = __shfl_up(v.z, off, 64); t.w = __shfl_up(v.w, off, 64);
        if (lane >= off) v = add4(v, t);
    }
    return v;
}
__global__ __launch_bounds__(1024) void k_scan4_local(int4 *data, int n, int4 *tops) {
    __shared__ int4 wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * S4_BLK + threadIdx.x * S4_IT;
    int4 v[S4_IT], sum = make_int4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < S4_IT; ++k) { v[k] = (i0 + k < n) ? data[i0 + k] : make_int4(0, 0, 0, 0); sum = add4(sum, v[k]); }
    const int4 incl = wave_incl_scan4(sum);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int4 wb = make_int4(0, 0, 0, 0), tot = wb;
    for (int w = 0; w < 16; ++w) { const int4 t = wtot[w]; if (w < wave) wb = add4(wb, t); tot = add4(tot, t); }
    int4 run = make_int4(wb.x + incl.x - sum.x, wb.y + incl.y - sum.y, wb.z + incl.z - sum.z, wb.w + incl.w - sum.w);
#pragma unroll
    for (int k = 0; k < S4_IT; ++k) { if (i0 + k < n) data[i0 + k] = run; run = add4(run, v[k]); }
    if (threadIdx.x == 0) tops[blockIdx.x] = tot;
}
// the block sums (at most a few hundred), one block; the grand total goes to *total
__global__ __launch_bounds__(1024) void k_scan4_tops(int4 *tops, int nb, int4 *total) {
    __shared__ int4 wtot[16];
    __shared__ int4 carry_s;
    if (threadIdx.x == 0) carry_s = make_int4(0, 0, 0, 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int4 v = i < nb ? tops[i] : make_int4(0, 0, 0, 0);
        const int4 incl = wave_incl_scan4(v);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int4 wb = make_int4(0, 0, 0, 0), tot = wb;
        for (int w = 0; w < 16; ++w) { const int4 t = wtot[w]; if (w < wave) wb = add4(wb, t); tot = add4(tot, t); }
        const int4 c = carry_s;
        if (i < nb) tops[i] = make_int4(c.x + wb.x + incl.x - v.x, c.y + wb.y + incl.y - v.y, c.z + wb.z + incl.z - v.z, c.w + wb.w + incl.w - v.w);
        __syncthreads();
        if (threadIdx.x == 0) carry_s = add4(c, tot);
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
__global__ __launch_bounds__(1024) void k_scan4_add(int4 *data, int n, const int4 *tops) {
    const int4 off = tops[blockIdx.x];
    const int i0 = blockIdx.x * S4_BLK + threadIdx.x * S4_IT;
#pragma unroll
    for (int k = 0; k < S4_IT; ++k) if (i0 + k < n) data[i0 + k] = add4(data[i0 + k], off);
}

// ---- filters: which reads does mpileup see (flag_fails, --min-MQ, non-empty reference span)
// pass[i] = 1 / 0 (then scanned in place: rank among the passing reads); ekey[i] = the read's end, or ~0 for a read that fails
__global__ __launch_bounds__(256) void k_reads_pass(const c3r_read_t *reads, const int32_t *rend, int n_reads, int min_mq, int excl_flags,
                                                    int32_t *pass, uint32_t *ekey) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_reads) return;
    if (i == n_reads) { pass[i] = 0; return; }
    const c3r_read_t r = reads[i];
    const bool ok = !flag_fails(r.flag, excl_flags) && r.mapq >= min_mq && rend[i] > r.pos;
    pass[i] = ok ? 1 : 0;
    ekey[i] = ok ? (uint32_t)rend[i] : 0xffffffffu;
}
// Coverage is deepest at some read's start p: (#passing reads with pos <= p) - (#passing reads with end <= p).  rank[] = exclusive
// scan of the pass flags (rank[n_reads] = their number), ends_sorted = the keys above in ascending order.
__global__ __launch_bounds__(256) void k_cover_max(const c3r_read_t *reads, int n_reads, const int32_t *rank, const uint32_t *ends_sorted,
                                                   LoadStats *st) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cover = 0;
    if (i < n_reads && rank[i + 1] > rank[i]) {
        const int p = reads[i].pos;
        int lo = i + 1, hi = n_reads;                 // first read that starts after p
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (reads[mid].pos > p) hi = mid; else lo = mid + 1; }
        const int started = rank[lo];
        const int n_pass = rank[n_reads];
        int a = 0, b = n_pass;                        // passing reads whose end is <= p
        while (a < b) { const int mid = (a + b) >> 1; if (ends_sorted[mid] <= (uint32_t)p) a = mid + 1; else b = mid; }
        cover = started - a;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cover = max(cover, __shfl_xor(cover, off, 64));
    if ((threadIdx.x & 63) == 0 && cover > 0) atomicMax(&st->max_cover, cover);
}

// ---- second pass over the raw CIGARs: everything the counts were for
// off[i] = exclusive prefix {normalised ops, segments, indel ops, op records} of read i.
__global__ __launch_bounds__(256) void k_reads_write(const c3r_read_t *reads, int n_reads, const uint32_t *cigars, const int4 *off, const int32_t *rend,
                                                     uint32_t *ncig, DevRead *out, DevSeg *rsegs, uint32_t *rseg_first, uint32_t *skey, uint32_t *sval) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_reads) return;
    const int4 o = off[i];
    rseg_first[i] = (uint32_t)o.y;
    if (i == n_reads) return;
    const c3r_read_t r = reads[i];
    DevRead d;
    d.pos = r.pos; d.end = rend[i]; d.cig_off = (uint32_t)o.x; d.n_cig = (uint32_t)(off[i + 1].x - o.x); d.seq_off = r.seq_off;
    d.flag = r.flag; d.mapq = r.mapq; d.hp = r.hp; d.l_seq = r.l_seq;
    out[i] = d;
    uint32_t kout = (uint32_t)o.x, sout = (uint32_t)o.y;
    SegWalk w;
    w.begin(r.pos);
    auto seg = [&](uint32_t k0, uint32_t nk, long long x0, uint32_t q0, long long x1, bool lead_n, bool lead_indel, int) {
        DevSeg g;
        g.pos = (int32_t)x0;
        g.ext_start = g.pos - (lead_indel ? 1 : 0);
        const long long e = x1 > (long long)g.ext_start + 1 ? x1 : (long long)g.ext_start + 1;
        g.end = (int32_t)e;
        g.cig_off = (uint32_t)o.x + k0; g.qstart = q0; g.l_seq = r.l_seq; g.seq_off = r.seq_off; g.read_idx = (uint32_t)i;
        g.n_cig = (uint16_t)nk; g.flag = r.flag; g.mapq = r.mapq; g.hp = r.hp; g.lead_n = lead_n ? 1 : 0; g.pad = 0;
        rsegs[sout] = g;
        skey[sout] = (uint32_t)g.ext_start ^ 0x80000000u;
        sval[sout] = sout;
        ++sout;
    };
    (void)walk_norm(cigars + r.cigar_off, r.n_cigar, [&](uint32_t op, uint32_t len) {
        ncig[kout++] = (len << 4) | op;
        w.op(op, len, seg);
    });
    w.close(seg);
}

// segments into sorted order + the key of their prefix maximum
__global__ __launch_bounds__(256) void k_seg_gather(const DevSeg *rsegs, const uint32_t *perm, int n_segs, DevSeg *segs) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_segs) return;
    // 48-byte records as three 16-byte pieces
    const int4 *s = reinterpret_cast<const int4 *>(rsegs + perm[k]);
    int4 *d = reinterpret_cast<int4 *>(segs + k);
    const int4 a = s[0], b = s[1], c = s[2];
    d[0] = a; d[1] = b; d[2] = c;
}

// ---- inclusive prefix maximum of the ends of the items that pass the filters (INT_MIN before the first), three launches
//   WHAT = 0: DevRead (flag, mapq, end > pos)    WHAT = 1: DevSeg (flag, mapq)
constexpr int PM_IT = 8, PM_BLK = 1024 * PM_IT;
template <int WHAT>
__device__ __forceinline__ int pm_value(const void *items, int i, int min_mq, int excl) {
    if (WHAT == 0) { const DevRead r = static_cast<const DevRead *>(items)[i]; return read_passes(r, min_mq, excl) ? r.end : INT32_MIN; }
    const DevSeg *g = static_cast<const DevSeg *>(items) + i;
    return (!flag_fails(g->flag, excl) && g->mapq >= min_mq) ? g->end : INT32_MIN;
}
__device__ __forceinline__ int wave_incl_max(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(v, off, 64); if (lane >= off) v = max(v, t); }
    return v;
}
template <int WHAT>
__global__ __launch_bounds__(1024) void k_prefmax_local(const void *items, int n, int min_mq, int excl, int32_t *out, int32_t *tops) {
    __shared__ int wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * PM_BLK + threadIdx.x * PM_IT;
    int v[PM_IT], m = INT32_MIN;
#pragma unroll
    for (int k = 0; k < PM_IT; ++k) { v[k] = (i0 + k < n) ? pm_value<WHAT>(items, i0 + k, min_mq, excl) : INT32_MIN; m = max(m, v[k]); }
    const int incl = wave_incl_max(m);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = INT32_MIN, tot = INT32_MIN;
    for (int w = 0; w < 16; ++w) { const int t = wtot[w]; if (w < wave) before = max(before, t); tot = max(tot, t); }
    int run = max(before, __shfl_up(incl, 1, 64));
    if (lane == 0) run = before;
#pragma unroll
    for (int k = 0; k < PM_IT; ++k) { run = max(run, v[k]); if (i0 + k < n) out[i0 + k] = run; }
    if (threadIdx.x == 0) tops[blockIdx.x] = tot;
}
// tops[b] <- maximum of the blocks before b (one block; at most a few hundred entries)
__global__ __launch_bounds__(1024) void k_prefmax_tops(int32_t *tops, int nb) {
    __shared__ int wtot[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = INT32_MIN;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const int v = i < nb ? tops[i] : INT32_MIN;
        const int incl = wave_incl_max(v);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int before = INT32_MIN, tot = INT32_MIN;
        for (int w = 0; w < 16; ++w) { const int t = wtot[w]; if (w < wave) before = max(before, t); tot = max(tot, t); }
        int excl = max(before, __shfl_up(incl, 1, 64));
        if (lane == 0) excl = before;
        const int c = carry_s;
        if (i < nb) tops[i] = max(c, excl);
        __syncthreads();
        if (threadIdx.x == 0) carry_s = max(c, tot);
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_prefmax_add(int32_t *out, int n, const int32_t *tops) {
    const int off = tops[blockIdx.x];
    const int i0 = blockIdx.x * PM_BLK + threadIdx.x * PM_IT;
#pragma unroll
    for (int k = 0; k < PM_IT; ++k) if (i0 + k < n) out[i0 + k] = max(out[i0 + k], off);
}

}  // namespace c3r
