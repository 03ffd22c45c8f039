// reads_kernels.hpp — gfx950 kernels that turn the caller's flat read records (include/c3r_types.h: c3r_read_t, BAM-encoded
// CIGARs, 4-bit bases) into the device tables the tile kernels walk.  Included by c3r_lib.hip only.
//
// What it replaces: the BAM side of `samtools mpileup <bam> -r ...` (src/create_tensor_pileup.py:436-451) — htslib resolves
// each read's CIGAR with a per-read cursor while it streams the file; here every read is prepared once, on the device, when
// the contig's records arrive (c3r_load_reads), and the result is a PILE TABLE: one self-contained 32-byte record (PileRec,
// pileup_kernels.hpp) per piece of an aligned op, BINNED BY REFERENCE POSITION (32-bp bins, counting sort), so that the ops a span
// of the genome needs are one contiguous range of the table — no segment lists, no sort, no per-span searches.
//
//   k_prep<false>     16 lanes per read: validate, reference end, DevRead header, count the records of every bin (atomics)
//   k_prefmax_bins    inclusive prefix maximum of the passing reads' ends (one pass, decoupled look-back) + its histogram over the bins
//   k_bin_scan        exclusive prefix sums over the bins (one pass, decoupled look-back): first record / reads started before /
//                     reads whose prefix-max end lies before / reads ended before each bin; upper bound of the deepest coverage
//   k_prep<true>      the same walk again: every record takes its slot in its bin (the bin counters count down to zero)
// Round 3 prepared per-read normalised CIGARs, aligned segments, two rocPRIM radix sorts, prefix maxima, a bucket index and an op
// table in ~45 launches and 0.85 ms per chr20; those tables survive only as the LEGACY tables below, built on demand for the one
// consumer that still walks a single read's CIGAR (token_at: the ordered haplotype recompute of the 30-channel mode).
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
    int32_t max_cover;        // upper bound of htslib's read list at any read's start: the most reads that overlap one 256-bp coarse bin or end on its edge
    int32_t n_rec;            // records of the pile table
    int32_t n_indel;          // I + D ops of the passing reads (bounds the indel-event scratch of a scan)
    int32_t n_padreads;       // mpileup_compat = 1: reads with a pad inside a run of I ops (the host then builds the c3r_padins_t table)
    int32_t n_recount;        // workgroups of k_prep<false> whose bin table held more occupied slots than their slab of wg_tab takes (k_prep<true> counts theirs again)
};
static_assert(sizeof(LoadStats) == 32, "LoadStats layout");
enum { LD_OK = 0, LD_UNSORTED = 1, LD_CIGAR_RANGE, LD_SEQ_RANGE, LD_BAD_OP, LD_OP_LONG, LD_END_2G, LD_SEG_OPS, LD_RECORDS, LD_PAD_INS };

__device__ __forceinline__ void load_fail(LoadStats *st, int read, int code) {
    atomicMin(&st->err, ((unsigned long long)(unsigned)read << 8) | (unsigned)code);
}

// CIGAR normalisation as a stream: drop H, zero-length ops and pads (a pad is kept — as a 1-long op that consumes nothing —
// exactly when the next real op is a D and the op before it is not an I: htslib marks a deletion only when the D IMMEDIATELY
// follows the op that ends on the column, while insertions are found through pads; behind an insertion the column's indel IS the
// insertion either way, and samtools >= 1.11 reaches the D through the pads of the run, bam_plp_insertion), fold = / X into M, merge
// equal neighbours.  emit(op, len) receives every
// finalised op in order.  Returns LD_OK or the error code.  Serial: one lane walks one read.  The parallel walk of k_prep handles
// the reads whose CIGAR needs none of this (every op kept as it is); the others come here.
template <class Emit>
__device__ __forceinline__ int walk_norm(const uint32_t *cig, uint32_t n, Emit &&emit) {
    bool have = false;
    uint32_t cop = 0;
    unsigned long long clen = 0;
    // ops are fetched eight at a time: the loads of a batch are independent of each other, so a read's walk pays one memory round trip
    // per eight ops instead of one per op
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
            if (have && cop == C3R_CIG_I) continue;
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
// seg(first op index, op count, pos, qstart, end_x, lead_n, lead_indel) is called for every segment that is kept:
// one that holds an M or D, or that starts with an I / D right after an N (that indel is attached to the last intron column).
// (Legacy tables, and the "more than 65535 ops between two N ops" rule of the load-time validation.)
struct SegWalk {
    long long x;          // reference cursor
    uint32_t y;           // query cursor
    uint32_t k;           // index of the next normalised op
    // the open segment
    uint32_t k0, q0, first_op;
    long long x0;
    bool open, useful, after_n;
    int bad;              // LD_SEG_OPS when a segment holds more than 65535 ops
    __device__ __forceinline__ void begin(int32_t pos) { x = pos; y = 0; k = 0; open = false; useful = false; after_n = false; bad = 0; first_op = 15; k0 = 0; q0 = 0; x0 = pos; }
    template <class Seg>
    __device__ __forceinline__ void close(Seg &&seg) {
        if (!open) return;
        const bool lead_indel = after_n && (first_op == C3R_CIG_I || first_op == C3R_CIG_D);
        if (useful || lead_indel) {
            if (k - k0 > 0xffffu) bad = LD_SEG_OPS;
            seg(k0, k - k0, x0, q0, x, after_n, lead_indel);
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
        if (!open) { open = true; useful = false; k0 = k; q0 = y; x0 = x; first_op = o; }
        if (o == C3R_CIG_M) { x += len; y += len; useful = true; }
        else if (o == C3R_CIG_D) { x += len; useful = true; }
        else if (o == C3R_CIG_I) { y += len; }
        else if (o == C3R_CIG_S) y += len;
        ++k;
    }
};

// ---- the pile table ---------------------------------------------------------------------------------------------------------
#ifndef C3R_PREP_GRP
#define C3R_PREP_GRP 16
#endif
constexpr int PREP_GRP = C3R_PREP_GRP;        // lanes per read in k_prep

struct ReadInfo {
    const uint32_t *cig;
    int32_t pos;
    uint32_t n_cig, l_seq, read_idx, wbits;   // wbits: strand and haplotype bits of PileRec::w
    uint64_t seq_off;
    int32_t compat;                           // c3r_params_t::mpileup_compat
    uint32_t padbit;                          // PR_INS_PADS when a run of I ops of this read holds pads (mpileup_compat = 1), else 0
};
// an op's neighbours in the normalised CIGAR (15 = none)
struct OpCtx { uint32_t prev, prev2, nop, nlen, n2op, n2len; };

// The records of ONE normalised op (code, length, reference / query offsets of its first base, the read's previous and next op).
//   M, D: one record per OP_CHOP reference positions; the last piece carries the indel htslib attaches to the op's last column
//         (an I after M / D, a D after M); I: one record, only when its predecessor consumes the reference (M, D, N) — otherwise
//         samtools shows no insertion; N, S, P: none.
// mpileup_compat = 1 (samtools >= 1.11, bam_plp_insertion): a D that follows an insertion AT ONCE is shown on the insertion's column too
// (`C+2TT-1N`): the piece that carries the insertion also carries that deletion's length (M: aux, D: naddr; an I after a ref-skip: -nxt),
// and the D's first piece is marked (PR_DEL_AFTER_INS) as a deletion event on the column before it.
// emit(rstart, w, naddr, q, nxt, aux).
template <class Emit>
__device__ __forceinline__ void op_records(const ReadInfo &R, uint32_t op, uint32_t len, long long x, uint32_t y, const OpCtx c, Emit &&emit) {
    const uint32_t prev = c.prev, nop = c.nop, nlen = c.nlen;
    const uint32_t del_after = (R.compat && nop == C3R_CIG_I && c.n2op == C3R_CIG_D) ? c.n2len : 0u;     // (this op consumes the reference)
    if (op == C3R_CIG_M) {
        const int32_t nxt_last = nop == C3R_CIG_I ? (int32_t)nlen : nop == C3R_CIG_D ? -(int32_t)nlen : 0;
        for (uint32_t d = 0; d < len; d += OP_CHOP) {
            const uint32_t pl = min((uint32_t)OP_CHOP, len - d), q = y + d;
            const uint32_t avail = q >= R.l_seq ? 0u : min(pl, R.l_seq - q);           // (a CIGAR may claim more bases than SEQ holds)
            const uint32_t w = (uint32_t)C3R_CIG_M | ((d ? (uint32_t)C3R_CIG_M : prev) << 2) | R.wbits | (pl << 9) | (avail << 14);
            const bool last = d + OP_CHOP >= len;
            emit((int32_t)(x + d), w, 2ull * R.seq_off + q, q, last ? nxt_last : 0, last ? del_after : 0u);
        }
    } else if (op == C3R_CIG_D) {
        const int32_t nxt_last = nop == C3R_CIG_I ? (int32_t)nlen : 0;
        const bool ins_before = R.compat && prev == C3R_CIG_I && (c.prev2 == C3R_CIG_M || c.prev2 == C3R_CIG_D || c.prev2 == C3R_CIG_N);
        for (uint32_t d = 0; d < len; d += OP_CHOP) {
            const uint32_t pl = min((uint32_t)OP_CHOP, len - d);
            const uint32_t w = (uint32_t)C3R_CIG_D | ((d ? (uint32_t)C3R_CIG_D : prev) << 2) | R.wbits | (pl << 9) | ((!d && ins_before) ? PR_DEL_AFTER_INS : 0u);
            const bool last = d + OP_CHOP >= len;
            emit((int32_t)(x + d), w, last ? (unsigned long long)del_after : 0ull, y, last ? nxt_last : 0, len);
        }
    } else if (op == C3R_CIG_I) {
        if (prev == C3R_CIG_M || prev == C3R_CIG_D || prev == C3R_CIG_N) {
            const uint32_t avail = y >= R.l_seq ? 0u : min(31u, R.l_seq - y);
            emit((int32_t)x, (uint32_t)C3R_CIG_I | (prev << 2) | R.wbits | (avail << 14) | R.padbit, 2ull * R.seq_off + y, y,
                 (R.compat && nop == C3R_CIG_D) ? -(int32_t)nlen : 0, len);
        }
    }
}

__device__ __forceinline__ uint32_t fold_op(uint32_t c) { const uint32_t op = c & 15u; return (op == C3R_CIG_EQ || op == C3R_CIG_X) ? (uint32_t)C3R_CIG_M : op; }
__device__ __forceinline__ bool op_ref(uint32_t op) { return op == C3R_CIG_M || op == C3R_CIG_D || op == C3R_CIG_N; }
__device__ __forceinline__ bool op_qry(uint32_t op) { return op == C3R_CIG_M || op == C3R_CIG_I || op == C3R_CIG_S; }

// A first look at a read's CIGAR by its 16 lanes: total reference length (normalisation never changes it), the I / D ops, and whether
// the parallel walk may take it as it is — every op kept (H only as the first or last op), no zero-length op, no pad, no equal
// neighbours other than M-like ones (which pieces cut anyway), no unknown op code, at most 65535 ops and less than 2^28 reference
// positions (the limits of the normalised form are then met without looking).  Aligners emit nothing else; the rest takes the
// serial walk.  All lanes return the group's values.
__device__ __forceinline__ bool cigar_is_plain(const ReadInfo &R, int gl, long long &ref_len, int &n_indel) {
    unsigned long long rl = 0;
    int ni = 0;
    bool odd = R.n_cig > 0xffffu;
    for (uint32_t k0 = 0; k0 < R.n_cig; k0 += PREP_GRP) {
        const uint32_t k = k0 + (uint32_t)gl;
        if (k < R.n_cig) {
            const uint32_t c = R.cig[k], raw = c & 15u, op = fold_op(c), len = c >> 4;
            if (raw > C3R_CIG_X || len == 0 || raw == C3R_CIG_P || (raw == C3R_CIG_H && k != 0 && k + 1 != R.n_cig)) odd = true;
            if (k > 0 && op != C3R_CIG_M && op == fold_op(R.cig[k - 1])) odd = true;
            if (op_ref(op)) rl += len;
            if (op == C3R_CIG_I || op == C3R_CIG_D) ++ni;
        }
    }
    int o = odd ? 1 : 0;
#pragma unroll
    for (int off = PREP_GRP / 2; off > 0; off >>= 1) {
        rl += __shfl_xor(rl, off, PREP_GRP);
        ni += __shfl_xor(ni, off, PREP_GRP);
        o |= __shfl_xor(o, off, PREP_GRP);
    }
    ref_len = (long long)rl; n_indel = ni;
    return !o && rl < (1ull << 28);
}

// The parallel walk: lane gl of the group takes ops gl, gl + 16, ...; reference / query offsets by 16-lane prefix sums.
template <class Emit>
__device__ __forceinline__ void walk_plain(const ReadInfo &R, int gl, Emit &&emit) {
    uint32_t x = (uint32_t)R.pos, y = 0;                       // (less than 2^28 reference positions: 32 bits do)
    for (uint32_t k0 = 0; k0 < R.n_cig; k0 += PREP_GRP) {
        const uint32_t k = k0 + (uint32_t)gl;
        const bool in = k < R.n_cig;
        const uint32_t c = in ? R.cig[k] : 0u, op = in ? fold_op(c) : (uint32_t)C3R_CIG_H, len = c >> 4;
        OpCtx cx;
        cx.prev = 15u; cx.prev2 = 15u; cx.nop = 15u; cx.nlen = 0; cx.n2op = 15u; cx.n2len = 0;
        if (in && k > 0) { cx.prev = fold_op(R.cig[k - 1]); if (cx.prev == C3R_CIG_H) cx.prev = 15u; }
        if (in && k + 1 < R.n_cig) { const uint32_t cn = R.cig[k + 1]; cx.nop = fold_op(cn); cx.nlen = cn >> 4; if (cx.nop == C3R_CIG_H) cx.nop = 15u; }
        if (R.compat && in) {          // the op after an insertion that follows / the op before an insertion that precedes
            if (cx.nop == C3R_CIG_I && k + 2 < R.n_cig) { const uint32_t c2 = R.cig[k + 2]; cx.n2op = fold_op(c2); cx.n2len = c2 >> 4; if (cx.n2op == C3R_CIG_H) cx.n2op = 15u; }
            if (cx.prev == C3R_CIG_I && k > 1) { cx.prev2 = fold_op(R.cig[k - 2]); if (cx.prev2 == C3R_CIG_H) cx.prev2 = 15u; }
        }
        const uint32_t rl = op_ref(op) ? len : 0u, ql = op_qry(op) ? len : 0u;
        uint32_t ri = rl, qi = ql;
#pragma unroll
        for (int off = 1; off < PREP_GRP; off <<= 1) {
            const uint32_t tr = __shfl_up(ri, off, PREP_GRP), tq = __shfl_up(qi, off, PREP_GRP);
            if (gl >= off) { ri += tr; qi += tq; }
        }
        if (in) op_records(R, op, len, (long long)(int32_t)(x + ri - rl), y + qi - ql, cx, emit);
        x += __shfl(ri, PREP_GRP - 1, PREP_GRP);
        y += __shfl(qi, PREP_GRP - 1, PREP_GRP);
    }
}

// The serial walk (one lane): walk_norm's stream, two ops of look-ahead (the indel attached to an op's last column, and with
// mpileup_compat the deletion right behind that insertion).  Returns the error code of the normalised form (unknown op, a merged op of
// 2^28 or more, more than 65535 ops between two N ops).
// mpileup_compat = 1: samtools >= 1.11 prints the pads INSIDE a run of I ops as '*' (`+3T*T`, bam_plp_insertion).  The walk itself
// does not change (pads consume nothing; the record of the merged I op carries the bases); a read that has such a run is flagged
// (*pads = true: its I records carry PR_INS_PADS) and the host builds the table of those runs (c3r_padins_t) that ev_equal and the
// decoder look up.  A run of more than 64 characters is refused (LD_PAD_INS): the table describes the pads by a 64-bit mask.
template <class Emit>
__device__ __forceinline__ int walk_serial(const ReadInfo &R, Emit &&emit, bool *pads = nullptr) {
    if (R.compat) {
        uint32_t run_i = 0, run_p = 0;
        bool any = false;
        for (uint32_t k = 0; k <= R.n_cig; ++k) {
            const uint32_t c = k < R.n_cig ? R.cig[k] : (1u << 4) /* 1M: closes the last run */, op = c & 15u, len = c >> 4;
            if (len == 0 || op == C3R_CIG_H) continue;
            if (op == C3R_CIG_I) run_i += min(len, 1u << 20); else if (op == C3R_CIG_P) run_p += min(len, 1u << 20);
            else {
                if (run_i && run_p) { if (run_i + run_p > 64u) return LD_PAD_INS; any = true; }
                run_i = 0; run_p = 0;
            }
        }
        if (pads) *pads = any;
    }
    SegWalk w;
    w.begin(R.pos);
    auto seg = [](uint32_t, uint32_t, long long, uint32_t, long long, bool, bool) {};
    struct NOp { uint32_t op, len, y, prev, prev2; long long x; };
    NOp q0, q1;                               // (two named slots, not an array indexed by nq: that array was the kernels' 80 B of scratch)
    q0.op = q1.op = 15u; q0.len = q1.len = 0; q0.y = q1.y = 0; q0.prev = q1.prev = 15u; q0.prev2 = q1.prev2 = 15u; q0.x = q1.x = 0;
    int nq = 0;
    uint32_t prev = 15u, prev2 = 15u, y = 0;
    long long x = R.pos;
    auto flush = [&](const NOp &o, uint32_t nop, uint32_t nlen, uint32_t n2op, uint32_t n2len) {
        OpCtx cx;
        cx.prev = o.prev; cx.prev2 = o.prev2; cx.nop = nop; cx.nlen = nlen; cx.n2op = n2op; cx.n2len = n2len;
        op_records(R, o.op, o.len, o.x, o.y, cx, emit);
    };
    const int err = walk_norm(R.cig, R.n_cig, [&](uint32_t op, uint32_t len) {
        NOp c;
        c.op = op; c.len = len; c.y = y; c.prev = prev; c.prev2 = prev2; c.x = x;
        if (nq == 2) { flush(q0, q1.op, q1.len, op, len); q0 = q1; q1 = c; }
        else if (nq == 1) { q1 = c; nq = 2; }
        else { q0 = c; nq = 1; }
        if (op_ref(op)) x += len;
        if (op_qry(op)) y += len;
        prev2 = prev; prev = op;
        w.op(op, len, seg);
    });
    if (err) return err;
    if (nq == 2) { flush(q0, q1.op, q1.len, 15u, 0u); flush(q1, 15u, 0u, 15u, 0u); }
    else if (nq == 1) flush(q0, 15u, 0u, 15u, 0u);
    w.close(seg);
    return w.bad;
}

struct PrepArgs {
    const c3r_read_t *reads; int32_t n_reads;
    const uint32_t *cigars; long long n_cigar_ops, n_seq_bytes;
    int32_t min_mq, excl_flags, compat;
    BinGeo geo;
    uint32_t *cnt;            // [nb] records per bin: counted up by k_prep<false>, counted down to zero by k_prep<true>
    uint32_t *sc, *ec;        // [nbc] per coarse bin: reads that start in it / reads that end in the 256 positions up to its first
    const uint32_t *rec_off;  // [nb + 1] k_bin_scan's prefix sums (k_prep<true>)
    DevRead *out;             // [n_reads] headers, written by k_prep<false>
    uint8_t *serial;          // [n_reads] 1 = the read takes the serial walk (3: and holds a run of I ops with pads, mpileup_compat = 1)
    int32_t *nind;            // [n_reads] I + D ops of the read, 0 when the filters drop it (summed by k_prefmax_bins: 54 k atomics on one
                              // word would take 0.6 ms)
    PileRec *recs;
    LoadStats *st;
    uint32_t *wg_tab;         // [workgroups][2 HB + 4] what k_prep<false> counted per workgroup — [0] entries, then (bin + 1, records) pairs from word 4 on, only
                              // the occupied slots of its table (a couple of hundred: ~2 KB of the slab's 8 KB are ever touched): k_prep<true> rebuilds its
                              // table from them instead of walking its reads' CIGARs once more to count (null: it counts again)
};

// Bin counters are updated through a per-workgroup hash table in LDS: a workgroup's 64 reads are neighbours in the sorted input and
// pile their records into the same few hundred bins, so it counts them in LDS and touches each global counter once.  One global
// atomic per record (3.5 M per chr20 on ~60 k hot counters) ran at 35 G atomics/s and was 0.10 of the first pass's 0.145 ms
// (tools/prep_probe.hip: 0.044 ms without them); coarser bins were slower still — it is contention on the counters, not their number.
// (256 threads = 16 reads and 1024 slots measured best: 0.059 / 0.098 ms for the two passes against 0.085 / 0.128 with 1024 threads, where
// a workgroup waits for the slowest of 64 reads)
#ifndef C3R_PREP_THREADS
#define C3R_PREP_THREADS 256
#endif
#ifndef C3R_HB_LOG
#define C3R_HB_LOG 10
#endif
constexpr int PREP_THREADS = C3R_PREP_THREADS, PREP_READS = PREP_THREADS / PREP_GRP;
constexpr int HB_LOG = C3R_HB_LOG, HB = 1 << HB_LOG, HB_PROBES = 16;
struct BinHash { uint32_t key[HB], val[HB]; };              // key = bin + 1, 0 = empty
// Where bin b's record counter lies.  The bins of ONE locus are neighbours, and at a locus in the thousands every workgroup of the load adds to the
// same ~150 of them: 0.8 M atomics on five cache lines (atomics on one line serialise like atomics on one word).  So the counters are laid out in
// blocks of 1024 bins with the 256 groups of four bins transposed 32 x 8: neighbouring groups lie 128 bytes apart, a locus' counters on 32 lines.
// (k_bin_scan reads a group of four with one 16-byte load as before.)
// Capped 20,000x locus (88 k reads): k_prep<false> 0.64 -> 0.26 ms, k_prep<true> 0.69 -> 0.26 ms; chr20 at 20x and the 500x contig: unchanged.
// (-DC3R_CNT_SWZ=0: the plain layout.)
#ifndef C3R_CNT_SWZ
#define C3R_CNT_SWZ 1
#endif
__host__ __device__ __forceinline__ uint32_t cnt_at(uint32_t b) {          // (__host__: tests/c/layout_check.hip)
    if (!C3R_CNT_SWZ) return b;
    const uint32_t g = (b >> 2) & 255u;
    return (b & ~1023u) | ((((g & 31u) << 3) | (g >> 5)) << 2) | (b & 3u);
}
#ifndef C3R_WG_TAB_PAIRS
#define C3R_WG_TAB_PAIRS 510
#endif
constexpr int WG_TAB_PAIRS = C3R_WG_TAB_PAIRS, WG_TAB_WORDS = 2 * WG_TAB_PAIRS + 4;      // PrepArgs::wg_tab: a 4-KB slab per workgroup; a workgroup with more occupied slots
                                                                             // (long reads far apart) says so ([0] = ~0) and the second pass counts again
// slot of `bin`, or -1: not there (insert: and no free slot among its HB_PROBES places — such a bin goes to the global counter directly,
// in every walk alike, because an occupied slot never becomes free)
__device__ __forceinline__ int hb_find(BinHash &T, uint32_t bin, bool insert) {
    uint32_t h = (bin * 2654435761u) >> (32 - HB_LOG);
    for (int p = 0; p < HB_PROBES; ++p, h = (h + 1) & (HB - 1)) {
        uint32_t k = T.key[h];
        if (k == bin + 1u) return (int)h;
        if (k == 0u) {
            if (!insert) return -1;
            k = atomicCAS(&T.key[h], 0u, bin + 1u);
            if (k == 0u || k == bin + 1u) return (int)h;
        }
    }
    return -1;
}

template <bool WRITE>
__global__ __launch_bounds__(PREP_THREADS) void k_prep(const PrepArgs a) {
    __shared__ BinHash T;
    __shared__ uint32_t s_base[WRITE ? HB : 1];
    __shared__ uint32_t s_ntab;
    __shared__ int s_cs[WRITE ? 1 : PREP_READS], s_ce[WRITE ? 1 : PREP_READS];      // coarse bin of each read's start / end (-1: no read)
    const int tid = (int)threadIdx.x, gl = tid & (PREP_GRP - 1);
    const int i = (int)(blockIdx.x * PREP_READS + (tid / PREP_GRP));
    const bool valid = i < a.n_reads;
    uint32_t *const tab = a.wg_tab ? a.wg_tab + (size_t)blockIdx.x * WG_TAB_WORDS : nullptr;
    for (int h = tid; h < HB; h += PREP_THREADS) { T.key[h] = 0; T.val[h] = 0; }
    if (tid == 0) s_ntab = 0;
    __syncthreads();
    ReadInfo R;
    R.cig = a.cigars; R.pos = 0; R.n_cig = 0; R.l_seq = 0; R.read_idx = (uint32_t)i; R.seq_off = 0; R.wbits = 0; R.compat = a.compat; R.padbit = 0;
    // every record of a passing read, counted in the workgroup's table (a bin that finds no place there: `spill`)
    auto tally = [&](int32_t rstart, uint32_t, unsigned long long, uint32_t, int32_t, uint32_t) {
        const int b = bin_of(a.geo, rstart);
        const int s_ = hb_find(T, (uint32_t)b, true);
        if (s_ >= 0) atomicAdd(&T.val[s_], 1u);
        else if (!WRITE) __hip_atomic_fetch_add(&a.cnt[cnt_at((uint32_t)b)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if (!WRITE) {
        c3r_read_t r;
        r.pos = 0; r.cigar_off = 0; r.n_cigar = 0; r.l_seq = 0; r.seq_off = 0; r.flag = 4; r.mapq = 0; r.hp = 0;
        int err = LD_OK;
        long long ref_len = 0;
        int n_indel = 0;
        bool plain = true, pass = false, pads = false;
        if (valid) {
            r = a.reads[i];
            if (i > 0 && r.pos < a.reads[i - 1].pos) err = LD_UNSORTED;
            else if ((long long)r.cigar_off + r.n_cigar > a.n_cigar_ops) err = LD_CIGAR_RANGE;
            else if ((long long)r.seq_off + (r.l_seq + 1) / 2 > a.n_seq_bytes) err = LD_SEQ_RANGE;
            R.cig = a.cigars + r.cigar_off; R.pos = r.pos; R.n_cig = err ? 0u : r.n_cigar; R.l_seq = r.l_seq; R.seq_off = r.seq_off;
            R.wbits = ((r.flag & 16u) ? 64u : 0u) | ((r.hp == 1 ? 1u : r.hp == 2 ? 2u : 0u) << 7);
            plain = cigar_is_plain(R, gl, ref_len, n_indel);
            pass = !err && !flag_fails(r.flag, a.excl_flags) && r.mapq >= a.min_mq;
            if (!err) {
                if (plain) { if (pass) walk_plain(R, gl, tally); }
                else if (gl == 0) {                                        // (also for a read the filters drop: its CIGAR is validated all the same)
                    if (pass) err = walk_serial(R, tally, &pads);
                    else err = walk_serial(R, [](int32_t, uint32_t, unsigned long long, uint32_t, int32_t, uint32_t) {}, &pads);
                }
                if (!err && (long long)r.pos + ref_len > INT32_MAX) err = LD_END_2G;
            }
            if (gl == 0) {
                if (err) { load_fail(a.st, i, err); ref_len = 0; n_indel = 0; }
                DevRead d;
                d.pos = r.pos; d.end = (int32_t)(r.pos + ref_len); d.cig_off = r.cigar_off; d.n_cig = r.n_cigar; d.seq_off = r.seq_off;
                d.flag = r.flag; d.mapq = r.mapq; d.hp = r.hp; d.l_seq = r.l_seq;
                a.out[i] = d;
                a.serial[i] = plain ? 0 : (pads && !err) ? 3 : 1;                       // (3: serial walk, and its I records carry PR_INS_PADS)
                if (pads && !err) __hip_atomic_fetch_add(&a.st->n_padreads, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_cs[tid / PREP_GRP] = bin_of(a.geo, d.pos) >> CBIN_SHIFT;
                // ec: the coarse bin that holds the read's EXCLUSIVE end.  k_bin_scan's exclusive sum over ec[j], j < c, then counts the reads whose
                // end position lies before coarse bin c, and  B(c) = reads started up to the end of c - that sum  bounds htslib's read list for every
                // read that starts in c (the kept reads with end > start - 1: a read whose last base sits on start - 1 still has its end inside c,
                // also when start is the bin's first position) — what decides whether mpileup's depth cap can bite at all (LoadStats::max_cover)
                int je = (d.end >> BIN_SHIFT) - a.geo.base;
                je = je < 0 ? 0 : je >= a.geo.nb ? a.geo.nb - 1 : je;
                s_ce[tid / PREP_GRP] = je >> CBIN_SHIFT;
                a.nind[i] = pass ? n_indel : 0;
            }
        } else if (gl == 0) { s_cs[tid / PREP_GRP] = -1; s_ce[tid / PREP_GRP] = -1; }
        __syncthreads();
        // the reads' starts and ends per coarse bin: a workgroup's reads are neighbours in position order, and at a locus in the thousands every read
        // of the sample lands in the same handful of coarse bins — one atomic per read made those words the kernel's time (88 k reads on five words:
        // 0.8 ms).  The first read of a workgroup with a given coarse bin adds for all of them.
        if (tid < 2 * PREP_READS) {
            const int *v = tid < PREP_READS ? s_cs : s_ce;
            const int k = tid < PREP_READS ? tid : tid - PREP_READS;
            const int mine = v[k];
            int n = 0;
            bool first = mine >= 0;
            for (int j = 0; j < PREP_READS; ++j) { if (v[j] == mine) { n += 1; if (j < k) first = false; } }
            if (first) __hip_atomic_fetch_add(&(tid < PREP_READS ? a.sc : a.ec)[mine], (uint32_t)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        for (int h = tid; h < HB; h += PREP_THREADS)
            if (T.key[h]) __hip_atomic_fetch_add(&a.cnt[cnt_at(T.key[h] - 1u)], T.val[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tab) {
            // the occupied slots, densely (order does not matter): what the second pass needs to know instead of counting again
            for (int h = tid; h < HB; h += PREP_THREADS)
                if (T.key[h]) { const uint32_t k = atomicAdd(&s_ntab, 1u); if (k < (uint32_t)WG_TAB_PAIRS) reinterpret_cast<uint2 *>(tab + 4)[k] = make_uint2(T.key[h], T.val[h]); }
            __syncthreads();
            if (tid == 0) {
                tab[0] = s_ntab <= (uint32_t)WG_TAB_PAIRS ? s_ntab : ~0u;
                if (s_ntab > (uint32_t)WG_TAB_PAIRS) __hip_atomic_fetch_add(&a.st->n_recount, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        bool pass = false, serial = false;
        if (valid) {
            const DevRead d = a.out[i];
            pass = !(flag_fails(d.flag, a.excl_flags) || d.mapq < a.min_mq);
            serial = a.serial[i] != 0;
            R.padbit = a.serial[i] == 3 ? PR_INS_PADS : 0u;
            R.cig = a.cigars + d.cig_off; R.pos = d.pos; R.n_cig = d.n_cig; R.l_seq = d.l_seq; R.seq_off = d.seq_off;
            R.wbits = ((d.flag & 16u) ? 64u : 0u) | ((d.hp == 1 ? 1u : d.hp == 2 ? 2u : 0u) << 7);
        }
        // how many records this workgroup has for each of its bins: the table k_prep<false> left (or, without it, a first walk); one atomic
        // per bin takes that many slots (the counters count down: the workgroup's run in bin b is [rec_off[b] + left - n, rec_off[b] + left));
        // then the walk: every record into its run
        const uint32_t nt = tab ? tab[0] : ~0u;
        if (nt != ~0u) {
            // (an entry that finds no place among its HB_PROBES slots now — the insertion order differs — is simply left out: its records take
            // the direct path below, against the global counter that holds them since the first pass)
            for (uint32_t k = (uint32_t)tid; k < nt; k += PREP_THREADS) {
                const uint2 e = reinterpret_cast<const uint2 *>(tab + 4)[k];
                const int s_ = hb_find(T, e.x - 1u, true);
                if (s_ >= 0) T.val[s_] = e.y;
            }
        } else if (pass) { if (!serial) walk_plain(R, gl, tally); else if (gl == 0) (void)walk_serial(R, tally); }
        __syncthreads();
        for (int h = tid; h < HB; h += PREP_THREADS) {
            if (!T.key[h]) continue;
            const uint32_t b = T.key[h] - 1u, nrec = T.val[h];
            const uint32_t left = __hip_atomic_fetch_add(&a.cnt[cnt_at(b)], 0u - nrec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_base[h] = a.rec_off[b] + (left - nrec);
            T.val[h] = 0;
        }
        __syncthreads();
        auto put = [&](int32_t rstart, uint32_t w, unsigned long long naddr, uint32_t q, int32_t nxt, uint32_t aux) {
            const int b = bin_of(a.geo, rstart);
            const int s_ = hb_find(T, (uint32_t)b, false);
            size_t at;
            if (s_ >= 0) at = (size_t)s_base[s_] + atomicAdd(&T.val[s_], 1u);
            else at = (size_t)a.rec_off[b] + (__hip_atomic_fetch_add(&a.cnt[cnt_at((uint32_t)b)], ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u);
            int4 *dst = reinterpret_cast<int4 *>(a.recs + at);
            dst[0] = make_int4(rstart, (int)w, (int)(uint32_t)naddr, (int)(uint32_t)(naddr >> 32));
            dst[1] = make_int4((int)q, i, nxt, (int)aux);
        };
        if (pass) { if (!serial) walk_plain(R, gl, put); else if (gl == 0) (void)walk_serial(R, put); }
    }
}

// ---- decoupled look-back over 64-bit words (flag in the two top bits, 62 bits of payload), sums or maxima
template <bool MAX>
__device__ __forceinline__ unsigned long long lb_lookback_op(unsigned long long *state, int b, unsigned long long mine) {
    const int lane = (int)(threadIdx.x & 63);
    constexpr unsigned long long PAY = 0x3fffffffffffffffull;
    unsigned long long excl = 0;
    if (b > 0) {
        if (lane == 0) lb_store(&state[b], (1ull << 62) | mine);
        for (int top = b - 1; top >= 0; top -= 64) {
            const int i = top - lane;
            unsigned long long w = 0;
            bool incl_found = false;
            for (;;) {
                w = i >= 0 ? lb_load(&state[i]) : (2ull << 62);
                const unsigned long long not_ready = __ballot((w >> 62) == 0), incl = __ballot((w >> 62) == 2);
                const int first_incl = incl ? __ffsll((long long)incl) - 1 : 64;
                const unsigned long long need = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);
                if (!(not_ready & need)) { incl_found = incl != 0; w = (lane <= first_incl) ? (w & PAY) : 0ull; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long t = __shfl_xor(w, off, 64); w = MAX ? (t > w ? t : w) : w + t; }
            excl = MAX ? (w > excl ? w : excl) : excl + w;
            if (incl_found) break;
        }
    }
    const unsigned long long incl_v = MAX ? (mine > excl ? mine : excl) : excl + mine;
    if (lane == 0) lb_store(&state[b], (2ull << 62) | (incl_v & PAY));
    return excl;
}

// ---- inclusive prefix maximum of the ends of the reads that pass the filters (INT_MIN before the first), one pass, one read per
// thread; pc[c] += 1 for every read whose prefix maximum lies in coarse bin c (reads before the first passing read: bin 0) — a long
// read's end is the prefix maximum of all the reads that follow it until a longer one comes, so equal neighbours inside a wavefront
// are added with one atomic.  Also the two totals k_prep<false> left per read: the largest end of a passing read (= the last prefix
// maximum) and the I + D ops.
constexpr int PM_BLK = 1024;
__global__ __launch_bounds__(1024) void k_prefmax_bins(const DevRead *reads, int n, int min_mq, int excl, BinGeo geo, const int32_t *nind, int32_t *out, uint32_t *pc,
                                                       LoadStats *st, int32_t *ticket, unsigned long long *state) {
    __shared__ int s_b, wtot[16], s_ni[16];
    __shared__ unsigned long long s_excl;
    if (threadIdx.x == 0) s_b = atomicAdd(ticket, 1);
    __syncthreads();
    const int b = s_b, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = b * PM_BLK + (int)threadIdx.x;
    int v = INT32_MIN, ni = 0;
    if (i < n) { const DevRead r = reads[i]; if (read_passes(r, min_mq, excl)) v = r.end; ni = nind[i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ni += __shfl_xor(ni, off, 64);
    if (lane == 0) s_ni[wave] = ni;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off, 64); if (lane >= off) incl = max(incl, t); }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = INT32_MIN, tot = INT32_MIN;
    for (int w = 0; w < 16; ++w) { const int t = wtot[w]; if (w < wave) before = max(before, t); tot = max(tot, t); }
    if (wave == 0) {
        // payload: end + 1 (ends are positive), 0 = none
        const unsigned long long e = lb_lookback_op<true>(state, b, tot > 0 ? (unsigned long long)tot + 1ull : 0ull);
        if (lane == 0) s_excl = e;
    }
    __syncthreads();
    const int carry = s_excl ? (int)(s_excl - 1ull) : INT32_MIN;
    const int run = max(max(before, carry), incl);
    if (i < n) {
        out[i] = run;
        if (i == n - 1) st->max_end = max(run, 0);
    }
    // one atomic per run of equal values inside the wavefront (the values never decrease)
    const int key = i < n ? (run > 0 ? bin_of(geo, run) >> CBIN_SHIFT : 0) : -1;
    const int prev = __shfl_up(key, 1, 64);
    const unsigned long long heads = __ballot(key >= 0 && (lane == 0 || prev != key)), live = __ballot(key >= 0);
    if (key >= 0 && (lane == 0 || prev != key)) {
        const unsigned long long later = heads & ~((2ull << lane) - 1ull);        // the next head after this lane
        const int end_lane = later ? __ffsll((long long)later) - 1 : 64;
        const unsigned long long mine = live & (end_lane >= 64 ? ~0ull : ((1ull << end_lane) - 1ull)) & ~((1ull << lane) - 1ull);
        __hip_atomic_fetch_add(&pc[key], (uint32_t)__popcll(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += s_ni[w]; if (t) atomicAdd(&st->n_indel, t); }
}

// ---- the bins' prefix sums, one launch, decoupled look-back.  Blocks [0, nfine): records per bin -> rec_off (cnt stays: k_prep<true> counts
// it down to zero).  Blocks [nfine, ...): the three read counters per coarse bin -> rtab, zeroed on the way (the next load starts from
// clean counters), and the upper bound of the deepest coverage.  Totals to rec_off[nb], rtab[nbc] and the LoadStats.
constexpr int BS_IT = 4, BS_BLK = 1024 * BS_IT;
__global__ __launch_bounds__(1024) void k_bin_scan(const uint32_t *cnt, uint32_t *sc, uint32_t *pc, uint32_t *ec, BinGeo geo, int nfine, uint32_t *rec_off, int4 *rtab, LoadStats *st,
                                                   int32_t *tickets /* [2] */, unsigned long long *state_f, unsigned long long *state_a, unsigned long long *state_b) {
    __shared__ int s_b;
    __shared__ uint4 wtot[16];
    __shared__ unsigned long long s_ea, s_eb;
    __shared__ int s_cov[16];
    const bool fine = (int)blockIdx.x < nfine;
    if (threadIdx.x == 0) s_b = atomicAdd(&tickets[fine ? 0 : 1], 1);
    __syncthreads();
    const int b = s_b, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = b * BS_BLK + (int)threadIdx.x * BS_IT;
    const int n = fine ? geo.nb : geo.nbc;
    uint4 v[BS_IT], sum = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < BS_IT; ++k) {
        v[k] = make_uint4(0, 0, 0, 0);
        if (i0 + k < n) {
            if (fine) v[k].x = cnt[cnt_at((uint32_t)(i0 + k))];
            else { v[k] = make_uint4(sc[i0 + k], pc[i0 + k], ec[i0 + k], 0); sc[i0 + k] = 0; pc[i0 + k] = 0; ec[i0 + k] = 0; }
        }
        sum.x += v[k].x; sum.y += v[k].y; sum.z += v[k].z;
    }
    uint4 incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint4 t;
        t.x = __shfl_up(incl.x, off, 64); t.y = __shfl_up(incl.y, off, 64); t.z = __shfl_up(incl.z, off, 64);
        if (lane >= off) { incl.x += t.x; incl.y += t.y; incl.z += t.z; }
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint4 wb = make_uint4(0, 0, 0, 0), tot = wb;
    for (int w = 0; w < 16; ++w) { const uint4 t = wtot[w]; if (w < wave) { wb.x += t.x; wb.y += t.y; wb.z += t.z; } tot.x += t.x; tot.y += t.y; tot.z += t.z; }
    // look-back chains: fine blocks one sum; coarse blocks two 31-bit sums (starts | prefix-max histogram) on wave 0 and the ends on wave 1
    if (wave == 0) {
        const unsigned long long e = fine ? lb_lookback_op<false>(state_f, b, (unsigned long long)tot.x) : lb_lookback_op<false>(state_a, b, ((unsigned long long)tot.x << 31) | tot.y);
        if (lane == 0) s_ea = e;
    }
    if (wave == 1 && !fine) { const unsigned long long e = lb_lookback_op<false>(state_b, b, (unsigned long long)tot.z); if (lane == 0) s_eb = e; }
    __syncthreads();
    const unsigned long long ea = s_ea, eb = fine ? 0ull : s_eb;
    uint4 run;
    run.x = (fine ? (uint32_t)ea : (uint32_t)(ea >> 31)) + wb.x + incl.x - sum.x;
    run.y = (fine ? 0u : (uint32_t)(ea & 0x7fffffffu)) + wb.y + incl.y - sum.y;
    run.z = (uint32_t)eb + wb.z + incl.z - sum.z;
    run.w = 0;
    int cov = 0;
#pragma unroll
    for (int k = 0; k < BS_IT; ++k) {
        if (i0 + k < n) {
            if (fine) rec_off[i0 + k] = run.x;
            else {
                rtab[i0 + k] = make_int4((int)run.x, (int)run.y, (int)run.z, 0);
                cov = max(cov, (int)(run.x + v[k].x) - (int)run.z);                // reads started up to the end of the coarse bin - reads ended before its start
            }
        }
        run.x += v[k].x; run.y += v[k].y; run.z += v[k].z;
        if (i0 + k == n - 1) {
            if (fine) { rec_off[n] = run.x; st->n_rec = run.x >= 0x7fffffffu ? -1 : (int32_t)run.x; }
            else rtab[n] = make_int4((int)run.x, (int)run.y, (int)run.z, 0);
        }
    }
    if (!fine) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cov = max(cov, __shfl_xor(cov, off, 64));
        if (lane == 0) s_cov[wave] = cov;
        __syncthreads();
        if (threadIdx.x == 0) { int m = 0; for (int w = 0; w < 16; ++w) m = max(m, s_cov[w]); if (m > 0) atomicMax(&st->max_cover, m); }
    }
}

// ---- LEGACY tables: per-read normalised CIGARs and aligned segments in read order, for token_at (pileup_kernels.hpp) — the ordered
// haplotype recompute of flagged columns in the 30-channel mode.  Built on demand (c3r_lib.hip, ensure_legacy_tables), one lane per read.
//   k_legacy_count: int2 {normalised ops, segments} per read (+ a zero entry at n_reads for the exclusive scan)
__global__ __launch_bounds__(256) void k_legacy_count(const DevRead *reads, int n_reads, const uint32_t *cigars, int2 *cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_reads) return;
    int2 c = make_int2(0, 0);
    if (i < n_reads) {
        const DevRead r = reads[i];
        SegWalk w;
        w.begin(r.pos);
        auto seg = [&](uint32_t, uint32_t, long long, uint32_t, long long, bool, bool) { c.y += 1; };
        (void)walk_norm(cigars + r.cig_off, r.n_cig, [&](uint32_t op, uint32_t len) { c.x += 1; w.op(op, len, seg); });
        w.close(seg);
    }
    cnt[i] = c;
}
// single-block exclusive scan of the int2 counts in place; totals to *total.  Eight consecutive items per thread and round (8192 per round):
// the carry between rounds is the serial part — one item per thread made the 68 k reads of a MAS-Seq chr20 67 rounds of two barriers each,
// most of the 0.16 ms the legacy tables cost a 30-channel pass.
__global__ __launch_bounds__(1024) void k_legacy_scan(int2 *data, int n, int2 *total) {
    constexpr int IT = 8;
    __shared__ int2 wtot[16];
    __shared__ int2 carry_s;
    if (threadIdx.x == 0) carry_s = make_int2(0, 0);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024 * IT) {
        const int i0 = base + (int)threadIdx.x * IT;
        int2 v[IT], sum = make_int2(0, 0);
#pragma unroll
        for (int k = 0; k < IT; ++k) { v[k] = (i0 + k < n) ? data[i0 + k] : make_int2(0, 0); sum.x += v[k].x; sum.y += v[k].y; }
        int2 incl;
        incl.x = wave_incl_scan(sum.x); incl.y = wave_incl_scan(sum.y);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int2 wb = make_int2(0, 0), tot = wb;
        for (int w = 0; w < 16; ++w) { const int2 t = wtot[w]; if (w < wave) { wb.x += t.x; wb.y += t.y; } tot.x += t.x; tot.y += t.y; }
        const int2 c = carry_s;
        int2 run = make_int2(c.x + wb.x + incl.x - sum.x, c.y + wb.y + incl.y - sum.y);
#pragma unroll
        for (int k = 0; k < IT; ++k) { if (i0 + k < n) data[i0 + k] = run; run.x += v[k].x; run.y += v[k].y; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = make_int2(c.x + tot.x, c.y + tot.y);
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
// off[i] = exclusive prefix {normalised ops, segments} of read i
__global__ __launch_bounds__(256) void k_legacy_write(const DevRead *reads, int n_reads, const uint32_t *cigars, const int2 *off, uint32_t *ncig, DevSeg *rsegs,
                                                      uint32_t *rseg_first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_reads) return;
    const int2 o = off[i];
    rseg_first[i] = (uint32_t)o.y;
    if (i == n_reads) return;
    const DevRead r = reads[i];
    uint32_t kout = (uint32_t)o.x, sout = (uint32_t)o.y;
    SegWalk w;
    w.begin(r.pos);
    auto seg = [&](uint32_t k0, uint32_t nk, long long x0, uint32_t q0, long long x1, bool lead_n, bool lead_indel) {
        DevSeg g;
        g.pos = (int32_t)x0;
        g.ext_start = g.pos - (lead_indel ? 1 : 0);
        const long long e = x1 > (long long)g.ext_start + 1 ? x1 : (long long)g.ext_start + 1;
        g.end = (int32_t)e;
        g.cig_off = (uint32_t)o.x + k0; g.qstart = q0; g.l_seq = r.l_seq; g.seq_off = r.seq_off; g.read_idx = (uint32_t)i;
        g.n_cig = (uint16_t)nk; g.flag = r.flag; g.mapq = r.mapq; g.hp = r.hp; g.lead_n = lead_n ? 1 : 0; g.pad = 0;
        rsegs[sout++] = g;
    };
    (void)walk_norm(cigars + r.cig_off, r.n_cig, [&](uint32_t op, uint32_t len) {
        ncig[kout++] = (len << 4) | op;
        w.op(op, len, seg);
    });
    w.close(seg);
}

}  // namespace c3r
