// pileup_kernels.hpp — gfx950 tensor-build kernels (K1): CIGAR walk -> per-position channel counts
// in LDS -> candidate gates -> window gather.  Included by c3r_lib.hip only.
//
// What it replaces (reference, /root/reference):
//   samtools mpileup column semantics ........ src/create_tensor_pileup.py:436-451 (third-party htslib)
//   generate_tensor ........................... src/create_tensor_pileup.py:85-302
//   sliding-window / candidate driver ......... src/create_tensor_pileup.py:463-637
//   depth>216 rescale ......................... clair3_rna/utils.py:88-92,120
//
// Design.  A workgroup of 256 threads owns a run of at most TILE reference positions, one per thread: per-position
// accumulators [TILE][C] int32 live in LDS, the ops that touch the run are ONE contiguous range of the pile table (PileRec: a
// self-contained record per piece of an aligned op, binned by reference position when the reads are loaded — reads_kernels.hpp), one
// lane per record; base / deletion / indel events are LDS atomics, the gates run per position (tile_columns).  Two drivers sit on top:
//   k_fused_tiles (plain mode): the run = a span of 224 positions + 16 on either side, so every candidate's 33-column window is in
//     LDS; windows are written from there, in arrival order, the candidates' read tokens right after them (tile_tokens), and
//     k_order_spans / k_finalize_sites put the small records in position order (win_idx maps sites to window rows);
//   k_scan_tiles (head/tail calling, splice padding, genotyping, c3r_get_columns): the finished tile is written to HBM once,
//     coalesced, and selection, ordered compaction, the window gather (one wavefront per candidate) and the tokens are separate kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/c3r_types.h"

namespace c3r {

constexpr int TILE = 256;      // reference positions per workgroup
constexpr int FUSE_IN = TILE - 2 * C3R_FLANK;     // 224: the positions of a fused span that can be candidates (k_fused_tiles)
constexpr int SCAN_THREADS = 256;
constexpr int WAVES = SCAN_THREADS / 64;

// Read header as k_prep leaves it (`end` = pos + reference length; cig_off / n_cig: the caller's raw CIGAR).
struct DevRead {
    int32_t pos;       // 0-based
    int32_t end;       // 0-based, exclusive
    uint32_t cig_off;
    uint32_t n_cig;
    uint64_t seq_off;  // byte offset into the 4-bit packed bases
    uint16_t flag;
    uint8_t mapq;
    uint8_t hp;
    uint32_t l_seq;
};
static_assert(sizeof(DevRead) == 32, "DevRead must be 32 bytes");

// LEGACY (token_at only: the ordered haplotype recompute of the 30-channel mode): one aligned segment of a read = the normalised CIGAR
// ops between two N (ref-skip) ops, in read order (k_legacy_write, built on demand).
struct DevSeg {
    int32_t ext_start;  // pos, or pos-1 when the segment starts with an I/D right after an N (indel attached to the
                        // last intron column)
    int32_t end;        // 0-based exclusive reference end of the segment's ops (>= ext_start + 1)
    int32_t pos;        // reference position of the first op
    uint32_t cig_off;   // absolute index of the first op in the normalised CIGAR array
    uint32_t qstart;    // query offset at the first op
    uint32_t l_seq;
    uint64_t seq_off;
    uint32_t read_idx;  // BAM-order ordinal (first-seen order, token order)
    uint16_t n_cig;
    uint16_t flag;
    uint8_t mapq;
    uint8_t hp;
    uint8_t lead_n;     // the op before the segment is an N
    uint8_t pad;
};
static_assert(sizeof(DevSeg) == 48, "DevSeg must be 48 bytes");

// One piece of an aligned CIGAR op of a read that passes the filters, with everything a lane needs to process it on its own: absolute
// reference position, the address of its bases, strand and haplotype, the op before it, and the indel htslib attaches to its last
// column.  M and D ops are cut into pieces of at most OP_CHOP reference positions (one 16-byte load of packed bases covers an M
// piece; a low-error read's few-hundred-base M ops spread over several lanes); an I leaves a record only when samtools shows the
// insertion (its predecessor consumes the reference: M, D, N); N / S / P ops leave none.  Records are binned by `rstart` (32-bp bins,
// k_prep / k_bin_scan): every record that can touch the positions [e0, e1) — pieces starting up to OP_CHOP - 1 before e0, indels
// anchored on e1 - 1 — lies in ONE contiguous range of the table.  The order inside a bin is arbitrary (whoever's atomic came first);
// nothing downstream depends on it: counts are sums, alleles are compared as sets, first-seen order and tokens go by read_idx.
struct PileRec {
    int32_t rstart;     // 0-based reference position of the piece's first base (I: of the base that follows the insertion; the
                        // insertion sits on the column rstart - 1)
    uint32_t w;         // bits 0-1 op (C3R_CIG_M / _I / _D); 2-5 the op before it (15: none; M / D for the later pieces of a cut op);
                        // 6 reverse strand; 7-8 haplotype (1, 2, else 0); 9-13 reference positions of the piece (M, D); 14-18 bases of
                        // the piece that SEQ really holds (M; I: of its first 31)
    uint64_t naddr;     // nibble index of the piece's first base in the packed bases (2 * seq_off + query offset; M, I)
    uint32_t q;         // query offset of the piece's first base (D: of the base that follows the deletion)
    uint32_t read_idx;  // BAM-order ordinal (first-seen order, token order, depth-cap mask)
    int32_t nxt;        // the indel attached to the piece's LAST column: +length of the insertion, -length of the deletion, 0 none
    uint32_t aux;       // I: inserted bases; D: length of the whole deletion
};
static_assert(sizeof(PileRec) == 32, "PileRec must be 32 bytes");
constexpr uint32_t PR_DEL_AFTER_INS = 1u << 19;    // w, first piece of a D that follows an insertion at once (mpileup_compat = 1): samtools >= 1.11 shows
                                                    // the deletion on the insertion's column as well
constexpr uint32_t PR_INS_PADS = 1u << 20;         // w, I record of a read that has a run of I ops with pads in it (mpileup_compat = 1): the printed insertion may
                                                    // hold '*' / '#' characters — look (read_idx, q) up in the c3r_padins_t table
constexpr int OP_CHOP = 30;    // 30 bases + an odd start nibble fit the 32 nibbles of a 16-byte load

// The bins of the pile table: 32 reference positions each, covering [base << 5, (base + nb) << 5).  Two tables of prefix sums (k_bin_scan):
//   rec_off[b], b <= nb    first record of bin b
//   rtab[c],  c <= nbc     per COARSE bin (8 bins = 256 positions): {reads that start before it, reads whose prefix-max end lies before it,
//                          reads that end at or before its first position} — a span's reads are a superset anyway, the coarse grid adds a few
// Four loads per span replace the four binary searches per tile of rounds 1-3.
#ifndef C3R_BIN_SHIFT
#define C3R_BIN_SHIFT 5
#endif
constexpr int BIN_SHIFT = C3R_BIN_SHIFT;
constexpr int CBIN_SHIFT = 3;
struct BinGeo { int32_t base, nb, nbc, pad; };          // nbc = coarse bins (>= ceil(nb / 8))
__host__ __device__ __forceinline__ int bin_of(const BinGeo g, int p) {          // the bin that holds position p, clamped into the table
    const int b = (p >> BIN_SHIFT) - g.base;
    return b < 0 ? 0 : b >= g.nb ? g.nb - 1 : b;
}
__host__ __device__ __forceinline__ int bin_edge(const BinGeo g, long long p) {  // tab index of the bin that holds p ("everything before it"), clamped to [0, nb]
    const long long b = (p >> BIN_SHIFT) - g.base;
    return b < 0 ? 0 : b > g.nb ? g.nb : (int)b;
}

struct EvRec {          // one indel event, bucketed by position inside a tile
    uint64_t key;       // insertion: first <=16 base codes, 4 bits each; deletion: 0
    uint32_t len;
    uint32_t read_idx;
    uint32_t qpos;      // query offset of the first inserted base
    uint16_t pl;        // position inside the tile
    uint8_t kind;       // bit0 = reverse strand, bit1 = insertion, bit2 = the insertion holds pads (its c3r_padins_t entry exists)
    uint8_t ch;         // channel receiving the max-multiplicity (I1 / i1 / D1 / d1)
};
static_assert(sizeof(EvRec) == 24, "EvRec must be 24 bytes");

// One scan can cover several regions (the reference's chunks, each ctg_start-33 .. ctg_end+33) in ONE set of launches:
// their position slots are laid out back to back, every region padded to whole tiles plus one empty guard tile, so that
// slot arithmetic (slot +- 16) never crosses from one region into the next and the 33-contiguous-rows rule sees a gap there.
// slot = tile * TILE + (p - p0).  Thirteen chunk launches of ~1.5 k tiles each were latency-bound; one launch of ~20 k tiles
// fills the chip.
struct TileGeo {
    int32_t p0;        // 0-based genome position of the tile's first slot
    int32_t p1;        // end (exclusive) of the positions that belong to the region; p1 == p0 for a guard tile
    int32_t region;
    int32_t pad;
};
static_assert(sizeof(TileGeo) == 16, "TileGeo must be 16 bytes");

struct ScanArgs {
    const DevRead *reads;
    const uint8_t *seq;
    int32_t n_reads;
    int32_t compat;               // c3r_params_t::mpileup_compat (the records were built for it)
    const PileRec *recs;          // the pile table
    const uint32_t *rec_off;      // [bins.nb + 1] first record of every bin
    const int4 *rtab;             // [bins.nbc + 1] read-range prefix sums per coarse bin (BinGeo)
    BinGeo bins;
    uint8_t *tile_cols;           // [n_tiles] 1 = this tile's columns were written (0: implicitly all-zero)
    int4 *tile_rng;               // [n_tiles] {lo, hi, rlo, rhi} from k_tile_ranges: reads / records that can touch the tile
    int32_t *tile_list;           // compact list of tiles covered by at least one read span
    int32_t *n_tile_list;
    // prune: intron-only tiles with no aligned segment within 16 bp are not scanned at all (their rows cannot be in any candidate's
    // window); they go to tile_list2, which c3r_get_columns completes on demand
    int32_t prune;
    int32_t *tile_list2;
    int32_t *n_tile_list2;
    int32_t n_tiles;
    int32_t head_tail;            // last_row (end of the row stream) is only needed for the head/tail flush rule
    int32_t abl;                  // timing-only ablation bits (env C3R_SCAN_ABL, 0 in production)
    int32_t deep_min;             // a span whose record range holds at least this many records is left to k_fused_deep; env C3R_DEEP_MIN
    int32_t no_shift;             // env C3R_NO_SHIFT: spans stay on the regions' fixed grid (k_tile_ranges_fused)
    unsigned long long *dbg;      // null in production; env C3R_SCAN_DBG: per-phase wall-clock sums of k_scan_tiles' heavy tiles (100 MHz ticks)
    const uint8_t *ref;           // upper-cased reference slice
    int32_t ref_beg0;             // 0-based position of ref[0]
    int32_t ref_len;
    const TileGeo *geo;           // [n_tiles] where each tile sits on the genome; several regions (chunks) share one launch
    int32_t *cols;                // [n_pos][C]
    int32_t *depth;               // [n_pos]
    int32_t *ncov;                // [n_pos] number of reads covering (incl. ref-skips)
    uint8_t *flags;               // [n_pos] bit0 row, bit1 candidate gate, bit2 emitted, bit3 phased channels to be recomputed in order
    const int32_t *lbed; int32_t n_lbed;   // -l column filter (merged, sorted, half-open 0-based)
    const int32_t *cbed; int32_t n_cbed;   // confident bed
    const int32_t *sites; int32_t n_sites; // genotyping mode (sorted, 1-based)
    int32_t has_lbed, has_cbed, genotyping;
    int32_t min_mq, excl_flags, min_cov;
    double snp_af, indel_af;
    const uint32_t *af_tab;       // [AF_TAB] per depth d: the smallest count c >= 1 with (double)c / (double)max(d, 1) >= snp_af (low half) / indel_af (high
                                  // half), 65535: none — the reference's float64 AF gates (src/create_tensor_pileup.py:267-285) as integer compares; built on the
                                  // host with the very division (c3r_lib.hip, af_table); deeper positions divide
    EvRec *ev;                    // scratch: ev_cap records, bump-allocated per tile through ev_cursor
    unsigned long long *ev_cursor;
    unsigned long long ev_cap;    // a reservation past it is refused: the tile skips its events and raises *ev_overflow
    EvRec *ev_wg;                 // k_fused_deep: per-workgroup arrival-order event buffers, ev_wg_cap records each (null: none — deep tiles walk twice)
    int32_t ev_wg_cap;
    // giant spans (split_min records or more in range): their walk is done ahead of k_fused_deep by several workgroups of k_deep_walk, each over a
    // slice of the span's records, into one of GIANT_SLOTS global accumulators (null: no pool yet — the span is walked by its own workgroup)
    int32_t split_min;            // env C3R_SPLIT_MIN
    int32_t split_slice;          // records per slice (GIANT_SLICE; env C3R_SPLIT_SLICE: tests cut small spans into many slices)
    int32_t *giant_acc;           // [GIANT_SLOTS][GIANT_STRIDE] counts, max deletion, order flags, {events met, first event in giant_ev, events it may hold}; zeroed before every scan
    EvRec *giant_ev;              // [giant_pool] the giant spans' events in arrival order: a span reserves as many as it has records in range (a record shows at most one)
    int32_t giant_pool;           //   (no room left: the span is walked by its own workgroup); *n_giant_ev: reserved so far
    int32_t *n_giant_ev;
    int4 *help_list;              // [GIANT_SLOTS * GIANT_MAX_HELP] {span's list position, slice, slices, slot}
    uint2 *giant_tab;             // [2 * giant_pool] {1 + first event of an allele, its channel << 30 | its multiplicity}: the giant spans' allele counts (k_deep_alleles);
                                  // all-zero between scans — k_fused_deep clears what it reads
    int32_t *n_giant, *n_help;
    // ... and only where it pays: splitting a span helps when CUs would idle beside it, not when thousands of deep spans keep them all busy (the
    // slices' kernels run before k_fused_deep and delay it).  A listed giant span is split iff its records x split_cus >= the records of all deep
    // spans of the scan (*deep_recs, summed by the list kernel): it alone is more than a CU's share.  k_deep_walk, k_deep_alleles and k_fused_deep
    // all decide by this after the list kernel has finished (giant_on).
    unsigned long long *deep_recs;
    int32_t split_cus;            // compute units (env C3R_SPLIT_CUS)
    int32_t *ev_overflow;         // bit 0: the event scratch; bit 1: a position covered by 32768 reads or more — its counts may not fit the 16-bit windows
    int32_t *last_row;            // [n_regions] atomicMax of the last SLOT (index into the position arrays) with a row
    const uint32_t *drop;         // mpileup depth cap: [n_regions][drop_words] bit per read = discarded in that region; null: none
    int32_t drop_words;
    int32_t splice;               // --enable_padding_in_splice_junction_regions: also produce skipmax[], materialise every tile
    int32_t *skipmax;             // [n_pos] max(#read starts, #read ends, #fwd ref-skips, #rev ref-skips) of the row
    const c3r_padins_t *padins;   // mpileup_compat = 1: insertions with pads, sorted by (read_idx, qpos); null / 0 for every CIGAR an aligner emits
    int32_t n_padins;
};
__device__ __forceinline__ bool giant_on(const ScanArgs &a, int nrec) {
    return (unsigned long long)(unsigned)nrec * (unsigned long long)(unsigned)a.split_cus >= *a.deep_recs;
}
// the table entry of the insertion of read r at query offset q, or null
__device__ __forceinline__ const c3r_padins_t *padins_find(const c3r_padins_t *tab, int n, uint32_t r, uint32_t q) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const c3r_padins_t &e = tab[mid];
        if (e.read_idx < r || (e.read_idx == r && e.qpos < q)) lo = mid + 1; else hi = mid;
    }
    return (lo < n && tab[lo].read_idx == r && tab[lo].qpos == q) ? &tab[lo] : nullptr;
}
// Diagnostics of the tile kernels — per-phase clocks (C3R_SCAN_DBG) and timing / traffic ablations (C3R_SCAN_ABL) — are compiled in only
// with -DC3R_SCAN_DIAG=1 (tools/build_variant.sh diag -DC3R_SCAN_DIAG=1): in the product build they fold away, which frees the scalar
// registers their pointer and flags held in a kernel that parks scalars in vector lanes as it is.
#ifndef C3R_SCAN_DIAG
#define C3R_SCAN_DIAG 0
#endif
#define C3R_DBG(a_) (C3R_SCAN_DIAG ? (a_).dbg : (unsigned long long *)nullptr)
#define C3R_ABL(a_) (C3R_SCAN_DIAG ? (a_).abl : 0)


// Inclusive prefix sum over the 64 lanes of a wavefront on the DPP cross-lane path: row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (the sequence LLVM's atomic optimizer emits for gfx9) — six v_add with a
// DPP operand.  The __shfl_up version was six ds_bpermute round trips through the LDS pipe (~100 cycles each, one after the other) per scan,
// four scans per span and wavefront.
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    return v;
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, *total = sum
// (NT > SCAN_THREADS — the deep-span workgroup of k_fused_deep: every thread takes part in the barriers, the first SCAN_THREADS threads hold the values)
template <int NT = SCAN_THREADS>
__device__ __forceinline__ int block_excl_scan(int v, int *wave_tot /* LDS [WAVES] */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    if (lane == 63 && (NT == SCAN_THREADS || wave < WAVES)) wave_tot[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { const int t = wave_tot[w]; if (w < wave) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// two ints per thread, ONE barrier: `slot` (LDS, 2 * WAVES ints) must belong to this call site alone — the trailing barrier of
// block_excl_scan only protects a shared scratch word against the next scan
template <int NT = SCAN_THREADS>
__device__ __forceinline__ int2 block_excl_scan2(int v0, int v1, int *slot, int *total0, int *total1) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = wave_incl_scan(v0), i1 = wave_incl_scan(v1);
    if (lane == 63 && (NT == SCAN_THREADS || wave < WAVES)) { slot[wave] = i0; slot[WAVES + wave] = i1; }
    __syncthreads();
    int b0 = 0, b1 = 0, t0 = 0, t1 = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { const int x = slot[w], y = slot[WAVES + w]; if (w < wave) { b0 += x; b1 += y; } t0 += x; t1 += y; }
    *total0 = t0; *total1 = t1;
    return make_int2(b0 + i0 - v0, b1 + i1 - v1);
}

__device__ __forceinline__ bool read_dropped(const uint32_t *drop, int words, int region, int r) {
    return drop != nullptr && ((drop[(size_t)region * words + (r >> 5)] >> (r & 31)) & 1u);
}
// samtools mpileup's read filter as the reference invokes it (src/create_tensor_pileup.py:436-451): --excl-flags (replaces the
// default mask), unmapped reads, --min-MQ, and — because the reference never passes -A / --count-orphans — "anomalous read
// pairs": reads with FLAG 0x1 (paired) set and 0x2 (proper pair) clear are skipped.
__host__ __device__ __forceinline__ bool flag_fails(unsigned flag, int excl) {
    return (flag & (unsigned)excl) || (flag & 4u) || ((flag & 1u) && !(flag & 2u));
}
__device__ __forceinline__ bool read_passes(const DevRead &r, int min_mq, int excl) {
    return !flag_fails(r.flag, excl) && r.mapq >= min_mq && r.end > r.pos;
}
// The reads [lo, hi) and the records [rlo, rhi) that can touch the positions [e0, e1): supersets by less than a bin on either side.
//   reads: everything up to the last read whose prefix-max end is <= e0 ends before e0; reads from the first one with pos >= e1 on
//   start after it.  records: pieces that start up to OP_CHOP - 1 before e0; indels anchored on e1 - 1 have rstart = e1.
__device__ __forceinline__ int4 span_ranges(const uint32_t *rec_off, const int4 *rtab, const BinGeo g, int e0, int e1) {
    int4 r;
    const int c0 = bin_edge(g, (long long)e0 + 1) >> CBIN_SHIFT;                                                   // coarse bin that starts at or before e0 + 1
    const int c1 = min(g.nbc, (bin_edge(g, (long long)e1 + (1 << BIN_SHIFT) - 1) + (1 << CBIN_SHIFT) - 1) >> CBIN_SHIFT);      // ... at or after e1
    r.x = rtab[c0].y;
    r.y = rtab[c1].x;
    r.z = (int)rec_off[bin_edge(g, (long long)e0 - (OP_CHOP - 1))];
    r.w = (int)rec_off[bin_edge(g, (long long)e1 + (1 << BIN_SHIFT))];
    return r;
}

__device__ __forceinline__ int base_code(const uint8_t *seq, uint64_t off, uint32_t q, uint32_t l_seq) {
    if (q >= l_seq) return 15;
    uint8_t b = seq[off + (q >> 1)];
    return (q & 1) ? (b & 0xf) : (b >> 4);
}
// BAM code -> 0..3 for A,C,G,T; -1 otherwise ('=', N and IUPAC codes contribute nothing,
// src/create_tensor_pileup.py:149,247-258)
__device__ __forceinline__ int acgt_index(int code) {
    return code == 1 ? 0 : code == 2 ? 1 : code == 4 ? 2 : code == 8 ? 3 : -1;
}
__device__ __forceinline__ int ref_index(uint8_t c) {   // evc_base_from: anything not ACGT counts as 'A'
    return c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 0;
}

// merged, sorted, disjoint half-open intervals: does [b,e) overlap any?
__device__ __forceinline__ bool intervals_overlap(const int32_t *iv, int n, int b, int e) {
    int lo = 0, hi = n;   // first interval with end > b
    while (lo < hi) { int mid = (lo + hi) >> 1; if (iv[2 * mid + 1] > b) hi = mid; else lo = mid + 1; }
    return lo < n && iv[2 * lo] < e;
}
__device__ __forceinline__ bool sorted_contains(const int32_t *a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] >= v) hi = mid; else lo = mid + 1; }
    return lo < n && a[lo] == v;
}

constexpr int AF_TAB = 8192;
constexpr int EV_HASH_MIN = 1024;      // indel events of a tile from which their alleles are counted through a hash table (tile_columns)
constexpr int DEEP_MIN_RECORDS = 2048; // records in a span's range from which it is left to k_fused_deep; ScanArgs::deep_min
enum WalkMode { ACCUM = 0, SCATTER = 1, FIRSTSEEN = 2 };
constexpr int FS_CAP = 32;     // positions per batch of the first-seen (tie-break) pass
// giant spans: tens of thousands of records in range (a locus at mpileup's depth cap: 270 k) — one CU walks those for a millisecond while the others idle
constexpr int SPLIT_MIN_RECORDS = 8192;    // ScanArgs::split_min
constexpr int GIANT_SLOTS = 256;           // giant spans of one scan that are split (the others are walked by their own workgroup)
constexpr int GIANT_SLICE = 4096;          // records per slice,
constexpr int GIANT_MAX_HELP = 32;         //   at most this many slices per span (longer slices beyond)
constexpr int GIANT_STRIDE = TILE * C3R_CH_PHASED + 2 * TILE + 16;     // int32 words per slot: cnt[TILE][C <= 30], maxdel[TILE], odd[TILE], GIANT_META..
constexpr int GIANT_META = TILE * C3R_CH_PHASED + 2 * TILE;            //   {events met, the span's first event in the pool, events reserved}
// slices of a giant span with n records (n >= 1) at `slice` records each: 1 .. GIANT_MAX_HELP; slice k of G over n items = [lo, hi), the G of them a
// partition of [0, n) (the last ones may be empty).  (__host__: tests/c/layout_check.hip)
__host__ __device__ __forceinline__ int giant_slices(int n, int slice) { const int g = n / slice + (n % slice != 0); return g < GIANT_MAX_HELP ? g : GIANT_MAX_HELP; }
__host__ __device__ __forceinline__ void giant_slice(int n, int k, int G, int &lo, int &hi) {
    const long long per = n / G + (n % G != 0), l = (long long)k * per, h = l + per;          // (n may be close to 2^31)
    lo = (int)(l < n ? l : n);
    hi = (int)(h < n ? h : n);
}
constexpr int GIANT_POOL_EVENTS = 12 << 20;                            // 40 bytes each (event + two table slots): 480 MB, allocated by a context that has met a giant span
constexpr int DEEP_EVG_CAP = 49152;      // events of a span that a workgroup's global buffer holds (ScanArgs::ev_wg, ::giant_ev): a span at mpileup's depth cap has ~34 k

struct TileLds {
    int32_t *cnt;      // [TILE][C]
    int32_t *cov;      // [TILE+1] coverage difference array, then inclusive-scanned in place
    int32_t *evoff;    // [TILE]
    int32_t *evfill;   // [TILE]
    int32_t *maxdel;   // [TILE]
    uint32_t *first;   // [FS_CAP][6] first-seen token index per class A,C,G,T,I,D of the positions with a tie at the top
    uint8_t *amb;      // [TILE] 0, or 1 + the position's row in `first` (current batch of the tie-break pass)
    uint8_t *odd;      // [TILE] phased mode: the column's haplotype channels need the ordered recompute (k_phase_recompute)
    EvRec *evq;        // [evq_cap] the tile's indel events in arrival order, captured by the ACCUM pass (null / 0: not captured)
    int32_t *evn;      // events met so far
    int32_t evq_cap;
    EvRec *evg;        // k_fused_deep: the workgroup's own event buffer in global memory — every event of the tile in arrival order, beside the LDS store
    int32_t evg_cap;   //   (a span whose events outgrow the LDS store is then bucketed from there, without a second walk over its records)
};

// Coverage of the tile from whole-read spans (incl. introns): header-only, one lane per read.  A read that starts at or before the tile's
// first position adds to cov[0]: at a deep locus that is most of a wavefront's reads on ONE LDS word (a 64-way serialised atomic), so those
// are counted by ballot and added once per wavefront.
__device__ __forceinline__ void cover_span(const TileLds &s, bool covers, int pos, int end, int t0, int t1) {
    const bool early = covers && pos <= t0;
    const unsigned long long m = __ballot(early);
    if (m && (int)(threadIdx.x & 63) == __builtin_ctzll(m)) atomicAdd(&s.cov[0], __popcll(m));
    if (covers && !early) atomicAdd(&s.cov[pos - t0], 1);
    if (covers && end < t1) atomicAdd(&s.cov[end - t0], -1);
}
template <int NT = SCAN_THREADS>
__device__ __forceinline__ void cover_reads(const ScanArgs &a, const TileLds &s, int lo, int hi, int t0, int t1, int region) {
    // (k_fused_deep: a span at mpileup's cap has 28 k reads in range — four headers per lane and round, their loads issued together)
    constexpr int U = NT > SCAN_THREADS ? 4 : 1;
    for (int base = lo; base < hi; base += NT * U) {          // (uniform trip count: the ballot sees whole wavefronts)
        DevRead rd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) rd[u] = a.reads[min(base + u * NT + (int)threadIdx.x, hi - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = base + u * NT + (int)threadIdx.x;
            const bool covers = r < hi && read_passes(rd[u], a.min_mq, a.excl_flags) && rd[u].end > t0 && rd[u].pos < t1 && !read_dropped(a.drop, a.drop_words, region, r);
            cover_span(s, covers, rd[u].pos, rd[u].end, t0, t1);
        }
    }
}

// ---- the tile's walk: one lane per record of the tile's range of the pile table.  A record is independent of every other one: two
// 16-byte loads, for an M piece (an I) one 16-byte load of packed bases, then the LDS atomics — two dependent cold misses per tile
// (records -> bases) whatever the tile holds.  (Round 2-3: segment headers -> scanned list in LDS -> binary search per lane -> op records
// -> bases; round 1: a 16-lane group per segment with prefix sums over its ops.)
//   M -> A/C/G/T or a/c/g/t (+ phased AP..TM), D -> '*' / '#', an I / a first D piece whose previous op is M, D, N (resp. M, N) ->
//   insertion / deletion on the preceding column (htslib semantics).
// the record's two 16-byte halves: ra = {rstart, w, naddr lo, naddr hi}, rb = {q, read_idx, nxt, aux}
__device__ __forceinline__ int nibble_at(uint64_t w0, uint64_t w1, int ni) {     // nibble ni of the 16 loaded bytes (high nibble of a byte first)
    const uint64_t w = ni < 16 ? w0 : w1;
    return (int)((w >> (8 * ((ni & 15) >> 1) + ((ni & 1) ? 0 : 4))) & 15u);
}

// The indel a record attaches to the column BEFORE its op (htslib: the next op is peeked at on the last position of the current one): the second
// half of a record's walk (walk_rec).
template <int C, int MODE>
__device__ __forceinline__ void walk_event(const ScanArgs &a, const TileLds &s, const int4 ra, const int4 rb, uint64_t w0, uint64_t w1, int t0, int t1,
                                           EvRec *ev) {
    const uint32_t w = (uint32_t)ra.y;
    const int op = (int)(w & 3u), prev = (int)((w >> 2) & 15u), hp = (int)((w >> 7) & 3u), avail = (int)((w >> 14) & 31u);
    const bool rev = (w & 64u) != 0;
    const int rstart = ra.x, r = rb.y;
    // indel attached to the column BEFORE the op (htslib: peek the next op at the last position of the current one).
    // I needs a ref-consuming predecessor (k_prep leaves no record otherwise), D needs an M or N predecessor (its first piece only: the
    // later pieces of a cut deletion have D before them).
    const bool is_ins = (op == C3R_CIG_I) && (prev == C3R_CIG_M || prev == C3R_CIG_D || prev == C3R_CIG_N);
    const bool is_del = (op == C3R_CIG_D) && (prev == C3R_CIG_M || prev == C3R_CIG_N || (w & PR_DEL_AFTER_INS));
    if (!(is_ins || is_del)) return;
    const int anchor = rstart - 1;
    if (anchor < t0 || anchor >= t1) return;
    const int pl = anchor - t0;
    const int ilen = (int)(uint32_t)rb.w;                              // inserted bases / length of the whole deletion
    const int odd = (int)((uint32_t)ra.z & 1u);
    if (MODE == FIRSTSEEN) {
        const int ai = s.amb[pl];
        if (ai) atomicMin(&s.first[(ai - 1) * 6 + (is_ins ? 4 : 5)], 2u * (uint32_t)r + 1u);
        return;
    }
    // the event record: allele key (the first <= 16 inserted base codes), channel of the max multiplicity
    EvRec e;
    e.key = 0;
    int fc = 15;
    if (is_ins) {
        // the first nk <= 16 inserted base codes, code j in bits 4j .. 4j + 3 — exactly the nibble-swapped, aligned form of the loaded bytes
        // (see the base loop above); codes SEQ does not hold (beyond `avail`) read as N
        const int nk = ilen < 16 ? ilen : 16;
        constexpr uint64_t LOWN = 0x0F0F0F0F0F0F0F0Full;
        uint64_t s0 = ((w0 & LOWN) << 4) | ((w0 >> 4) & LOWN);
        if (odd) s0 = (s0 >> 4) | ((((w1 & LOWN) << 4) | ((w1 >> 4) & LOWN)) << 60);
        const uint64_t m_nk = nk >= 16 ? ~0ull : ((1ull << (4 * nk)) - 1ull), m_av = avail >= 16 ? ~0ull : ((1ull << (4 * avail)) - 1ull);
        e.key = (s0 & m_nk & m_av) | (m_nk & ~m_av);
        fc = (int)(e.key & 15u);
    }
    // 'I' iff the first inserted char is one of "ACGTN*" (upper case => forward strand), src/create_tensor_pileup.py:227-232; a leading pad
    // prints as '*' on the forward strand (in the list) and as '#' on the reverse strand (not in it)
    const c3r_padins_t *pe = (is_ins && (w & PR_INS_PADS)) ? padins_find(a.padins, a.n_padins, (uint32_t)r, (uint32_t)rb.x) : nullptr;
    const bool up = !rev && (acgt_index(fc) >= 0 || fc == 15 || (pe && (pe->pad_mask & 1ull)));
    e.len = (uint32_t)ilen; e.read_idx = (uint32_t)r; e.qpos = (uint32_t)rb.x;
    e.pl = (uint16_t)pl; e.kind = (uint8_t)((rev ? 1 : 0) | (is_ins ? 2 : 0) | (pe ? 4 : 0));
    e.ch = (uint8_t)(is_ins ? (up ? C3R_I1 : C3R_i1) : (rev ? C3R_d1 : C3R_D1));
    if (MODE == ACCUM) {
        // an indel on a ref-skip column takes the haplotype of the previous token-list ENTRY (:183,189)
        if (C == C3R_CH_PHASED && prev == C3R_CIG_N) s.odd[pl] = 1;
        int ch;
        if (is_ins) ch = up ? C3R_I : C3R_i;
        else { ch = rev ? C3R_d : C3R_D; atomicMax(&s.maxdel[pl], ilen); }
        atomicAdd(&s.cnt[pl * C + ch], 1);
        // (a deletion shown right behind an insertion takes the haplotype of the previous token-list ENTRY — the insertion's, '0':
        // src/create_tensor_pileup.py:188-193)
        if (C == C3R_CH_PHASED && !(w & PR_DEL_AFTER_INS)) {
            if (hp == 1) atomicAdd(&s.cnt[pl * C + (is_ins ? C3R_IP : C3R_DP)], 1);
            else if (hp == 2) atomicAdd(&s.cnt[pl * C + (is_ins ? C3R_IM : C3R_DM)], 1);
        }
        if (s.evq_cap > 0) {
            const int at = atomicAdd(s.evn, 1);
            if (at < s.evq_cap) s.evq[at] = e;
            if (at < s.evg_cap) s.evg[at] = e;
        } else if (s.evq_cap < 0) {
            // k_deep_walk: a slice of a giant span — the span's buffer and its cursor are global and shared with the other slices' workgroups: one
            // returning atomic per wavefront for the lanes that are here
            const unsigned long long m = __ballot(1);
            const int lane = (int)(threadIdx.x & 63), lead = __builtin_ctzll(m);
            int base = 0;
            if (lane == lead) base = atomicAdd(s.evn, __popcll(m));
            base = __builtin_amdgcn_readlane(base, lead);
            const int at = base + __popcll(m & ((1ull << lane) - 1ull));
            if (at < s.evg_cap) s.evg[at] = e;
        }
    } else {  // SCATTER
        const int slot = s.evoff[pl] + atomicAdd(&s.evfill[pl], 1);
        ev[slot] = e;
    }
}

template <int C, int MODE>
__device__ __forceinline__ void walk_rec(const ScanArgs &a, const TileLds &s, const int4 ra, const int4 rb, uint64_t w0, uint64_t w1, int t0, int t1,
                                         EvRec *ev) {
    const uint32_t w = (uint32_t)ra.y;
    const int op = (int)(w & 3u), hp = (int)((w >> 7) & 3u), len = (int)((w >> 9) & 31u), avail = (int)((w >> 14) & 31u);
    const bool rev = (w & 64u) != 0;
    const int rstart = ra.x, r = rb.y;
    if (op == C3R_CIG_M) {
        if (MODE == SCATTER) return;
        const int b0 = max(rstart, t0), b1 = min(rstart + len, t1);
        if (b0 >= b1) return;
        const int off = b0 - rstart, nb = b1 - b0;
        // The 16 loaded bytes hold the piece's bases from nibble naddr + off on, BAM order (the first base of a byte in its HIGH nibble).
        // Swapping the nibbles of every byte and dropping the odd start nibble puts base u into bits 4u .. 4u + 3 of a 128-bit value:
        // one v_bfe_u32 with constant operands per base instead of a variable 64-bit shift and a select.  A base is one of A C G T iff
        // its code has exactly one bit set (1, 2, 4, 8), and then its channel is the bit's index: no chain of compares, and the only
        // thing left under a condition is the LDS atomic itself (the chain compiled to ~36 instructions and four branches per base).
        constexpr uint64_t LOWN = 0x0F0F0F0F0F0F0F0Full;
        uint64_t s0 = ((w0 & LOWN) << 4) | ((w0 >> 4) & LOWN), s1 = ((w1 & LOWN) << 4) | ((w1 >> 4) & LOWN);
        if (((uint32_t)ra.z + (uint32_t)off) & 1u) { s0 = (s0 >> 4) | (s1 << 60); s1 >>= 4; }
        const uint32_t q4[4] = {(uint32_t)s0, (uint32_t)(s0 >> 32), (uint32_t)s1, (uint32_t)(s1 >> 32)};
        const int lim = min(nb, max(avail - off, 0));                  // (a CIGAR may claim more bases than SEQ holds: those show as N)
        const int pl0 = b0 - t0;
        int32_t *const c0 = &s.cnt[pl0 * C + (rev ? 9 : 0)];
        int32_t *const ch = &s.cnt[pl0 * C + (hp == 2 ? (int)C3R_AM : (int)C3R_AP)];
#pragma unroll
        for (int u = 0; u < OP_CHOP; ++u) {
            const uint32_t nib = (q4[u >> 3] >> (4 * (u & 7))) & 15u;
            const bool in = u < lim;
            const bool acgt = nib != 0u && (nib & (nib - 1u)) == 0u;
            const int bi = __builtin_ctz(nib | 16u);                      // 0..3 for A C G T
            if (MODE == ACCUM) {
                if (in && acgt && !(C3R_ABL(a) & 4096)) atomicAdd(&c0[u * C + bi], 1);
                if (C == C3R_CH_PHASED) {
                    if (in && acgt && hp != 0) atomicAdd(&ch[u * C + bi], 1);
                    // '=' / IUPAC letters are ignored by the reference's token scan WITHOUT consuming their HP entry: every
                    // later read of the column is then phased with its predecessor's tag (:116-145)
                    if (in && !acgt && nib != 15u) s.odd[pl0 + u] = 1;
                }
            } else if (MODE == FIRSTSEEN) {
                if (in && acgt) {
                    const int ai = s.amb[pl0 + u];
                    if (ai) atomicMin(&s.first[(ai - 1) * 6 + bi], 2u * (uint32_t)r);
                }
            }
        }
        return;
    }
    if (op == C3R_CIG_D && MODE == ACCUM) {
        const int b0 = max(rstart, t0), b1 = min(rstart + len, t1);
        for (int p = b0; p < b1; ++p) atomicAdd(&s.cnt[(p - t0) * C + (rev ? C3R_HASH : C3R_STAR)], 1);
    }
    walk_event<C, MODE>(a, s, ra, rb, w0, w1, t0, t1, ev);
}

// All records of [rlo, rhi), WALK_UNR per lane and round: their loads (record, then bases) are issued together.
#ifndef C3R_WALK_UNR
#define C3R_WALK_UNR 1
#endif
constexpr int WALK_UNR = C3R_WALK_UNR;
#ifndef C3R_DEEP_WALK_UNR
#define C3R_DEEP_WALK_UNR 1
#endif
constexpr int DEEP_WALK_UNR = C3R_DEEP_WALK_UNR;        // k_fused_deep's first walk, records per lane and round (2 and 4 measured no faster at 500x: the LDS atomics bound it)
template <int C, int MODE, int NT = SCAN_THREADS, int WALK_UNR = c3r::WALK_UNR>
__device__ __forceinline__ void walk_records(const ScanArgs &a, const TileLds &s, int rlo, int rhi, int t0, int t1, int region, EvRec *ev) {
    const int tid = (int)threadIdx.x;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    for (int base = rlo; base < rhi; base += NT * WALK_UNR) {
        int4 ra[WALK_UNR], rb[WALK_UNR];
        bool have[WALK_UNR];
#pragma unroll
        for (int u = 0; u < WALK_UNR; ++u) {
            const int iu = base + u * NT + tid;
            have[u] = iu < rhi;
            const int4 *rec = reinterpret_cast<const int4 *>(a.recs + (have[u] ? iu : rhi - 1));      // (idle lanes re-read the last record)
            ra[u] = rec[0]; rb[u] = rec[1];
        }
        uint64_t w0[WALK_UNR], w1[WALK_UNR];
#pragma unroll
        for (int u = 0; u < WALK_UNR; ++u) {
            w0[u] = 0; w1[u] = 0;
            const uint32_t w = (uint32_t)ra[u].y;
            const int op = (int)(w & 3u), len = (int)((w >> 9) & 31u), avail = (int)((w >> 14) & 31u);
            // does the record touch the tile at all?  (the range is a superset by up to a bin on either side)
            const bool body = op != C3R_CIG_I && ra[u].x < t1 && ra[u].x + len > t0;
            const bool anchored = op != C3R_CIG_M && ra[u].x - 1 >= t0 && ra[u].x - 1 < t1;
            if (!(body || anchored)) have[u] = false;
            if (have[u] && a.drop && read_dropped(a.drop, a.drop_words, region, rb[u].y)) have[u] = false;
            int off = -1;                                    // first base to fetch, relative to the piece's first base
            if (have[u] && op == C3R_CIG_M && MODE != SCATTER) off = max(ra[u].x, t0) - ra[u].x;
            if (have[u] && op == C3R_CIG_I && MODE != FIRSTSEEN) off = 0;
            if (off >= 0 && off < avail && !(C3R_ABL(a) & 8192)) {
                const uint64_t na = ((uint64_t)(uint32_t)ra[u].z | ((uint64_t)(uint32_t)ra[u].w << 32)) + (uint64_t)off;
                u64x2 w;                                     // (the packed-base buffer is padded: the load may run past a read's last byte)
                __builtin_memcpy(&w, a.seq + (na >> 1), 16);
                w0[u] = w[0]; w1[u] = w[1];
            }
        }
#pragma unroll
        for (int u = 0; u < WALK_UNR; ++u)
            if (have[u]) walk_rec<C, MODE>(a, s, ra[u], rb[u], w0[u], w1[u], t0, t1, ev);
    }
}

// Two indel events are the same allele iff their mpileup texts are equal (Counter keys, src/create_tensor_pileup.py:179): same
// kind and length, same bases, and the same letter case = strand — except that '=' (BAM base code 0) has no case, so an
// insertion made of '=' only reads the same on both strands (and lands on channel i for both: '=' is not upper-case, :221-230).
__device__ __forceinline__ bool ev_equal(const ScanArgs &a, const EvRec &x, const EvRec &y) {
    if (((x.kind ^ y.kind) & 6) || x.len != y.len || x.key != y.key) return false;
    bool caseless = (x.kind & 2) && x.key == 0;
    if (x.kind & 4) {                  // pads inside both insertions: the same number of them at the same places; '*' and '#' tell the strands apart
        const c3r_padins_t *px = padins_find(a.padins, a.n_padins, x.read_idx, x.qpos), *py = padins_find(a.padins, a.n_padins, y.read_idx, y.qpos);
        if (!px || !py || px->total != py->total || px->pad_mask != py->pad_mask) return false;
        caseless = false;
    }
    if ((x.kind & 2) && x.len > 16) {
        const DevRead rx = a.reads[x.read_idx], ry = a.reads[y.read_idx];
        for (uint32_t j = 16; j < x.len; ++j) {
            const int cx = base_code(a.seq, rx.seq_off, x.qpos + j, rx.l_seq);
            if (cx != base_code(a.seq, ry.seq_off, y.qpos + j, ry.l_seq)) return false;
            caseless = caseless && cx == 0;
        }
    }
    return !((x.kind ^ y.kind) & 1) || caseless;
}

// equal alleles (ev_equal) hash alike: kind, length, the first 16 bases, and the strand unless the insertion may be the '='-only kind that has none
__device__ __forceinline__ uint64_t ev_hash(const EvRec &me) {
    const bool maybe_caseless = (me.kind & 2) && me.key == 0;
    uint64_t h = me.key * 0x9E3779B97F4A7C15ull + (uint64_t)me.len * 0xC2B2AE3D27D4EB4Full + (uint64_t)((me.kind & 6) | (maybe_caseless ? 0 : (me.kind & 1)));
    h ^= h >> 29;
    return h * 0xBF58476D1CE4E5B9ull;
}

// One thread per tile: the reads and records that can touch it (span_ranges: four table entries), thousands at a time instead of by
// one lane at the head of every tile workgroup.  Tiles that no read span covers are dropped from the work list.
__global__ __launch_bounds__(256) void k_tile_ranges(const ScanArgs a) {     // (256 threads: the list append below counts four wavefronts)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bool listed = false, pruned = false;
    if (t < a.n_tiles) {
        const TileGeo tg = a.geo[t];
        const int t0 = tg.p0, t1 = tg.p1;
        if (t1 > t0) {                        // (not a guard tile)
            const int4 r = span_ranges(a.rec_off, a.rtab, a.bins, t0, t1);
            if (r.x < r.y) {
                a.tile_rng[t] = r;
                listed = true;
                if (a.prune && r.z >= r.w) {
                    // intron-only tile.  A candidate is a position with aligned bases (depth > 0) and its window reaches 16
                    // positions to either side: this tile's rows matter only if an aligned base comes within 16 bp of it (a
                    // superset test: whole bins, records of every passing read)
                    const int4 q = span_ranges(a.rec_off, a.rtab, a.bins, t0 - C3R_FLANK - 1, t1 + C3R_FLANK);
                    if (q.z >= q.w) { listed = false; pruned = true; }
                }
            }
        }
    }
    // both lists are appended with ONE atomic per workgroup and list (the ~4 k per-wavefront atomics on one counter were most of
    // this kernel's time)
    __shared__ int s_cnt[2][4], s_base[2];
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const unsigned long long m0 = __ballot(listed), m1 = __ballot(pruned);
    if (lane == 0) { s_cnt[0][wave] = __popcll(m0); s_cnt[1][wave] = __popcll(m1); }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int w = (int)threadIdx.x, tot = s_cnt[w][0] + s_cnt[w][1] + s_cnt[w][2] + s_cnt[w][3];
        s_base[w] = tot ? atomicAdd(w ? a.n_tile_list2 : a.n_tile_list, tot) : 0;
    }
    __syncthreads();
    int b0 = s_base[0], b1 = s_base[1];
    for (int w = 0; w < wave; ++w) { b0 += s_cnt[0][w]; b1 += s_cnt[1][w]; }
    if (listed) a.tile_list[b0 + __popcll(m0 & ((1ull << lane) - 1ull))] = t;
    if (pruned) a.tile_list2[b1 + __popcll(m1 & ((1ull << lane) - 1ull))] = t;
}

// look-back words are read and written at device scope (every XCD has its own L2)
__device__ __forceinline__ unsigned long long lb_load(unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_store(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Decoupled look-back by one whole wavefront: entry b publishes its own counts (flag 1), sums the entries before it — 64 at a time, every
// lane one uncached load — down to the nearest entry that already holds an inclusive prefix (flag 2), then publishes its own
// inclusive prefix.  Returns the exclusive prefix (the 62 payload bits) to every lane.  A single thread walking the words one
// round trip at a time met ~1000 not-yet-inclusive predecessors whenever the chip's resident workgroups finished together.
__device__ __forceinline__ unsigned long long lb_lookback(unsigned long long *state, int b, unsigned long long mine) {
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
                w = i >= 0 ? lb_load(&state[i]) : (2ull << 62);            // (before the first entry: an inclusive prefix of zero)
                const unsigned long long not_ready = __ballot((w >> 62) == 0), incl = __ballot((w >> 62) == 2);
                // lanes are ordered nearest predecessor first: everything up to the nearest inclusive entry must be there
                const int first_incl = incl ? __ffsll((long long)incl) - 1 : 64;
                const unsigned long long need = first_incl >= 63 ? ~0ull : ((2ull << first_incl) - 1ull);
                if (!(not_ready & need)) { incl_found = incl != 0; w = (lane <= first_incl) ? (w & PAY) : 0ull; break; }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
            excl += w;
            if (incl_found) break;
        }
    }
    if (lane == 0) lb_store(&state[b], (2ull << 62) | ((excl + mine) & PAY));
    return excl;
}

// The same look-back with MAX for +: the largest payload among the entries before b (0: none).
__device__ __forceinline__ unsigned long long lb_lookback_max(unsigned long long *state, int b, unsigned long long mine) {
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
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(w, off, 64); w = o > w ? o : w; }
            excl = w > excl ? w : excl;
            if (incl_found) break;
        }
    }
    if (lane == 0) lb_store(&state[b], (2ull << 62) | ((excl > mine ? excl : mine) & PAY));
    return excl;
}

// The fused path's list of spans (k_fused_tiles): only spans that hold aligned bases, in ASCENDING order (= output order), with the
// read / record ranges of the span plus C3R_FLANK on either side.  Workgroups take blocks of 256 spans by ticket and place their
// listed spans behind those of the blocks before them (decoupled look-back over one word per block, as in k_fused_tiles).
// Everything a workgroup of k_fused_tiles needs to know about its span, in one 48-byte record indexed by LIST position: the tile
// kernel's per-span start-up was a chain of dependent loads (ticket -> tile_list -> geo -> region bounds, tile ranges) = several
// microseconds before the first useful instruction, 16 us of fixed latency per span in all (ablation: 0.26 of the kernel's 0.58 ms).
struct SpanRec { int32_t tile, p0, p1, region; int4 rng; int32_t reg_lo, reg_hi, pad0, pad1; };
static_assert(sizeof(SpanRec) == 48, "SpanRec must be 48 bytes");

__global__ __launch_bounds__(256) void k_tile_ranges_fused(const ScanArgs a, int32_t *ticket, unsigned long long *rstate, int nblk, const int2 *reg_bounds,
                                                           SpanRec *span_rec, int32_t *deep_list, int32_t *n_deep, unsigned long long *mstate) {
    __shared__ int s_b, s_base, s_cnt[4];
    __shared__ unsigned s_kmax[4], s_kexcl;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_b = atomicAdd(ticket, 1);
    __syncthreads();
    const int b = s_b;
    const int t = b * 256 + tid;
    bool listed = false;
    SpanRec rec;
    rec.tile = 0; rec.rng = make_int4(0, 0, 0, 0);
    // ---- spans follow the reads, not a grid.  The tiles of a region are a fixed grid of FUSE_IN positions, and an exon of 150 bp that straddles a
    // grid line costs two spans (chr20: 20.3 k spans for 12.4 k runs of aligned bases).  So a RUN of tiles — consecutive tiles of a region whose
    // own record ranges are not empty — is shifted as a whole to the first bin that holds one of its records: every tile of the run starts `shift`
    // positions later (0 .. FUSE_IN - 1), the run's spans still lie back to back, every covered position still lies in exactly one of them, and the
    // run's last tile often covers nothing any more (15.3 k spans).  The shift of a tile's run = that of the most recent run start at or before it:
    // a max-scan over (tile + 1) << 8 | shift of the run starts (block scan + decoupled look-back).
    TileGeo tg; tg.p0 = 0; tg.p1 = 0; tg.region = 0; tg.pad = 0;
    bool nonempty = false;
    unsigned key = 0;
    if (t < a.n_tiles) {
        tg = a.geo[t];
        if (tg.p1 > tg.p0) {
            const int4 own0 = span_ranges(a.rec_off, a.rtab, a.bins, tg.p0, tg.p1);
            nonempty = own0.z < own0.w;
            if (nonempty && !a.no_shift) {
                bool prev = false;
                if (t > 0) {
                    const TileGeo pg = a.geo[t - 1];
                    if (pg.region == tg.region && pg.p1 > pg.p0) { const int4 po = span_ranges(a.rec_off, a.rtab, a.bins, pg.p0, pg.p1); prev = po.z < po.w; }
                }
                if (!prev) {
                    int shift = 0;
                    const int bl = bin_edge(a.bins, (long long)tg.p0 - (OP_CHOP - 1)), bh = bin_edge(a.bins, (long long)tg.p1 - 1);
                    for (int bb = bl; bb <= bh && bb < a.bins.nb; ++bb)
                        if (a.rec_off[bb + 1] > a.rec_off[bb]) { shift = min(max(((bb + a.bins.base) << BIN_SHIFT) - tg.p0, 0), FUSE_IN - 1); break; }
                    key = ((unsigned)(t + 1) << 8) | (unsigned)shift;
                }
            }
        }
    }
    // inclusive max over the tiles up to mine
    unsigned kin = key;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const unsigned o = (unsigned)__shfl_up((int)kin, off, 64); if (lane >= off && o > kin) kin = o; }
    if (lane == 63) s_kmax[wave] = kin;
    __syncthreads();
    if (wave == 0) {
        const unsigned bm = max(max(s_kmax[0], s_kmax[1]), max(s_kmax[2], s_kmax[3]));
        const unsigned long long ex = lb_lookback_max(mstate, b, (unsigned long long)bm);
        if (lane == 0) s_kexcl = (unsigned)ex;
    }
    __syncthreads();
    unsigned run = max(kin, s_kexcl);
    for (int w = 0; w < wave; ++w) run = max(run, s_kmax[w]);
    if (nonempty) {
        const int shift = a.no_shift ? 0 : (int)(run & 255u);
        const int2 rb = reg_bounds[tg.region];
        const int t0 = tg.p0 + shift, t1 = min(t0 + FUSE_IN, rb.y);
        if (t1 > t0) {
            // a candidate needs aligned bases on its own position, min_cov of them: spans whose own range meets no record — or fewer records than
            // the coverage gate asks reads for (a read shows at most one piece on a position) — are not listed.  Real RNA-seq is full of
            // them: the one-to-three-read islands between the expressed loci
            const int4 own = span_ranges(a.rec_off, a.rtab, a.bins, t0, t1);
            if (own.z < own.w && own.w - own.z >= a.min_cov) {
                const int4 r = span_ranges(a.rec_off, a.rtab, a.bins, t0 - C3R_FLANK, t1 + C3R_FLANK);
                a.tile_rng[t] = r;
                listed = r.x < r.y;
                rec.tile = t; rec.p0 = t0; rec.p1 = t1; rec.region = tg.region; rec.rng = r;
                rec.reg_lo = rb.x; rec.reg_hi = rb.y; rec.pad0 = 0; rec.pad1 = 0;
            }
        }
    }
    const unsigned long long m = __ballot(listed);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (wave == 0) {
        const unsigned long long mine = (unsigned long long)(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
        const unsigned long long excl = lb_lookback(rstate, b, mine);
        if (lane == 0) {
            s_base = (int)excl;
            if (b == nblk - 1) *a.n_tile_list = (int32_t)(excl + mine);
        }
    }
    __syncthreads();
    int at = s_base;
    for (int w = 0; w < wave; ++w) at += s_cnt[w];
    if (listed) {
        const int at2 = at + __popcll(m & ((1ull << lane) - 1ull));
        a.tile_list[at2] = t;
        // a giant span takes one of the GIANT_SLOTS accumulators and lists its slices for k_deep_walk (no slot left: it is walked by its own workgroup)
        int gslot = 0;
        const int nrec = rec.rng.w - rec.rng.z;
        if (nrec >= a.split_min && nrec >= a.deep_min) {
            const int g = atomicAdd(a.n_giant, 1);          // (counted with or without a pool: the host allocates one after the first scan that met any)
            // (the pool's cursor: at most GIANT_SLOTS reservations of at most giant_pool events each — below 2^32, read as unsigned)
            const bool may = a.giant_acc && g < GIANT_SLOTS && nrec <= a.giant_pool;
            const int ev0 = may ? atomicAdd(a.n_giant_ev, nrec) : a.giant_pool;
            if (may && (unsigned)ev0 <= (unsigned)(a.giant_pool - nrec)) {
                gslot = g + 1;
                int32_t *meta = a.giant_acc + (size_t)g * GIANT_STRIDE + GIANT_META;
                meta[1] = ev0; meta[2] = nrec;
                const int G = giant_slices(nrec, a.split_slice);
                const int h0 = atomicAdd(a.n_help, G);
                for (int k = 0; k < G; ++k) a.help_list[h0 + k] = make_int4(at2, k, G, g);
            }
        }
        int4 *dst = reinterpret_cast<int4 *>(span_rec + at2);
        dst[0] = make_int4(rec.tile, rec.p0, rec.p1, rec.region); dst[1] = rec.rng; dst[2] = make_int4(rec.reg_lo, rec.reg_hi, gslot, 0);
    }
    // the deep spans' list positions, in any order, for k_fused_deep (one atomic per wavefront that holds any)
    const bool deep = listed && rec.rng.w - rec.rng.z >= a.deep_min;
    if (deep && a.giant_acc) atomicAdd(a.deep_recs, (unsigned long long)(rec.rng.w - rec.rng.z));
    const unsigned long long md = __ballot(deep);
    if (md) {
        int base = 0;
        if (lane == __builtin_ctzll(md)) base = atomicAdd(n_deep, __popcll(md));
        base = __shfl(base, __builtin_ctzll(md), 64);
        if (deep) deep_list[base + __popcll(md & ((1ull << lane) - 1ull))] = at + __popcll(m & ((1ull << lane) - 1ull));
    }
}

// The tile kernels run over the compact tile list with a FIXED grid (LIST_GRID workgroups, each taking every LIST_GRID-th list
// entry): the list's length lives on the device, and a grid of one workgroup per tile of the scan — 250 k for chr20, of which
// 40 k are listed — spent ~0.09 ms per kernel dispatching workgroups that left at once.
constexpr int LIST_GRID = 8192;

// LDS of one tile workgroup.  Indel events of a tile stay in LDS when there are at most EV_LDS of them (a 20x ONT tile holds ~100): the
// first walk captures them as it meets them, and they are bucketed by position without a second walk over the records; deeper tiles
// walk again and bump-allocate global scratch.  (30 channels: the accumulators leave no room at four workgroups per CU; a store of 96 events fits and
// was measured: no faster on the MAS-Seq contig, whose spans hold a dozen events — the second walk is not what its time is.)
template <int C, int EVL = (C == C3R_CH ? 192 : 0)>
struct alignas(16) TileMem {
    static constexpr int EV_LDS = EVL;
    typedef typename std::conditional<(EVL > 256), uint16_t, uint8_t>::type evord_t;
    int32_t cnt[TILE * C];
    int32_t cov[TILE + 1];
    int32_t evoff[TILE];
    int32_t evfill[TILE];
    int32_t maxdel[TILE];
    uint32_t first[FS_CAP * 6];
    uint8_t amb[TILE];
    uint8_t odd[TILE];
    int misc[8];           // [0] events captured by the first walk, [2..5] block-scan scratch
    int scan_slot[2][2 * WAVES];                // block_excl_scan2: one slot per call site
    unsigned long long rowmask[WAVES];          // k_fused_tiles: which positions hold a pileup row, one bit each
    unsigned long long ambmask[WAVES];          // k_fused_tiles: positions with a tie at the top of the allele counts (first-seen pass)
    unsigned long long evbase;
    alignas(16) EvRec ev[EV_LDS > 0 ? EV_LDS : 1];
    evord_t evord[EV_LDS > 0 ? EV_LDS : 4];     // the captured events' indices, bucketed by position
    // the deep kernel's event store (EV_LDS > 256): per bucket slot the event's allele hash and, for the first event of an allele, its multiplicity
    alignas(16) uint32_t esig[EV_LDS > 256 ? EV_LDS : 1];
    int32_t ecnt[EV_LDS > 256 ? EV_LDS : 1];
    evord_t elead[EV_LDS > 256 ? EV_LDS : 4];   // per bucket slot: the slot that stands for its allele
};
struct TileOut { bool is_row, cand; int depth, cov; };     // cov: the token slots the position needs if it becomes a candidate — an upper bound of the reads that show
                                                           // something other than the reference base or a ref-skip there (tile_tokens)

// The columns of the positions [t0, t1) (at most TILE of them, thread tid <-> position t0 + tid) from the reads [lo, hi) and the
// records [slo, shi) of the pile table: accumulators in LDS, indel alleles, the per-position gates (src/create_tensor_pileup.py:259-299, :536-556),
// the reference-channel overwrite.  On return M.cnt holds the finished columns, M.odd the phased columns that need the ordered
// recompute, and the thread its position's verdict.  Positions below pmin hold no rows (they lie before the region).
// Used by the column-store kernel (k_scan_tiles) and by the fused kernel (k_fused_tiles), whose "tile" is a window-complete span.
// FUSED (k_fused_tiles): the row mask of the window rule is published with the ambiguity votes (one barrier instead of two).
// The first slot in [b, i] whose event is slot i's allele (i itself when none before it is): the slots' 32-bit allele hashes four at a time — one
// 16-byte LDS read per step, the next one issued before this one is looked at —, `same(j)` (ev_equal) only where the hashes agree.
template <class Same>
__device__ __forceinline__ int first_of_allele(const uint32_t *sig, int b, int i, Same &&same) {
    const uint32_t mine = sig[i];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    int j = b & ~3;
    u32x4 cur = *reinterpret_cast<const u32x4 *>(sig + j);
#pragma unroll 1
    for (; j < i; j += 4) {
        const u32x4 nxt = *reinterpret_cast<const u32x4 *>(sig + min(j + 4, i & ~3));
        uint32_t hit = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) hit |= (j + q >= b && j + q < i && cur[q] == mine) ? (1u << q) : 0u;
        while (hit) {
            const int q = __builtin_ctz(hit);
            hit &= hit - 1u;
            if (same(j + q)) return j + q;
        }
        cur = nxt;
    }
    return i;
}

template <int C, int NT, int EVL>
__device__ __forceinline__ void tile_zero(TileMem<C, EVL> &M) {
    const int tid = threadIdx.x;
    for (int i = tid; i < TILE * C; i += NT) M.cnt[i] = 0;
    if (NT == SCAN_THREADS || tid < TILE) {
        M.cov[tid] = 0; if (tid == 0) M.cov[TILE] = 0;
        M.evfill[tid] = 0; M.maxdel[tid] = 0; M.amb[tid] = 0; M.odd[tid] = 0;
        if (tid == 0) M.misc[0] = 0;
    }
}
// NT: threads of the workgroup.  SCAN_THREADS everywhere but in k_fused_deep, whose workgroups of DEEP_THREADS keep thread tid <-> position t0 + tid for the
// first TILE threads (`pos_thread`) and put all sixteen wavefronts on everything that goes record by record, read by read or event by event.
template <int C, bool FUSED = false, int NT = SCAN_THREADS, int EVL = TileMem<C>::EV_LDS>
__device__ __forceinline__ TileOut tile_columns(const ScanArgs &a, TileMem<C, EVL> &M, int t0, int t1, int pmin, int region, int lo, int hi, int slo, int shi,
                                                int cand_lo, int cand_hi, const int giant = 0) {
    constexpr int EV_LDS = EVL;
    const int tid = threadIdx.x;
    const bool pos_thread = NT == SCAN_THREADS || tid < TILE;
    // giant (k_fused_deep, 1 + slot): the span's records were walked by k_deep_walk — counts, deletion lengths, order flags and events lie in its slot
    const bool split = NT > SCAN_THREADS && giant > 0;
    // (k_fused_deep: workgroup blockIdx.x owns a.ev_wg_cap event slots of a.ev_wg)
    const int32_t *const gmeta = split ? a.giant_acc + (size_t)(giant - 1) * GIANT_STRIDE + GIANT_META : nullptr;
    const int g_ev0 = split ? gmeta[1] : 0;
    EvRec *const evg = split ? a.giant_ev + g_ev0 : (NT > SCAN_THREADS && a.ev_wg) ? a.ev_wg + (size_t)blockIdx.x * (size_t)a.ev_wg_cap : nullptr;
    TileLds s{M.cnt, M.cov, M.evoff, M.evfill, M.maxdel, M.first, M.amb, M.odd, M.ev, &M.misc[0], EV_LDS, evg, split ? gmeta[2] : evg ? a.ev_wg_cap : 0};
    unsigned long long tprev = C3R_DBG(a) ? wall_clock64() : 0ull;
#define C3R_PHASE(K) do { if (C3R_DBG(a) && tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&C3R_DBG(a)[K], now_ - tprev); tprev = now_; } } while (0)
    tile_zero<C, NT, EVL>(M);
    // loads that nothing before the gates depends on leave now: this position's reference base, and the first round of read headers
    // (their round trip runs beside the walk's record and base loads instead of before them)
    const int p = t0 + tid;
    const int rp = p - a.ref_beg0;
    const uint8_t rb = (pos_thread && rp >= 0 && rp < a.ref_len) ? a.ref[rp] : (uint8_t)'N';
    const int r0 = lo + tid;
    int rd_pos = 0, rd_end = INT32_MIN;                       // read lo + tid as the coverage sees it (end = INT32_MIN: not a covering read)
    if (r0 < hi && !(C3R_ABL(a) & 4)) {
        const DevRead rd0 = a.reads[r0];
        rd_pos = rd0.pos;
        if (read_passes(rd0, a.min_mq, a.excl_flags) && !read_dropped(a.drop, a.drop_words, region, r0)) rd_end = rd0.end;
    }
    __syncthreads();

    C3R_PHASE(0);
    if (split) {
        const int32_t *acc = a.giant_acc + (size_t)(giant - 1) * GIANT_STRIDE;
        for (int i = tid; i < TILE * C; i += NT) M.cnt[i] = acc[i];
        if (tid < TILE) { M.maxdel[tid] = acc[TILE * C3R_CH_PHASED + tid]; M.odd[tid] = (uint8_t)(acc[TILE * C3R_CH_PHASED + TILE + tid] != 0); }
    } else
    if (!(C3R_ABL(a) & 1)) walk_records<C, ACCUM, NT, (NT > SCAN_THREADS ? DEEP_WALK_UNR : c3r::WALK_UNR)>(a, s, slo, shi, t0, t1, region, nullptr);
    if (!(C3R_ABL(a) & 4)) {
        cover_span(s, rd_end > t0 && rd_pos < t1, rd_pos, rd_end, t0, t1);
        cover_reads<NT>(a, s, lo + NT, hi, t0, t1, region);
    }
    __syncthreads();
    C3R_PHASE(1);

    // coverage: inclusive scan of the difference array; indel events: exclusive scan of per-position counts (one scan, one barrier)
    int *wave_tot = &M.misc[2];
    const int my_cov_d = pos_thread ? M.cov[tid] : 0;
    const int32_t *row = &M.cnt[(pos_thread ? tid : 0) * C];
    const int nev = pos_thread ? row[C3R_I] + row[C3R_i] + row[C3R_D] + row[C3R_d] : 0;
    int tot, ev_total;
    const int2 ex = block_excl_scan2<NT>(my_cov_d, nev, M.scan_slot[0], &tot, &ev_total);
    const int my_cov = (C3R_ABL(a) & 256) ? 1 : ex.x + my_cov_d;
    if (pos_thread && my_cov >= 32768) atomicOr(a.ev_overflow, 2);          // (every count of a column is a count of reads that cover it: below this, the windows fit int16)
    if (pos_thread) M.evoff[tid] = ex.y;
    C3R_PHASE(2);
    if (C3R_ABL(a) & 524288) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
    if (ev_total > 0 && !(C3R_ABL(a) & 2)) {
        // the tile's indel events, bucketed by position (counting sort through evoff / evfill), then the max multiplicity of one
        // allele per (position, channel): I1 / i1 / D1 / d1
        __syncthreads();                       // (evoff of every position is in place)
        if (split && a.giant_tab) {
            // k_deep_alleles has counted the span's alleles: per position a region of 2 n slots behind those of the positions before it, a used slot =
            // {1 + the allele's first event, which of I1 / i1 / D1 / d1 it counts for << 30 | how many events show it}.  The slot's position: the last
            // one whose region starts at or before it (evoff is non-decreasing; positions without events have empty regions)
            uint2 *tab = a.giant_tab + 2 * (size_t)g_ev0;
#pragma unroll 4
            for (int i = tid; i < 2 * ev_total; i += NT) {
                const uint2 v = tab[i];
                if (v.x) {
                    tab[i] = make_uint2(0u, 0u);
                    int pl = 0;
#pragma unroll
                    for (int st = TILE / 2; st > 0; st >>= 1) if (2 * M.evoff[pl + st] <= i) pl += st;
                    constexpr int CH1[4] = {C3R_I1, C3R_i1, C3R_D1, C3R_d1};
                    atomicMax(&M.cnt[pl * C + CH1[v.y >> 30]], (int)(v.y & 0x3fffffffu));
                }
            }
        } else {
        if (split && ev_total <= EV_LDS) {
            for (int e = tid; e < ev_total; e += NT) M.ev[e] = evg[e];
            __syncthreads();
        }
        if (ev_total <= EV_LDS) {
            // the usual case: the first walk has captured every event (ev_total of them, in arrival order); bucket their indices
            for (int e = (C3R_ABL(a) & 262144) ? ev_total : tid; e < ev_total; e += NT) {
                const int pl = M.ev[e].pl;
                const int slot = M.evoff[pl] + atomicAdd(&M.evfill[pl], 1);
                M.evord[slot] = (typename TileMem<C, EVL>::evord_t)e;
                if (EV_LDS > 256 && !(C3R_ABL(a) & 32768)) { M.esig[slot] = (uint32_t)(ev_hash(M.ev[e]) >> 32); M.ecnt[slot] = 0; }
            }
            __syncthreads();
            C3R_PHASE(5);
            if (EV_LDS > 256) {
                // The deep kernel's store holds thousands of events and a true indel at 500x puts hundreds of them into ONE bucket, whose threads would
                // each compare with all of it.  Instead an allele is stood for by its FIRST event in the bucket: an event looks for the first slot before
                // its own that holds its allele (the 32-bit hashes first, ev_equal — an equivalence — on a match), adds one to that slot's count, and
                // reads the count back after the barrier.  Members of a frequent allele stop after a few slots, the stragglers only compare hashes.
                // (loops kept rolled and ev_equal at ONE call site: a span runs this code once, and its size is instruction-cache misses)
#pragma unroll 1
                for (int i = (C3R_ABL(a) & 131072) ? ev_total : tid; i < ev_total; i += NT) {
                    const EvRec me = M.ev[M.evord[i]];
                    const int L = (C3R_ABL(a) & 16384) ? i : first_of_allele(M.esig, M.evoff[me.pl], i, [&](int j) { return ev_equal(a, me, M.ev[M.evord[j]]); });
                    atomicAdd(&M.ecnt[L], 1);
                    M.elead[i] = (typename TileMem<C, EVL>::evord_t)L;
                }
                __syncthreads();
#pragma unroll 1
                for (int i = (C3R_ABL(a) & 65536) ? ev_total : tid; i < ev_total; i += NT) {
                    const EvRec &me = M.ev[M.evord[i]];
                    atomicMax(&M.cnt[(int)me.pl * C + me.ch], M.ecnt[M.elead[i]]);
                }
            } else
            for (int e = tid; e < ev_total; e += NT) {
                const EvRec me = M.ev[e];
                const int pl = me.pl;
                const int b = M.evoff[pl];
                const int32_t *rw = &M.cnt[pl * C];
                const int n = rw[C3R_I] + rw[C3R_i] + rw[C3R_D] + rw[C3R_d];
                int eq = 0;
                for (int j = 0; j < n; ++j) eq += ev_equal(a, me, M.ev[M.evord[b + j]]) ? 1 : 0;
                atomicMax(&M.cnt[pl * C + me.ch], eq);
            }
        } else {
            // A deep tile (thousands of events: coverage in the thousands) counts its alleles through a hash table in the global scratch,
            // one region of 2 n slots per position behind the tile's events: an allele's FIRST event claims a slot (CAS) and stands for it,
            // every event of the same allele (ev_equal against the slot's representative — an equivalence: kind, length, bases, pads and, but
            // for the '='-only insertions, strand) adds one.  The all-pairs count below is quadratic in the depth: at mpileup's cap of 8000
            // reads a span spent 14 of its 22 ms there (profiles/r5/deep_locus_phases.txt); it stays for the shallow tiles, where it is cheaper.
            // (k_fused_deep) up to LEAD_CAP events the alleles are counted as in the LDS store above — first event of an allele in its bucket stands for
            // it — with the events themselves in the global buckets and their hashes / counts where the (outgrown) LDS store lay
            constexpr int LEAD_CAP = EV_LDS > 256 ? (int)(sizeof(M.ev) / 8) : 0;
            const bool lead = EV_LDS > 256 && ev_total <= LEAD_CAP;
            const bool hashed = !lead && ev_total > EV_HASH_MIN;
            const unsigned long long ev_units = (unsigned long long)((ev_total + 15) & ~15);
            const unsigned long long tab_units = hashed ? ((unsigned long long)ev_total * 2ull * sizeof(uint2) + sizeof(EvRec) - 1) / sizeof(EvRec) : 0ull;
            auto events = [&](EvRec *ev) __attribute__((always_inline)) {
                if (s.evg && ev_total <= s.evg_cap) {
                    // every event of the tile lies in the workgroup's buffer in arrival order (walk_event): into the buckets from there
                    __threadfence_block();
#pragma unroll 1
                    for (int e = tid; e < ev_total; e += NT) {
                        // (the buffer is re-used span after span by this CU: read past its L1, which may hold the last span's lines)
                        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(s.evg + e);
                        unsigned long long w3[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q) w3[q] = __hip_atomic_load(src + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        EvRec me;
                        __builtin_memcpy(&me, w3, sizeof me);
                        ev[M.evoff[me.pl] + atomicAdd(&M.evfill[me.pl], 1)] = me;
                    }
                } else walk_records<C, SCATTER, NT>(a, s, slo, shi, t0, t1, region, ev);
                uint2 *tab = reinterpret_cast<uint2 *>(ev + ev_units);
                if (hashed) for (int i = tid; i < 2 * ev_total; i += NT) tab[i] = make_uint2(0xffffffffu, 0u);
                __threadfence_block();
                __syncthreads();
                if (lead) {
                    // per bucket slot: the event's hash and its count where the events of the LDS store lay, the slot that stands for its allele where that
                    // store's own tables lay (evord, esig, ecnt, elead: contiguous, 12 bytes per event of the store)
                    uint32_t *gsig = reinterpret_cast<uint32_t *>(M.ev), *gcnt = gsig + LEAD_CAP, *glead = reinterpret_cast<uint32_t *>(M.evord);
                    typedef TileMem<C, EVL> TM;
                    static_assert(offsetof(TM, elead) + sizeof(M.elead) - offsetof(TM, evord) >= (EV_LDS > 256 ? LEAD_CAP : 0) * sizeof(uint32_t) &&
                                  sizeof(M.ev) >= 2 * (EV_LDS > 256 ? LEAD_CAP : 0) * sizeof(uint32_t), "the leader words alias the LDS event store");
#pragma unroll 1
                    for (int i = tid; i < ev_total; i += NT) { gsig[i] = (uint32_t)(ev_hash(ev[i]) >> 32); gcnt[i] = 0; }
                    __syncthreads();
#pragma unroll 1
                    for (int i = tid; i < ev_total; i += NT) {
                        const EvRec me = ev[i];
                        const int L = first_of_allele(gsig, M.evoff[me.pl], i, [&](int j) { return ev_equal(a, me, ev[j]); });
                        atomicAdd(&gcnt[L], 1u);
                        glead[i] = (uint32_t)L;
                    }
                    __syncthreads();
#pragma unroll 1
                    for (int i = tid; i < ev_total; i += NT) {
                        const EvRec me = ev[i];
                        atomicMax(&M.cnt[(int)me.pl * C + me.ch], (int)gcnt[glead[i]]);
                    }
                    return;
                }
                if (!hashed) {
                    for (int e = tid; e < ev_total; e += NT) {
                        const EvRec me = ev[e];
                        const int pl = me.pl;
                        const int b = M.evoff[pl];
                        const int32_t *rw = &M.cnt[pl * C];
                        const int n = rw[C3R_I] + rw[C3R_i] + rw[C3R_D] + rw[C3R_d];
                        int eq = 0;
                        for (int j = 0; j < n; ++j) eq += ev_equal(a, me, ev[b + j]) ? 1 : 0;
                        atomicMax(&M.cnt[pl * C + me.ch], eq);
                    }
                    return;
                }
                auto slot_of = [&](const EvRec &me, int n) -> uint32_t { return (uint32_t)(ev_hash(me) >> 33) % (uint32_t)(2 * n); };
                for (int pass = 0; pass < 2; ++pass) {
                    // pass 0: claim or join the allele's slot; pass 1: every event reads its allele's count
                    for (int e = tid; e < ev_total; e += NT) {
                        const EvRec me = ev[e];
                        const int pl = me.pl;
                        const int b = M.evoff[pl];
                        const int32_t *rw = &M.cnt[pl * C];
                        const int n = rw[C3R_I] + rw[C3R_i] + rw[C3R_D] + rw[C3R_d];
                        uint2 *reg = tab + 2 * b;
                        uint32_t h = slot_of(me, n);
                        for (int probe = 0; probe < 2 * n; ++probe) {
                            uint32_t rep;
                            if (pass == 0) rep = atomicCAS(&reg[h].x, 0xffffffffu, (uint32_t)e);
                            else rep = __hip_atomic_load(&reg[h].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const bool mine = pass == 0 ? (rep == 0xffffffffu) : (rep == (uint32_t)e);
                            if (mine || (rep != 0xffffffffu && ev_equal(a, me, ev[rep]))) {
                                if (pass == 0) atomicAdd(&reg[h].y, 1u);
                                else atomicMax(&M.cnt[pl * C + me.ch], (int)__hip_atomic_load(&reg[h].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                break;
                            }
                            h = h + 1u == (uint32_t)(2 * n) ? 0u : h + 1u;
                        }
                    }
                    __threadfence_block();
                    __syncthreads();
                }
            };
            if (tid == 0) M.evbase = atomicAdd(a.ev_cursor, ev_units + tab_units);
            __syncthreads();
            const unsigned long long evb = M.evbase;
            if (evb + ev_units + tab_units > a.ev_cap) {
                // cannot happen with the host's sizing (c3r_pileup_scan_regions); if it ever does, no write leaves the buffer and the
                // scan call fails instead of corrupting device memory
                if (tid == 0) atomicOr(a.ev_overflow, 1);
            } else {
                events(a.ev + evb);
            }
        }
        }
        __syncthreads();
    }

    C3R_PHASE(3);
    // ---- per-position gates (src/create_tensor_pileup.py:259-299, :536-556)
    bool is_row = false, cand = false, ambiguous = false;
    int depth = 0, refi = 0, ntok_bound = 0;
    int cls[6] = {0, 0, 0, 0, 0, 0};
    bool gates_ok = false;
    if (pos_thread && p >= pmin && p < t1 && my_cov > 0) {
        is_row = !a.has_lbed || intervals_overlap(a.lbed, a.n_lbed, p, p + 1);
    }
    if (is_row) {
        int32_t *c = &M.cnt[tid * C];                    // (is_row: a position thread)
        const int up = c[C3R_A] + c[C3R_C] + c[C3R_G] + c[C3R_T];
        const int lw = c[C3R_a] + c[C3R_c] + c[C3R_g] + c[C3R_t];
        depth = up + lw + c[C3R_STAR] + c[C3R_HASH];
        const bool ref_acgt = (rb == 'A' || rb == 'C' || rb == 'G' || rb == 'T');
        refi = ref_index(rb);
        cls[0] = c[C3R_A] + c[C3R_a]; cls[1] = c[C3R_C] + c[C3R_c]; cls[2] = c[C3R_G] + c[C3R_g]; cls[3] = c[C3R_T] + c[C3R_t];
        cls[4] = c[C3R_I] + c[C3R_i]; cls[5] = c[C3R_D] + c[C3R_d];
        // every read that gets a token (tile_tokens) is counted at least once here: a non-reference base, a '*' / '#', an indel
        ntok_bound = cls[4] + cls[5] + c[C3R_STAR] + c[C3R_HASH];
        for (int x = 0; x < 4; ++x) if (x != refi) ntok_bound += cls[x];
        const bool may_be_cand = p >= cand_lo && p < cand_hi;       // (the fused kernel decides candidates for its inner span only)
        const double denom = depth > 0 ? (double)depth : 1.0;
        bool pass = (C3R_ABL(a) & 1024) != 0;
        if (!pass && depth < AF_TAB) {
            // (six float64 divisions per position otherwise: ~70 double-rate instructions in a kernel whose time follows its instruction count)
            const uint32_t th = a.af_tab[depth];
            const int ts = (int)(th & 0xffffu), ti = (int)(th >> 16);
            for (int x = 0; x < 4; ++x)
                if (x != refi && cls[x] >= ts) pass = true;
            if (cls[4] >= ti || cls[5] >= ti) pass = true;
        } else if (!pass) {
            for (int x = 0; x < 4; ++x)
                if (x != refi && cls[x] > 0 && (double)cls[x] / denom >= a.snp_af) pass = true;
            if (!pass && cls[4] > 0 && (double)cls[4] / denom >= a.indel_af) pass = true;
            if (!pass && cls[5] > 0 && (double)cls[5] / denom >= a.indel_af) pass = true;
        }
        if (depth > 0 && (a.snp_af == 0.0 || a.indel_af == 0.0)) pass = true;
        if (!pass) {
            // pileup_list[0][0] != reference_base: top class by count, ties broken by first occurrence
            int m = 0;
            for (int x = 0; x < 6; ++x) m = max(m, cls[x]);
            if (m > 0) {
                if (cls[refi] < m) pass = true;
                else {
                    for (int x = 0; x < 6; ++x) if (x != refi && cls[x] == m) ambiguous = true;
                }
            }
        }
        bool site_ok;
        if (a.genotyping) site_ok = sorted_contains(a.sites, a.n_sites, p + 1);
        else {
            gates_ok = may_be_cand && ref_acgt && depth >= a.min_cov &&
                       (!a.has_cbed || intervals_overlap(a.cbed, a.n_cbed, p, p + M.maxdel[tid] + 2));
            site_ok = gates_ok && pass;
            if (!gates_ok) ambiguous = false;
        }
        if (a.genotyping) ambiguous = false;
        cand = site_ok && may_be_cand;
        // reference-base channels are overwritten with minus the strand totals (:296-297).  BASE2INDEX is keyed by channel
        // NAME, so an IUPAC 'D' (or an 'I') in the reference lands on the D / d (I / i) channels; other letters count as 'A'
        // (the reference raises KeyError there)
        const int ch_up = rb == 'D' ? (int)C3R_D : rb == 'I' ? (int)C3R_I : refi;
        const int ch_lo = rb == 'D' ? (int)C3R_d : rb == 'I' ? (int)C3R_i : 9 + refi;
        c[ch_up] = -up;
        c[ch_lo] = -lw;
    }
    if (C3R_ABL(a) & 512) ambiguous = false;
    bool any_amb;
    if (FUSED) {
        // the window rule's row mask travels with the votes: 33 contiguous rows = 33 set bits (k_fused_tiles)
        const unsigned long long rm = __ballot(is_row), am = __ballot(ambiguous);
        if ((tid & 63) == 0 && pos_thread) { M.rowmask[tid >> 6] = rm; M.ambmask[tid >> 6] = am; }
        __syncthreads();
        any_amb = (M.ambmask[0] | M.ambmask[1] | M.ambmask[2] | M.ambmask[3]) != 0ull;
    } else any_amb = __syncthreads_or(ambiguous ? 1 : 0) != 0;
    C3R_PHASE(4);

    if (any_amb) {
        // "top allele != reference" with a tie at the top: the reference's stable sort keeps the class seen first in the column.  A
        // third walk records, for the tied positions only (FS_CAP at a time), the first read that shows each class
        int n_amb;
        const int arank = block_excl_scan<NT>(ambiguous ? 1 : 0, wave_tot, &n_amb);
        for (int base = 0; base < n_amb; base += FS_CAP) {
            const bool mine = ambiguous && arank >= base && arank < base + FS_CAP;
            if (pos_thread) M.amb[tid] = mine ? (uint8_t)(1 + arank - base) : (uint8_t)0;
            for (int i = tid; i < FS_CAP * 6; i += NT) M.first[i] = 0xffffffffu;
            __syncthreads();
            walk_records<C, FIRSTSEEN, NT>(a, s, slo, shi, t0, t1, region, nullptr);
            __syncthreads();
            if (mine) {
                int m = 0;
                for (int x = 0; x < 6; ++x) m = max(m, cls[x]);
                const uint32_t *fs = &M.first[(arank - base) * 6];
                const uint32_t fr = fs[refi];
                bool top_ne_ref = false;
                for (int x = 0; x < 6; ++x)
                    if (x != refi && cls[x] == m && fs[x] < fr) top_ne_ref = true;
                cand = gates_ok && top_ne_ref;
            }
            __syncthreads();
        }
    }
    C3R_PHASE(6);
    if (C3R_DBG(a) && tid == 0) { atomicAdd(&C3R_DBG(a)[14], (unsigned long long)(shi - slo)); atomicAdd(&C3R_DBG(a)[15], 1ull); atomicAdd(&C3R_DBG(a)[13], (unsigned long long)(hi - lo)); atomicAdd(&C3R_DBG(a)[12], (unsigned long long)ev_total); }
#undef C3R_PHASE
    TileOut o;
    o.is_row = is_row; o.cand = cand; o.depth = depth; o.cov = ntok_bound;
    return o;
}

// One tile of the column store (the path of head/tail calling, splice padding, genotyping mode and c3r_get_columns): the tile's
// columns, depth, covering reads and flags go to HBM; selection, compaction and the window gather are separate kernels.
template <int C>
__device__ __forceinline__ void scan_tile(const ScanArgs &a, const int tile, TileMem<C> &M) {
    const int tid = threadIdx.x;
    const TileGeo tg = a.geo[tile];
    const int t0 = tg.p0, t1 = tg.p1;
    const int slot0 = tile * TILE;        // index of the tile's first position in cols / depth / ncov / flags
    const int4 rng = a.tile_rng[tile];
    if ((C3R_ABL(a) & 16) && rng.z >= rng.w) return;   // ablation: skip intron-only tiles
    if ((C3R_ABL(a) & 32) && rng.z < rng.w) return;    // ablation: skip tiles with aligned bases
    const int lo = rng.x, hi = rng.y;       // reads whose span can overlap [t0,t1)
    const int slo = rng.z, shi = rng.w;     // records of the pile table that can touch it
    if (slo >= shi) {
        // intron-only tile: rows exist (ref-skip columns) but every count is zero.  Only the flags are written; the
        // gather treats the columns of such a tile as zeros (tile_cols stays 0).
        TileLds s{M.cnt, M.cov, M.evoff, M.evfill, M.maxdel, M.first, M.amb, M.odd, nullptr, nullptr, 0, nullptr, 0};
        M.cov[tid] = 0; if (tid == 0) M.cov[TILE] = 0;
        __syncthreads();
        cover_reads(a, s, lo, hi, t0, t1, tg.region);
        __syncthreads();
        int tot;
        const int d = M.cov[tid];
        const int cov = block_excl_scan(d, &M.misc[2], &tot) + d;
        const int p = t0 + tid;
        bool is_row = false;
        if (p < t1 && cov > 0) is_row = !a.has_lbed || intervals_overlap(a.lbed, a.n_lbed, p, p + 1);
        if (p < t1) {
            const int gi = slot0 + tid;
            // depth / ncov of a position are read for candidates (rescale, token count) and, in splice-padding mode, for window
            // slots.  An intron-only tile has candidates only in genotyping mode, so in the plain mode its 2 KB of depth / ncov
            // stay unwritten (230 k such tiles per chr20 pass = 0.47 GB of stores nobody reads)
            if (a.genotyping || a.splice) { a.depth[gi] = 0; a.ncov[gi] = is_row ? cov : 0; }
            bool cand = false;
            if (is_row && a.genotyping) cand = sorted_contains(a.sites, a.n_sites, p + 1);
            a.flags[gi] = (uint8_t)((is_row ? 1 : 0) | (cand ? 2 : 0));
        }
        if (a.head_tail) {
            int mx = is_row ? slot0 + tid : -1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
            // racy pre-check is safe (the value only grows) and keeps ~10^5 tiles from serialising on one L2 atomic
            if ((tid & 63) == 0 && mx > *(volatile int32_t *)&a.last_row[tg.region]) atomicMax(&a.last_row[tg.region], mx);
        }
        if (a.splice) {
            // splice padding writes into low-depth columns in place: they must exist
            int32_t *gcol = a.cols + (size_t)slot0 * C;
            for (int i = tid; i < (t1 - t0) * C; i += SCAN_THREADS) gcol[i] = 0;
            if (tid == 0) a.tile_cols[tile] = 1;
        }
        return;
    }
    const TileOut o = tile_columns<C>(a, M, t0, t1, t0, tg.region, lo, hi, slo, shi, t0, t1);
    // ---- write the tile's columns, coalesced
    const int npos = t1 - t0;
    int32_t *gcol = a.cols + (size_t)slot0 * C;
    if (!(C3R_ABL(a) & 8))
    for (int i = tid; i < npos * C; i += SCAN_THREADS) gcol[i] = M.cnt[i];
    if (tid == 0) a.tile_cols[tile] = 1;
    // ---- per-position metadata
    if (t0 + tid < t1) {
        const int gi = slot0 + tid;
        a.depth[gi] = o.depth;
        a.ncov[gi] = o.is_row ? o.cov : 0;
        a.flags[gi] = (uint8_t)((o.is_row ? 1 : 0) | (o.cand ? 2 : 0) | ((o.is_row && M.odd[tid]) ? 8 : 0));
    }
    if (a.head_tail) {
        int mx = o.is_row ? slot0 + tid : -1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
        if ((tid & 63) == 0 && mx > *(volatile int32_t *)&a.last_row[tg.region]) atomicMax(&a.last_row[tg.region], mx);
    }
}

// (register budget = the occupancy LDS allows: five workgroups per CU at 18 channels, four at 30)
template <int C>
#ifndef C3R_SCAN_OCC30
#define C3R_SCAN_OCC30 4
#endif
__global__ __launch_bounds__(SCAN_THREADS, (C == C3R_CH ? 5 : C3R_SCAN_OCC30)) void k_scan_tiles(const ScanArgs a) {
    __shared__ TileMem<C> M;
    const int n = *a.n_tile_list;
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        scan_tile<C>(a, a.tile_list[b], M);
        __syncthreads();                  // the next tile re-uses the LDS arrays
    }
}

// -------------------------------------------------------------------------------------------------
// Splice-junction padding, part 1 (src/create_tensor_pileup.py:151-178, :532-534): per row
//   max_skip_count = max(#'$', #'^', #'<', #'>')
// i.e. reads ending here, reads starting here, reverse / forward reads showing a ref-skip here.  Starts, ends and per-strand read
// coverage come from the read spans, per-strand ALIGNED coverage from the M / D pieces of the pile table (two loads per record, no
// bases), and ref-skips = covering - aligned.  Same tile list as k_scan_tiles; only launched in splice-padding mode.
__global__ __launch_bounds__(SCAN_THREADS) void k_skip_counts(const ScanArgs a) {
    __shared__ int32_t s_cov[2][TILE + 1];     // reads covering, by strand (difference arrays)
    __shared__ int32_t s_seg[2][TILE + 1];     // aligned segments covering, by strand
    __shared__ int32_t s_start[TILE], s_end[TILE];
    __shared__ int s_w[WAVES];
    const int tid = threadIdx.x;
    const int n_list = *a.n_tile_list;
    for (int lb = blockIdx.x; lb < n_list; lb += gridDim.x) {
    const int tile = a.tile_list[lb];
    const TileGeo tg = a.geo[tile];
    const int t0 = tg.p0, t1 = tg.p1;
    const int slot0 = tile * TILE;        // index of the tile's first position in cols / depth / ncov / flags
    const int4 rng = a.tile_rng[tile];
    __syncthreads();
    for (int i = tid; i < 2 * (TILE + 1); i += SCAN_THREADS) { (&s_cov[0][0])[i] = 0; (&s_seg[0][0])[i] = 0; }
    s_start[tid] = 0; s_end[tid] = 0;
    __syncthreads();
    for (int r = rng.x + tid; r < rng.y; r += SCAN_THREADS) {
        const DevRead rd = a.reads[r];
        if (!read_passes(rd, a.min_mq, a.excl_flags) || rd.end <= t0 || rd.pos >= t1) continue;
        if (read_dropped(a.drop, a.drop_words, tg.region, r)) continue;
        const int st = (rd.flag & 16) ? 1 : 0;
        atomicAdd(&s_cov[st][max(rd.pos, t0) - t0], 1);
        if (rd.end < t1) atomicAdd(&s_cov[st][rd.end - t0], -1);
        if (rd.pos >= t0) atomicAdd(&s_start[rd.pos - t0], 1);
        if (rd.end - 1 < t1) atomicAdd(&s_end[rd.end - 1 - t0], 1);
    }
    for (int g = rng.z + tid; g < rng.w; g += SCAN_THREADS) {
        const int4 *rec = reinterpret_cast<const int4 *>(a.recs + g);
        const int4 ra = rec[0];
        const uint32_t w = (uint32_t)ra.y;
        const int len = (int)((w >> 9) & 31u);
        if ((w & 3u) == C3R_CIG_I || ra.x >= t1 || ra.x + len <= t0) continue;
        if (a.drop && read_dropped(a.drop, a.drop_words, tg.region, rec[1].y)) continue;
        const int st = (w & 64u) ? 1 : 0;
        atomicAdd(&s_seg[st][max(ra.x, t0) - t0], 1);
        if (ra.x + len < t1) atomicAdd(&s_seg[st][ra.x + len - t0], -1);
    }
    __syncthreads();
    int tot, v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = (k < 2) ? s_cov[k][tid] : s_seg[k - 2][tid];
        v[k] = block_excl_scan(d, s_w, &tot) + d;
    }
    const int p = t0 + tid;
    if (p < t1) {
        const int gi = slot0 + tid;
        int m = max(s_start[tid], s_end[tid]);
        m = max(m, max(v[0] - v[2], v[1] - v[3]));
        a.skipmax[gi] = (a.flags[gi] & 1) ? m : 0;
    }
    }
}

// -------------------------------------------------------------------------------------------------
// Window selection (src/create_tensor_pileup.py:512-516,565-568,613-637): a candidate is emitted iff
// the 33 positions centre-16..centre+16 are contiguous rows; with head_tail the ring is pre-filled
// with zero columns after every gap (left side always OK) and the stream end is flushed with 16
// zero columns (right side OK only when the run reaches the last row of the stream).
// heavy: null, or tile_cols — outside genotyping mode only tiles that hold aligned bases (tile_cols = 1) can have candidates, and
// 93 % of an RNA contig's covered tiles are intron-only: the three kernels below leave their blocks early there instead of
// reading 64 MB of flags each.
__global__ __launch_bounds__(TILE) void k_select(uint8_t *flags, int n_pos, const TileGeo *geo, int head_tail, const int32_t *last_row, const uint8_t *heavy,
                                                 const int32_t *tile_list, const int32_t *n_tile_list) {
    static_assert(TILE == 256, "one 256-thread block per tile");
    const int n_list = *n_tile_list;
    for (int lb = blockIdx.x; lb < n_list; lb += gridDim.x) {
        const int tile = tile_list[lb];
        if (heavy && !heavy[tile]) continue;
        const int i = tile * TILE + (int)threadIdx.x;
        if (i >= n_pos) continue;
        const uint8_t f = flags[i];
        if (!(f & 2)) continue;
        bool ok = true;
        if (!head_tail) {
            if (i - C3R_FLANK < 0 || i + C3R_FLANK >= n_pos) ok = false;
            else for (int q = i - C3R_FLANK; q <= i + C3R_FLANK; ++q) if (!(flags[q] & 1)) { ok = false; break; }
        } else {
            const int last = last_row[geo[tile].region];
            const int hi = min(i + C3R_FLANK, last);
            for (int q = i + 1; q <= hi; ++q) if (!(flags[q] & 1)) { ok = false; break; }
        }
        if (ok) flags[i] = f | 4;
    }
}

// ordered stream compaction of emitted positions: count -> scan -> write
constexpr int CMP_THREADS = 256;
constexpr int CMP_ITEMS = 4;                      // positions per thread
constexpr int CMP_BLOCK = CMP_THREADS * CMP_ITEMS;

__device__ __forceinline__ bool cmp_block_empty(const uint8_t *heavy, int n_pos) {
    if (!heavy) return false;
    const int t0 = blockIdx.x * (CMP_BLOCK / TILE), nt = (n_pos + TILE - 1) / TILE;
    bool any = false;
#pragma unroll
    for (int k = 0; k < CMP_BLOCK / TILE; ++k) if (t0 + k < nt && heavy[t0 + k]) any = true;
    return !any;
}

__global__ __launch_bounds__(CMP_THREADS) void k_compact_count(const uint8_t *flags, int n_pos, int32_t *block_cnt, const uint8_t *heavy) {
    static_assert(CMP_BLOCK % TILE == 0, "a compaction block covers whole tiles");
    if (cmp_block_empty(heavy, n_pos)) { if (threadIdx.x == 0) block_cnt[blockIdx.x] = 0; return; }
    const int base = blockIdx.x * CMP_BLOCK + threadIdx.x * CMP_ITEMS;
    int c = 0;
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; ++j) { const int i = base + j; if (i < n_pos && (flags[i] & 4)) ++c; }
    __shared__ int wsum[CMP_THREADS / 64];
    int v = c;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < CMP_THREADS / 64; ++w) t += wsum[w]; block_cnt[blockIdx.x] = t; }
}

// single-block exclusive scan of n ints (in place); total written to *total
__global__ __launch_bounds__(1024) void k_excl_scan(int32_t *data, int n, int32_t *total) {
    // 8 consecutive items per thread and round (8192 per round): the carry chain between rounds is the serial part, so fewer,
    // fatter rounds (the 202 k token counts of a chr20 pass: 25 rounds instead of 198)
    constexpr int IT = 8;
    __shared__ int wtot[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = 0; base < n; base += 1024 * IT) {
        const int i0 = base + threadIdx.x * IT;
        int v[IT], sum = 0;
#pragma unroll
        for (int k = 0; k < IT; ++k) { v[k] = (i0 + k < n) ? data[i0 + k] : 0; sum += v[k]; }
        const int incl = wave_incl_scan(sum);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int wb = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int t = wtot[w]; if (w < wave) wb += t; tot += t; }
        const int carry = carry_s;
        int run = carry + wb + incl - sum;
#pragma unroll
        for (int k = 0; k < IT; ++k) { if (i0 + k < n) data[i0 + k] = run; run += v[k]; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

// The same scan in three short launches for long inputs (the 200 k token counts, the 63 k block counts of a chr20 pass): the
// single-block version pays ~5 us of barriers per 8192 items, one round after the other — 0.25 ms per pass.
//   k_scan_local: each block scans its 8192 items in place (exclusive) and leaves their sum in tops[block]
//   k_excl_scan:  the <= few hundred sums, one block
//   k_scan_add:   block offsets added back
constexpr int SCAN_IT = 8, SCAN_BLK = 1024 * SCAN_IT;
__global__ __launch_bounds__(1024) void k_scan_local(int32_t *data, int n, int32_t *tops) {
    __shared__ int wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_IT;
    int v[SCAN_IT], sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IT; ++k) { v[k] = (i0 + k < n) ? data[i0 + k] : 0; sum += v[k]; }
    const int incl = wave_incl_scan(sum);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int wb = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { const int t = wtot[w]; if (w < wave) wb += t; tot += t; }
    int run = wb + incl - sum;
#pragma unroll
    for (int k = 0; k < SCAN_IT; ++k) { if (i0 + k < n) data[i0 + k] = run; run += v[k]; }
    if (threadIdx.x == 0) tops[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_scan_add(int32_t *data, int n, const int32_t *tops) {
    const int off = tops[blockIdx.x];
    const int i0 = blockIdx.x * SCAN_BLK + threadIdx.x * SCAN_IT;
#pragma unroll
    for (int k = 0; k < SCAN_IT; ++k) if (i0 + k < n) data[i0 + k] += off;
}

// tile_cand[tile] = {index of the tile's first candidate in cand_idx, number of candidates}: a wavefront's 64 x 4 positions are
// exactly one tile (zeroed beforehand: blocks without aligned bases leave early)
__global__ __launch_bounds__(CMP_THREADS) void k_compact_write(const uint8_t *flags, int n_pos, const int32_t *block_off,
                                                                 int32_t *cand_idx /* region-relative index */, const uint8_t *heavy, int2 *tile_cand) {
    static_assert(64 * CMP_ITEMS == TILE, "one wavefront per tile");
    __shared__ int wsum[CMP_THREADS / 64];
    if (cmp_block_empty(heavy, n_pos)) return;
    const int base = blockIdx.x * CMP_BLOCK + threadIdx.x * CMP_ITEMS;
    int c = 0;
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; ++j) { const int i = base + j; if (i < n_pos && (flags[i] & 4)) ++c; }
    const int incl = wave_incl_scan(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = block_off[blockIdx.x] + incl - c;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (lane == 0 && base < n_pos) tile_cand[base / TILE] = make_int2(off, wsum[wave]);
#pragma unroll
    for (int j = 0; j < CMP_ITEMS; ++j) { const int i = base + j; if (i < n_pos && (flags[i] & 4)) cand_idx[off++] = i; }
}

// -------------------------------------------------------------------------------------------------
// Window gather: one wavefront per candidate.  Copies 33 consecutive columns (zero-filled outside the
// run when head_tail), applies the A5 rescale (clair3_rna/utils.py:88-92: tensor / (depth/144) in
// float64, truncated toward zero by the int32 store) and writes the site record.
struct GatherArgs {
    const int32_t *cols; const int32_t *depth; const int32_t *ncov; const uint8_t *flags; const uint8_t *tile_cols;
    const int32_t *cand_idx; int32_t n_cand; int32_t n_pos; const TileGeo *geo;
    const uint8_t *ref; int32_t ref_beg0; int32_t ref_len;
    int32_t head_tail; const int32_t *last_row;
    int32_t rescale; int32_t max_depth;   // 144
    void *tensors;         // [n][33][C]: int16 when x16 (the resident, rescaled windows), else int32 (c3r_get_tensors' raw export)
    int32_t x16;
    int32_t *raw;          // [n][33][C] un-rescaled copy (may be null)
    const int32_t *skipmax; // splice-padding mode only
    c3r_site_t *sites;     // [n] (may be null)
    int32_t *tok_cnt;      // [n] (may be null): number of tokens of the centre column
};

template <int C>
__device__ __forceinline__ void gather_window(const GatherArgs &g, int w, int ci, int lane, const int32_t *zcol = nullptr) {
    int lo_valid = ci - C3R_FLANK, hi_valid = ci + C3R_FLANK;
    if (g.head_tail) {
        int q = ci;
        while (q - 1 >= 0 && q - 1 >= ci - C3R_FLANK && (g.flags[q - 1] & 1)) --q;
        lo_valid = q;
        hi_valid = min(ci + C3R_FLANK, g.last_row[g.geo[ci / TILE].region]);
    }
    const int pc = g.geo[ci / TILE].p0 + (ci % TILE);      // genome position of the centre; slots +-16 are positions +-16
    const int dep = g.depth[ci];
    const bool scale = g.rescale && dep > 0 && (double)dep > (double)g.max_depth * 1.5;
    const double sf = (double)dep / (double)g.max_depth;
    int32_t *out = (int32_t *)g.tensors + (size_t)w * C3R_WINDOW * C;
    int16_t *out16 = (int16_t *)g.tensors + (size_t)w * C3R_WINDOW * C;
    int32_t *raw = g.raw ? g.raw + (size_t)w * C3R_WINDOW * C : nullptr;
    const int first = ci - C3R_FLANK;
    // two channels (8 bytes) per lane and round: C is even, so a pair never straddles two columns, and every window, column and
    // tensor starts on an 8-byte boundary (C * 4 = 72 / 120 bytes per column).  All rounds' loads are issued before the first store;
    // a window touches at most two tiles, whose "columns exist" bytes are read once
    static_assert(C % 2 == 0, "channel pairs");
    typedef int int2v __attribute__((ext_vector_type(2)));
    constexpr int NP = C3R_WINDOW * C / 2, NIT = (NP + 63) / 64;
    const int tl = max(first, 0) / TILE, th = min(first + 2 * C3R_FLANK, g.n_pos - 1) / TILE;
    const bool have_l = g.tile_cols[tl] != 0, have_h = g.tile_cols[th] != 0;
    int2v v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = 2 * (lane + 64 * it);
        const int q = first + i / C, ch = i % C;
        v[it] = int2v{0, 0};
        if (i < 2 * NP) {
            if (q >= lo_valid && q <= hi_valid) { if (q / TILE == tl ? have_l : have_h) v[it] = *(const int2v *)(g.cols + (size_t)q * C + ch); }
            else if (zcol && q < lo_valid) { v[it][0] = zcol[ch]; v[it][1] = zcol[ch + 1]; }      // the run's shared pre-fill column (splice padding edits it)
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = 2 * (lane + 64 * it);
        if (i < 2 * NP) {
            if (raw) *(int2v *)(raw + i) = v[it];
            if (scale) { v[it][0] = (int32_t)((double)v[it][0] / sf); v[it][1] = (int32_t)((double)v[it][1] / sf); }
            if (g.x16) *(int *)(out16 + i) = (v[it][0] & 0xffff) | (v[it][1] << 16);
            else *(int2v *)(out + i) = v[it];
        }
    }
    if (g.sites) {
        c3r_site_t *s = &g.sites[w];
        if (lane < C3R_WINDOW) {
            const int rp = pc - C3R_FLANK + lane - g.ref_beg0;
            s->ref33[lane] = (rp >= 0 && rp < g.ref_len) ? (char)g.ref[rp] : 'A';
        } else if (lane < C3R_WINDOW + 3) {
            s->ref33[lane] = 0;
        }
        if (lane == 0) { s->pos = pc + 1; s->depth = dep; s->n_tok = g.ncov[ci]; s->tok_off = 0; }
    }
    if (g.tok_cnt && lane == 0) g.tok_cnt[w] = g.ncov[ci];
}

template <int C>
__global__ __launch_bounds__(256) void k_gather(const GatherArgs g) {
    const int w = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));      // (wave-uniform: scalar loads)
    const int lane = threadIdx.x & 63;
    if (w >= g.n_cand) return;
    gather_window<C>(g, w, g.cand_idx[w], lane);
}

// -------------------------------------------------------------------------------------------------
// Splice-junction padding, part 2 (src/create_tensor_pileup.py:573-593, :611).  The reference edits the ring's column
// lists IN PLACE while it emits candidates in position order, so a window sees the edits made for every earlier
// candidate within 32 bp, and `del depth_dict[center]` makes an emitted centre count as depth 0 for later windows.
// Emitted candidates less than 33 bp apart therefore form a chain that has to be processed in order; chains are
// independent.  One wavefront per chain (the wavefront of the chain's first candidate; the others exit), lane = window
// slot: decide, edit the columns in HBM, then gather that candidate's window before moving on.
__device__ __forceinline__ int pad_channel(uint8_t up, bool lower) {   // BASE2INDEX of a reference letter
    switch (up) {
        case 'A': return lower ? C3R_a : C3R_A;  case 'C': return lower ? C3R_c : C3R_C;
        case 'G': return lower ? C3R_g : C3R_G;  case 'T': return lower ? C3R_t : C3R_T;
        case 'I': return lower ? C3R_i : C3R_I;  case 'D': return lower ? C3R_d : C3R_D;   // IUPAC letters that are keys too
        default: return -1;                                                                // (the reference raises KeyError)
    }
}

template <int C>
__global__ __launch_bounds__(256) void k_splice_gather(const GatherArgs g) {
    // with head/tail calling the ring is pre-filled after every gap with 33 references to ONE zero list
    // ([[0]*C]*33, :467,:514): padding a slot left of the run start edits that shared list, and every such slot of
    // this and later windows of the run shows it.  One copy per wavefront, keyed by the run start.
    __shared__ int32_t s_z[4][C];
    const int w0 = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    int32_t *zc = s_z[threadIdx.x >> 6];
    if (w0 >= g.n_cand) return;
    if (w0 > 0 && g.cand_idx[w0] - g.cand_idx[w0 - 1] <= 2 * C3R_FLANK) return;     // not the head of a chain
    int32_t *cols = const_cast<int32_t *>(g.cols);
    int z_run = INT32_MIN;
    for (int w = w0; w < g.n_cand; ++w) {
        const int ci = g.cand_idx[w];
        if (w > w0 && ci - g.cand_idx[w - 1] > 2 * C3R_FLANK) break;
        int lo_valid = ci - C3R_FLANK, hi_valid = ci + C3R_FLANK;
        if (g.head_tail) {
            int qq = ci;
            while (qq - 1 >= 0 && qq - 1 >= ci - C3R_FLANK && (g.flags[qq - 1] & 1)) --qq;
            lo_valid = qq;
            hi_valid = min(ci + C3R_FLANK, g.last_row[g.geo[ci / TILE].region]);
            if (lo_valid > ci - C3R_FLANK && lo_valid != z_run) {       // first window of a new run: fresh zero list
                for (int i = lane; i < C; i += 64) zc[i] = 0;
                z_run = lo_valid;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        const int q = ci - C3R_FLANK + lane;                   // lane < 33: window slot
        const bool in_win = lane < C3R_WINDOW;
        const bool real = in_win && q >= lo_valid && q <= hi_valid;      // the ring slot holds this position's own column
        // depth_dict / max_skip_count_dict are keyed by POSITION and survive ring resets: a slot left of the run start
        // (head/tail mode: it shows the shared pre-fill column) still finds the depth of that position if an EARLIER run
        // had a row there (gap < 16 bp), and counts in the window maxima
        const bool has_row = in_win && q >= 0 && q <= hi_valid && (real || (g.flags[q] & 1));
        // entries of already-emitted centres (all of them left of ci) have been deleted from depth_dict
        const bool deleted = has_row && q < ci && (g.flags[q] & 4);
        const int cur = (has_row && !deleted) ? g.depth[q] : 0;
        int md = (has_row && !deleted) ? cur : INT32_MIN;
        int ms = has_row ? g.skipmax[q] : INT32_MIN;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { md = max(md, __shfl_xor(md, off, 64)); ms = max(ms, __shfl_xor(ms, off, 64)); }
        const int cdepth = g.depth[ci];
        // candidates closer than 16 rows to the end of the stream leave through the tail flush, which does not pad (:613-637)
        const bool pads = hi_valid == ci + C3R_FLANK;
        if (pads && (double)ms / (double)md > 0.2) {
            const int pc = g.geo[ci / TILE].p0 + (ci % TILE);
            const int rp = pc - g.ref_beg0;
            const uint8_t rc = (rp >= 0 && rp < g.ref_len) ? g.ref[rp] : (uint8_t)'N';
            const int cu = pad_channel(rc, false), cl = pad_channel(rc, true);
            int sf = cu >= 0 ? cols[(size_t)ci * C + cu] : 0, sr = cl >= 0 ? cols[(size_t)ci * C + cl] : 0;
            sf = sf < 0 ? -sf : sf; sr = sr < 0 ? -sr : sr;
            const double fpct = (sf + sr > 0) ? (double)sf / (double)(sf + sr) : 0.0;
            const double rpct = 1 - fpct;
            if (in_win && lane != C3R_FLANK && (double)cur < (double)cdepth * 0.2) {
                int rq = pc + (q - ci) - g.ref_beg0;
                if (rq < 0) rq += g.ref_len;                  // Python negative index (slots left of the contig start)
                const uint8_t rb = (rq >= 0 && rq < g.ref_len) ? g.ref[rq] : (uint8_t)'N';
                const int u = pad_channel(rb, false), l = pad_channel(rb, true);
                if (u >= 0) {
                    const int vf = -1 * (int)((double)cdepth * fpct), vr = -1 * (int)((double)cdepth * rpct);
                    if (real) { cols[(size_t)q * C + u] = vf; cols[(size_t)q * C + l] = vr; }
                    else if (q < lo_valid) { zc[u] = vf; zc[l] = vr; }     // (every such lane writes the same two values)
                }
            }
            __threadfence();
            __builtin_amdgcn_wave_barrier();
        }
        gather_window<C>(g, w, ci, lane, g.head_tail ? zc : nullptr);
        __threadfence();
        __builtin_amdgcn_wave_barrier();
    }
}

// What read r (header rd) shows at reference position p: base code (0..15 BAM nibble, 16 = deleted base, 17 = ref-skip)
// and the indel attached to the column (+len insertion with its query offset, -len deletion, 0 none).  The read's aligned
// segment that holds p is found through the read-ordered segment list (a handful per read), then only its ops are walked.
struct TokenAt { uint8_t base; int32_t indel; uint32_t qpos; };
__device__ __forceinline__ TokenAt token_at(const DevRead &rd, int r, int p, const DevSeg *rsegs, const uint32_t *rseg_first,
                                            const uint32_t *cigar, const uint8_t *seq) {
    TokenAt tk; tk.base = 17; tk.indel = 0; tk.qpos = 0;
    const uint32_t s0 = rseg_first[r], s1 = rseg_first[r + 1];
    for (uint32_t si = s0; si < s1; ++si) {
        const DevSeg sg = rsegs[si];
        if (sg.ext_start > p) break;
        if (p >= sg.end) continue;
        int x = sg.pos, y = (int)sg.qstart;
        if (p < sg.pos) {          // p == pos-1: I/D right after an N, attached to the last intron column
            const uint32_t c0 = cigar[sg.cig_off];
            const int op0 = (int)(c0 & 15u), len0 = (int)(c0 >> 4);
            if (op0 == C3R_CIG_I) { tk.indel = len0; tk.qpos = (uint32_t)y; }
            else if (op0 == C3R_CIG_D) tk.indel = -len0;
            break;
        }
        // ops five at a time: the loads of a batch are independent of each other (one load latency per batch instead of one per
        // op); the fifth is the op the fourth may have to peek at
        const uint32_t nc = sg.n_cig;
        bool done = false;
        for (uint32_t k0 = 0; k0 < nc && !done; k0 += 4) {
            uint32_t cc[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) cc[i] = cigar[sg.cig_off + min(k0 + (uint32_t)i, nc - 1)];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t k = k0 + (uint32_t)i;
                if (done || k >= nc) break;
                const int op = (int)(cc[i] & 15u), len = (int)(cc[i] >> 4);
                if (op == C3R_CIG_M || op == C3R_CIG_D) {
                    if (p < x + len) {
                        if (op == C3R_CIG_M) tk.base = (uint8_t)base_code(seq, rd.seq_off, (uint32_t)(y + (p - x)), rd.l_seq);
                        else tk.base = 16;
                        if (p == x + len - 1 && k + 1 < nc) {
                            const int op2 = (int)(cc[i + 1] & 15u), len2 = (int)(cc[i + 1] >> 4);
                            if (op2 == C3R_CIG_I) { tk.indel = len2; tk.qpos = (uint32_t)(y + (op == C3R_CIG_M ? len : 0)); }
                            else if (op2 == C3R_CIG_D && op != C3R_CIG_D) tk.indel = -len2;
                        }
                        done = true;
                        break;
                    }
                    x += len;
                    if (op == C3R_CIG_M) y += len;
                } else if (op == C3R_CIG_I || op == C3R_CIG_S) {
                    y += len;
                }
            }
        }
        break;
    }
    return tk;
}

// -------------------------------------------------------------------------------------------------
// Alt tokens: for every emitted candidate, in BAM order, what each covering read shows at the centre column (base / '*' / ref-skip,
// strand, indel length, query offset).  The host rebuilds the ordered alt_info dictionary from these
// (src/create_tensor_pileup.py:179,221-261,595-596; decode.hpp / altinfo.py), because order drives tie-breaks in the decoder
// (clair3_rna/call_variants.py:144,151,187,196).
// tile_tokens: the candidates of ONE tile / span, by the workgroup that has just decided them (k_fused_tiles: the records are still
// in the cache, the candidates in LDS) or by k_tile_tokens (column-store path).
// A candidate's tokens are what the ordered alt_info is made of (src/create_tensor_pileup.py:221-258): one token per read that shows
// something OTHER than the reference base or a ref-skip on the column — a non-reference A / C / G / T, a '*' / '#', an indel attached to
// the column (also one sitting behind a ref-skip).  Reads that show the reference base, an N, an IUPAC letter or '>' / '<' add nothing to
// alt_info beyond the depth, which the site record carries.  (Rounds 1-4 wrote a token for EVERY covering read: 5.6 M tokens per chr20
// pass of which 0.5 M said anything, a cover pass over the span's read headers per batch of candidates, and records walked once per
// batch x read chunk — at a deep locus with many candidates that product ran to 100 ms, profiles/r5/depth_cap_zones.txt.)
// One pass: one lane per record of the tile's range; every candidate an M / D piece covers is looked at, and a read with something to
// say takes the candidate's next slot (LDS counter).  Slots: the candidate owns `cap` of them from `toff` on — the gates' bound
// (TileOut::cov) — so the order INSIDE a site is the order of arrival; c3r_get_tokens and the row snapshot sort a site's tokens by read
// index, i.e. BAM order (k_export_tokens / k_pack_*).  done(k, n) reports the tokens written for candidate k.
struct TokLds {
    unsigned long long cmask[TILE / 64];       // the candidates as a bit per position of the tile
    int32_t toff[TILE], cap[TILE], cnt[TILE];  // per candidate (rank by position): first slot, slots owned, tokens written
    uint8_t refn[TILE];                        // its reference base as a 4-bit code (evc_base_from: anything but C / G / T counts as A)
    int32_t lost;                              // tokens that found no slot (the bound was wrong: the scan fails)
};
// index of the candidate at tile position x (its bit in cmask is set)
__device__ __forceinline__ int tok_cand(const TokLds &K, int x) {
    const int w = x >> 6;
    int c = __popcll(K.cmask[w] & ((1ull << (x & 63)) - 1ull));
#pragma unroll
    for (int k = 0; k < TILE / 64 - 1; ++k) if (k < w) c += __popcll(K.cmask[k]);
    return c;
}
// the candidates among the tile positions [x0, x0 + n), n <= 64, as bits 0 .. n - 1
__device__ __forceinline__ unsigned long long tok_cand_bits(const TokLds &K, int x0, int n) {
    const int w = x0 >> 6, sh = x0 & 63;
    unsigned long long bits = K.cmask[w] >> sh;
    if (sh + n > 64) bits |= K.cmask[w + 1] << (64 - sh);          // (x0 + n <= TILE: w + 1 is a word of the mask)
    return n >= 64 ? bits : bits & ((1ull << n) - 1ull);
}

__device__ __forceinline__ void tok_emit(TokLds &K, c3r_token_t *tok, long long tok_cap, int c, int r, int indel, uint32_t qpos, int base, bool rev,
                                         uint32_t del_after = 0) {
    const int i = atomicAdd(&K.cnt[c], 1);
    if (i >= K.cap[c]) { atomicAdd(&K.lost, 1); return; }
    const long long slot = (long long)K.toff[c] + i;
    if (slot >= tok_cap) { atomicAdd(&K.lost, 1); return; }
    int4 v;
    v.x = r; v.y = indel; v.z = (int)qpos; v.w = base | ((rev ? 1 : 0) << 8) | (int)(min(del_after, 65535u) << 16);
    *reinterpret_cast<int4 *>(&tok[slot]) = v;
}

// One record of the tile's range against the candidates.  `bits`: the candidates the piece covers (tok_cand_bits over its positions
// inside the tile, bit j = position b0 + j) — the caller has tested them before it fetched any bases: most records cover none.
__device__ __forceinline__ void tok_rec(TokLds &K, c3r_token_t *tok, long long tok_cap, const int4 ra, const int4 rb, uint64_t w0, uint64_t w1, int boff,
                                        int t0, int t1, unsigned long long bits) {
    const uint32_t w = (uint32_t)ra.y;
    const int op = (int)(w & 3u), prev = (int)((w >> 2) & 15u), len = (int)((w >> 9) & 31u), avail = (int)((w >> 14) & 31u);
    const bool rev = (w & 64u) != 0;
    const int rstart = ra.x, r = rb.y;
    if (op != C3R_CIG_I && bits) {
        const int b0 = max(rstart, t0);
        const int odd = (int)(((uint32_t)ra.z + (uint32_t)boff) & 1u);      // the bases were loaded from nibble naddr + boff on
        while (bits) {
            const int j = __builtin_ctzll(bits);
            bits &= bits - 1ull;
            const int p = b0 + j, c = tok_cand(K, p - t0);
            int base = 16;
            if (op == C3R_CIG_M) base = (p - rstart) < avail ? nibble_at(w0, w1, odd + (p - rstart) - boff) : 15;
            int indel = 0; uint32_t qpos = 0, dafter = 0;
            if (p == rstart + len - 1 && rb.z != 0) {
                // htslib: the op after the one that ends on the column (k_prep has looked ahead)
                indel = rb.z;
                if (indel > 0) { qpos = (uint32_t)rb.x + (op == C3R_CIG_M ? (uint32_t)len : 0u); dafter = op == C3R_CIG_M ? (uint32_t)rb.w : (uint32_t)ra.z; }
            }
            // (base: one bit set = A / C / G / T; 16 = inside a deletion; 15 = N and the IUPAC codes say nothing)
            const bool says = indel != 0 || base == 16 || ((base == 1 || base == 2 || base == 4 || base == 8) && base != (int)K.refn[c]);
            if (says) tok_emit(K, tok, tok_cap, c, r, indel, qpos, base, rev, dafter);
        }
    }
    if (prev == C3R_CIG_N && (op == C3R_CIG_I || op == C3R_CIG_D)) {
        // I / D right after a ref-skip: attached to the last intron column, which shows the ref-skip itself
        const int ax = rstart - 1 - t0;
        if (ax >= 0 && ax < t1 - t0 && ((K.cmask[ax >> 6] >> (ax & 63)) & 1ull))
            tok_emit(K, tok, tok_cap, tok_cand(K, ax), r, op == C3R_CIG_I ? (int)(uint32_t)rb.w : -(int)(uint32_t)rb.w, op == C3R_CIG_I ? (uint32_t)rb.x : 0u, 17, rev,
                     (op == C3R_CIG_I && rb.z < 0) ? (uint32_t)(-rb.z) : 0u);
    }
}

// cand(k, lpos, toff, cap): position (relative to t0), first token slot and slots owned of the tile's k-th candidate, ascending.
// done(k, n): the tokens written for it.  Returns (to every thread) the tokens that found no slot — zero unless the bound is wrong.
// HAVE_MASK: thread tid stands for position t0 + tid and `mine` says whether it is a candidate (k_fused_tiles): the mask is four ballots,
// no clearing, no atomics, one barrier less.
template <bool HAVE_MASK, int NT = SCAN_THREADS, class CandFn, class DoneFn>
__device__ __forceinline__ int tile_tokens(const ScanArgs &a, TokLds &K, int t0, int t1, int region, int rlo, int rhi, int nc, CandFn &&cand, DoneFn &&done,
                                           c3r_token_t *tok, long long tok_cap, bool mine = false) {
    const int tid = (int)threadIdx.x;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const unsigned long long my_wave = HAVE_MASK ? __ballot(mine) : 0ull;
    __syncthreads();                                        // (K lies where the accumulators lay: everyone is done with them)
    if (HAVE_MASK) { if ((tid & 63) == 0 && (NT == SCAN_THREADS || tid < TILE)) K.cmask[tid >> 6] = my_wave; }
    else {
        if (tid < TILE / 64) K.cmask[tid] = 0ull;
        __syncthreads();
    }
    if (tid == 0) K.lost = 0;
    if (tid < nc) {
        int lp, off, cp;
        cand(tid, lp, off, cp);
        K.toff[tid] = off; K.cap[tid] = cp; K.cnt[tid] = 0;
        const int rp = t0 + lp - a.ref_beg0;
        const uint8_t rb = (rp >= 0 && rp < a.ref_len) ? a.ref[rp] : (uint8_t)'N';
        K.refn[tid] = (uint8_t)(1u << ref_index(rb));
        if (!HAVE_MASK) atomicOr(&K.cmask[lp >> 6], 1ull << (lp & 63));
    }
    __syncthreads();
    // a record matters only if its piece covers a candidate (or, an indel behind a ref-skip, sits on one): the candidates' bit mask says
    // so from the record's first half alone, before its second half or any base is fetched
    if (NT > SCAN_THREADS) {
        // k_fused_deep: tens or hundreds of thousands of records, of which the few that cover a candidate matter — the walk is round trips (a giant
        // span with one candidate: 260 rounds, 0.5 ms).  Four first halves per lane and round, the second half and the bases only for a record that counts.
        constexpr int U = 4;
#pragma unroll 1
        for (int base = rlo; base < rhi; base += NT * U) {
            int4 ra[U];
#pragma unroll
            for (int u = 0; u < U; ++u) ra[u] = *reinterpret_cast<const int4 *>(a.recs + min(base + u * NT + tid, rhi - 1));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (base + u * NT + tid >= rhi) continue;
                const uint32_t w = (uint32_t)ra[u].y;
                const int op = (int)(w & 3u), len = (int)((w >> 9) & 31u), avail = (int)((w >> 14) & 31u);
                const int b0 = max(ra[u].x, t0), b1 = min(ra[u].x + len, t1);
                unsigned long long bits = 0ull;
                if (op != C3R_CIG_I && b0 < b1) bits = tok_cand_bits(K, b0 - t0, b1 - b0);
                const int ax = ra[u].x - 1 - t0;
                const bool anchored = op != C3R_CIG_M && ((w >> 2) & 15u) == (uint32_t)C3R_CIG_N && ax >= 0 && ax < t1 - t0 && ((K.cmask[ax >> 6] >> (ax & 63)) & 1ull);
                if (!(bits || anchored)) continue;
                const int4 rb = reinterpret_cast<const int4 *>(a.recs + (base + u * NT + tid))[1];
                if (a.drop && read_dropped(a.drop, a.drop_words, region, rb.y)) continue;
                uint64_t w0 = 0, w1 = 0;
                int boff = 0;
                if (op == C3R_CIG_M && bits) {
                    boff = b0 - ra[u].x;
                    if (boff < avail) {
                        const uint64_t na = ((uint64_t)(uint32_t)ra[u].z | ((uint64_t)(uint32_t)ra[u].w << 32)) + (uint64_t)boff;
                        u64x2 ww;
                        __builtin_memcpy(&ww, a.seq + (na >> 1), 16);
                        w0 = ww[0]; w1 = ww[1];
                    }
                }
                tok_rec(K, tok, tok_cap, ra[u], rb, w0, w1, boff, t0, t1, bits);
            }
        }
    } else
    for (int base = rlo; base < rhi; base += NT * WALK_UNR) {
        int4 ra[WALK_UNR], rb[WALK_UNR];
        bool have[WALK_UNR];
        unsigned long long bits[WALK_UNR];
#pragma unroll
        for (int u = 0; u < WALK_UNR; ++u) {
            const int iu = base + u * NT + tid;
            have[u] = iu < rhi;
            const int4 *rec = reinterpret_cast<const int4 *>(a.recs + (have[u] ? iu : rhi - 1));
            ra[u] = rec[0]; rb[u] = rec[1];
        }
        uint64_t w0[WALK_UNR], w1[WALK_UNR];
        int boff[WALK_UNR];
#pragma unroll
        for (int u = 0; u < WALK_UNR; ++u) {
            w0[u] = 0; w1[u] = 0; boff[u] = 0; bits[u] = 0ull;
            const uint32_t w = (uint32_t)ra[u].y;
            const int op = (int)(w & 3u), len = (int)((w >> 9) & 31u), avail = (int)((w >> 14) & 31u);
            const int b0 = max(ra[u].x, t0), b1 = min(ra[u].x + len, t1);
            if (have[u] && op != C3R_CIG_I && b0 < b1) bits[u] = tok_cand_bits(K, b0 - t0, b1 - b0);
            const int ax = ra[u].x - 1 - t0;
            const bool anchored = op != C3R_CIG_M && ((w >> 2) & 15u) == (uint32_t)C3R_CIG_N && ax >= 0 && ax < t1 - t0 && ((K.cmask[ax >> 6] >> (ax & 63)) & 1ull);
            if (!(bits[u] || anchored)) have[u] = false;
            // (mpileup's depth cap: the records of a read this region's scan has discarded say nothing)
            if (have[u] && a.drop && read_dropped(a.drop, a.drop_words, region, rb[u].y)) have[u] = false;
            if (have[u] && op == C3R_CIG_M && bits[u]) {
                boff[u] = b0 - ra[u].x;
                if (boff[u] < avail) {
                    const uint64_t na = ((uint64_t)(uint32_t)ra[u].z | ((uint64_t)(uint32_t)ra[u].w << 32)) + (uint64_t)boff[u];
                    u64x2 w;
                    __builtin_memcpy(&w, a.seq + (na >> 1), 16);
                    w0[u] = w[0]; w1[u] = w[1];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < WALK_UNR; ++u)
            if (have[u]) tok_rec(K, tok, tok_cap, ra[u], rb[u], w0[u], w1[u], boff[u], t0, t1, bits[u]);
    }
    __syncthreads();
    if (tid < nc) done(tid, min(K.cnt[tid], K.cap[tid]));
    return K.lost;
}

// the column-store path's token kernel: one workgroup per tile that holds candidates (tile_cand from the compaction)
struct TileTokArgs {
    ScanArgs a;                    // the scan's own arguments (reads, pile table, tile list and ranges, filters, depth cap)
    const int32_t *cand_idx; const int2 *tile_cand; const int32_t *tok_off /* [n_cand + 1]: exclusive sums of the candidates' slot counts */; c3r_site_t *sites; c3r_token_t *tok;
    int32_t tok_base;              // token slots already taken by earlier scans of the batch
    int32_t *totals;               // [0] += tokens written, [1] += tokens that found no slot (the host fails the scan)
};
#ifndef C3R_TOK_OCC
#define C3R_TOK_OCC 6
#endif
__global__ __launch_bounds__(SCAN_THREADS, C3R_TOK_OCC) void k_tile_tokens(const TileTokArgs t) {
    __shared__ TokLds K;
    const ScanArgs &a = t.a;
    const int n_list = *a.n_tile_list;
    for (int lb = blockIdx.x; lb < n_list; lb += gridDim.x) {
        const int tile = a.tile_list[lb];
        const int2 tc = t.tile_cand[tile];
        if (tc.y <= 0) continue;
        const TileGeo tg = a.geo[tile];
        const int slot0 = tile * TILE;
        const int4 rng = a.tile_rng[tile];
        const int lost = tile_tokens<false>(a, K, tg.p0, tg.p1, tg.region, rng.z, rng.w, tc.y, [&](int k, int &lp, int &off, int &cp) {
            const int w = tc.x + k;
            lp = t.cand_idx[w] - slot0;
            off = t.tok_base + t.tok_off[w];
            cp = t.tok_off[w + 1] - t.tok_off[w];
            if (t.sites) t.sites[w].tok_off = (uint32_t)off;
        }, [&](int k, int n) {
            if (t.sites) t.sites[tc.x + k].n_tok = n;
            if (n) atomicAdd(&t.totals[0], n);
        }, t.tok, (long long)INT32_MAX);
        if (lost && threadIdx.x == 0) atomicAdd(&t.totals[1], lost);
    }
}

// -------------------------------------------------------------------------------------------------
// Phased channels, the two places where the ORDER of the column's token list matters (src/create_tensor_pileup.py:113-145,
// :180-217).  The reference walks the column's reads in BAM order with an index into the HP list that advances on base
// tokens (ACGTN acgtn * #) and on ref-skips, but NOT on letters it ignores ('=' / IUPAC): after such a read every later
// read of the column is phased with its predecessor's tag.  And an indel token takes phasing[idx-1], the previous list
// ENTRY: its own read's base normally, but for an indel sitting on a ref-skip column whatever came before ('0' if that
// was another indel token).  k_scan_tiles counts haplotypes with unordered atomics (each read's own tag) and flags the
// columns where either case occurs (flags bit 3); here one thread per flagged column redoes the 12 haplotype channels
// in order.  Both cases are rare (aligners do not emit them), so the serial walk over the column's reads is affordable.
struct PhaseArgs {
    const int32_t *tile_list; const int32_t *n_tile_list; const int4 *tile_rng; const TileGeo *geo;
    const DevRead *reads; const DevSeg *rsegs; const uint32_t *rseg_first; const uint32_t *cigar; const uint8_t *seq;
    const uint8_t *flags; int32_t *cols;
    int32_t min_mq, excl_flags;
    const uint32_t *drop; int32_t drop_words;
};
// the 12 haplotype channels of position p, in order, from the reads [lo, hi)
__device__ __forceinline__ void phase_column(const PhaseArgs &a, int p, int lo, int hi, int region, int (&cnt)[12]) {
    auto next_cov = [&](int r) {            // next read after r (BAM order) that passes the filters and covers p
        for (++r; r < hi; ++r) {
            const DevRead rd = a.reads[r];
            if (read_passes(rd, a.min_mq, a.excl_flags) && rd.pos <= p && rd.end > p && !read_dropped(a.drop, a.drop_words, region, r)) break;
        }
        return r;
    };
#pragma unroll
    for (int k = 0; k < 12; ++k) cnt[k] = 0;
    int r2 = next_cov(lo - 1);              // cursor into the HP list (one entry per covering read)
    int prev = 0; bool have_prev = false;
    for (int r = next_cov(lo - 1); r < hi; r = next_cov(r)) {
        const DevRead rd = a.reads[r];
        const TokenAt tk = token_at(rd, r, p, a.rsegs, a.rseg_first, a.cigar, a.seq);
        const int bi = acgt_index(tk.base);
        if (tk.base == 16 || tk.base == 15 || bi >= 0) {          // * / #, N, A C G T: a list entry that consumes a tag
            const int hp = r2 < hi ? (int)a.reads[r2].hp : 0;
            r2 = next_cov(r2);
            if (bi >= 0) { if (hp == 1) cnt[bi]++; else if (hp == 2) cnt[6 + bi]++; }
            prev = hp; have_prev = true;
        } else if (tk.base == 17) {
            r2 = next_cov(r2);                                    // a ref-skip consumes a tag, adds no entry
        }
        if (tk.indel != 0) {
            if (have_prev) {
                const int k = tk.indel > 0 ? 4 : 5;               // IP / DP (+6: IM / DM)
                if (prev == 1) cnt[k]++; else if (prev == 2) cnt[6 + k]++;
            }
            prev = 0; have_prev = true;                           // the indel token's own entry is phased '0'
        }
    }
}
__global__ __launch_bounds__(TILE) void k_phase_recompute(const PhaseArgs a) {
    const int n_list = *a.n_tile_list;
    for (int lb = blockIdx.x; lb < n_list; lb += gridDim.x) {
    const int tile = a.tile_list[lb];
    const int slot = tile * TILE + (int)threadIdx.x;
    if (!(a.flags[slot] & 8)) continue;
    const int p = a.geo[tile].p0 + (int)threadIdx.x;
    const int4 rng = a.tile_rng[tile];
    int cnt[12];
    phase_column(a, p, rng.x, rng.y, a.geo[tile].region, cnt);
    int32_t *c = a.cols + (size_t)slot * C3R_CH_PHASED + C3R_AP;
#pragma unroll
    for (int k = 0; k < 12; ++k) c[k] = cnt[k];
    }
}

// -------------------------------------------------------------------------------------------------
// The plain mode (no head/tail calling, no splice padding, no genotyping list) in ONE tile kernel: k_fused_tiles.
//
// A workgroup takes a span of FUSE_IN = 224 positions that holds aligned bases and builds the columns of those positions plus
// C3R_FLANK on either side — 256 positions, one per thread — in LDS (tile_columns).  Every candidate of the inner span then has its
// whole 33-column window in LDS: the window rule (33 contiguous rows, :565-568), the rescale (clair3_rna/utils.py:88-92) and the int32
// window are produced right there, and the columns never go to HBM (round 2 wrote 644 MB of columns per chr20 pass and read them
// back in k_gather: 2.6x the algorithmic bytes).
// Order.  Candidates must come out in position order, but a span's first output index is the sum of the counts of all spans before
// it, and a span's run time varies 10x with depth: making each span wait for its predecessors (decoupled look-back inside this
// kernel, the first version) let the chip's ~1300 resident workgroups retire only as fast as the slowest of them — 1.35 ms per
// chr20 pass against 0.61 ms without the ordering.  So the WINDOWS — and, since round 4, the candidates' TOKENS — are written where
// they arrive: a span takes its rows and its token slots with one atomic add each and notes (first row, candidates, tokens, first
// token) in span_info and a 20-byte record per candidate in `meta`.  k_order_spans then sums the candidate counts in span order
// (uniform work: look-back costs nothing there), and k_finalize_sites writes everything that is small — site records with their
// token offsets, slots — in position order, plus win_idx[i] = the row of the i-th site's window, through which layer 1 of the
// network (and c3r_get_tensors) reads the tensors.  A site's tokens are found through its tok_off (contiguous, BAM order), so the
// token array needs no global order either; c3r_get_tokens exports it in site order.  No count -> scan -> write over flag arrays, no host
// round trip for sizes: outputs are bounds-checked against the buffers' capacity, and the host learns the totals (and whether
// anything did not fit: then it grows the buffers and repeats the scan) from the single read-back at the end of the scan.
#ifndef C3R_TICKET_RUN
#define C3R_TICKET_RUN 4
#endif
constexpr int TICKET_RUN = C3R_TICKET_RUN;
constexpr int TICKET_Q = 16, TICKET_STRIDE = 64;     // ticket words 256 bytes apart: atomics on one cache line serialise like atomics on one word
// Output rows and token slots are handed out by up to ALLOC_SHARDS sub-allocators, each owning an equal part of the scan's row / token
// space and ONE 64-bit word (tokens << 32 | rows; the words 256 bytes apart): a span takes both with one returning atomic on its
// workgroup's shard.  With one `arrived` and one `tok_arrived` word for the whole scan (in one cache line) the 2 x 23 k returning atomics
// of a chr20 pass were the kernel's floor: a word takes ~88 of them per microsecond (MI355X_MICROARCH.md, "dequeue") = 0.52 ms, and the
// kernel ran 0.65 ms with candidates against 0.33 ms without.  Rows and tokens are reached through win_idx / tok_off, so the holes
// between the shards' runs cost address space only.
constexpr int ALLOC_SHARDS = 16, ALLOC_STRIDE = 32;  // (u64 words)
struct CandMeta { int32_t slot, depth, ncov, tpre, span, pos0; };      // per arrived candidate: slot (tile * TILE + offset), depth, its tokens (the slots reserved until the token pass has counted), token slots of the
                                                                       // span's earlier candidates, list index of its span, its 0-based position
struct FusedArgs {
    ScanArgs a;                   // tile_list: spans with aligned bases, ascending; tile_rng: reads / segments of the span + flanks
    int32_t *ticket;              // [TICKET_Q * TICKET_STRIDE] tickets handed out per queue
    const SpanRec *span_rec;      // [listed spans] where the span lies, its region's bounds (rows exist only inside their region), its read / segment ranges
    unsigned long long *alloc;    // [n_shards * ALLOC_STRIDE] per shard: tokens << 32 | rows handed out so far
    int32_t n_shards;             // 1 for small scans (dense output), ALLOC_SHARDS for large ones
    int32_t shard_rows, shard_toks;   // rows / token slots a shard owns: shard s hands out rows [s * shard_rows, (s + 1) * shard_rows)
    int32_t *overflow;            // bit 0: a span's candidates or tokens did not fit into its shard (nothing was written past it)
    int32_t cand_cap;             // = n_shards * shard_rows
    int32_t rescale, max_depth;   // A5: windows with depth > 1.5 x max_depth are rescaled
    void *tensors;                // [cand_cap][33][C], rows in arrival order: int16 (x16: the resident, rescaled windows — a count never exceeds the reads that
                                  // cover its position, and a position that 32768 reads cover fails the scan) or int32 (the raw re-run of c3r_get_tensors)
    int32_t x16;
    int4 *span_info;              // [listed spans] {first row, candidates, tokens, first token (scan-relative)}
    CandMeta *meta;               // [cand_cap]
    c3r_token_t *tok;             // the batch's token array (null: no tokens wanted — the raw re-run of c3r_get_tensors)
    int32_t tok_base;             // token slots taken by earlier scans of the batch
    int32_t tok_cap;              // = n_shards * shard_toks
    PhaseArgs ph;                 // 30 channels: the ordered recompute of flagged columns
};

// One span, by the whole workgroup: columns (tile_columns), the window rule, rows and token slots, the windows, `between()` (k_fused_tiles
// fetches the next span's record there), the candidates' tokens.  b: the span's list position.
struct SpanShared { int row0, tok0, fits; };
template <int C, int NT, int EVL, class Between>
__device__ __forceinline__ void fused_span(const FusedArgs &f, TileMem<C, EVL> &M, SpanShared &S, const int b, const int shard, const int4 r0, const int4 rng, const int4 r2,
                                           Between &&between) {
    const ScanArgs &a = f.a;
    const int tid = (int)threadIdx.x;
    {
        const int tile = r0.x;
        TileGeo tg; tg.p0 = r0.y; tg.p1 = r0.z; tg.region = r0.w; tg.pad = 0;
        const int2 rb = make_int2(r2.x, r2.y);
        const int x0 = tg.p0 - C3R_FLANK, x1 = min(tg.p1 + C3R_FLANK, rb.y);       // thread tid <-> position x0 + tid
        const TileOut o = tile_columns<C, true, NT, EVL>(a, M, x0, x1, rb.x, tg.region, rng.x, rng.y, rng.z, rng.w, tg.p0, tg.p1,
                                                         (NT > SCAN_THREADS && r2.z > 0 && giant_on(a, rng.w - rng.z)) ? r2.z : 0);
        unsigned long long t_tail = C3R_DBG(a) ? wall_clock64() : 0ull;
        int dbg_slot = 7;
        if (C == C3R_CH_PHASED) {
            // a column whose haplotype channels depend on the ORDER of the reads (see k_phase_recompute): redone in place, one thread
            // per flagged column
            if (o.is_row && M.odd[tid]) {
                if (!f.ph.rsegs) atomicOr(f.overflow, 4);         // no op / segment tables yet: the host builds them and repeats the scan
                else {
                    int cnt[12];
                    phase_column(f.ph, x0 + tid, rng.x, rng.y, tg.region, cnt);
#pragma unroll
                    for (int k = 0; k < 12; ++k) M.cnt[tid * C + C3R_AP + k] = cnt[k];
                }
            }
        }
        // ---- the window rule: 33 contiguous rows = 33 set bits in the span's row mask (published by tile_columns with its last barrier)
        bool emit = false;
        if (o.cand && tid >= C3R_FLANK && tid + C3R_FLANK < TILE) {
            const int first = tid - C3R_FLANK, w = first >> 6, sh = first & 63;
            unsigned long long bits = M.rowmask[w] >> sh;
            if (sh > 64 - C3R_WINDOW) bits |= M.rowmask[w + 1] << (64 - sh);          // (first + 32 <= 255: w + 1 <= 3)
            constexpr unsigned long long ALL = (1ull << C3R_WINDOW) - 1ull;
            emit = (bits & ALL) == ALL;
            if (C3R_ABL(a) & 2048) emit = false;
        }
        if (C3R_DBG(a) && tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&C3R_DBG(a)[9], now_ - t_tail); t_tail = now_; }
        int nc, nt;
        const int2 ex = block_excl_scan2<NT>(emit ? 1 : 0, emit ? o.cov : 0, M.scan_slot[1], &nc, &nt);
        if (C3R_DBG(a) && tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&C3R_DBG(a)[10], now_ - t_tail); t_tail = now_; }
        const int rank = ex.x, tpre = ex.y;
        // per candidate (by rank): position in the span, depth for the window copy, tokens of the span's earlier candidates — the event
        // arrays are free by now
        if (emit) { M.amb[rank] = (uint8_t)tid; M.evfill[rank] = o.depth; M.evoff[rank] = tpre; }
        if (tid == 0) {
            int lrow = 0, ltok = 0;
            if (nc) {
                const unsigned long long got = atomicAdd(&f.alloc[shard * ALLOC_STRIDE], ((unsigned long long)(unsigned)(f.tok ? nt : 0) << 32) | (unsigned)nc);
                lrow = (int)(unsigned)got; ltok = (int)(unsigned)(got >> 32);
            }
            const bool fits = (long long)lrow + nc <= (long long)f.shard_rows && (!f.tok || (long long)ltok + nt <= (long long)f.shard_toks);
            S.fits = fits ? 1 : 0;
            S.row0 = shard * f.shard_rows + lrow;
            S.tok0 = shard * f.shard_toks + ltok;
            f.span_info[b] = make_int4(S.row0, nc, 0, S.tok0);                // (.z: tokens actually written, added up by the token pass)
            if (C3R_DBG(a)) { const unsigned long long now_ = wall_clock64(); atomicAdd(&C3R_DBG(a)[11], now_ - t_tail); t_tail = now_; }
        }
        __syncthreads();
        const int row0 = S.row0, tok0 = S.tok0;
        const bool fits = S.fits != 0;
        if (nc > 0 && !fits && tid == 0) atomicOr(f.overflow, 1);
        if (nc > 0 && fits) {
        if (emit) {
            CandMeta m;
            m.slot = tile * TILE + (tid - C3R_FLANK); m.depth = o.depth; m.ncov = o.cov; m.tpre = tpre; m.span = b; m.pos0 = x0 + tid;
            f.meta[row0 + rank] = m;
        }
        // ---- the span's nc windows are ONE contiguous run of the output (rows [row0, row0 + nc) x 33 x C int32) and each window is
        // one contiguous run of LDS: all threads copy the run, 16 bytes per lane and store (a window is 8 bytes short of a multiple
        // of 16, so whole-window copies could only use 8-byte stores)
        constexpr int WIN = C3R_WINDOW * C;
        const int resc_thr = f.rescale ? 3 * f.max_depth : INT32_MAX;                                // depth > 1.5 x max_depth  <=>  2 depth > 3 max_depth
        auto fetch = [&](int g) -> int {
            const int k = g / WIN, oo = g - k * WIN;
            int v = M.cnt[((int)M.amb[k] - C3R_FLANK) * C + oo];
            const int dep = M.evfill[k];
            if (2 * (long long)dep > (long long)resc_thr) v = (int32_t)((double)v / ((double)dep / (double)f.max_depth));
            return v;
        };
        const int total = nc * WIN;
        if (f.x16) {
            // eight 16-bit values per lane and store: the run starts on a 4-byte boundary (a window is 1188 / 1980 bytes), so the values before
            // the first 16-byte boundary are an even number, every group of eight starts on an even offset of its window (8-byte LDS reads),
            // and a group straddles two windows once in ~70
            int16_t *out = (int16_t *)f.tensors + (size_t)row0 * WIN;
            const int head = min(total, (int)((16u - ((unsigned)(uintptr_t)out & 15u)) & 15u) >> 1);
            if (tid < head && !(C3R_ABL(a) & 64)) out[tid] = (int16_t)fetch(tid);
            const int n8 = (C3R_ABL(a) & 64) ? 0 : (total - head) >> 3;
            constexpr int STEP = 8 * NT, KSTEP = STEP / WIN, OSTEP = STEP % WIN;
            int g = head + 8 * tid;
            int k = g / WIN, oo = g - k * WIN;
            for (int c8 = tid; c8 < n8; c8 += NT, g += STEP) {
                int v[8];
                if (oo + 7 < WIN) {
                    // (an even offset into an even-strided row of a 16-byte-aligned array: 8-byte LDS reads)
                    typedef int int2v __attribute__((ext_vector_type(2)));
                    const int2v *src = reinterpret_cast<const int2v *>(&M.cnt[((int)M.amb[k] - C3R_FLANK) * C + oo]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const int2v t = src[e]; v[2 * e] = t[0]; v[2 * e + 1] = t[1]; }
                    const int dep = M.evfill[k];
                    if (2 * (long long)dep > (long long)resc_thr) {
                        const double sf = (double)dep / (double)f.max_depth;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = (int32_t)((double)v[e] / sf);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fetch(g + e);
                }
                int4 pk;                                       // low halves of two values side by side: one v_perm_b32 per pair
                pk.x = (int)__builtin_amdgcn_perm((unsigned)v[1], (unsigned)v[0], 0x05040100u); pk.y = (int)__builtin_amdgcn_perm((unsigned)v[3], (unsigned)v[2], 0x05040100u);
                pk.z = (int)__builtin_amdgcn_perm((unsigned)v[5], (unsigned)v[4], 0x05040100u); pk.w = (int)__builtin_amdgcn_perm((unsigned)v[7], (unsigned)v[6], 0x05040100u);
                *reinterpret_cast<int4 *>(out + g) = pk;
                oo += OSTEP; k += KSTEP;
                if (oo >= WIN) { oo -= WIN; ++k; }
            }
            const int gt = head + 8 * n8 + tid;
            if (gt < total && !(C3R_ABL(a) & 64)) out[gt] = (int16_t)fetch(gt);
        } else {
        int32_t *out = (int32_t *)f.tensors + (size_t)row0 * WIN;
        const int head = min(total, (int)((16u - ((unsigned)(uintptr_t)out & 15u)) & 15u) >> 2);      // ints before the first 16-byte boundary
        if (tid < head) out[tid] = fetch(tid);
        const int n4 = (C3R_ABL(a) & 64) ? 0 : (total - head) >> 2;          // (ablation 64: no window store — byte attribution, tools/pmc_bytes.sh)
        {
            // a thread's groups of four lie 4 * SCAN_THREADS ints apart: window index and offset inside the window are carried along
            // (one division per thread instead of one per int); a group that straddles two windows (1 in ~150) takes the general path
            constexpr int STEP = 4 * NT, KSTEP = STEP / WIN, OSTEP = STEP % WIN;
            int g = head + 4 * tid;
            int k = g / WIN, oo = g - k * WIN;
            for (int c4 = tid; c4 < n4; c4 += NT, g += STEP) {
                int4 v;
                if (oo + 3 < WIN) {
                    const int32_t *src = &M.cnt[((int)M.amb[k] - C3R_FLANK) * C + oo];
                    v.x = src[0]; v.y = src[1]; v.z = src[2]; v.w = src[3];
                    const int dep = M.evfill[k];
                    if (2 * (long long)dep > (long long)resc_thr) {
                        const double sf = (double)dep / (double)f.max_depth;
                        v.x = (int32_t)((double)v.x / sf); v.y = (int32_t)((double)v.y / sf); v.z = (int32_t)((double)v.z / sf); v.w = (int32_t)((double)v.w / sf);
                    }
                } else {
                    v.x = fetch(g); v.y = fetch(g + 1); v.z = fetch(g + 2); v.w = fetch(g + 3);
                }
                *reinterpret_cast<int4 *>(out + g) = v;
                oo += OSTEP; k += KSTEP;
                if (oo >= WIN) { oo -= WIN; ++k; }
            }
        }
        const int gt = head + 4 * n4 + tid;
        if (gt < total && !(C3R_ABL(a) & 64)) out[gt] = fetch(gt);
        }
        }
        between();
        if (nc > 0 && fits) {
        // ---- the candidates' tokens, while the span's records are still in the cache (k_tile_tokens walked the op table a second time:
        // 0.29 ms and 291 MB per chr20 pass).  The accumulators are dead once the windows are out: their LDS holds the token pass's tables
        if (f.tok && !(C3R_ABL(a) & 128)) {                                   // (ablation 128: no token pass)
            if (C3R_DBG(a) && tid == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&C3R_DBG(a)[7], now_ - t_tail); t_tail = now_; dbg_slot = 8; }
            TokLds &K = *reinterpret_cast<TokLds *>(M.cnt);
            const int lost = tile_tokens<true, NT>(a, K, x0, x1, tg.region, rng.z, rng.w, nc, [&](int k, int &lp, int &off, int &cp) {
                lp = (int)M.amb[k]; off = f.tok_base + tok0 + M.evoff[k];
                cp = (k + 1 < nc ? M.evoff[k + 1] : nt) - M.evoff[k];
            }, [&](int k, int n) {
                f.meta[row0 + k].ncov = n;                                  // (the site's n_tok: what was written, not what was reserved)
                if (n) atomicAdd(&f.span_info[b].z, n);                     // the span's tokens: k_order_spans sums them into the scan's total
            }, f.tok, (long long)f.tok_base + (long long)(shard + 1) * f.shard_toks, emit);
            if (lost && tid == 0) atomicOr(f.overflow, 8);
        }
        }
        if (C3R_DBG(a) && tid == 0) atomicAdd(&C3R_DBG(a)[dbg_slot], wall_clock64() - t_tail);
    }
}

template <int C>
#ifndef C3R_FUSED_OCC
#define C3R_FUSED_OCC 5
#endif
__global__ __launch_bounds__(SCAN_THREADS, (C == C3R_CH ? C3R_FUSED_OCC : C3R_SCAN_OCC30)) void k_fused_tiles(const FusedArgs f) {
    __shared__ TileMem<C> M;
    __shared__ int s_ticket;
    __shared__ SpanShared S;
    __shared__ int4 s_rec[3];
    const int shard = (int)(blockIdx.x % (unsigned)f.n_shards);
    static_assert(sizeof(TokLds) <= sizeof(M.cnt), "the token pass re-uses the accumulators' LDS");
    const ScanArgs &a = f.a;
    const int tid = (int)threadIdx.x;
    const int n = *a.n_tile_list;
    // Spans are taken by ticket — dealing them out by stride is 15 % slower: heavy spans differ 10x and a static deal leaves the last
    // workgroups alone with theirs.  ONE ticket word saturates at ~88 dequeues per microsecond (MI355X_MICROARCH.md, "dequeue"): 20 k
    // spans = 0.23 ms of atomic throughput, 0.14 ms of it visible in the kernel's time.  So the list is dealt into TICKET_Q interleaved
    // queues (queue q = list positions q, q + TICKET_Q, ...), a workgroup draws from its home queue and moves on to the next one when
    // that is empty; the NEXT ticket is drawn before the current span is worked on, so its round trip is hidden; the span's record is
    // one wave-uniform 48-byte load.
    // Neighbouring spans share the records of their flanks (a third of what a span reads): runs of TICKET_RUN consecutive list positions
    // belong to ONE queue, whose workgroups (blockIdx % TICKET_Q, TICKET_Q a multiple of the 8 XCDs) sit on one XCD = behind one L2.
    auto queue_len = [&](int q) { return n / (TICKET_RUN * TICKET_Q) * TICKET_RUN + min(max(n % (TICKET_RUN * TICKET_Q) - q * TICKET_RUN, 0), TICKET_RUN); };
    auto list_pos = [&](int q, int t) { return t / TICKET_RUN * (TICKET_RUN * TICKET_Q) + q * TICKET_RUN + t % TICKET_RUN; };
    int home = (int)(blockIdx.x % TICKET_Q);
    auto take = [&]() -> int {                            // (thread 0) next list position, or n when every queue is empty
        for (int tries = 0; tries < TICKET_Q; ++tries) {
            const int len = queue_len(home);
            if (len > 0) { const int t = atomicAdd(&f.ticket[home * TICKET_STRIDE], 1); if (t < len) return list_pos(home, t); }
            home = (home + 1) % TICKET_Q;
        }
        return n;
    };
    // The span's record (48 bytes: where it lies, its read / record ranges) is fetched one span AHEAD by three lanes — issued when the
    // current span's windows are out, so that its round trip runs under the token pass, and parked in LDS (s_rec) across the hand-over:
    // the next span starts without a dependent load.  (Round 4 tried the same with the fetch at the TOP of the span: the three registers
    // it kept alive through the whole span cost more than the round trip.)
    if (tid == 0) s_ticket = take();
    __syncthreads();
    int b = s_ticket;
    if (b < n && tid < 3) s_rec[tid] = reinterpret_cast<const int4 *>(f.span_rec + b)[tid];
    __syncthreads();
    while (b < n) {
        int t_next = 0;
        const int q_next = home;
        const int4 r0 = s_rec[0], rng = s_rec[1], r2 = s_rec[2];
        // a deep span (ScanArgs::deep_min records or more in its range) is left to k_fused_deep, which runs behind this kernel with workgroups of
        // sixteen wavefronts: the span would keep these four for milliseconds
        const bool deep = rng.w - rng.z >= a.deep_min;
        if (tid == 0) t_next = atomicAdd(&f.ticket[q_next * TICKET_STRIDE], 1);
        int b_next = 0;
        int4 nrec = make_int4(0, 0, 0, 0);
        // ---- the next span: three lanes of the first wavefront fetch its record when this span's windows are out (the round trip runs under the
        // token pass; the list position travels from lane 0 by a cross-lane read, no barrier)
        auto fetch_next = [&]() __attribute__((always_inline)) {
            if (tid < 64) {
                if (tid == 0) b_next = t_next < queue_len(q_next) ? list_pos(q_next, t_next) : take();
                b_next = __shfl(b_next, 0, 64);
                if (b_next < n && tid < 3) nrec = reinterpret_cast<const int4 *>(f.span_rec + b_next)[tid];
            }
        };
        if (!deep) fused_span<C, SCAN_THREADS, TileMem<C>::EV_LDS>(f, M, S, b, shard, r0, rng, r2, fetch_next);
        else {
            // (every wavefront has read THIS span's record out of s_rec before the hand-over below overwrites it: a span that is worked on passes
            // a dozen barriers in between, a skipped one none — without this one a late wavefront took the next span's record for the current)
            __syncthreads();
            fetch_next();
        }
        // ---- hand over: the next span's record and list position into LDS (s_rec was last read at the top of this span, many barriers
        // ago); the barrier also frees this span's LDS
        if (tid < 3) s_rec[tid] = nrec;
        if (tid == 0) s_ticket = b_next;
        __syncthreads();
        b = s_ticket;
    }
}

// The deep spans of the list (ScanArgs::deep_min records or more in their range), one at a time per workgroup of DEEP_THREADS = 1024 threads =
// sixteen wavefronts, one workgroup per CU.  Everything a span does record by record, read by read or event by event — the walks, the coverage, the
// event buckets, the window copy, the token pass — runs on all sixteen; the per-position steps on the first four (tile_columns, `pos_thread`).
// A span's time at depth is rounds x memory latency while its workgroup has the CU to itself and LDS-atomic throughput once the CU is full (a
// wavefront's 30 atomic adds per round cost ~9 LDS cycles each on random banks, 3.8 without conflicts: profiles/r6/lds_atomic_probe.txt): four
// times the wavefronts are a quarter of the rounds.  (Sixteen wavefronts leave the span body 128 registers, which it overruns by 12 / 20 bytes of scratch
// per lane at 18 / 30 channels — the ISA shows one 64-bit address hoisted out of the span loop, stored once and reloaded once per span, and the rest
// of the pressure taken by scalar spills into lanes of a vector register.  -DC3R_DEEP_THREADS=768, twelve wavefronts at 168 registers, has no scratch
// and is 8-10 % slower at depth: scan of bench.py's stress_500x 1.25 against 1.14 ms, capped locus 4.9 against 4.3 ms.)
// With one workgroup per CU the LDS holds DEEP_EV_LDS captured events: up to there a span's
// alleles are counted out of LDS as in the shallow kernel, without the second walk and the global hash table of the deeper ones.
#ifndef C3R_DEEP_THREADS
#define C3R_DEEP_THREADS 1024
#endif
constexpr int DEEP_THREADS = C3R_DEEP_THREADS;      // sixteen wavefronts, four per SIMD: 128 registers each
constexpr int DEEP_EV_LDS = 3072;
struct DeepArgs { const int32_t *list; const int32_t *n_list; int32_t *ticket; };
template <int C>
__global__ __launch_bounds__(DEEP_THREADS, DEEP_THREADS / 256) void k_fused_deep(const FusedArgs f, const DeepArgs d) {
    __shared__ TileMem<C, DEEP_EV_LDS> M;
    __shared__ int s_b;
    __shared__ SpanShared S;
    __shared__ int4 s_rec[3];
    static_assert(sizeof(TokLds) <= sizeof(M.cnt), "the token pass re-uses the accumulators' LDS");
    const int shard = (int)(blockIdx.x % (unsigned)f.n_shards);
    const int tid = (int)threadIdx.x;
    const int n = *d.n_list;
    for (;;) {
        if (tid == 0) { const int i = atomicAdd(d.ticket, 1); s_b = i < n ? d.list[i] : -1; }
        __syncthreads();
        const int b = s_b;
        if (b < 0) return;
        if (tid < 3) s_rec[tid] = reinterpret_cast<const int4 *>(f.span_rec + b)[tid];
        __syncthreads();
        // the span's record is the same in every lane: say so, and its twelve words live in scalar registers for the length of the span
        auto uni = [](const int4 v) { return make_int4(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y),
                                                       __builtin_amdgcn_readfirstlane(v.z), __builtin_amdgcn_readfirstlane(v.w)); };
        const int4 r0 = uni(s_rec[0]), rng = uni(s_rec[1]), r2 = uni(s_rec[2]);
        const unsigned long long t_span = C3R_DBG(f.a) ? wall_clock64() : 0ull;
        fused_span<C, DEEP_THREADS, DEEP_EV_LDS>(f, M, S, __builtin_amdgcn_readfirstlane(b), shard, r0, rng, r2, [] {});
        __syncthreads();                      // (s_b, s_rec and the span's LDS are free)
        // (diag build: the longest span of the launch, in ticks << 24 | its records in range >> 4)
        if (C3R_DBG(f.a) && tid == 0) atomicMax(&C3R_DBG(f.a)[18], ((wall_clock64() - t_span) << 24) | (unsigned long long)((unsigned)(rng.w - rng.z) >> 4));
    }
}

// The walk of the giant spans (ScanArgs::split_min records or more in range), ahead of k_fused_deep: a workgroup takes one slice of one span's records
// by ticket, walks it into LDS exactly as the span's own workgroup would (walk_records, ACCUM), and adds what it found to the span's slot — counts by
// global adds (a slot is 4.6 k words; a slice leaves most of them non-zero once), the longest deletion per position by max, the order flags of the
// 30-channel mode by store, the indel events into the slot's buffer behind a global cursor.  Every one of these is a sum, a max or a set: the order
// of the slices does not show.  k_fused_deep then starts such a span from its slot instead of walking (tile_columns, `giant`).
struct WalkArgs { const SpanRec *span_rec; int32_t *ticket; };
template <int C>
__global__ __launch_bounds__(DEEP_THREADS, DEEP_THREADS / 256) void k_deep_walk(const ScanArgs a, const WalkArgs d) {
    __shared__ int32_t s_cnt[TILE * C];
    __shared__ int32_t s_maxdel[TILE];
    __shared__ uint8_t s_odd[TILE];
    __shared__ int s_i;
    const int tid = (int)threadIdx.x;
    const int n = min(*a.n_help, GIANT_SLOTS * GIANT_MAX_HELP);
    for (;;) {
        if (tid == 0) s_i = atomicAdd(d.ticket, 1);
        __syncthreads();
        const int i = s_i;
        if (i >= n) return;
        const int4 h = a.help_list[i];                       // {span, slice, slices, slot}
        const int4 *rec = reinterpret_cast<const int4 *>(d.span_rec + h.x);
        const int4 r0 = rec[0], rng = rec[1], r2 = rec[2];
        if (!giant_on(a, rng.w - rng.z)) { __syncthreads(); continue; }
        for (int j = tid; j < TILE * C; j += DEEP_THREADS) s_cnt[j] = 0;
        if (tid < TILE) { s_maxdel[tid] = 0; s_odd[tid] = 0; }
        __syncthreads();
        const int t0 = r0.y - C3R_FLANK, t1 = min(r0.z + C3R_FLANK, r2.y);
        int lo, hi;
        giant_slice(rng.w - rng.z, h.y, h.z, lo, hi);
        lo += rng.z; hi += rng.z;
        int32_t *acc = a.giant_acc + (size_t)h.w * GIANT_STRIDE;
        TileLds s{s_cnt, nullptr, nullptr, nullptr, s_maxdel, nullptr, nullptr, s_odd, nullptr, acc + GIANT_META, -1, a.giant_ev + acc[GIANT_META + 1], acc[GIANT_META + 2]};
        walk_records<C, ACCUM, DEEP_THREADS, DEEP_WALK_UNR>(a, s, lo, hi, t0, t1, r0.w, nullptr);
        __syncthreads();
        for (int j = tid; j < TILE * C; j += DEEP_THREADS) { const int v = s_cnt[j]; if (v) atomicAdd(&acc[j], v); }
        if (tid < TILE) {
            if (s_maxdel[tid]) atomicMax(&acc[TILE * C3R_CH_PHASED + tid], s_maxdel[tid]);
            if (s_odd[tid]) acc[TILE * C3R_CH_PHASED + TILE + tid] = 1;
        }
        __syncthreads();
    }
}

// The allele counts of the giant spans, between k_deep_walk and k_fused_deep: the slices of a span's EVENTS (in the slot's buffer, arrival order) go to
// workgroups by ticket; an event claims or joins its allele's slot in the span's table — per position a region of 2 n slots, n the position's events,
// at twice the exclusive sum of the positions before it (the very offsets tile_columns derives from the merged counts) — and adds one.  One workgroup
// took 0.6 ms per span at mpileup's cap for this (34 k events, four dependent round trips each, on sixteen wavefronts).
template <int C>
__global__ __launch_bounds__(DEEP_THREADS, DEEP_THREADS / 256) void k_deep_alleles(const ScanArgs a, int32_t *ticket) {
    __shared__ int s_evoff[TILE], s_nev[TILE], s_wt[WAVES], s_i;
    const int tid = (int)threadIdx.x;
    const int n = min(*a.n_help, GIANT_SLOTS * GIANT_MAX_HELP);
    for (;;) {
        if (tid == 0) s_i = atomicAdd(ticket, 1);
        __syncthreads();
        const int i = s_i;
        if (i >= n) return;
        const int4 h = a.help_list[i];                       // {span, slice, slices, slot}
        const int32_t *acc = a.giant_acc + (size_t)h.w * GIANT_STRIDE;
        if (!giant_on(a, acc[GIANT_META + 2])) { __syncthreads(); continue; }          // ([2]: the events reserved = the span's records)
        int nev = 0;
        if (tid < TILE) { const int32_t *row = acc + tid * C; nev = row[C3R_I] + row[C3R_i] + row[C3R_D] + row[C3R_d]; }
        int n_ev;
        const int off = block_excl_scan<DEEP_THREADS>(nev, s_wt, &n_ev);
        if (tid < TILE) { s_evoff[tid] = off; s_nev[tid] = nev; }
        __syncthreads();
        {
            const int ev0 = acc[GIANT_META + 1];
            const EvRec *ev = a.giant_ev + ev0;
            uint2 *tab = a.giant_tab + 2 * (size_t)ev0;
            int e0, e1;
            giant_slice(n_ev, h.y, h.z, e0, e1);
#pragma unroll 1
            for (int e = e0 + tid; e < e1; e += DEEP_THREADS) {
                const EvRec me = ev[e];
                const int nn = 2 * s_nev[me.pl];
                uint2 *reg = tab + 2 * s_evoff[me.pl];
                uint32_t hs = (uint32_t)(ev_hash(me) >> 33) % (uint32_t)nn;
                for (int probe = 0; probe < nn; ++probe) {
                    const uint32_t rep = atomicCAS(&reg[hs].x, 0u, (uint32_t)e + 1u);
                    // (the allele's first event also says which channel the count is for)
                    if (rep == 0u) { atomicAdd(&reg[hs].y, 1u | ((me.ch == C3R_I1 ? 0u : me.ch == C3R_i1 ? 1u : me.ch == C3R_D1 ? 2u : 3u) << 30)); break; }
                    if (ev_equal(a, me, ev[rep - 1u])) { atomicAdd(&reg[hs].y, 1u); break; }
                    hs = hs + 1u == (uint32_t)nn ? 0u : hs + 1u;
                }
            }
        }
        __syncthreads();
    }
}

// Spans in list order -> where each span's candidates and tokens start in position order.  256 spans per workgroup, taken by ticket;
// the workgroups' sums meet by decoupled look-back (uniform, tiny work per workgroup: nobody waits for long).
//   span_base[i] = first candidate of span i;  totals = {candidates, tokens}
__global__ __launch_bounds__(256) void k_order_spans(const int4 *span_info, const int32_t *n_list_p, int32_t *ticket, unsigned long long *ostate,
                                                     int32_t *span_base, int32_t *totals) {
    __shared__ int s_b, s_w[WAVES];
    __shared__ unsigned long long s_excl;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = *n_list_p;
    if (tid == 0) s_b = atomicAdd(ticket, 1);
    __syncthreads();
    const int b = s_b;
    if (b * 256 >= n) { if (n == 0 && b == 0 && tid == 0) { totals[0] = 0; totals[1] = 0; } return; }
    const int i = b * 256 + tid;
    int4 v = make_int4(0, 0, 0, 0);
    if (i < n) v = span_info[i];
    int tc, tt;
    const int ec = block_excl_scan(v.y, s_w, &tc);
    (void)block_excl_scan(v.z, s_w, &tt);
    if (wave == 0) {
        const unsigned long long mine = ((unsigned long long)(unsigned)tc << 32) | (unsigned)tt;
        const unsigned long long excl = lb_lookback(ostate, b, mine);
        if (lane == 0) {
            s_excl = excl;
            if ((b + 1) * 256 >= n) { totals[0] = (int32_t)((excl + mine) >> 32); totals[1] = (int32_t)(unsigned)(excl + mine); }
        }
    }
    __syncthreads();
    if (i < n) span_base[i] = (int)(s_excl >> 32) + ec;
}

// Everything small about a candidate, in position order; one wavefront per arrived row.
//   i = span_base[span] + (row - first row of the span):  sites[i] (tok_off = the span's first token + the tokens of its earlier
//   candidates), cand_idx[i] = slot, win_idx[i] = row_base + row
struct FinalizeArgs {
    const CandMeta *meta; const int4 *span_info; const int32_t *span_base; const unsigned long long *alloc; int32_t n_shards, shard_rows; const int32_t *overflow;
    const uint8_t *ref; int32_t ref_beg0, ref_len;
    c3r_site_t *sites; int32_t *cand_idx; int32_t *win_idx; int32_t row_base; int32_t tok_base;
};
__global__ __launch_bounds__(256) void k_finalize_sites(const FinalizeArgs g) {
    if (*g.overflow) return;                                  // (some rows were never written: the host repeats the scan with larger buffers)
    const int n = g.n_shards * g.shard_rows;                  // the scan's row space; a shard's rows beyond what it handed out are holes
    // sixteen lanes per arrived row: the 52-byte site record leaves as thirteen dwords of one store instruction
    static_assert(sizeof(c3r_site_t) == 52 && offsetof(c3r_site_t, ref33) == 8 && offsetof(c3r_site_t, n_tok) == 44, "c3r_site_t layout");
    const int gl = (int)(threadIdx.x & 15), ng = (int)(gridDim.x * (blockDim.x >> 4));
    for (int row = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 4); row < n; row += ng) {
        const int sh = row / g.shard_rows;
        if (row - sh * g.shard_rows >= (int)(unsigned)g.alloc[sh * ALLOC_STRIDE]) continue;
        const CandMeta m = g.meta[row];
        const int4 si = g.span_info[m.span];
        const int i = g.span_base[m.span] + (row - si.x);
        const int pc = m.pos0;
        if (g.sites && gl < 13) {
            uint32_t v;
            if (gl == 0) v = (uint32_t)(pc + 1);
            else if (gl == 1) v = (uint32_t)m.depth;
            else if (gl == 11) v = (uint32_t)m.ncov;
            else if (gl == 12) v = (uint32_t)(g.tok_base + si.w + m.tpre);
            else {
                v = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = 4 * (gl - 2) + k;           // character of ref33[36]: 33 reference bases ('A' beyond the contig), then NULs
                    const int rp = pc - C3R_FLANK + c - g.ref_beg0;
                    const uint32_t ch = c >= C3R_WINDOW ? 0u : (rp >= 0 && rp < g.ref_len) ? (uint32_t)g.ref[rp] : (uint32_t)'A';
                    v |= ch << (8 * k);
                }
            }
            reinterpret_cast<uint32_t *>(&g.sites[i])[gl] = v;
        }
        if (gl == 0) {
            if (g.cand_idx) g.cand_idx[i] = m.slot;
            g.win_idx[i] = g.row_base + row;
        }
    }
}

// rows of a [n][row_ints] int32 table through an index (c3r_get_tensors: windows in position order), and the identity index the
// column-store path leaves for its candidates
// (x16: the source rows are int16 — the resident windows; idx null: row i)
__global__ __launch_bounds__(256) void k_gather_rows(const void *src, const int32_t *idx, int n, int row_ints, int32_t *dst, int x16) {
    const int lane = (int)(threadIdx.x & 63), nw = (int)(gridDim.x * (blockDim.x >> 6));
    for (int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); i < n; i += nw) {
        const size_t row = idx ? (size_t)idx[i] : (size_t)i;
        int32_t *d_ = dst + (size_t)i * row_ints;
        if (x16) {
            const int16_t *s_ = (const int16_t *)src + row * row_ints;
            for (int k = lane; k < row_ints; k += 64) d_[k] = (int32_t)s_[k];
        } else {
            const int32_t *s_ = (const int32_t *)src + row * row_ints;
            for (int k = lane; k < row_ints; k += 64) d_[k] = s_[k];
        }
    }
}
__global__ __launch_bounds__(256) void k_iota(int32_t *dst, int n, int base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = base + i;
}
// c3r_get_tokens: the tokens of site 0, then of site 1, ... (the fused path writes a span's tokens where they arrive); n_tok per site first
__global__ __launch_bounds__(256) void k_site_ntok(const c3r_site_t *sites, int n, int32_t *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sites[i].n_tok;
    else if (i == n) out[i] = 0;
}
// A site's tokens lie in the order they arrived (tile_tokens); whoever takes them out puts them into BAM order = by read index (a read has
// at most one token per site).  One wavefront per site: rank of token k0 + lane among the site's n tokens, and among those that carry an
// indel (the packed form keeps the indel records apart).  Sites hold two or three tokens; a deep site's thousands cost n^2 / 64 steps.
__device__ __forceinline__ void site_tok_rank(const c3r_token_t *s_, int n, uint32_t my_read, int my_k, int lane, int &rank, int &rank_ind) {
    rank = 0; rank_ind = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        uint32_t oi = 0xffffffffu; int oind = 0;
        if (j < n) { oi = s_[j].read_idx; oind = s_[j].indel != 0 ? 1 : 0; }
        const int m = min(64, n - j0);
        for (int l = 0; l < m; ++l) {
            const uint32_t x = (uint32_t)__shfl((int)oi, l, 64);
            const int xi = __shfl(oind, l, 64);
            // (equal read indices cannot occur — one token per read and site — but a rank must be a permutation whatever the input)
            if (x < my_read || (x == my_read && j0 + l < my_k)) { ++rank; rank_ind += xi; }
        }
    }
}
__global__ __launch_bounds__(256) void k_export_tokens(const c3r_site_t *sites, const int32_t *dst_off, int n, const c3r_token_t *tok, c3r_token_t *out) {
    const int lane = (int)(threadIdx.x & 63), nw = (int)(gridDim.x * (blockDim.x >> 6));
    for (int i = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6); i < n; i += nw) {
        const c3r_token_t *s_ = tok + sites[i].tok_off;
        int4 *d_ = reinterpret_cast<int4 *>(out + dst_off[i]);
        const int nt = sites[i].n_tok;
        for (int k0 = 0; k0 < nt; k0 += 64) {
            const int k = k0 + lane;
            int4 v = make_int4(0, 0, 0, 0);
            if (k < nt) v = reinterpret_cast<const int4 *>(s_)[k];
            int rank, rind;
            site_tok_rank(s_, nt, k < nt ? (uint32_t)v.x : 0xffffffffu, k, lane, rank, rind);
            if (k < nt) d_[rank] = v;
        }
    }
}


// ---- tokens for the row decoder, packed.  A token is 16 bytes (c3r_token_t) but the decoder reads the read index, the indel length
// and the query offset only of the few tokens that carry an indel; of all others it needs the base code.  One wavefront per site
// turns its tokens into one byte each (base code | 0x20 on the reverse strand | 0x80 when an indel record follows) and appends the indel records (12 bytes,
// token order kept by a ballot prefix) to a contiguous range it draws from one counter: a 250-Mb contig copies out ~60 MB instead
// of 760 MB.
struct TokRec { uint32_t read_idx; int32_t indel; uint32_t qpos; uint32_t del_after; };
static_assert(sizeof(TokRec) == 16, "TokRec must be 16 bytes");

__global__ __launch_bounds__(256) void k_pack_tokens(const c3r_site_t *__restrict__ sites, const c3r_token_t *__restrict__ tok, int64_t n_sites,
                                                      uint8_t *__restrict__ bytes, TokRec *__restrict__ recs, uint32_t *__restrict__ rec_off,
                                                      unsigned long long *__restrict__ counter) {
    const int lane = threadIdx.x & 63;
    const int64_t site = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (site >= n_sites) return;
    const uint32_t off = sites[site].tok_off;
    const int n_tok = sites[site].n_tok;
    int cnt = 0;
    for (int i = lane; i < n_tok; i += 64) cnt += tok[off + i].indel != 0;
    for (int d = 32; d; d >>= 1) cnt += __shfl_xor(cnt, d);
    uint32_t base = 0;
    if (lane == 0) {
        base = cnt ? (uint32_t)atomicAdd(counter, (unsigned long long)cnt) : 0u;
        rec_off[site] = base;
    }
    base = __shfl(base, 0);
    for (int i0 = 0; i0 < n_tok; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < n_tok;
        c3r_token_t t{};
        if (valid) t = tok[off + i];
        int rank, rind;
        site_tok_rank(tok + off, n_tok, valid ? t.read_idx : 0xffffffffu, i, lane, rank, rind);     // BAM order
        const bool f = valid && t.indel != 0;
        if (valid) bytes[off + rank] = (uint8_t)((t.base & 31) | (t.rev ? 0x20 : 0) | (f ? 0x80 : 0));
        if (f) recs[base + (uint32_t)rind] = TokRec{t.read_idx, t.indel, t.qpos, t.del_after};
    }
}

// ---- row snapshots without the sites the decoder prints nothing for.  Without --show_ref a site whose probabilities take the early
// RefCall exit of the decoder (clair3_rna/call_variants.py:540-542: P(0/0) >= 0.5 and P(gt21 = ref ref) >= 0.5, decode.hpp call_site) leaves
// no row, whatever its alt_info says, and so does a site whose reference class wins the first round of the decoder's arg-max — together 97 %
// of the candidates of a trained model.  k_row_keep applies those very tests, an exclusive scan turns the flags into row numbers, and k_pack_rows moves only the kept sites: site record,
// probabilities, packed tokens.  keep[n] is the scan's sentinel.
__global__ __launch_bounds__(256) void k_row_keep(const c3r_site_t *__restrict__ sites, const float *__restrict__ probs, int n, int32_t *__restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    if (i == n) { keep[i] = 0; return; }
    const char c = sites[i].ref33[C3R_FLANK];
    // iupac2acgt (decode.hpp): ACGTURYSWKMBDHVN -> ACGTTACCAGACAAAA, anything else A; then gt21 of the homozygous reference genotype
    int rr = 0;                                                               // AA
    if (c == 'C' || c == 'Y' || c == 'S' || c == 'B') rr = 4;                 // CC
    else if (c == 'G' || c == 'K') rr = 7;                                    // GG
    else if (c == 'T' || c == 'U') rr = 9;                                    // TT
    const float *p = probs + (size_t)i * C3R_NPROB;
    const float z0 = p[21], z1 = p[22], z2 = p[23];
    bool ref_call = z0 >= 0.5f && p[rr] >= 0.5f;
    // ... and the first round of the decoder's arg-max (call_site: `top == p_ref`): when no class product exceeds P(0/0) * P(gt21 = ref ref)
    // the site is a RefCall before alt_info is looked at.  The same float32 products (one IEEE multiply each); p_ref must be a normal
    // number so that a flushed denormal on either side cannot turn the comparison.
    if (!ref_call) {
        const float p_ref = z0 * p[rr];
        float top = 0.f;
        const int homo[4] = {0, 4, 7, 9}, het[6] = {1, 2, 3, 5, 6, 8};
#pragma unroll
        for (int k = 0; k < 4; ++k) top = fmaxf(top, z1 * p[homo[k]]);
#pragma unroll
        for (int k = 0; k < 6; ++k) top = fmaxf(top, z2 * p[het[k]]);
        top = fmaxf(top, fmaxf(z1 * p[15], z2 * p[15]));                      // InsIns
        top = fmaxf(top, fmaxf(z1 * p[10], z2 * p[10]));                      // DelDel
#pragma unroll
        for (int k = 0; k < 4; ++k) top = fmaxf(top, fmaxf(p[16 + k] * z2, p[11 + k] * z2));      // base + Ins, base + Del
        top = fmaxf(top, z2 * p[20]);                                         // InsDel
        ref_call = p_ref >= 1e-30f && p_ref >= top;
    }
    keep[i] = ref_call ? 0 : 1;
}
// idx: the exclusive scan of the keep flags ([n + 1]); counters: [0] indel records, [1] token bytes handed out
__global__ __launch_bounds__(256) void k_pack_rows(const c3r_site_t *__restrict__ sites, const c3r_token_t *__restrict__ tok, const float *__restrict__ probs,
                                                    const int32_t *__restrict__ idx, int64_t n_sites, c3r_site_t *__restrict__ sites_c, float *__restrict__ probs_c,
                                                    uint8_t *__restrict__ bytes, TokRec *__restrict__ recs, uint32_t *__restrict__ rec_off,
                                                    unsigned long long *__restrict__ counters) {
    const int lane = threadIdx.x & 63;
    const int64_t site = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (site >= n_sites) return;
    const int j = idx[site];
    if (idx[site + 1] == j) return;                                           // not kept
    const uint32_t off = sites[site].tok_off;
    const int n_tok = sites[site].n_tok;
    int cnt = 0;
    for (int i = lane; i < n_tok; i += 64) cnt += tok[off + i].indel != 0;
    for (int d = 32; d; d >>= 1) cnt += __shfl_xor(cnt, d);
    uint32_t base = 0, bbase = 0;
    if (lane == 0) {
        base = cnt ? (uint32_t)atomicAdd(&counters[0], (unsigned long long)cnt) : 0u;
        bbase = (uint32_t)atomicAdd(&counters[1], (unsigned long long)n_tok);
        rec_off[j] = base;
    }
    base = __shfl(base, 0); bbase = __shfl(bbase, 0);
    static_assert(sizeof(c3r_site_t) == 52 && offsetof(c3r_site_t, tok_off) == 48, "c3r_site_t layout");
    if (lane < 13) reinterpret_cast<uint32_t *>(&sites_c[j])[lane] = lane == 12 ? bbase : reinterpret_cast<const uint32_t *>(&sites[site])[lane];
    if (lane < C3R_NPROB) probs_c[(size_t)j * C3R_NPROB + lane] = probs[(size_t)site * C3R_NPROB + lane];
    for (int i0 = 0; i0 < n_tok; i0 += 64) {
        const int i = i0 + lane;
        const bool valid = i < n_tok;
        c3r_token_t t{};
        if (valid) t = tok[off + i];
        int rank, rind;
        site_tok_rank(tok + off, n_tok, valid ? t.read_idx : 0xffffffffu, i, lane, rank, rind);     // BAM order
        const bool f = valid && t.indel != 0;
        if (valid) bytes[bbase + rank] = (uint8_t)((t.base & 31) | (t.rev ? 0x20 : 0) | (f ? 0x80 : 0));
        if (f) recs[base + (uint32_t)rind] = TokRec{t.read_idx, t.indel, t.qpos, t.del_after};
    }
}

}  // namespace c3r
