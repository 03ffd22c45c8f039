/* c3r.h — C-ABI of libc3r.so, the MI355X-native pileup variant-calling hot path for Clair3-RNA.
 *
 * The reference has NO plugin / FFI seam for this path: the boundary is a pair of sub-processes joined
 * by a text pipe, launched per (contig, chunk) by clair3_rna/call_var_bam.py:288-295.  This header is
 * therefore the seam the build defines; each entry point names the reference code it
 * replaces.  INTEGRATION.md shows the ctypes stub a maintainer adds to call_var_bam.py.
 *
 * Conventions: plain C, no exceptions across the boundary.  Every call returns 0 on success or a
 * negative C3R_E* code; c3r_last_error(ctx) returns a human-readable message.  The caller owns all
 * host buffers; the library owns all device memory inside the opaque c3r_ctx.  One context = one
 * GPU = one HIP stream (plus, inside the library, a second one on which the tensor build's deep-span kernel runs beside the
 * other); contexts are independent (one process or thread per GPU, no collectives: chunks are independent work items,
 * run_clair3_rna:681-706).  Several contexts of one process may share a device, each driven by its own host thread (two
 * pipelined contexts: one uploads while the other computes): c3r_load_reads lets only one of them upload at a time per device, so
 * that they stay out of phase on the PCIe link.  A single context is not re-entrant.
 *
 * There is NO CPU fallback: every compute entry point fails with C3R_ENODEVICE when no gfx950 device
 * is usable.
 */
#ifndef C3R_H
#define C3R_H

#include <stddef.h>
#include <stdint.h>

#include "c3r_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define C3R_OK 0
#define C3R_EINVAL (-1)       /* bad argument / call order */
#define C3R_ENODEVICE (-2)    /* no usable HIP device */
#define C3R_EHIP (-3)         /* a HIP runtime call failed */
#define C3R_ENOMEM (-4)
#define C3R_EUNSUPPORTED (-5) /* valid in the reference, not implemented on the GPU path yet */
#define C3R_EOVERFLOW (-6)    /* caller buffer too small */

typedef struct c3r_ctx c3r_ctx;
typedef struct c3r_rows c3r_rows;   /* a batch's decode inputs on the host, detached from the context (c3r_rows_begin) */

/* ---- lifetime ------------------------------------------------------------------------------ */
/* Library version string, e.g. "c3r 0.1 (gfx950)". */
const char *c3r_version(void);
/* Create a context on HIP device `device_id`.  `stream` may be NULL (the library creates its own
 * non-blocking stream) or an existing hipStream_t the caller wants the work ordered on. */
int c3r_create(int device_id, void *stream, c3r_ctx **out);
void c3r_destroy(c3r_ctx *ctx);
/* c3r_destroy keeps up to two of the largest device blocks (the 8.9-GB layer-1 output of a full network slice) for the process's next
 * context: the driver clears freed memory before it hands it out again, 0.8 s for a block of that size.  c3r_trim() gives them back
 * (for a host application that destroys its contexts to return HBM to other users of the GPU); returns the bytes released.
 * C3R_NO_BLOCK_CACHE=1 turns the cache off. */
int64_t c3r_trim(void);
const char *c3r_last_error(const c3r_ctx *ctx);
/* Block until all work queued on the context's stream is complete. */
int c3r_synchronize(c3r_ctx *ctx);
/* The hipStream_t the context launches on (for event timing by the caller). */
void *c3r_stream(c3r_ctx *ctx);

/* ---- configuration ------------------------------------------------------------------------- */
/* Fill `p` with the defaults run_clair3_rna forwards to call_var_bam (run_clair3_rna:684-705;
 * shared/param_p.py:20,41,88-90). */
void c3r_default_params(c3r_params_t *p);
/* Replaces the argparse surface of src/create_tensor_pileup.py:660-778 that affects tensors.
 * splice_padding=1 (src/create_tensor_pileup.py:573-593) is supported, also together with head_tail=1
 * (including the reference's shared pre-fill column, [[0]*C]*33, that padding edits in place). */
int c3r_set_params(c3r_ctx *ctx, const c3r_params_t *p);

/* ---- inputs -------------------------------------------------------------------------------- */
/* Hand over one contig's aligned reads, sorted by pos (BAM order), as flat host-resident records.  Replaces the BAM side of
 * `samtools mpileup <bam> -r ...` (src/create_tensor_pileup.py:446-451): the records are copied to the device as they are and
 * everything htslib's per-read CIGAR cursor would do while streaming them — dropping pads / hard clips / empty ops, folding
 * = and X, the split at N ops into aligned segments with their reference / query offsets, the segment order, the expanded op
 * table — happens there (csrc/reads_kernels.hpp), with one host synchronisation to report sizes and validation errors.
 * Filtering by excl_flags / min_mq happens on the device too.  cigars: BAM-encoded ops; seq4: 4-bit packed bases.
 * This call is part of the measured path (bench.py times it inside every step). */
int c3r_load_reads(c3r_ctx *ctx, const c3r_read_t *reads, int64_t n_reads,
                   const uint32_t *cigars, int64_t n_cigar_ops, const uint8_t *seq4, int64_t n_seq_bytes);
/* Page-locked host memory for the arrays handed to c3r_load_reads: from such buffers the three uploads are truly asynchronous
 * DMA transfers (~55 GB/s); from ordinary pageable memory the HIP runtime stages them through its own buffer first.  Optional —
 * any host pointer works.  The caller frees with c3r_host_free. */
void *c3r_host_alloc(size_t bytes);
void c3r_host_free(void *p);
/* Upload the reference slice covering the region.  ref[0] is 1-based position `ref_start`;
 * replaces reference_sequence_from / `samtools faidx` (shared/utils.py:168-194,
 * src/create_tensor_pileup.py:424-428).  Upper-cased on upload like the reference does. */
int c3r_set_reference(c3r_ctx *ctx, int64_t ref_start, const char *ref, int64_t len);
/* The same for a slice the caller already holds UPPER-CASED (c3r_fasta_fetch, include/c3r_io.h) and keeps alive and unchanged until
 * the context has been given another reference AND every row snapshot (c3r_rows_begin) taken meanwhile has been freed: no host copy
 * is made — the upload reads the caller's bytes and so does the decoder.  (c3r_set_reference's pass over a 250-Mb chromosome is
 * 40-250 ms of the calling thread.) */
int c3r_set_reference_view(c3r_ctx *ctx, int64_t ref_start, const char *ref_upper, int64_t len);
/* Optional interval filters, 0-based half-open, for the current contig.  which=0: `-l` column
 * filter (mpileup -l extend_bed, src/create_tensor_pileup.py:443); which=1: confident-bed candidate
 * filter (is_region_in, src/create_tensor_pileup.py:551-554).  n=0 clears. */
int c3r_set_bed(c3r_ctx *ctx, int which, const int32_t *start_end_pairs, int64_t n);
/* Genotyping mode site list (--vcf_fn; src/create_tensor_pileup.py:399-407,555-556), 1-based. */
int c3r_set_sites(c3r_ctx *ctx, const int32_t *sites, int64_t n);

/* ---- tensor build (A1-A5) ------------------------------------------------------------------ */
/* Phase 1+2: CIGAR walk over the reads overlapping [ctg_start-33, ctg_end+33] (1-based, clamped
 * at 1; src/create_tensor_pileup.py:411-415), per-position channel counts, candidate gates, window
 * selection, window gather with the depth>216 rescale (clair3_rna/utils.py:88-92).  Tensors and
 * site records stay resident on the device.  Returns the number of emitted candidates.
 * max_depth (samtools mpileup -d, default 8000): htslib's rule — a read that is not the first one pushed for its start
 * position is discarded while more than max_depth reads are live — is applied per region before the walk.
 * A scan that meets a position covered by more than 32,767 kept reads (only with the cap off or raised) is repeated with 32-bit resident windows,
 * which the context keeps from then on; C3R_EOVERFLOW only when the batch already holds 16-bit windows. */
int c3r_pileup_scan(c3r_ctx *ctx, int64_t ctg_start, int64_t ctg_end, int64_t *n_candidates);
/* The same for several regions (the chunks of one contig) in ONE set of kernel launches: results are exactly those of
 * n_regions successive c3r_pileup_scan calls in batch mode — candidates of region 0 first, each region with its own
 * +-33 bp halo, head/tail flush and window rule — but the chip sees ~20 k tiles at once instead of 13 latency-bound
 * launches of ~1.5 k.  *n_candidates receives the total. */
int c3r_pileup_scan_regions(c3r_ctx *ctx, int32_t n_regions, const int64_t *ctg_starts, const int64_t *ctg_ends, int64_t *n_candidates);
/* Batch mode: between c3r_batch_begin and c3r_batch_end every scan APPENDS its candidates (tensors, sites,
 * tokens) to the device-resident batch instead of replacing it, so that the chunks of a whole contig go through
 * the network in one launch per layer (the reference batches 200 sites, shared/param_p.py:51; 288 GB of HBM let
 * us batch a whole contig).  c3r_batch_count returns the totals; c3r_infer(NULL, n_total) and the c3r_get_*
 * calls then cover all accumulated candidates in scan order. */
int c3r_batch_begin(c3r_ctx *ctx);
int c3r_batch_end(c3r_ctx *ctx);
int c3r_batch_count(c3r_ctx *ctx, int64_t *n_sites, int64_t *n_tokens);
/* Copy out what the last scan (or the current batch) produced.  Any pointer may be NULL.  tensors: int32 [n][33][C]
 * row-major (the 594/990 integers of a create_tensor line, after the A5 rescale when
 * `rescaled` != 0, raw otherwise); sites: [n]; tokens: [n_tokens] (see c3r_token_count), site after site (tok_off / n_tok of
 * c3r_get_sites), each site's in BAM order. */
int c3r_get_tensors(c3r_ctx *ctx, int rescaled, int32_t *tensors, int64_t cap_sites);
int c3r_get_sites(c3r_ctx *ctx, c3r_site_t *sites, int64_t cap_sites);
int c3r_token_count(c3r_ctx *ctx, int64_t *n_tokens);
int c3r_get_tokens(c3r_ctx *ctx, c3r_token_t *tokens, int64_t cap_tokens);
/* mpileup_compat = 1: the insertions of the loaded reads that hold pads (c3r_padins_t, sorted by read and query offset) — what a caller that
 * rebuilds alt_info from c3r_get_tokens needs beside the read bases (`+3T*T`, samtools >= 1.11).  out may be NULL to ask for the count; empty
 * for every CIGAR an aligner emits.  Valid after c3r_load_reads / c3r_set_params. */
int c3r_get_pad_insertions(c3r_ctx *ctx, c3r_padins_t *out, int64_t cap, int64_t *n);
/* Debug / parity: per-position columns of the last scan.  cols: int32 [n_pos][C]; depth: int32
 * [n_pos]; flags: uint8 [n_pos] (bit0 = row exists, bit1 = candidate gate passed, bit2 = emitted).
 * Position i is 1-based ctg position region_start + i where region_start is returned. */
int c3r_get_columns(c3r_ctx *ctx, int64_t *region_start, int64_t *n_pos, int32_t *cols, int32_t *depth,
                    uint8_t *flags, int64_t cap_pos);

/* ---- network (A6/A7) ----------------------------------------------------------------------- */
/* Upload network weights: a flat fp32 blob in Keras layout (documented in DESIGN.md §weights):
 * per LSTM layer and direction K[in,4H], R[H,4H], b[4H] (gate order i,f,c,o); then L4, L5_1, L5_2,
 * Y_gt21, Y_genotype (W[in,out], b[out]).  Replaces m.load_weights (clair3_rna/call_variants.py:1472).
 * `channels` is 18 or 30 and must match c3r_set_params. */
int c3r_load_weights(c3r_ctx *ctx, const float *blob, int64_t n_floats, int channels);
int64_t c3r_weight_count(int channels);
/* Arithmetic of the network GEMMs: 0 = fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32); 1 (default) = split-f16:
 * every fp32 operand carried as hi + lo halves, products hi*hi + hi*lo + lo*hi accumulated in fp32 on the f16 matrix
 * pipe (fp32-equivalent: both modes meet the 1e-4 probability tolerance against the fp32 oracle);
 * 2 = f16 main term + both correction terms on the block-scaled fp8 pipe (v_mfma_scale_f32_32x32x64_f8f6f4): ~1.15x the
 * throughput of mode 1, max |dP| 2-3e-5 on N(0, 0.05) weights but NOT robust to weights of 2-3x that norm — opt-in;
 * 3 = auto: mode 2 if it agrees with mode 1 to 4e-5 on 2048 calibration windows run through the loaded weights (measured at
 * c3r_load_weights / here), else mode 1.  Modes 2 and 3 need 18- or 30-channel weights like the others. */
int c3r_set_precision(c3r_ctx *ctx, int mode);
/* The mode the network runs in (after "auto" and the split-f16 guard have decided) and the fp8 calibration's max |dP| (-1: not measured). */
int c3r_get_precision(c3r_ctx *ctx, int *mode_in_use, double *calibration_err);
/* The guard of the split-f16 arithmetic itself.  Nothing in clair3_rna/model.py:126-172 bounds a trained model's weights, and f16 ends at
 * 65504: c3r_load_weights refuses non-finite values, packs every layer (LSTM 1, LSTM 2, L4) with the largest power-of-two scale <= 2^12
 * that keeps 2^s max|w| <= 2^15 (scale_log2[3]; 12 12 12 for ordinary weights), and runs 2048 pileup-shaped calibration windows through the
 * fp32 MFMA path and the split-f16 path: f16_err = max |dP| between them.  Above 1e-4 (the parity tolerance: the two paths drift apart with
 * the weights' gain exactly as each drifts from an fp32 CPU evaluation) a request for mode 1, 2 or 3 is served by mode 0
 * (fp32 MFMA, about a third of the speed), *fell_back = 1, and a warning goes to stderr.  Any pointer may be NULL. */
int c3r_get_precision_guard(c3r_ctx *ctx, double *f16_err, int32_t *scale_log2, int *fell_back);
/* Optional: size the network's device buffers for batches of up to n_sites candidates now (after c3r_load_weights) instead of
 * inside the first c3r_infer.  The layer-1 output of a full 2^18-site slice is 8.9 GB, and a first hipMalloc of that size takes
 * 0.25-0.4 s: a caller that is still waiting for its input (a BAM fetch) spends them here for free. */
int c3r_reserve(c3r_ctx *ctx, int64_t n_sites);
/* Forward pass over tensors.  tensors==NULL: use the device-resident tensors of the last scan.
 * Otherwise `tensors` is a host int32 [n][33][C] array.  probs (host, [n][24]) may be NULL to keep
 * the result on the device only.  Replaces m.predict_on_batch (clair3_rna/call_variants.py:1505). */
int c3r_infer(c3r_ctx *ctx, const int32_t *tensors, int64_t n, float *probs);
/* Fetch the [n][24] probabilities of the last c3r_infer(…, probs = NULL) — lets the caller queue the network on this
 * context's stream, do other work (e.g. the tensor build of the next contig on a second context), and collect later. */
int c3r_get_probs(c3r_ctx *ctx, float *probs, int64_t n);

/* ---- decode on the host (A8) --------------------------------------------------------------- */
/* Probabilities -> genotype / ALT / QUAL -> VCF text rows, replacing batch_output / output_with / output_from
 * (clair3_rna/call_variants.py:1077-1392, :684-1020).  c3r_call_rows works on the resident candidates after
 * c3r_infer: it rebuilds each site's ordered alt_info from the per-read tokens (src/create_tensor_pileup.py:221-261,
 * 595-596), decodes on host threads and caches the '\n'-terminated rows; c3r_get_rows copies them out.
 * qual < 0 means "no quality cut-off" (--qual None); show_ref != 0 keeps RefCall rows (--showRef). */
int c3r_call_rows(c3r_ctx *ctx, const char *ctg, int qual, int show_ref, int64_t *out_len, int64_t *n_rows);
int c3r_get_rows(c3r_ctx *ctx, char *out, int64_t cap);
/* The same in two steps, so that the context can go on to the next contig while host threads decode this one: c3r_rows_begin (after
 * c3r_infer) copies sites, tokens, probabilities and the read bases to the host and returns a snapshot; from then on the context may load
 * new reads / a new reference (the snapshot keeps its contig's reference buffer alive).  c3r_rows_decode / c3r_rows_get work on the
 * snapshot from ANY thread, no GPU involved.  c3r_rows_free must be called before c3r_destroy of the context that made the snapshot. */
int c3r_rows_begin(c3r_ctx *ctx, c3r_rows **out);
/* The same with two ways of moving less.  drop_ref_calls != 0: the snapshot holds only the sites that can print a row when the decoder runs
 * WITHOUT show_ref — a site whose probabilities take the decoder's early RefCall exit (P(0/0) >= 0.5 and P(gt21 = ref ref) >= 0.5,
 * clair3_rna/call_variants.py:540-542) or whose reference class wins the first round of its arg-max (:730-760: no class product exceeds
 * P(0/0) * P(ref ref)) prints nothing then, whatever its alt_info holds, so neither its record nor its tokens nor its
 * probabilities leave the device (a trained model calls ~97 % of the candidates that way); decoding such a snapshot with show_ref != 0 is
 * an error of the caller (those rows are gone).  host_reads / host_seq: the arrays that were handed to c3r_load_reads for the loaded contig,
 * if the caller keeps them alive and unchanged until c3r_rows_free — the decoder then reads inserted bases from them in place instead of
 * from a copy fetched back from the device (both NULL: fetched back, as c3r_rows_begin does). */
int c3r_rows_begin_ex(c3r_ctx *ctx, int drop_ref_calls, const c3r_read_t *host_reads, const uint8_t *host_seq, c3r_rows **out);
int c3r_rows_decode(c3r_rows *rows, const char *ctg, int qual, int show_ref, int64_t *out_len, int64_t *n_rows);
int c3r_rows_get(c3r_rows *rows, char *out, int64_t cap);
void c3r_rows_free(c3r_rows *rows);
/* The same decoder on caller-supplied text (no GPU, no context): n sites, alt_info strings "<depth>-<k v ...>".
 * Returns C3R_EOVERFLOW (with *out_len = bytes needed, excluding the NUL) when `out` is too small. */
int c3r_decode_text(const char *ctg, int64_t n, const int32_t *pos, const char *ref33s, int ref33_stride, const char *const *alt_infos,
                    const float *probs, int qual, int show_ref, char *out, int64_t cap, int64_t *out_len);

/* ---- measurement --------------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by HIP events on the context's stream and the
 * elapsed times are accumulated per kernel name. */
int c3r_set_profiling(c3r_ctx *ctx, int enabled);
int c3r_reset_kernel_stats(c3r_ctx *ctx);
/* Writes up to cap entries; returns the number of distinct kernels via *n.  names[i] points into
 * storage owned by the context. */
int c3r_get_kernel_stats(c3r_ctx *ctx, const char **names, double *total_ms, int64_t *launches, int cap, int *n);

#ifdef __cplusplus
}
#endif
#endif /* C3R_H */
