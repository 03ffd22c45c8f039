/* c3r_io.h — C ABI of libc3r_io.so: BAM/BGZF/BAI -> the flat read records c3r_load_reads takes; merged, bgzipped,
 * tabix-indexed VCF out.
 *
 * Replaces the input side of the reference's `samtools mpileup <bam> -r ctg:beg-end` subprocess
 * (src/create_tensor_pileup.py:436-451; region set-up :409-428): open the BAM, use its .bai to find the
 * alignments overlapping the region, and hand back exactly the fields the tensor builder reads — core.pos,
 * flag, MAPQ, CIGAR (incl. the CG:B,I long-CIGAR convention), 4-bit SEQ and the HP aux tag
 * (--output-extra HP, create_tensor_pileup.py:440).  Flag / MAPQ filtering is NOT done here: it belongs to
 * the scan (c3r_params_t.excl_flags / min_mq), as it belongs to mpileup in the reference.
 *
 * Host-only C++ (zlib), no GPU, no torch types.  Every call returns 0 or a negative C3R_E* code
 * (include/c3r.h); c3r_bam_last_error gives the message.
 */
#ifndef C3R_IO_H
#define C3R_IO_H

#include <stdint.h>

#include "c3r_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c3r_bam c3r_bam;

/* Open `path`, parse the header; loads `<path>.bai` or `<path without .bam>.bai` when present.
 * n_threads: BGZF inflate threads for whole-file loads (<=0: hardware concurrency). */
int c3r_bam_open(const char *path, int n_threads, c3r_bam **out);
void c3r_bam_close(c3r_bam *b);
const char *c3r_bam_last_error(c3r_bam *b);

int c3r_bam_n_contigs(c3r_bam *b);
/* name points into the handle (valid until close) */
int c3r_bam_contig(c3r_bam *b, int i, const char **name, int64_t *length);
int c3r_bam_has_index(c3r_bam *b);
/* How much work contig i holds, from the index (-1: not known — no index, or it has no metadata pseudo-bin): the number of mapped
 * reads (BAI pseudo-bin 37450) and the compressed bytes its records span.  What the sample's contigs are dealt to the GPUs by
 * (the reference fans out by chunk: run_clair3_rna:441-449,681-706; SURVEY.md 8e: "by read count ... from the read index"). */
int c3r_bam_contig_weight(c3r_bam *b, int i, int64_t *n_mapped, int64_t *file_bytes);

/* Collect the alignments of `contig` overlapping the 0-based half-open region [beg0, end0) in file
 * (coordinate) order; end0 <= 0 means the whole contig.  Uses the .bai when loaded, else inflates the
 * whole file (in parallel) and filters.  Unmapped-position records (pos < 0) and records without CIGAR are
 * skipped, as mpileup never sees them.  Sizes of the three flat arrays are returned. */
int c3r_bam_fetch(c3r_bam *b, const char *contig, int64_t beg0, int64_t end0, int64_t *n_reads, int64_t *n_cigar,
                  int64_t *n_seq_bytes);
/* Copy the result of the last fetch: reads[n_reads] (cigar_off / seq_off index the other two arrays),
 * cigar[n_cigar] BAM-encoded (len<<4|op), seq[n_seq_bytes] 4-bit packed, (l_seq+1)/2 bytes per read. */
int c3r_bam_copy(c3r_bam *b, c3r_read_t *reads, uint32_t *cigar, uint8_t *seq);

/* `samtools index` equivalent (samtools is not a dependency of this path): write a .bai for a
 * coordinate-sorted BAM. */
int c3r_bam_index_build(const char *bam_path, const char *bai_path);

/* ---- output side (csrc/vcfio.cpp): what `sort_vcf` does after the per-chunk calls (src/sort_vcf.py:123-292).
 *
 * c3r_vcf_merge: `rows` = the newline-terminated VCF records of ONE contig in the order its chunks produced them.
 * Applies src/sort_vcf.py:204-236 to every record — dropped when ALT is "." or equals REF unless show_ref; FILTER set
 * to LowQual when it is a variant, qual != 0 and QUAL <= qual; FILTER set to RNAEditing when (pos, REF, ALT) is in the
 * REDIportal entries of this contig (edit_pos ascending, edit_ref / edit_alt parallel to it; n_edit 0 = no tagging)
 * and the record holds neither "Germline" nor "RefCall" — then keeps the LAST record of every position and writes them
 * in position order to out[cap].  out_nt / out_nt_len (may be NULL): the same records with RNAEditing shown as PASS
 * (the reference's *_no_tagging.vcf).  counts[3] = records read, kept (before the duplicate rule), tagged.
 * Returns C3R_EOVERFLOW with *out_len / *out_nt_len set when a buffer is too small (call twice). */
int c3r_vcf_merge(const char *rows, int64_t n_bytes, int qual, int show_ref, const int32_t *edit_pos, const char *const *edit_ref,
                  const char *const *edit_alt, int64_t n_edit, char *out, int64_t cap, int64_t *out_len, char *out_nt, int64_t cap_nt,
                  int64_t *out_nt_len, int64_t *counts);

/* `bgzip -f <path>` + `tabix -f -p vcf <path>.gz` (src/sort_vcf.py:70-75): writes <path>.gz (BGZF, 0xff00-byte blocks
 * deflated on `threads` threads, <= 0: hardware concurrency up to 32) and <path>.gz.tbi (TBI v1, VCF preset), removes
 * <path>. */
int c3r_vcf_compress(const char *path, int threads);

/* The same compressor fed piece by piece (the whole-sample driver hands it the header and then every contig's merged records as they
 * are written, so that no bgzip pass is left after the last contig): c3r_vcfz_write takes newline-terminated text in file order,
 * deflates every complete 0xff00-byte block on `threads` threads and writes it; c3r_vcfz_close(keep = 1) writes the last block, the
 * EOF block and <gz_path>.tbi — byte for byte what c3r_vcf_compress makes of the concatenated text; keep = 0 removes the file. */
typedef struct c3r_vcfz c3r_vcfz;
int c3r_vcfz_open(const char *gz_path, int threads, c3r_vcfz **out);
int c3r_vcfz_write(c3r_vcfz *z, const char *text, int64_t n_bytes);
int c3r_vcfz_close(c3r_vcfz *z, int keep);
/* Pieces: c3r_vcfz_piece_make compresses and indexes a run of whole lines (one contig's merged records) on its own — any thread, no
 * writer involved — and c3r_vcfz_append puts it into the file in calling order: the bytes written so far are closed into a block of
 * their own, the piece's blocks follow, its index entries are shifted to where it landed.  Same decompressed text and an equally
 * valid index as feeding the lines through c3r_vcfz_write, different block boundaries; only the append stays on the ordering thread.
 * A piece may be appended once and is freed by the caller. */
typedef struct c3r_vcfz_piece c3r_vcfz_piece;
int c3r_vcfz_piece_make(const char *text, int64_t n_bytes, int threads, c3r_vcfz_piece **out);
int c3r_vcfz_append(c3r_vcfz *z, const c3r_vcfz_piece *piece);
void c3r_vcfz_piece_free(c3r_vcfz_piece *piece);

/* Host memory for the large arrays that cross this interface (a contig's records, its reference slice, its rows): 2-MB aligned and
 * advised huge, so that a fresh 100-MB array is a few hundred page faults instead of tens of thousands where transparent huge pages
 * are available.  Optional — every entry point takes any host pointer.  NULL when out of memory. */
void *c3r_io_alloc(size_t bytes);
void c3r_io_free(void *p);

/* ---- reference side: `samtools faidx <fasta> ctg:beg-end` (shared/utils.py:168-193 reference_sequence_from) for an uncompressed,
 * faidx-indexed FASTA.  The caller passes the contig's .fai geometry (file offset of its first base, bases per line, bytes per line
 * including the line end); bases [beg0, end0) (0-based, half-open, inside the contig) are written to out[0, end0 - beg0) with the line
 * ends dropped, upper-cased when `upper` (the reference upper-cases, :185), read with pread on `threads` threads (<= 0: up to 8).
 * C3R_EINVAL when the file is shorter than the index says or a line does not end where the index says it does. */
int c3r_fasta_fetch(const char *path, int64_t offset, int32_t linebases, int32_t linewidth, int64_t beg0, int64_t end0, int upper,
                    int threads, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
