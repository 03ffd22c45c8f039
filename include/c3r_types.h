/* c3r_types.h — plain-C record layouts shared by the C-ABI (include/c3r.h), the host code and the
 * test oracle.  No torch / HIP types.  All integers little-endian, structs naturally aligned.
 *
 * The read record is the flat, device-friendly form of one BAM alignment: exactly the fields
 * `samtools mpileup` consumes when the reference spawns it at src/create_tensor_pileup.py:436-451
 * (core.pos, flag, MAPQ, CIGAR, 4-bit SEQ, optional HP aux tag).  Base qualities are not carried:
 * the reference always runs with --min-BQ 0 (shared/param_p.py:21, call_var_bam.py:205-228) and
 * never reads the QUAL column.
 */
#ifndef C3R_TYPES_H
#define C3R_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C3R_WINDOW 33          /* shared/param_p.py:34-35  no_of_positions = 2*16+1 */
#define C3R_FLANK 16           /* shared/param_p.py:34     flankingBaseNum          */
#define C3R_CH 18              /* shared/param_p.py:31     channel tuple            */
#define C3R_CH_PHASED 30       /* + phased_channel_size 12 (shared/param_p.py:33)   */
#define C3R_NPROB 24           /* 21 gt21 + 3 zygosity (shared/param_p.py:37)       */

/* channel indices, order of shared/param_p.py:31 */
enum {
    C3R_A = 0, C3R_C, C3R_G, C3R_T, C3R_I, C3R_I1, C3R_D, C3R_D1, C3R_STAR,
    C3R_a, C3R_c, C3R_g, C3R_t, C3R_i, C3R_i1, C3R_d, C3R_d1, C3R_HASH,
    /* phased extras, order of src/create_tensor_pileup.py:181,217 */
    C3R_AP, C3R_CP, C3R_GP, C3R_TP, C3R_IP, C3R_DP, C3R_AM, C3R_CM, C3R_GM, C3R_TM, C3R_IM, C3R_DM
};

/* BAM CIGAR op codes (SAM spec): MIDNSHP=X */
enum { C3R_CIG_M = 0, C3R_CIG_I, C3R_CIG_D, C3R_CIG_N, C3R_CIG_S, C3R_CIG_H, C3R_CIG_P, C3R_CIG_EQ, C3R_CIG_X };

typedef struct c3r_read {
    int32_t  pos;        /* 0-based leftmost reference position (BAM core.pos)            */
    uint32_t cigar_off;  /* index of this read's first op in the cigar array              */
    uint32_t n_cigar;    /* number of ops; each op is BAM-encoded: len << 4 | op          */
    uint32_t l_seq;      /* query length in bases                                         */
    uint64_t seq_off;    /* BYTE offset of base 0 in the packed sequence array; BAM nibble
                            order (base 2k in the high nibble of byte k), codes =ACMGRSVTWYHKDBN */
    uint16_t flag;       /* SAM flag                                                      */
    uint8_t  mapq;
    uint8_t  hp;         /* HP aux tag: 0 = absent, else the tag value (1, 2, ...)        */
    uint32_t reserved;   /* pad to 32 bytes; must be 0                                    */
} c3r_read_t;

/* Parameters of the tensor-build stage; defaults are the values run_clair3_rna forwards
 * (run_clair3_rna:684-705) and shared/param_p.py. */
typedef struct c3r_params {
    int32_t  channels;        /* 18, or 30 when --enable_phasing_model / --add_phasing_feature   */
    int32_t  min_mq;          /* --minMQ, default 5 (param_p.py:20)                                */
    int32_t  excl_flags;      /* samtools --excl-flags, 2316 (param_p.py:41); unmapped reads and anomalous pairs
                                 (FLAG 0x1 without 0x2: mpileup without -A) are always skipped            */
    int32_t  min_coverage;    /* --minCoverage, default 4 (param_p.py:90)                          */
    double   snp_min_af;      /* --snp_min_af, default 0.08 (param_p.py:88)                        */
    double   indel_min_af;    /* --indel_min_af, default 0.15 (param_p.py:89)                      */
    int32_t  head_tail;       /* --enable_variant_calling_at_sequence_head_and_tail               */
    int32_t  splice_padding;  /* --enable_padding_in_splice_junction_regions                       */
    int32_t  genotyping_mode; /* 1: candidates = the supplied site list (--vcf_fn), gates ignored  */
    int32_t  max_depth_rescale; /* 144 (param_p.py:14); windows with depth > 1.5x are rescaled     */
    int32_t  max_depth;       /* samtools mpileup -d: reads beyond this many live reads are discarded (htslib's
                                 rule, see c3r_pileup_scan); default 8000 = mpileup's own default, which the reference
                                 leaves in force (src/create_tensor_pileup.py:442); 0 = no cap.  The resident windows are
                                 16-bit until a scan meets a position that more than 32,767 kept reads cover (cannot
                                 happen at the default cap): that scan is repeated with 32-bit windows, which the context
                                 keeps from then on; only a BATCH that already holds 16-bit windows fails there
                                 (C3R_EOVERFLOW: scan the deep region first, or alone)                              */
    int32_t  mpileup_compat;  /* which samtools the column text is restated from (run_clair3_rna:159,166 only sets a floor of 1.10):
                                 0 = samtools <= 1.10 (default): an I immediately followed by a D shows the insertion only (`C+2TT`);
                                 1 = samtools >= 1.11 (bam_plp_insertion): it shows both (`C+2TT-1N`), which the reference's parser
                                     (src/create_tensor_pileup.py:151-163) reads as an insertion token AND a deletion token; pads (P ops) inside the
                                     run of I ops are printed as '*' ('#' on the reverse strand: --reverse-del), `+3T*T` (c3r_padins_t)       */
} c3r_params_t;

/* One emitted candidate site (the non-tensor fields of a create_tensor output line,
 * src/create_tensor_pileup.py:595-605). */
typedef struct c3r_site {
    int32_t pos;          /* 1-based centre position */
    int32_t depth;        /* depth at the centre (first field of alt_info) */
    char    ref33[C3R_WINDOW + 3]; /* reference +-16, 'A'-padded at contig ends; NUL-terminated, padded to 36 */
    int32_t n_tok;        /* number of per-read tokens recorded for the centre column (c3r_token_t: only the reads with something to say) */
    uint32_t tok_off;     /* offset of the first token in the token array */
} c3r_site_t;

/* One read's contribution to a candidate's centre column, in BAM order — what the host needs to
 * rebuild the ordered alt_info dictionary (src/create_tensor_pileup.py:179,221-261).  A site holds a token
 * for every read that shows something OTHER than the reference base or a ref-skip on the column: a
 * non-reference A / C / G / T, a '*' / '#', or an indel attached to the column (base 17 when that indel sits
 * behind a ref-skip).  Reads that show the reference base, N or an IUPAC letter add nothing to alt_info but
 * the depth, which c3r_site_t carries: R<ref> = max(0, depth - deletions - insertions - mismatches)
 * (:259).  (Rounds 1-4 returned one token per covering read.) */
typedef struct c3r_token {
    uint32_t read_idx;    /* index into the loaded read array */
    int32_t  indel;       /* >0 insertion length, <0 deletion length, 0 none */
    uint32_t qpos;        /* query offset of the first inserted base (valid when indel > 0) */
    uint8_t  base;        /* 4-bit BAM base code; 16 = '*'/'#' (inside deletion); 17 = ref-skip (with an indel behind it) */
    uint8_t  rev;         /* 1 = reverse strand */
    uint16_t del_after;   /* mpileup_compat = 1, indel > 0: length of the deletion that follows the insertion at once (0: none;
                             saturates at 65535) — a second indel token of the read on this column, after the insertion */
} c3r_token_t;

/* mpileup_compat = 1 only: an insertion whose run of I ops holds pads (CIGAR `2M1I1P1I2M`).  samtools >= 1.11 (bam_plp_insertion)
 * prints the run as ONE insertion of `total` characters with the pads as '*' — '#' on the reverse strand, because the reference passes
 * --reverse-del (src/create_tensor_pileup.py:436-451) — and the reference's parser keeps that text as the allele (:151-163, :221-232).  A token
 * carries the inserted BASES only (indel = n_bases, qpos); whoever rebuilds the allele text looks the pair (read_idx, qpos) up here.  Sorted
 * by (read_idx, qpos).  No aligner for long RNA reads emits pads: the table is empty except for hand-made CIGARs. */
typedef struct c3r_padins {
    uint32_t read_idx;    /* index into the loaded read array                                        */
    uint32_t qpos;        /* query offset of the first inserted base                                 */
    uint32_t n_bases;     /* inserted bases = sum of the run's I ops                                 */
    uint32_t total;       /* printed length = bases + pads, at most 64                               */
    uint64_t pad_mask;    /* bit j: character j of the printed insertion is a pad                    */
} c3r_padins_t;

#ifdef __cplusplus
}
#endif
#endif /* C3R_TYPES_H */
