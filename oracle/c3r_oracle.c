/* c3r_oracle.c — CPU restatement of the Clair3-RNA pileup hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (clair3_rna_amd/, libc3r.so) never links, imports or calls it.
 *
 * It mirrors the reference's structure stage by stage (all citations relative to /root/reference):
 *
 *   A1  orc_mpileup          reads -> `samtools mpileup` text rows.  The reference shells out to
 *                            samtools (src/create_tensor_pileup.py:436-451); samtools/htslib are an
 *                            un-vendored third-party dependency (>= 1.10, otherwise unpinned,
 *                            run_clair3_rna:159) that is absent from this image.  This function
 *                            restates htslib's published pileup algorithm (per-read CIGAR cursor,
 *                            indel attached to the preceding column, '^'/'$' markers, '*'/'#' with
 *                            --reverse-del, '>'/'<' for N ops, literal bases because no -f is given).
 *                            PARITY UNPINNED for this stage: pinned only by hand-derived CIGAR
 *                            known-answer cases in tests/test_oracle_mpileup.py.
 *   A2  orc_generate_tensor  one pileup column -> channel vector + gates
 *                            (src/create_tensor_pileup.py:85-302).        pinned: golden G1
 *   A3  orc_create_tensor    sliding window / candidate driver, emits the reference's text lines
 *                            (src/create_tensor_pileup.py:463-637).        pinned: golden G2
 *   A4  orc_chunk_region     chunk -> coordinates (src/create_tensor_pileup.py:379-422)  pinned: G2
 *   A5  orc_batch_from_lines text lines -> int32 batch, depth>216 rescale
 *                            (clair3_rna/utils.py:64-138).                 pinned: golden G3
 *   A6  orc_forward          Bi-LSTM x2 + dense heads, fp32 (clair3_rna/model.py:126-216; Keras
 *                            LSTM equations).  TensorFlow is absent: PARITY UNPINNED against the
 *                            reference itself; cross-checked against torch.nn.LSTM (golden G5).
 *
 * Plain C11, no dependencies.  Strings returned by orc_* are malloc'd; free with orc_free().
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/c3r_types.h"

/* ------------------------------------------------------------------------------------------ util */
typedef struct { char *p; size_t n, cap; } sbuf;

static void sb_reserve(sbuf *s, size_t extra) {
    if (s->n + extra + 1 > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 4096;
        while (nc < s->n + extra + 1) nc *= 2;
        s->p = (char *)realloc(s->p, nc);
        s->cap = nc;
    }
}
static void sb_putc(sbuf *s, char c) { sb_reserve(s, 1); s->p[s->n++] = c; s->p[s->n] = 0; }
static void sb_put(sbuf *s, const char *t, size_t len) { sb_reserve(s, len); memcpy(s->p + s->n, t, len); s->n += len; s->p[s->n] = 0; }
static void sb_puts(sbuf *s, const char *t) { sb_put(s, t, strlen(t)); }
static void sb_putl(sbuf *s, long long v) { char b[32]; int n = snprintf(b, sizeof b, "%lld", v); sb_put(s, b, (size_t)n); }

void orc_free(void *p) { free(p); }

/* ============================================================================ A1: mpileup text */
static const char NT16[] = "=ACMGRSVTWYHKDBN";

static inline int cig_op(uint32_t c) { return (int)(c & 0xf); }
static inline int cig_len(uint32_t c) { return (int)(c >> 4); }
static inline int is_refop(int op) { return op == C3R_CIG_M || op == C3R_CIG_D || op == C3R_CIG_N || op == C3R_CIG_EQ || op == C3R_CIG_X; }
static inline int is_matchop(int op) { return op == C3R_CIG_M || op == C3R_CIG_EQ || op == C3R_CIG_X; }
static inline int seq_code(const uint8_t *seq, uint64_t off, uint32_t i) {
    uint8_t b = seq[off + (i >> 1)];
    return (i & 1) ? (b & 0xf) : (b >> 4);
}

/* reference length consumed by a read's CIGAR */
static int64_t cigar_rlen(const uint32_t *c, uint32_t n) {
    int64_t l = 0;
    for (uint32_t k = 0; k < n; ++k) if (is_refop(cig_op(c[k]))) l += cig_len(c[k]);
    return l;
}

/* per-read cursor, advanced column by column (htslib keeps the same three numbers per read) */
typedef struct {
    int64_t idx;   /* read index */
    int k;         /* current ref-consuming op, -1 = not started */
    int64_t x;     /* reference position where op k starts */
    int64_t y;     /* query offset where op k starts */
    int64_t end;   /* last reference position covered (inclusive) */
} cursor_t;

typedef struct { int is_del, is_refskip, indel, is_head, is_tail; int64_t qpos; int ins_k; /* first op behind the one on the column (indel > 0) */ } plp_t;

/* Resolve what read r shows at reference position `pos` (0-based).  Restates the htslib column
 * resolution: locate the op covering pos; at the LAST position of that op peek the following op:
 * D (when the current op is not D) => deletion of the merged run of D ops; I => insertion of the
 * merged run of I ops (pads skipped); P then I's => insertion. */
static void resolve(const c3r_read_t *r, const uint32_t *cg, cursor_t *s, int64_t pos, plp_t *p) {
    int n = (int)r->n_cigar;
    int k;
    if (s->k == -1) {
        s->x = r->pos; s->y = 0;
        for (k = 0; k < n; ++k) {
            int op = cig_op(cg[k]), l = cig_len(cg[k]);
            if (is_refop(op)) break;
            if (op == C3R_CIG_I || op == C3R_CIG_S) s->y += l;
        }
        s->k = k;
    }
    /* advance to the op covering pos (htslib steps one op per column; a loop lets the cursor start
     * in the middle of a read when the region begins inside it) */
    while (pos - s->x >= cig_len(cg[s->k])) {
        int l = cig_len(cg[s->k]);
        if (is_matchop(cig_op(cg[s->k]))) s->y += l;
        s->x += l;
        for (k = s->k + 1; k < n; ++k) {
            int op = cig_op(cg[k]), l2 = cig_len(cg[k]);
            if (is_refop(op)) break;
            if (op == C3R_CIG_I || op == C3R_CIG_S) s->y += l2;
        }
        s->k = k;
    }
    int op = cig_op(cg[s->k]), l = cig_len(cg[s->k]);
    p->is_del = p->indel = p->is_refskip = 0;
    p->ins_k = s->k + 1;
    if (s->x + l - 1 == pos && s->k + 1 < n) {
        int op2 = cig_op(cg[s->k + 1]), l2 = cig_len(cg[s->k + 1]);
        if (op2 == C3R_CIG_D && op != C3R_CIG_D) {
            p->indel = -l2;
            for (k = s->k + 2; k < n; ++k) {
                if (cig_op(cg[k]) == C3R_CIG_D) p->indel -= cig_len(cg[k]); else break;
            }
        } else if (op2 == C3R_CIG_I) {
            p->indel = l2;
            for (k = s->k + 2; k < n; ++k) {
                int o = cig_op(cg[k]);
                if (o == C3R_CIG_I) p->indel += cig_len(cg[k]);
                else if (o != C3R_CIG_P) break;
            }
        } else if (op2 == C3R_CIG_P && s->k + 2 < n) {
            int l3 = 0;
            for (k = s->k + 2; k < n; ++k) {
                int o = cig_op(cg[k]);
                if (o == C3R_CIG_I) l3 += cig_len(cg[k]);
                else if (is_refop(o)) break;
            }
            if (l3 > 0) p->indel = l3;
        }
    }
    if (is_matchop(op)) {
        p->qpos = s->y + (pos - s->x);
    } else {
        p->is_del = 1; p->qpos = s->y;
        p->is_refskip = (op == C3R_CIG_N);
    }
    p->is_head = (pos == r->pos);
    p->is_tail = (pos == s->end);
}

static int bed_contains(const int32_t *bed, int n_bed, int64_t pos0) {
    /* half-open 0-based intervals in any order, possibly overlapping (samtools -l takes them as they come) */
    for (int i = 0; i < n_bed; ++i)
        if (pos0 >= bed[2 * i] && pos0 < bed[2 * i + 1]) return 1;
    return 0;
}

/* reads must be sorted by pos (BAM order).  beg1/end1: 1-based inclusive region (-r).  bed: the
 * `-l` file's intervals for this contig (NULL = none).  Returns text rows
 * "ctg\tpos\tN\tn\tBASES\tQUALS[\tHP,...]\n". */
char *orc_mpileup_d(const c3r_read_t *reads_in, int64_t n_reads, const uint32_t *cigar_in, const uint8_t *seq,
                    const char *ctg, int64_t beg1, int64_t end1, int min_mq, int excl_flags,
                    const int32_t *bed, int n_bed, int with_hp, int max_depth, int64_t *out_len);
/* compat: which samtools printer is restated (the reference only sets a floor of 1.10, run_clair3_rna:159,166):
 *   0  samtools <= 1.10 pileup_seq: `+<n><n query bases>` for an insertion (pads skipped), nothing for a deletion behind it;
 *   1  samtools >= 1.11 (htslib bam_plp_insertion): the run of I and P ops behind the column's op is printed as ONE insertion with the
 *      pads as '*' — '#' for a reverse-strand read under --reverse-del, which the reference always passes (src/create_tensor_pileup.py:448) —
 *      (`+3T*T`), and when a D ends that run its length follows (`+2TT-1N`).  (Adjacent D ops count as one deletion, as
 *      everywhere in this restatement.) */
char *orc_mpileup_c(const c3r_read_t *reads_in, int64_t n_reads, const uint32_t *cigar_in, const uint8_t *seq,
                    const char *ctg, int64_t beg1, int64_t end1, int min_mq, int excl_flags,
                    const int32_t *bed, int n_bed, int with_hp, int max_depth, int compat, int64_t *out_len);

/* samtools mpileup's default -d 8000 (the reference passes --max-depth only on request, src/create_tensor_pileup.py:442) */
char *orc_mpileup(const c3r_read_t *reads_in, int64_t n_reads, const uint32_t *cigar_in, const uint8_t *seq,
                  const char *ctg, int64_t beg1, int64_t end1, int min_mq, int excl_flags,
                  const int32_t *bed, int n_bed, int with_hp, int64_t *out_len) {
    return orc_mpileup_d(reads_in, n_reads, cigar_in, seq, ctg, beg1, end1, min_mq, excl_flags, bed, n_bed, with_hp, 8000, out_len);
}

/* The flag part of samtools mpileup's read filter as the reference runs it (src/create_tensor_pileup.py:436-451):
 * --excl-flags replaces the default mask, unmapped reads never pile up, and since the reference never passes -A
 * (--count-orphans) mpileup also skips "anomalous read pairs": FLAG 0x1 (paired) set with 0x2 (proper pair) clear
 * (htslib-based mpileup, mplp_func; third-party, absent here: restated from its documented behaviour). */
static int flag_fails(unsigned flag, int excl_flags) {
    return (flag & (unsigned)excl_flags) || (flag & 4u) || ((flag & 1u) && !(flag & 2u));
}

/* Depth cap, restated from htslib's pileup engine (bam_plp_push / bam_plp_next; third-party, absent here: parity unpinned).
 * Reads reach the engine in file order, already filtered by flag / MAPQ, and only those overlapping the region.  A read is
 * discarded iff it starts at the position the engine currently stands on — i.e. it is NOT the first read pushed for its start
 * position — and the engine's node pool holds more than max_depth nodes at that moment (`iter->mp->cnt > iter->maxcnt`).  The pool
 * holds the read list plus the list's empty tail node (allocated when the iterator is made and again after every kept read), so the
 * test is  list + 1 > max_depth:  reads that all start on one position (an amplicon) pile up to exactly max_depth, the 8000 that
 * samtools users see.  (Rounds 1-4 restated it as list > max_depth, one read more; htslib is absent here, so this stays unpinned —
 * the constant below is the one place to change.)  The list holds every kept read that has not been retired yet; a read is retired
 * while the column at or after its (exclusive) end is processed, and all columns left of the new read's start have been processed by
 * then: the list is the kept reads with exclusive end > start - 1.
 * max_depth <= 0: no cap.  Marks dropped[i] = 1. */
#define ORC_PLP_POOL_EXTRA 1      /* nodes of the pool that are not reads: the list's tail */
static void depth_cap(const c3r_read_t *reads, int64_t n_reads, const uint32_t *cigar, int64_t beg0, int64_t end0_incl,
                      int min_mq, int excl_flags, int max_depth, uint8_t *dropped) {
    memset(dropped, 0, (size_t)(n_reads > 0 ? n_reads : 1));
    if (max_depth <= 0) return;
    int64_t *ends = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_reads > 0 ? n_reads : 1));   /* exclusive ends of kept reads */
    int64_t n_live = 0, last_pos = INT64_MIN;
    for (int64_t i = 0; i < n_reads; ++i) {
        const c3r_read_t *r = &reads[i];
        if (r->pos > end0_incl) break;
        if (flag_fails(r->flag, excl_flags) || r->mapq < min_mq || r->n_cigar == 0) continue;
        const int64_t rl = cigar_rlen(cigar + r->cigar_off, r->n_cigar);
        if (rl <= 0 || r->pos + rl <= beg0) continue;                 /* not fetched for this region */
        int64_t w = 0;                                                /* retire: keep exclusive end > pos - 1 */
        for (int64_t k = 0; k < n_live; ++k) if (ends[k] > (int64_t)r->pos - 1) ends[w++] = ends[k];
        n_live = w;
        const int first = (r->pos != last_pos);
        last_pos = r->pos;
        if (!first && n_live + ORC_PLP_POOL_EXTRA > max_depth) { dropped[i] = 1; continue; }
        ends[n_live++] = r->pos + rl;
    }
    free(ends);
}

char *orc_mpileup_d(const c3r_read_t *reads_in, int64_t n_reads, const uint32_t *cigar_in, const uint8_t *seq,
                    const char *ctg, int64_t beg1, int64_t end1, int min_mq, int excl_flags,
                    const int32_t *bed, int n_bed, int with_hp, int max_depth, int64_t *out_len) {
    return orc_mpileup_c(reads_in, n_reads, cigar_in, seq, ctg, beg1, end1, min_mq, excl_flags, bed, n_bed, with_hp, max_depth, 0, out_len);
}

char *orc_mpileup_c(const c3r_read_t *reads_in, int64_t n_reads, const uint32_t *cigar_in, const uint8_t *seq,
                    const char *ctg, int64_t beg1, int64_t end1, int min_mq, int excl_flags,
                    const int32_t *bed, int n_bed, int with_hp, int max_depth, int compat, int64_t *out_len) {
    /* Zero-length CIGAR ops are dropped before the walk (conscious deviation): BAM writers do not emit
     * them, and what htslib's cursor does with them is an accident of its peek-next-op logic (e.g. `3M0I2D` loses the
     * deletion marker, `3M0D2I` the insertion).  The product path drops them at load, so the checker does too. */
    int64_t n_ops_total = 0;
    for (int64_t i = 0; i < n_reads; ++i) n_ops_total += reads_in[i].n_cigar;
    c3r_read_t *reads = (c3r_read_t *)malloc(sizeof(c3r_read_t) * (size_t)(n_reads > 0 ? n_reads : 1));
    uint32_t *cigar = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n_ops_total > 0 ? n_ops_total : 1));
    {
        int64_t w = 0;
        for (int64_t i = 0; i < n_reads; ++i) {
            reads[i] = reads_in[i];
            reads[i].cigar_off = (uint32_t)w;
            for (uint32_t k = 0; k < reads_in[i].n_cigar; ++k) {
                const uint32_t c = cigar_in[reads_in[i].cigar_off + k];
                if (cig_len(c) > 0) cigar[w++] = c;
            }
            reads[i].n_cigar = (uint32_t)(w - reads[i].cigar_off);
        }
    }
    uint8_t *dropped = (uint8_t *)malloc((size_t)(n_reads > 0 ? n_reads : 1));
    depth_cap(reads, n_reads, cigar, beg1 - 1, end1 - 1, min_mq, excl_flags, max_depth, dropped);
    sbuf out = {0}, bases = {0}, hps = {0};
    sb_reserve(&out, 1);
    out.p[0] = 0;
    cursor_t *act = NULL; size_t n_act = 0, cap_act = 0;
    int64_t next = 0;
    int64_t pos = beg1 - 1;              /* 0-based column */
    const int64_t end0 = end1 - 1;       /* inclusive */
    while (pos <= end0) {
        /* admit reads starting at or before pos */
        while (next < n_reads && reads[next].pos <= pos) {
            const c3r_read_t *r = &reads[next];
            int ok = !flag_fails(r->flag, excl_flags) && r->mapq >= min_mq && r->n_cigar > 0 && !dropped[next];
            if (ok) {
                int64_t rl = cigar_rlen(cigar + r->cigar_off, r->n_cigar);
                if (rl > 0 && r->pos + rl - 1 >= pos) {
                    if (n_act == cap_act) { cap_act = cap_act ? cap_act * 2 : 256; act = (cursor_t *)realloc(act, cap_act * sizeof *act); }
                    cursor_t c = { next, -1, 0, 0, r->pos + rl - 1 };
                    act[n_act++] = c;
                }
            }
            ++next;
        }
        if (n_act == 0) {
            if (next >= n_reads) break;
            if (reads[next].pos > pos) { pos = reads[next].pos; continue; }
        }
        /* drop finished reads, preserving order */
        size_t w = 0;
        for (size_t i = 0; i < n_act; ++i) if (act[i].end >= pos) act[w++] = act[i];
        n_act = w;
        if (n_act == 0) continue;
        int in_bed = (bed == NULL) || bed_contains(bed, n_bed, pos);
        bases.n = 0; hps.n = 0;
        for (size_t i = 0; i < n_act; ++i) {
            const c3r_read_t *r = &reads[act[i].idx];
            const uint32_t *cg = cigar + r->cigar_off;
            plp_t p;
            resolve(r, cg, &act[i], pos, &p);   /* always advance the cursor, even outside the bed */
            if (!in_bed) continue;
            int rev = (r->flag & 16) != 0;
            if (p.is_head) { sb_putc(&bases, '^'); sb_putc(&bases, (char)(r->mapq > 93 ? 126 : r->mapq + 33)); }
            if (!p.is_del) {
                int c = (p.qpos < (int64_t)r->l_seq) ? NT16[seq_code(seq, r->seq_off, (uint32_t)p.qpos)] : 'N';
                if (c == '=') c = rev ? ',' : '.';
                else c = rev ? tolower(c) : toupper(c);
                sb_putc(&bases, (char)c);
            } else {
                sb_putc(&bases, p.is_refskip ? (rev ? '<' : '>') : (rev ? '#' : '*'));
            }
            if (p.indel > 0 && compat) {
                /* bam_plp_insertion: the run of I / P ops behind the op on the column; a D that ends it is printed too */
                int n_ops = (int)r->n_cigar, kk, total = 0, del_after = 0;
                for (kk = p.ins_k; kk < n_ops; ++kk) {
                    int o = cig_op(cg[kk]);
                    if (o == C3R_CIG_I || o == C3R_CIG_P) total += cig_len(cg[kk]);
                    else { if (o == C3R_CIG_D) { for (int k2 = kk; k2 < n_ops && cig_op(cg[k2]) == C3R_CIG_D; ++k2) del_after += cig_len(cg[k2]); } break; }
                }
                sb_putc(&bases, '+'); sb_putl(&bases, total);
                int64_t q = p.qpos + 1 - p.is_del;
                for (kk = p.ins_k; kk < n_ops; ++kk) {
                    int o = cig_op(cg[kk]), ln = cig_len(cg[kk]);
                    if (o == C3R_CIG_P) { for (int j = 0; j < ln; ++j) sb_putc(&bases, rev ? '#' : '*'); }      /* (pileup_seq: `pad = rev_del ? '#' : '*'` on the reverse strand; the reference passes --reverse-del) */
                    else if (o == C3R_CIG_I) {
                        for (int j = 0; j < ln; ++j, ++q) {
                            int c = (q < (int64_t)r->l_seq) ? NT16[seq_code(seq, r->seq_off, (uint32_t)q)] : 'N';
                            sb_putc(&bases, (char)(rev ? tolower(c) : toupper(c)));
                        }
                    } else break;
                }
                if (del_after > 0) {
                    sb_putc(&bases, '-'); sb_putl(&bases, del_after);
                    for (int j = 0; j < del_after; ++j) sb_putc(&bases, rev ? 'n' : 'N');
                }
            } else if (p.indel > 0) {
                sb_putc(&bases, '+'); sb_putl(&bases, p.indel);
                for (int j = 1; j <= p.indel; ++j) {
                    int64_t q = p.qpos + j - p.is_del;
                    int c = (q < (int64_t)r->l_seq) ? NT16[seq_code(seq, r->seq_off, (uint32_t)q)] : 'N';
                    sb_putc(&bases, (char)(rev ? tolower(c) : toupper(c)));
                }
            } else if (p.indel < 0) {
                sb_putc(&bases, '-'); sb_putl(&bases, -p.indel);
                for (int j = 0; j < -p.indel; ++j) sb_putc(&bases, rev ? 'n' : 'N');
            }
            if (p.is_tail) sb_putc(&bases, '$');
            if (with_hp) {
                if (hps.n) sb_putc(&hps, ',');
                if (r->hp) sb_putl(&hps, r->hp); else sb_putc(&hps, '*');
            }
        }
        if (in_bed) {
            sb_puts(&out, ctg); sb_putc(&out, '\t'); sb_putl(&out, pos + 1); sb_puts(&out, "\tN\t");
            sb_putl(&out, (long long)n_act); sb_putc(&out, '\t');
            sb_put(&out, bases.p, bases.n); sb_putc(&out, '\t');
            for (size_t i = 0; i < n_act; ++i) sb_putc(&out, '~');
            if (with_hp) { sb_putc(&out, '\t'); sb_put(&out, hps.p ? hps.p : "", hps.n); }
            sb_putc(&out, '\n');
        }
        ++pos;
    }
    free(act); free(bases.p); free(hps.p); free(reads); free(cigar); free(dropped);
    if (out_len) *out_len = (int64_t)out.n;
    return out.p;
}

/* ==================================================================== A2: column -> channel vector */
typedef struct { const char *s; int len; int count; char ph; } tok_t;
typedef struct { char *key; int count; } kv_t;
typedef struct { kv_t *v; int n, cap; } odict;   /* insertion-ordered dict with tiny linear lookup */

static void od_add(odict *d, const char *key, int klen, int count) {
    for (int i = 0; i < d->n; ++i)
        if ((int)strlen(d->v[i].key) == klen && memcmp(d->v[i].key, key, (size_t)klen) == 0) { d->v[i].count += count; return; }
    if (d->n == d->cap) { d->cap = d->cap ? d->cap * 2 : 8; d->v = (kv_t *)realloc(d->v, (size_t)d->cap * sizeof(kv_t)); }
    d->v[d->n].key = (char *)malloc((size_t)klen + 1);
    memcpy(d->v[d->n].key, key, (size_t)klen); d->v[d->n].key[klen] = 0;
    d->v[d->n].count = count; d->n++;
}
static void od_free(odict *d) { for (int i = 0; i < d->n; ++i) free(d->v[i].key); free(d->v); d->v = NULL; d->n = d->cap = 0; }

static const char *CHN[18] = {"A","C","G","T","I","I1","D","D1","*","a","c","g","t","i","i1","d","d1","#"};
static int chan_of_char(char c) {
    for (int i = 0; i < 18; ++i) if (CHN[i][1] == 0 && CHN[i][0] == c) return i;
    return -1;
}

/* evc_base_from, src/create_tensor_pileup.py:64-74 */
static char evc_base(char b) {
    if (b == 'N') return 'A';
    if (b == 'n') return 'a';
    if (strchr("ACGTacgt", b) && b) return b;
    return isupper((unsigned char)b) ? 'A' : 'a';
}

typedef struct {
    int32_t tensor[C3R_CH_PHASED];
    odict alt;           /* ordered alt_dict */
    odict plist;         /* pileup_list (sorted, stable) */
    int depth, pass_af, max_del_length, max_skip_count;
    double af;
} column_t;

static void column_free(column_t *c) { od_free(&c->alt); od_free(&c->plist); }

/* hp: array of n_hp NUL-terminated strings flattened as pointers, or NULL when unphased */
static void generate_tensor(const char *s, int n, char **hp, int n_hp, int64_t pos, const char *ref_seq, int64_t ref_len,
                            int64_t ref_start, char reference_base, double snp_af, double indel_af, double indel_af_default,
                            column_t *out) {
    memset(out, 0, sizeof *out);
    reference_base = evc_base(reference_base);
    tok_t *base_list = (tok_t *)malloc(sizeof(tok_t) * (size_t)(n + 1));
    int nb = 0, phasing_idx = 0;
    int read_end = 0, read_start = 0, skip_start = 0, skip_end = 0;
    int i = 0;
    while (i < n) {
        char b = s[i];
        if (b && strchr("ACGTNacgtn#*", b)) {
            base_list[nb].s = s + i; base_list[nb].len = 1; base_list[nb].count = 0;
            base_list[nb].ph = '?';
            if (hp) {
                const char *hv = (phasing_idx < n_hp) ? hp[phasing_idx] : "?";
                base_list[nb].ph = !strcmp(hv, "1") ? '1' : !strcmp(hv, "2") ? '2' : 'x';
                phasing_idx++;
            }
            nb++;
        } else if (b == '+' || b == '-') {
            int start_sign = i;
            i += 1;
            int advance = 0;
            while (i < n && isdigit((unsigned char)s[i])) { advance = advance * 10 + (s[i] - '0'); i++; }
            /* token = sign + s[i : i+advance] (digits dropped); represent as (sign, ptr, len) */
            int avail = n - i; if (avail < 0) avail = 0;
            int tl = advance < avail ? advance : avail;
            base_list[nb].s = s + i; base_list[nb].len = -(tl + 1);   /* negative len marks an indel token */
            base_list[nb].count = (s[start_sign] == '+') ? 1 : 2;      /* 1 = ins, 2 = del (temp use) */
            base_list[nb].ph = '0';
            nb++;
            i += advance - 1;
        } else if (b == '^') {
            i += 1; read_start++;
        } else if (b == '<' || b == '>') {
            if (b == '<') skip_start++; else skip_end++;
            if (hp) phasing_idx++;
        } else if (b == '$') {
            read_end++;
        }
        i += 1;
    }
    int msk = read_end; if (read_start > msk) msk = read_start; if (skip_start > msk) msk = skip_start; if (skip_end > msk) msk = skip_end;
    out->max_skip_count = msk;

    /* Counter(base_list): distinct tokens in first-seen order */
    typedef struct { int sign; const char *s; int len; int count; } dtok;   /* sign 0 = plain base */
    dtok *dt = (dtok *)malloc(sizeof(dtok) * (size_t)(nb + 1));
    int nd = 0;
    for (int t = 0; t < nb; ++t) {
        int sign = 0, len = base_list[t].len; const char *p = base_list[t].s;
        if (len < 0) { sign = base_list[t].count; len = -len - 1; }
        int f = -1;
        for (int d = 0; d < nd; ++d)
            if (dt[d].sign == sign && dt[d].len == len && memcmp(dt[d].s, p, (size_t)len) == 0) { f = d; break; }
        if (f < 0) { dt[nd].sign = sign; dt[nd].s = p; dt[nd].len = len; dt[nd].count = 1; nd++; }
        else dt[f].count++;
    }

    int nch = 18;
    if (hp) {
        int ph[12] = {0};  /* AP CP GP TP IP DP AM CM GM TM IM DM */
        for (int t = 0; t < nb; ++t) {
            int isindel = base_list[t].len < 0;
            int sign = isindel ? base_list[t].count : 0;
            char p = base_list[t].ph;
            if (sign == 1 && t > 0) {
                char pp = base_list[t - 1].ph;
                if (pp == '1') ph[4]++; else if (pp == '2') ph[10]++;
            } else if (sign == 2 && t > 0) {
                char pp = base_list[t - 1].ph;
                if (pp == '1') ph[5]++; else if (pp == '2') ph[11]++;
            } else if (!isindel) {
                char u = (char)toupper((unsigned char)base_list[t].s[0]);
                int bi = (u == 'A') ? 0 : (u == 'C') ? 1 : (u == 'G') ? 2 : (u == 'T') ? 3 : -1;
                if (bi >= 0) { if (p == '1') ph[bi]++; else if (p == '2') ph[6 + bi]++; }
            }
        }
        for (int j = 0; j < 12; ++j) out->tensor[18 + j] = ph[j];
        nch = 30;
    }
    (void)nch;

    int depth = 0, max_ins_0 = 0, max_del_0 = 0, max_ins_1 = 0, max_del_1 = 0, max_del_length = 0;
    int alt_count = 0, ins_count = 0, del_count = 0;
    odict pileup_dict = {0};
    char keybuf_static[256];
    for (int d = 0; d < nd; ++d) {
        int count = dt[d].count;
        if (dt[d].sign == 1) {
            int kl = dt[d].len + 2;
            char *kb = kl < (int)sizeof keybuf_static ? keybuf_static : (char *)malloc((size_t)kl + 1);
            kb[0] = 'I'; kb[1] = reference_base;
            for (int j = 0; j < dt[d].len; ++j) kb[2 + j] = (char)toupper((unsigned char)dt[d].s[j]);
            od_add(&out->alt, kb, kl, count);
            if (kb != keybuf_static) free(kb);
            od_add(&pileup_dict, "I", 1, count);
            ins_count += count;
            if (dt[d].len >= 1 && strchr("ACGTN*", dt[d].s[0])) { out->tensor[C3R_I] += count; if (count > max_ins_0) max_ins_0 = count; }
            else { out->tensor[C3R_i] += count; if (count > max_ins_1) max_ins_1 = count; }
        } else if (dt[d].sign == 2) {
            int64_t a = pos - ref_start + 1, b = pos - ref_start + dt[d].len + 1;
            if (a < 0) a = 0; if (b > ref_len) b = ref_len; if (a > ref_len) a = ref_len; if (b < a) b = a;
            int dl = (int)(b - a);
            int kl = dl + 1;
            char *kb = kl < (int)sizeof keybuf_static ? keybuf_static : (char *)malloc((size_t)kl + 1);
            kb[0] = 'D'; memcpy(kb + 1, ref_seq + a, (size_t)dl);
            od_add(&out->alt, kb, kl, count);
            if (kb != keybuf_static) free(kb);
            od_add(&pileup_dict, "D", 1, count);
            if (dl > max_del_length) max_del_length = dl;
            del_count += count;
            if (dt[d].len >= 1 && strchr("N*ACGT", dt[d].s[0])) { out->tensor[C3R_D] += count; if (count > max_del_0) max_del_0 = count; }
            else { out->tensor[C3R_d] += count; if (count > max_del_1) max_del_1 = count; }
        } else {
            char k = dt[d].s[0];
            char u = (char)toupper((unsigned char)k);
            if (u == 'A' || u == 'C' || u == 'G' || u == 'T') {
                od_add(&pileup_dict, &u, 1, count);
                depth += count;
                if (u != reference_base) { char kb[2] = { 'X', u }; od_add(&out->alt, kb, 2, count); alt_count += count; }
                out->tensor[chan_of_char(k)] += count;
            } else if (k == '#' || k == '*') {
                del_count += count;
                out->tensor[chan_of_char(k)] += count;
                depth += count;
            }
        }
    }
    int ref_count = depth - del_count - ins_count - alt_count; if (ref_count < 0) ref_count = 0;
    if (ref_count > 0) { char kb[2] = { 'R', reference_base }; od_add(&out->alt, kb, 2, ref_count); }
    out->tensor[C3R_I1] = max_ins_0; out->tensor[C3R_i1] = max_ins_1;
    out->tensor[C3R_D1] = max_del_0; out->tensor[C3R_d1] = max_del_1;
    int denominator = depth > 0 ? depth : 1;

    /* stable sort by count, descending */
    for (int a = 0; a < pileup_dict.n; ++a) {
        int best = -1;
        for (int b = 0; b < pileup_dict.n; ++b) {
            if (pileup_dict.v[b].count < 0) continue;  /* consumed */
            if (best < 0 || pileup_dict.v[b].count > pileup_dict.v[best].count) best = b;
        }
        od_add(&out->plist, pileup_dict.v[best].key, (int)strlen(pileup_dict.v[best].key), 0);
        out->plist.v[out->plist.n - 1].count = pileup_dict.v[best].count;
        pileup_dict.v[best].count = -1 - pileup_dict.v[best].count;
    }
    od_free(&pileup_dict);

    if (snp_af < 0) snp_af = 0.08;
    if (indel_af < 0) indel_af = indel_af_default;
    int pass_snp = 0, pass_indel = 0;
    int pass_af = out->plist.n > 0 && !(out->plist.v[0].key[0] == reference_base && out->plist.v[0].key[1] == 0);
    for (int a = 0; a < out->plist.n; ++a) {
        const char *item = out->plist.v[a].key; int count = out->plist.v[a].count;
        if (item[0] == reference_base) continue;
        if (item[0] == 'I' || item[0] == 'D') { pass_indel = pass_indel || ((double)count / denominator >= indel_af); continue; }
        pass_snp = pass_snp || ((double)count / denominator >= snp_af);
    }
    double af = out->plist.n > 1 ? (double)out->plist.v[1].count / denominator : 0.0;
    if (out->plist.n >= 1 && out->plist.v[0].key[0] != reference_base) af = (double)out->plist.v[0].count / denominator;
    out->af = af;

    /* BASE2INDEX[reference_base] (src/create_tensor_pileup.py:296-297): the channel names double as keys, so an IUPAC 'D' in
     * the reference indexes the D / d channels; any other non-ACGT letter raises KeyError in the reference.  There the
     * build (oracle and HIP path alike) falls back to evc_base_from's rule: counts as 'A' / 'a'. */
    int up = chan_of_char(reference_base), lo = chan_of_char((char)tolower((unsigned char)reference_base));
    if (up < 0) up = chan_of_char(evc_base(reference_base));
    if (lo < 0) lo = chan_of_char(evc_base((char)tolower((unsigned char)reference_base)));
    out->tensor[up] = -(out->tensor[C3R_A] + out->tensor[C3R_C] + out->tensor[C3R_G] + out->tensor[C3R_T]);
    out->tensor[lo] = -(out->tensor[C3R_a] + out->tensor[C3R_c] + out->tensor[C3R_g] + out->tensor[C3R_t]);

    out->pass_af = pass_af || pass_snp || pass_indel;
    out->depth = depth;
    out->max_del_length = max_del_length;
    free(base_list); free(dt);
}

static void alt_to_text(const odict *d, sbuf *sb) {
    for (int i = 0; i < d->n; ++i) { if (i) sb_putc(sb, ' '); sb_puts(sb, d->v[i].key); sb_putc(sb, ' '); sb_putl(sb, d->v[i].count); }
}

static char **split_hp(const char *hp_csv, int *n_out, char **storage) {
    /* returns array of pointers into a malloc'd copy */
    size_t L = strlen(hp_csv);
    char *cp = (char *)malloc(L + 1); memcpy(cp, hp_csv, L + 1);
    int n = 1; for (size_t i = 0; i < L; ++i) if (cp[i] == ',') n++;
    char **v = (char **)malloc(sizeof(char *) * (size_t)n);
    int k = 0; v[k++] = cp;
    for (size_t i = 0; i < L; ++i) if (cp[i] == ',') { cp[i] = 0; v[k++] = cp + i + 1; }
    *n_out = n; *storage = cp;
    return v;
}

/* ctypes entry for golden G1.  hp_csv NULL = unphased.  tensor must hold 30 ints.  Returns a
 * malloc'd text "alt k v k v ...\nplist k v ...\n". */
char *orc_generate_tensor(const char *bases, const char *hp_csv, int64_t pos, const char *ref_seq, int64_t ref_start,
                          char ref_base, double snp_af, double indel_af, int32_t *tensor, int32_t *depth, int32_t *pass_af,
                          int32_t *max_del_len, int32_t *max_skip, double *af) {
    column_t c; int n_hp = 0; char **hp = NULL; char *st = NULL;
    if (hp_csv) hp = split_hp(hp_csv, &n_hp, &st);
    generate_tensor(bases, (int)strlen(bases), hp, n_hp, pos, ref_seq, (int64_t)strlen(ref_seq), ref_start, ref_base, snp_af, indel_af, 0.15, &c);
    memcpy(tensor, c.tensor, sizeof(int32_t) * 30);
    *depth = c.depth; *pass_af = c.pass_af; *max_del_len = c.max_del_length; *max_skip = c.max_skip_count; *af = c.af;
    sbuf sb = {0}; sb_reserve(&sb, 1); sb.p[0] = 0;
    alt_to_text(&c.alt, &sb); sb_putc(&sb, '\n'); alt_to_text(&c.plist, &sb); sb_putc(&sb, '\n');
    column_free(&c); free(hp); free(st);
    return sb.p;
}

/* ================================================================================ A4: chunk -> region */
/* src/create_tensor_pileup.py:379-422.  mode 0: no bed (contig_len used); mode 1: bed given
 * (bed_start/bed_end of the extended split bed).  chunk_id is 1-based as on the CLI, or 0 = no chunking
 * (ctg_start/ctg_end passed through in io[0], io[1]).  out: ctg_start, ctg_end, extend_start, extend_end,
 * reference_start, reference_end (all 1-based as the reference uses them). */
void orc_chunk_region(int mode, int64_t contig_len, int64_t bed_start, int64_t bed_end, int chunk_id, int chunk_num, int64_t *io) {
    int64_t ctg_start = io[0], ctg_end = io[1];
    if (chunk_id > 0) {
        int id0 = chunk_id - 1;
        if (mode == 0) {
            int64_t cs = (contig_len % chunk_num) ? contig_len / chunk_num + 1 : contig_len / chunk_num;
            ctg_start = cs * id0; ctg_end = ctg_start + cs;
        } else {
            int64_t span = bed_end - bed_start;
            int64_t cs = (span % chunk_num) ? span / chunk_num + 1 : span / chunk_num;
            ctg_start = bed_start + 1 + cs * id0; ctg_end = ctg_start + cs;
        }
    }
    int64_t es = ctg_start - C3R_WINDOW, ee = ctg_end + C3R_WINDOW;
    if (es < 1) es = 1;
    int64_t rs = ctg_start - 1000, re = ctg_end + 1000;
    if (rs < 1) rs = 1;
    io[0] = ctg_start; io[1] = ctg_end; io[2] = es; io[3] = ee; io[4] = rs; io[5] = re;
}

/* ============================================================================ A3: window driver */
typedef struct {
    double snp_af, indel_af;
    int32_t min_coverage;
    int32_t head_tail, splice_padding, phased;
    int32_t has_bed;  int32_t n_bed;  const int32_t *bed;     /* confident bed, 0-based half-open, any order */
    int32_t has_sites; int32_t n_sites; const int32_t *sites; /* genotyping mode: 1-based positions */
    int32_t platform_hifi;  /* indel_af default when < 0 */
} orc_ct_params;

typedef struct { int32_t *v; int refcnt_unused; } colbuf;

static int bed_overlap(const int32_t *bed, int n, int64_t b, int64_t e) {
    for (int i = 0; i < n; ++i) if (bed[2 * i] < e && bed[2 * i + 1] > b) return 1;
    return 0;
}
static int site_in(const int32_t *s, int n, int64_t pos) { for (int i = 0; i < n; ++i) if (s[i] == pos) return 1; return 0; }

typedef struct { int64_t pos; int depth; int has_depth; int skip; int has_skip; } posinfo;

/* get_flanked_sequence, src/create_tensor_pileup.py:313-331 */
static void flanked(const char *ref, int64_t ref_len, int64_t center, int64_t reference_start, char *out33) {
    int64_t l = center - C3R_FLANK - reference_start, r = center + C3R_FLANK + 1 - reference_start;
    int k = 0;
    for (int64_t i = l; i < r; ++i) out33[k++] = (i < 0 || i >= ref_len) ? 'A' : ref[i];
    out33[k] = 0;
}

static void emit_line(sbuf *out, const char *ctg, int64_t center, const char *ref33, int32_t **win, int C, int depth, const odict *alt) {
    sb_puts(out, ctg); sb_putc(out, '\t'); sb_putl(out, center); sb_putc(out, '\t'); sb_puts(out, ref33); sb_putc(out, '\t');
    for (int i = 0; i < C3R_WINDOW; ++i)
        for (int j = 0; j < C; ++j) { if (i || j) sb_putc(out, ' '); sb_putl(out, win[i][j]); }
    sb_putc(out, '\t'); sb_putl(out, depth); sb_putc(out, '-'); alt_to_text(alt, out); sb_putc(out, '\n');
}

/* rows_text: mpileup rows.  ref_seq[0] is 1-based position reference_start.  Returns the lines the
 * reference would write to its stdout pipe. */
char *orc_create_tensor(const char *rows_text, const char *ctg, const char *ref_seq, int64_t reference_start,
                        const orc_ct_params *P, int64_t *out_len) {
    const int C = P->phased ? C3R_CH_PHASED : C3R_CH;
    const int64_t ref_len = (int64_t)strlen(ref_seq);
    sbuf out = {0}; sb_reserve(&out, 1); out.p[0] = 0;
    /* the ring holds POINTERS so that Python's list aliasing ([[0]*C]*33 and in-place padding writes
     * that persist in later windows, src/create_tensor_pileup.py:467,571,592-593) is reproduced */
    int32_t *ring[C3R_WINDOW];
    /* all column buffers are kept until the end (simple arena) */
    int32_t **arena = NULL; size_t n_arena = 0, cap_arena = 0;
#define NEWCOL(dst) do { if (n_arena == cap_arena) { cap_arena = cap_arena ? cap_arena * 2 : 1024; arena = (int32_t **)realloc(arena, cap_arena * sizeof *arena); } \
        arena[n_arena] = (int32_t *)calloc((size_t)C, sizeof(int32_t)); (dst) = arena[n_arena++]; } while (0)
#define RESET_RING() do { if (P->head_tail) { int32_t *z; NEWCOL(z); for (int i_ = 0; i_ < C3R_WINDOW; ++i_) ring[i_] = z; } \
        else for (int i_ = 0; i_ < C3R_WINDOW; ++i_) ring[i_] = NULL; } while (0)
    RESET_RING();
    int pos_offset = 0; int64_t pre_pos = -1;
    /* candidate queue + per-candidate alt dict / depth */
    typedef struct { int64_t pos; odict alt; int depth; int alive; } cand_t;
    cand_t *cands = NULL; size_t n_c = 0, cap_c = 0, head_c = 0;
    /* depth_dict / max_skip_count_dict for splice padding: sparse by position, kept in a growing array */
    posinfo *pinfo = NULL; size_t n_pi = 0, cap_pi = 0;

    const char *p = rows_text;
    while (*p) {
        const char *eol = strchr(p, '\n'); size_t ll = eol ? (size_t)(eol - p) : strlen(p);
        /* split columns by tab, maxsplit 6 */
        const char *col[7]; int clen[7]; int nc = 0; const char *q = p, *end = p + ll;
        /* row.strip(): trim trailing whitespace */
        while (end > p && (end[-1] == ' ' || end[-1] == '\t' || end[-1] == '\r')) end--;
        while (nc < 6) { const char *t = memchr(q, '\t', (size_t)(end - q)); if (!t) break; col[nc] = q; clen[nc] = (int)(t - q); nc++; q = t + 1; }
        col[nc] = q; clen[nc] = (int)(end - q); nc++;
        p = eol ? eol + 1 : p + ll;
        if (nc < 5) continue;
        int64_t pos = strtoll(col[1], NULL, 10);
        const char *bases = col[4]; int blen = clen[4];
        char **hp = NULL; int n_hp = 0; char *hp_st = NULL; char *hpcopy = NULL;
        if (nc >= 7) { hpcopy = (char *)malloc((size_t)clen[6] + 1); memcpy(hpcopy, col[6], (size_t)clen[6]); hpcopy[clen[6]] = 0; hp = split_hp(hpcopy, &n_hp, &hp_st); }
        char reference_base = (char)toupper((unsigned char)ref_seq[pos - reference_start]);

        if (pre_pos + 1 != pos) {
            pos_offset = 0; RESET_RING();
            for (size_t i = head_c; i < n_c; ++i) if (cands[i].alive) { cands[i].alive = 0; }
            head_c = n_c;   /* candidate_position = [] (dict entries intentionally linger, as in the reference) */
        }
        pre_pos = pos;
        column_t c;
        char *bcopy = (char *)malloc((size_t)blen + 1); memcpy(bcopy, bases, (size_t)blen); bcopy[blen] = 0;
        generate_tensor(bcopy, blen, hp, n_hp, pos, ref_seq, ref_len, reference_start, reference_base, P->snp_af, P->indel_af,
                        P->platform_hifi ? 0.08 : 0.15, &c);
        free(bcopy); free(hp); free(hp_st); free(hpcopy);
        int depth = c.depth, pass_af = c.pass_af;
        if (P->splice_padding) {
            if (n_pi == cap_pi) { cap_pi = cap_pi ? cap_pi * 2 : 4096; pinfo = (posinfo *)realloc(pinfo, cap_pi * sizeof *pinfo); }
            posinfo pi = { pos, depth, 1, c.max_skip_count, 1 }; pinfo[n_pi++] = pi;
        }
        double eff_snp = P->snp_af, eff_indel = P->indel_af;
        if (depth > 0 && (eff_snp == 0.0 || eff_indel == 0.0)) pass_af = 1;
        int pass_bed = !P->has_bed || bed_overlap(P->bed, P->n_bed, pos - 1, pos + c.max_del_length + 1);
        int is_cand;
        if (P->has_sites) is_cand = site_in(P->sites, P->n_sites, pos);
        else is_cand = pass_bed && strchr("ACGT", reference_base) && reference_base && pass_af && depth >= P->min_coverage;
        if (is_cand) {
            if (n_c == cap_c) { cap_c = cap_c ? cap_c * 2 : 256; cands = (cand_t *)realloc(cands, cap_c * sizeof *cands); }
            cands[n_c].pos = pos; cands[n_c].alt = c.alt; cands[n_c].depth = depth; cands[n_c].alive = 1; n_c++;
            c.alt.v = NULL; c.alt.n = c.alt.cap = 0;   /* ownership moved */
            if (!P->splice_padding) { /* depth_dict[pos] only matters for splice padding */ }
        }
        int32_t *colv; NEWCOL(colv); memcpy(colv, c.tensor, sizeof(int32_t) * (size_t)C);
        ring[pos_offset] = colv;
        column_free(&c);
        pos_offset = (pos_offset + 1) % C3R_WINDOW;
        if (head_c < n_c && pos - cands[head_c].pos == C3R_FLANK) {
            cand_t *cd = &cands[head_c++];
            int64_t center = cd->pos;
            int has_empty = 0; for (int i = 0; i < C3R_WINDOW; ++i) if (!ring[i]) has_empty = 1;
            if (!has_empty) {
                int cdepth = cd->depth;
                char ref33[C3R_WINDOW + 1]; flanked(ref_seq, ref_len, center, reference_start, ref33);
                int32_t *win[C3R_WINDOW];
                for (int i = 0; i < C3R_WINDOW; ++i) win[i] = ring[(pos_offset + i) % C3R_WINDOW];
                if (P->splice_padding) {
                    int max_depth = -1, max_skip = -1;
                    for (size_t k = n_pi; k-- > 0;) {
                        if (pinfo[k].pos < center - C3R_FLANK) break;   /* rows arrive in increasing position order */
                        if (pinfo[k].pos >= center - C3R_FLANK && pinfo[k].pos <= center + C3R_FLANK) {
                            if (pinfo[k].has_depth && pinfo[k].depth > max_depth) max_depth = pinfo[k].depth;
                            if (pinfo[k].has_skip && pinfo[k].skip > max_skip) max_skip = pinfo[k].skip;
                        }
                    }
                    if ((double)max_skip / (double)max_depth > 0.2) {
                        /* BASE2INDEX[letter]: a reference letter that is no channel name raises KeyError in the reference;
                         * the build (oracle and HIP path) reads 0 / pads nothing there */
                        char rc = ref_seq[center - reference_start];
                        int cu = chan_of_char((char)toupper((unsigned char)rc)), cl = chan_of_char((char)tolower((unsigned char)rc));
                        int sf = cu >= 0 ? win[C3R_FLANK][cu] : 0;
                        int sr = cl >= 0 ? win[C3R_FLANK][cl] : 0;
                        if (sf < 0) sf = -sf; if (sr < 0) sr = -sr;
                        double fpct = (sf + sr > 0) ? sf / (double)(sf + sr) : 0.0;
                        double rpct = 1 - fpct;
                        for (int idx = 0; idx < C3R_WINDOW; ++idx) {
                            int64_t pp = center - C3R_FLANK + idx;
                            int cur = 0;
                            for (size_t k = n_pi; k-- > 0;) {
                                if (pinfo[k].pos < center - C3R_FLANK) break;
                                if (pinfo[k].pos == pp) { cur = pinfo[k].has_depth ? pinfo[k].depth : 0; break; }
                            }
                            if (cur < cdepth * 0.2 && idx != C3R_FLANK) {
                                int64_t ri = pp - reference_start;
                                if (ri < 0) ri += ref_len;   /* Python negative index: head slots left of the reference slice */
                                char rb = (char)toupper((unsigned char)ref_seq[ri]);
                                int pu = chan_of_char(rb), pl = chan_of_char((char)tolower((unsigned char)rb));
                                if (pu >= 0 && pl >= 0) {
                                    win[idx][pu] = -1 * (int)(cdepth * fpct);
                                    win[idx][pl] = -1 * (int)(cdepth * rpct);
                                }
                            }
                        }
                    }
                }
                emit_line(&out, ctg, center, ref33, win, C, cdepth, &cd->alt);
                /* del depth_dict[center] (src/create_tensor_pileup.py:611) */
                if (P->splice_padding)
                    for (size_t k = n_pi; k-- > 0;) { if (pinfo[k].pos == center) { pinfo[k].has_depth = 0; break; } if (pinfo[k].pos < center) break; }
            }
            od_free(&cd->alt); cd->alive = 0;
        }
    }
    if (P->head_tail && pre_pos >= 0) {
        int64_t ens = pre_pos + C3R_FLANK;
        for (int64_t pos = pre_pos + 1; pos <= ens; ++pos) {
            int32_t *z; NEWCOL(z); ring[pos_offset] = z;
            pos_offset = (pos_offset + 1) % C3R_WINDOW;
            int64_t center = pos - C3R_FLANK;
            for (size_t i = head_c; i < n_c; ++i) {
                if (cands[i].alive && cands[i].pos == center) {
                    int has_empty = 0; for (int k = 0; k < C3R_WINDOW; ++k) if (!ring[k]) has_empty = 1;
                    if (!has_empty) {
                        char ref33[C3R_WINDOW + 1]; flanked(ref_seq, ref_len, center, reference_start, ref33);
                        int32_t *win[C3R_WINDOW];
                        for (int k = 0; k < C3R_WINDOW; ++k) win[k] = ring[(pos_offset + k) % C3R_WINDOW];
                        emit_line(&out, ctg, center, ref33, win, C, cands[i].depth, &cands[i].alt);
                    }
                    break;
                }
            }
        }
    }
    for (size_t i = 0; i < n_c; ++i) if (cands[i].alt.v) od_free(&cands[i].alt);
    for (size_t i = 0; i < n_arena; ++i) free(arena[i]);
    free(arena); free(cands); free(pinfo);
    if (out_len) *out_len = (int64_t)out.n;
    return out.p;
#undef NEWCOL
#undef RESET_RING
}

/* ==================================================================== A5: lines -> int32 batch */
/* clair3_rna/utils.py:64-138.  Parses candidate lines; out must hold n_lines*33*C int32.  depth_out
 * (optional) receives the parsed depth.  Returns the number of rows written (rows whose centre
 * reference base is not an IUPAC letter are skipped, utils.py:113). */
int64_t orc_batch_from_lines(const char *lines, int C, int32_t *out, int32_t *depth_out) {
    const char *p = lines; int64_t n = 0;
    const int W = C3R_WINDOW * C;
    while (*p) {
        const char *eol = strchr(p, '\n'); size_t ll = eol ? (size_t)(eol - p) : strlen(p);
        const char *f[5]; int nf = 0; const char *q = p, *end = p + ll;
        f[nf++] = q;
        while (nf < 5) { const char *t = memchr(q, '\t', (size_t)(end - q)); if (!t) break; q = t + 1; f[nf++] = q; }
        p = eol ? eol + 1 : p + ll;
        if (nf < 5) continue;
        const char *seq = f[2];
        if (!strchr("ACGTURYSWKMBDHVN", seq[C3R_FLANK])) continue;
        int depth = (int)strtol(f[4], NULL, 10);
        const char *t = f[3]; char *e;
        int32_t *row = out + n * W;
        for (int i = 0; i < W; ++i) { row[i] = (int32_t)strtol(t, &e, 10); t = e; }
        if (depth > 0 && (double)depth > 144 * 1.5) {
            double scale = (double)depth / 144;
            for (int i = 0; i < W; ++i) row[i] = (int32_t)((double)row[i] / scale);   /* C cast truncates toward zero, like numpy's int32 store */
        }
        if (depth_out) depth_out[n] = depth;
        n++;
    }
    return n;
}

/* ==================================================================== A6: network forward (fp32) */
/* Weight blob layout (floats, in this order; all matrices row-major [in][out], Keras layout):
 *   for layer in (LSTM1, LSTM2): for dir in (fwd, bwd): K[in,4H], R[H,4H], b[4H]     gate order i,f,c,o
 *   L4: W[33*320,128], b[128];  L5_1: W[128,128], b;  L5_2: W[128,128], b;
 *   Y_gt21: W[128,21], b[21];  Y_genotype: W[128,3], b[3]
 * (clair3_rna/model.py:126-170; LSTM1 128 units, LSTM2 160 units, params dict at :45-83) */
#define H1 128
#define H2 160
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float seluf_(float x) {
    const float scale = 1.0507009873554805f, alpha = 1.6732632423543772f;
    return x > 0 ? scale * x : scale * alpha * (expf(x) - 1.0f);
}

int64_t orc_weight_count(int C) {
    int64_t n = 0;
    n += 2 * ((int64_t)C * 4 * H1 + (int64_t)H1 * 4 * H1 + 4 * H1);
    n += 2 * ((int64_t)2 * H1 * 4 * H2 + (int64_t)H2 * 4 * H2 + 4 * H2);
    n += (int64_t)C3R_WINDOW * 2 * H2 * 128 + 128;
    n += 2 * (128 * 128 + 128);
    n += 128 * 21 + 21 + 128 * 3 + 3;
    return n;
}

/* one direction of one LSTM layer for one site: x[33][in] -> y[33][ystride] at column offset */
static void lstm_dir(const float *x, int in, int H, const float *K, const float *R, const float *b, int reverse,
                     float *y, int ystride, int yoff, float *z /* 4H scratch */) {
    float h[H2], c[H2];
    for (int j = 0; j < H; ++j) h[j] = c[j] = 0.f;
    for (int s = 0; s < C3R_WINDOW; ++s) {
        int t = reverse ? C3R_WINDOW - 1 - s : s;
        const float *xt = x + (size_t)t * in;
        for (int j = 0; j < 4 * H; ++j) z[j] = b[j];
        for (int k = 0; k < in; ++k) { float xv = xt[k]; const float *Kr = K + (size_t)k * 4 * H; for (int j = 0; j < 4 * H; ++j) z[j] += xv * Kr[j]; }
        for (int k = 0; k < H; ++k) { float hv = h[k]; const float *Rr = R + (size_t)k * 4 * H; for (int j = 0; j < 4 * H; ++j) z[j] += hv * Rr[j]; }
        for (int j = 0; j < H; ++j) {
            float ig = sigmoidf_(z[j]), fg = sigmoidf_(z[H + j]), gg = tanhf(z[2 * H + j]), og = sigmoidf_(z[3 * H + j]);
            c[j] = fg * c[j] + ig * gg;
            h[j] = og * tanhf(c[j]);
            y[(size_t)t * ystride + yoff + j] = h[j];
        }
    }
}

/* X: int32 [n][33][C]; probs: [n][24].  y1_dbg/y2_dbg optional ([n][33][256] / [n][33][320]). */
void orc_forward(const float *w, int C, const int32_t *X, int64_t n, float *probs, float *y1_dbg, float *y2_dbg) {
    const float *K1[2], *R1[2], *b1[2], *K2[2], *R2[2], *b2[2];
    const float *q = w;
    for (int d = 0; d < 2; ++d) { K1[d] = q; q += (size_t)C * 4 * H1; R1[d] = q; q += (size_t)H1 * 4 * H1; b1[d] = q; q += 4 * H1; }
    for (int d = 0; d < 2; ++d) { K2[d] = q; q += (size_t)2 * H1 * 4 * H2; R2[d] = q; q += (size_t)H2 * 4 * H2; b2[d] = q; q += 4 * H2; }
    const float *W4 = q; q += (size_t)C3R_WINDOW * 2 * H2 * 128; const float *b4 = q; q += 128;
    const float *W51 = q; q += 128 * 128; const float *b51 = q; q += 128;
    const float *W52 = q; q += 128 * 128; const float *b52 = q; q += 128;
    const float *Wg = q; q += 128 * 21; const float *bg = q; q += 21;
    const float *Wz = q; q += 128 * 3; const float *bz = q; q += 3;
#pragma omp parallel
    {
        float *x0 = (float *)malloc(sizeof(float) * C3R_WINDOW * (size_t)C);
        float *y1 = (float *)malloc(sizeof(float) * C3R_WINDOW * 2 * H1);
        float *y2 = (float *)malloc(sizeof(float) * C3R_WINDOW * 2 * H2);
        float *z = (float *)malloc(sizeof(float) * 4 * H2);
#pragma omp for schedule(dynamic, 4)
        for (int64_t s = 0; s < n; ++s) {
            const int32_t *xs = X + (size_t)s * C3R_WINDOW * C;
            for (int i = 0; i < C3R_WINDOW * C; ++i) x0[i] = (float)xs[i];
            for (int d = 0; d < 2; ++d) lstm_dir(x0, C, H1, K1[d], R1[d], b1[d], d, y1, 2 * H1, d * H1, z);
            for (int d = 0; d < 2; ++d) lstm_dir(y1, 2 * H1, H2, K2[d], R2[d], b2[d], d, y2, 2 * H2, d * H2, z);
            if (y1_dbg) memcpy(y1_dbg + (size_t)s * C3R_WINDOW * 2 * H1, y1, sizeof(float) * C3R_WINDOW * 2 * H1);
            if (y2_dbg) memcpy(y2_dbg + (size_t)s * C3R_WINDOW * 2 * H2, y2, sizeof(float) * C3R_WINDOW * 2 * H2);
            float a4[128], a51[128], a52[128], l21[21], l3[3];
            for (int j = 0; j < 128; ++j) a4[j] = b4[j];
            for (int k = 0; k < C3R_WINDOW * 2 * H2; ++k) { float v = y2[k]; const float *wr = W4 + (size_t)k * 128; for (int j = 0; j < 128; ++j) a4[j] += v * wr[j]; }
            for (int j = 0; j < 128; ++j) a4[j] = seluf_(a4[j]);
            for (int j = 0; j < 128; ++j) { a51[j] = b51[j]; a52[j] = b52[j]; }
            for (int k = 0; k < 128; ++k) { float v = a4[k]; for (int j = 0; j < 128; ++j) { a51[j] += v * W51[k * 128 + j]; a52[j] += v * W52[k * 128 + j]; } }
            for (int j = 0; j < 128; ++j) { a51[j] = seluf_(a51[j]); a52[j] = seluf_(a52[j]); }
            for (int j = 0; j < 21; ++j) l21[j] = bg[j];
            for (int j = 0; j < 3; ++j) l3[j] = bz[j];
            for (int k = 0; k < 128; ++k) { for (int j = 0; j < 21; ++j) l21[j] += a51[k] * Wg[k * 21 + j]; for (int j = 0; j < 3; ++j) l3[j] += a52[k] * Wz[k * 3 + j]; }
            float m = -1e30f, sum = 0.f;
            for (int j = 0; j < 21; ++j) { l21[j] = seluf_(l21[j]); if (l21[j] > m) m = l21[j]; }
            for (int j = 0; j < 21; ++j) { l21[j] = expf(l21[j] - m); sum += l21[j]; }
            for (int j = 0; j < 21; ++j) probs[s * C3R_NPROB + j] = l21[j] / sum;
            m = -1e30f; sum = 0.f;
            for (int j = 0; j < 3; ++j) { l3[j] = seluf_(l3[j]); if (l3[j] > m) m = l3[j]; }
            for (int j = 0; j < 3; ++j) { l3[j] = expf(l3[j] - m); sum += l3[j]; }
            for (int j = 0; j < 3; ++j) probs[s * C3R_NPROB + 21 + j] = l3[j] / sum;
        }
        free(x0); free(y1); free(y2); free(z);
    }
}

/* =============================================================================== end-to-end helper */
/* reads -> mpileup text -> candidate lines -> int32 tensors (+rescale) [-> probabilities].
 * This is the leg bench.py times as cpu_baseline (kind "port").  Returns number of candidate rows;
 * *lines_out receives the malloc'd candidate lines (free with orc_free). */
int64_t orc_pipeline(const c3r_read_t *reads, int64_t n_reads, const uint32_t *cigar, const uint8_t *seq,
                     const char *ctg, int64_t ctg_start, int64_t ctg_end, const char *ref_seq, int64_t reference_start,
                     int min_mq, int excl_flags, const int32_t *lbed, int n_lbed, const orc_ct_params *P,
                     char **lines_out, int32_t **tensors_out, const float *weights, float **probs_out) {
    int64_t es = ctg_start - C3R_WINDOW, ee = ctg_end + C3R_WINDOW; if (es < 1) es = 1;
    int64_t tl = 0, ll = 0;
    char *rows = orc_mpileup(reads, n_reads, cigar, seq, ctg, es, ee, min_mq, excl_flags, lbed, n_lbed, P->phased, &tl);
    char *lines = orc_create_tensor(rows, ctg, ref_seq, reference_start, P, &ll);
    free(rows);
    int64_t nl = 0; for (const char *p = lines; *p; ++p) if (*p == '\n') nl++;
    const int C = P->phased ? C3R_CH_PHASED : C3R_CH;
    int64_t n = 0;
    if (tensors_out) {
        int32_t *T = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nl ? nl : 1) * C3R_WINDOW * C);
        n = orc_batch_from_lines(lines, C, T, NULL);
        *tensors_out = T;
        if (weights && probs_out) {
            float *pr = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1) * C3R_NPROB);
            orc_forward(weights, C, T, n, pr, NULL, NULL);
            *probs_out = pr;
        }
    } else n = nl;
    if (lines_out) *lines_out = lines; else free(lines);
    return n;
}
