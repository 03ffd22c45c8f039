"""A8 — probabilities -> genotype / ALT / QUAL -> VCF row (host side of the drop-in).

Restates, for the configuration run_clair3_rna actually uses (pileup model, --add_indel_length off, --showRef on,
--qual 2, no haploid modes, no long-indel mode), the decision procedure of
    clair3_rna/call_variants.py:518-569   possible_outcome_probabilites_from  (class probabilities, early RefCall)
    clair3_rna/call_variants.py:684-1020  output_from                         (arg-max with retry-by-zeroing loop)
    clair3_rna/call_variants.py:112-196   insertion_/deletion_bases_using_alt_info_from
    clair3_rna/call_variants.py:670-681   find_alt_base
    clair3_rna/call_variants.py:1117-1392 output_with                         (GT, AD, AF, QUAL, FILTER, row text)
Pinned by golden G4 (tests/golden/g4_decode.json).  Numeric semantics: class probabilities are float32
products; the Phred transform is evaluated in float64 (what the reference computes under its numpy < 1.24).
"""
from math import e, log

import numpy as np

GT21_LABELS = ['AA', 'AC', 'AG', 'AT', 'CC', 'CG', 'CT', 'GG', 'GT', 'TT', 'DelDel', 'ADel', 'CDel', 'GDel', 'TDel',
               'InsIns', 'AIns', 'CIns', 'GIns', 'TIns', 'InsDel']      # clair3_rna/task/gt21.py:3-25
GT21 = {k: i for i, k in enumerate(GT21_LABELS)}
HOMO_SNP = ['AA', 'CC', 'GG', 'TT']
HETERO_SNP = ['AC', 'AG', 'AT', 'CG', 'CT', 'GT']
IUPAC2ACGT = dict(zip("ACGTURYSWKMBDHVN", "ACGTTACCAGACAAAA"))              # shared/utils.py:41-44
BASIC_BASES = set("ACGTU")
MAX_INFER_LEN = 50                                                        # shared/param_p.py:16
PHRED_TRANS = -10 * log(e, 10)                                            # call_variants.py:59
ACGT = "ACGT"


def parse_alt_info(alt_info):
    """'<depth>-<k v k v ...>' -> (depth, ordered dict)   (call_variants.py:1155-1159)"""
    parts = alt_info.rstrip().split('-')
    depth = int(parts[0])
    seqs = (parts[1] if len(parts) > 1 else '').split(' ')
    return depth, dict(zip(seqs[::2], [int(x) for x in seqs[1::2]]))


def _first_max_key(d):
    best, bk = None, ""
    for k, v in d.items():
        if best is None or v > best:
            best, bk = v, k
    return bk


def _indel_candidates(alt, tag):
    out = {}
    for raw, cnt in alt.items():
        if raw[0] != tag:
            continue
        key = raw[1:]
        if 1 <= len(key) <= MAX_INFER_LEN:
            out[key] = cnt
    return out


def _best_insertion(alt):
    d = _indel_candidates(alt, 'I')
    return _first_max_key(d) if d else ""


def _two_insertions(alt):
    items = list(_indel_candidates(alt, 'I').items())
    ranked = [k for k, _ in sorted(items, key=lambda x: x[1])[::-1]]     # ascending stable sort, reversed (ties flip)
    return ranked[:2] if ranked else ""


def _best_deletion(alt):
    d = _indel_candidates(alt, 'D')
    return _first_max_key(d) if d else ""


def _two_deletions(alt):
    items = list(_indel_candidates(alt, 'D').items())
    ranked = [k for k, _ in sorted(items, key=lambda x: x[1])[::-1]]
    if len(ranked) <= 1:
        return ""
    return [ranked[0], ranked[1]] if len(ranked[0]) > len(ranked[1]) else [ranked[1], ranked[0]]


def _find_alt_base(alt, proposed=None):
    ranked = sorted([(k[1], c) for k, c in alt.items() if k[0] == 'X'], key=lambda x: x[1], reverse=True)
    if not ranked:
        return [], None
    mine = [c for b, c in ranked if b == proposed]
    if not mine or ranked[0][1] - mine[0] >= 9:        # max_depth_gap (call_variants.py:672)
        proposed = ranked[0][0]
    return [b for b, _ in ranked], proposed


def quality_score_from(p):
    p = float(p)
    return float(round(max(PHRED_TRANS * log(((1.0 - p) + 1e-10) / (p + 1e-10)) + 10, 0), 2))


def convert_iupac_to_n(s):
    if s == ".":
        return s
    return "".join(c if c.upper() in "ACGTN,." else 'N' for c in s)


def call_site(gt21_p, zyg_p, ref33, alt):
    """-> (flags dict, ref_allele, alt_allele, probability).  gt21_p/zyg_p are float32 arrays."""
    center = ref33[16] if len(ref33) > 1 else ref33[0]
    ref_acgt = IUPAC2ACGT[center]
    z0, z1, z2 = zyg_p[0], zyg_p[1], zyg_p[2]
    rr = GT21[ref_acgt + ref_acgt]
    p_ref = z0 * gt21_p[rr]
    names = ("homo_snp", "het_snp", "homo_ins", "het_base_ins", "het_insins", "homo_del", "het_base_del", "het_deldel", "insdel")
    none = dict.fromkeys(names, False)
    if z0 >= 0.5 and gt21_p[rr] >= 0.5:
        return dict(none, ref=True), ref_acgt, ref_acgt, p_ref
    cls = {
        "homo_snp": [z1 * gt21_p[GT21[k]] for k in HOMO_SNP],
        "het_snp": [z2 * gt21_p[GT21[k]] for k in HETERO_SNP],
        "homo_ins": [z1 * gt21_p[GT21['InsIns']]],
        "het_insins": [z2 * gt21_p[GT21['InsIns']]],
        "het_base_ins": [gt21_p[GT21[b + 'Ins']] * z2 for b in ACGT],
        "homo_del": [z1 * gt21_p[GT21['DelDel']]],
        "het_deldel": [z2 * gt21_p[GT21['DelDel']]],
        "het_base_del": [gt21_p[GT21[b + 'Del']] * z2 for b in ACGT],
        "insdel": [z2 * gt21_p[GT21['InsDel']]],
    }
    # NOTE the loop condition: the reference loops `while reference_base is None or alternate_base is None`
    # (call_variants.py:730) and several branches assign both alleles BEFORE a late `continue`; such a
    # `continue` therefore leaves the loop with the alleles (and flags) of that iteration.  Reproduced as is.
    ref_allele = alt_allele = None
    flags, top = dict(none, ref=False), p_ref
    while ref_allele is None or alt_allele is None:
        top = max([p_ref] + [max(v) for v in cls.values()])
        if top == p_ref:
            return dict(none, ref=True), ref_acgt, ref_acgt, top
        hit = {k: (top in cls[k]) for k in names}
        flags = dict(hit, ref=False)
        if hit["homo_snp"]:
            probs = cls["homo_snp"]
            ref_allele = center
            idx = probs.index(top)
            lab = HOMO_SNP[int(np.argmax(probs))]
            alt_allele = lab[0] if lab[0] != ref_allele else lab[1]
            _, alt_allele = _find_alt_base(alt, alt_allele)
            if alt_allele is None or alt_allele == ref_allele:
                probs[idx] = 0
                continue
        elif hit["het_snp"]:
            probs = cls["het_snp"]
            lab = HETERO_SNP[int(np.argmax(probs))]
            idx = probs.index(top)
            ref_allele = center
            if lab[0] != ref_allele and lab[1] != ref_allele:
                ranked, _ = _find_alt_base(alt)
                if len(ranked) < 2:
                    probs[idx] = 0
                    continue
                alt_allele = ','.join(ranked[:2])
            else:
                alt_allele = lab[0] if lab[0] != ref_allele else lab[1]
                _, alt_allele = _find_alt_base(alt, alt_allele)
                if alt_allele is None or alt_allele == ref_allele:
                    probs[idx] = 0
                    continue
        elif hit["homo_ins"]:
            ins = _best_insertion(alt)
            if not ins:
                cls["homo_ins"][0] = 0
                continue
            ref_allele, alt_allele = center, ins
        elif hit["het_base_ins"]:
            probs = cls["het_base_ins"]
            idx = probs.index(top)
            ins = _best_insertion(alt)
            if not ins:
                probs[idx] = 0
                continue
            ref_allele, alt_allele = center, ins
            if ACGT[idx] != ref_allele:
                ranked, _ = _find_alt_base(alt)
                if not ranked:
                    probs[idx] = 0
                    continue
                alt_allele = "%s,%s" % (ranked[0], alt_allele)
        elif hit["het_insins"]:
            two = _two_insertions(alt)
            if len(two) < 2:
                cls["het_insins"][0] = 0
                continue
            ref_allele, alt_allele = center, two[0]
            if two[1] != alt_allele:
                alt_allele = "%s,%s" % (two[1], alt_allele)
            else:
                cls["het_insins"][0] = 0
                continue
        elif hit["homo_del"]:
            d = _best_deletion(alt)
            if not d:
                cls["homo_del"][0] = 0
                continue
            ref_allele = center + d
            alt_allele = ref_allele[0]
        elif hit["het_base_del"]:
            probs = cls["het_base_del"]
            idx = probs.index(top)
            d = _best_deletion(alt)
            if not d:
                probs[idx] = 0
                continue
            ref_allele = center + d
            alt_allele = ref_allele[0]
            if ACGT[idx] != ref_allele[0]:
                alt_allele = "%s,%s" % (alt_allele, ACGT[idx] + ref_allele[1:])
        elif hit["het_deldel"]:
            two = _two_deletions(alt)
            if len(two) < 2:
                cls["het_deldel"][0] = 0
                continue
            ref_allele = center + two[0]
            alt_allele = ref_allele[0]
            a1 = alt_allele
            a2 = ref_allele[0] + ref_allele[len(two[1]) + 1:]
            if a1 != a2 and ref_allele != a1 and ref_allele != a2:
                alt_allele = "%s,%s" % (a1, a2)
            else:
                cls["het_deldel"][0] = 0
                continue
        elif hit["insdel"]:
            ins, d = _best_insertion(alt), _best_deletion(alt)
            if not ins or not d:
                cls["insdel"][0] = 0
                continue
            ref_allele = center + d
            alt_allele = "%s,%s" % (ref_allele[0], ins + ref_allele[1:])
    return flags, ref_allele, alt_allele, top


def vcf_row(ctg, pos, ref33, alt_info, probs24, qual_for_pass=2, show_ref=True):
    """One candidate -> VCF text row (or None when the reference would print nothing)."""
    probs24 = np.asarray(probs24, dtype=np.float32)
    depth, alt = parse_alt_info(alt_info)
    f, ref_allele, alt_allele, p = call_site(probs24[:21], probs24[21:24], ref33, alt)
    is_ref = f["ref"]
    if (not show_ref and is_ref) or (not is_ref and ref_allele == alt_allele):
        return None
    if ref_allele is None or alt_allele is None:
        return None
    multi = "," in str(alt_allele)
    if is_ref:
        gt = "0/0"
    elif f["homo_snp"] or f["homo_ins"] or f["homo_del"]:
        gt = "1/1"
    elif f["het_snp"] or f["het_base_ins"] or f["het_insins"] or f["het_base_del"] or f["het_deldel"]:
        gt = "0/1"
    else:
        gt = None
    if multi:
        gt = "1/2"
    snp, ins, dele, ref_count = {}, {}, {}, 0
    for k, c in alt.items():
        if k[0] == 'X':
            snp[k[1]] = c
        elif k[0] == 'I':
            ins[k[1:]] = c
        elif k[0] == 'D':
            dele[k[1:]] = c
        elif k[0] == 'R':
            ref_count = c
    ref_count = max(0, ref_count)
    support, per_alt = 0, []
    if is_ref:
        support = ref_count
        alt_allele = "."
    elif f["homo_snp"] or f["het_snp"]:
        for b in str(alt_allele):
            if b == ',':
                continue
            support += snp.get(b, 0)
            per_alt.append(support)                       # cumulative, as the reference does (:1252-1255)
    elif f["homo_ins"] or f["het_insins"]:
        for s in alt_allele.split(','):
            n = ins.get(s, 0)
            support += n
            per_alt.append(n)
    elif f["het_base_ins"]:
        snp_base = alt_allele.split(",")[0][0] if multi else None
        s = alt_allele.split(",")[1] if multi else alt_allele
        n_snp = snp.get(snp_base, 0) if multi else 0
        n_ins = ins.get(s, 0)
        support = n_ins + n_snp
        if snp_base:
            per_alt.append(n_snp)
        per_alt.append(n_ins)
    elif f["homo_del"] or f["het_deldel"]:
        if len(dele) > 0:
            if f["homo_del"]:
                d = ref_allele[1:] if len(ref_allele) > 1 else None
                support = dele.get(d, 0)
                per_alt.append(support)
            elif f["het_deldel"] and len(dele) > 1:
                for a in alt_allele.split(','):
                    L = len(ref_allele) - len(a)
                    hits = [dele[k] for k in dele if len(k) == L]
                    n = hits[0] if hits else 0
                    per_alt.append(n)
                    support += n
    elif f["het_base_del"]:
        parts = alt_allele.split(",")
        snp_base = (parts[1][0] if len(parts) > 1 else None) if multi else None
        n_snp = snp.get(snp_base, 0) if multi else 0
        d = ref_allele[1:] if len(ref_allele) > 1 else None
        n_del = dele.get(d, 0)
        support = n_del + n_snp
        if snp_base:
            per_alt.append(n_snp)
        per_alt.append(n_del)
    elif f["insdel"]:
        for a in alt_allele.split(','):
            L = len(ref_allele) - len(a)
            if L < 0:
                s = a[:-(len(ref_allele) - 1)] if len(ref_allele) > 1 else a
                n = ins.get(s, 0)
            else:
                hits = [dele[k] for k in dele if len(k) == L]
                n = hits[0] if hits else 0
            per_alt.append(n)
            support += n
    af = ((support + 0.0) / depth) if depth != 0 else 0.0
    if af > 1:
        af = 1
    q = quality_score_from(p)
    filt = "RefCall" if is_ref else ("PASS" if (qual_for_pass is None or q >= qual_for_pass) else "LowQual")
    ref_allele = convert_iupac_to_n(ref_allele)
    alt_allele = convert_iupac_to_n(alt_allele)
    ad = str(ref_count) + ((',' + ','.join(str(x) for x in per_alt)) if per_alt else "")
    af_s = "%.4f" % af if len(per_alt) <= 1 else ','.join("%.4f" % min(1.0, 1.0 * x / depth) for x in per_alt)
    return "%s\t%d\t.\t%s\t%s\t%.2f\t%s\t%s\tGT:GQ:DP:AD:AF\t%s:%d:%d:%s:%s" % (
        ctg, pos, ref_allele, alt_allele, q, filt, ".", gt, q, depth, ad, af_s)


def vcf_rows(ctg, sites_pos, ref33_list, alt_info_list, probs, qual_for_pass=2, show_ref=True):
    """batch_output (call_variants.py:1077-1114): rows in input order, skipping sites that print nothing."""
    out = []
    for pos, r33, ai, y in zip(sites_pos, ref33_list, alt_info_list, probs):
        row = vcf_row(ctg, int(pos), r33, ai, y, qual_for_pass, show_ref)
        if row is not None:
            out.append(row)
    return out
