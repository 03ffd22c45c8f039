// decode.hpp — host-side A8: probabilities -> genotype / ALT / QUAL -> VCF row, and the ordered alt_info
// dictionary from per-read tokens.  C++ twin of clair3_rna_amd/decode.py + altinfo.py (same golden G4 pins both);
// restates clair3_rna/call_variants.py:518-569 (class probabilities, early RefCall), :684-1020 (arg-max with the
// retry-by-zeroing loop and its `while ref is None or alt is None` exit quirk), :112-196, :670-681, :1117-1392.
// Numerics: class probabilities are float32 products; the Phred transform is float64.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/c3r_types.h"

namespace c3r {

typedef std::vector<std::pair<std::string, int>> AltDict;   // insertion-ordered, keys unique

inline void alt_add(AltDict &d, const std::string &k, int c) {
    for (auto &kv : d) if (kv.first == k) { kv.second += c; return; }
    d.emplace_back(k, c);
}
inline void alt_set(AltDict &d, const std::string &k, int c) {   // dict(zip(...)): last assignment wins
    for (auto &kv : d) if (kv.first == k) { kv.second = c; return; }
    d.emplace_back(k, c);
}
inline int alt_get(const AltDict &d, const std::string &k, int dflt = 0) {
    for (auto &kv : d) if (kv.first == k) return kv.second;
    return dflt;
}

// "<depth>-<k v k v ...>" -> (depth, ordered dict)
inline void parse_alt_info(const char *s, int &depth, AltDict &alt) {
    std::string t(s);
    while (!t.empty() && (t.back() == '\n' || t.back() == ' ' || t.back() == '\r' || t.back() == '\t')) t.pop_back();
    const size_t dash = t.find('-');
    depth = atoi(t.substr(0, dash).c_str());
    alt.clear();
    if (dash == std::string::npos) return;
    // Python: parts[1] only (text up to the next '-'); keys never contain '-'
    std::string rest = t.substr(dash + 1);
    const size_t d2 = rest.find('-');
    if (d2 != std::string::npos) rest = rest.substr(0, d2);
    std::vector<std::string> tok;
    size_t p = 0;
    while (true) {
        const size_t q = rest.find(' ', p);
        tok.push_back(rest.substr(p, q == std::string::npos ? std::string::npos : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    for (size_t i = 0; i + 1 < tok.size(); i += 2) alt_set(alt, tok[i], atoi(tok[i + 1].c_str()));
}

inline std::string alt_info_string(int depth, const AltDict &alt) {
    std::string s = std::to_string(depth) + "-";
    bool first = true;
    for (auto &kv : alt) { if (!first) s += ' '; first = false; s += kv.first; s += ' '; s += std::to_string(kv.second); }
    return s;
}

static const char NT16_STR[] = "=ACMGRSVTWYHKDBN";

// tokens of one site (BAM order) -> ordered alt dict (src/create_tensor_pileup.py:179,221-261)
struct ReadView { const uint8_t *seq; uint64_t seq_off; uint32_t l_seq; };
// the upper-cased reference slice as the library holds it (a page-locked buffer: the upload's source and the decoder's view)
struct RefView {
    const char *p; size_t n;
    size_t size() const { return n; }
    char operator[](size_t i) const { return p[i]; }
    std::string substr(size_t a, size_t len) const { return std::string(p + a, len); }
};
// what the dictionary needs of one token; `next(i)` is called for i = 0 .. n-1 in order (a packed token stream keeps a cursor)
struct TokView { int base; int32_t indel; uint32_t read_idx, qpos, del_after; bool rev; };
// mpileup_compat = 1: the loaded reads' insertions that hold pads (c3r_padins_t, sorted by read and query offset; usually none)
struct PadView {
    const c3r_padins_t *p; size_t n;
    const c3r_padins_t *find(uint32_t r, uint32_t q) const {
        size_t lo = 0, hi = n;
        while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (p[mid].read_idx < r || (p[mid].read_idx == r && p[mid].qpos < q)) lo = mid + 1; else hi = mid; }
        return (lo < n && p[lo].read_idx == r && p[lo].qpos == q) ? &p[lo] : nullptr;
    }
};
constexpr int COUNT_DEPTH = -1;
template <typename NextTok, typename GetRead>
// site_depth >= 0 (what every caller passes): the tokens are only those of reads that show something other than the reference base or a ref-skip
// (what the kernels write since round 5) and the column's depth comes from the site record.  COUNT_DEPTH: one token per covering read, the
// depth is counted here (the legacy token stream; asked for by name, never by leaving the argument out).
inline void alt_from_stream(int n, NextTok next, GetRead get_read, const RefView &ref, int64_t ref_start1, int64_t pos1,
                            AltDict &alt, int &depth_out, const PadView pads, int site_depth) {
    alt.clear();
    const int64_t ri = pos1 - ref_start1;
    char rb = (ri >= 0 && ri < (int64_t)ref.size()) ? ref[(size_t)ri] : 'N';
    if (rb != 'A' && rb != 'C' && rb != 'G' && rb != 'T') rb = 'A';
    int depth = 0, alt_count = 0, ins_count = 0, del_count = 0;
    for (int i = 0; i < n; ++i) {
        const TokView t = next(i);
        const int b = t.base;
        if (b == 1 || b == 2 || b == 4 || b == 8) {
            depth++;
            const char u = NT16_STR[b];
            if (u != rb) { alt_add(alt, std::string("X") + u, 1); alt_count++; }
        } else if (b == 16) { depth++; del_count++; }
        if (t.indel > 0) {
            const ReadView rv = get_read(t.read_idx);
            std::string k = "I"; k += rb;
            // samtools >= 1.11 prints the pads of the run between the bases: '*', on the reverse strand '#' (--reverse-del); the key keeps them
            const c3r_padins_t *pe = pads.n ? pads.find(t.read_idx, t.qpos) : nullptr;
            const uint32_t total = pe ? pe->total : (uint32_t)t.indel;
            for (uint32_t ch = 0, j = 0; ch < total; ++ch) {
                if (pe && ((pe->pad_mask >> ch) & 1ull)) { k += t.rev ? '#' : '*'; continue; }
                const uint32_t q = t.qpos + j++;
                char c = 'N';
                if (q < rv.l_seq) { const uint8_t by = rv.seq[rv.seq_off + (q >> 1)]; c = NT16_STR[(q & 1) ? (by & 15) : (by >> 4)]; }
                k += c;
            }
            alt_add(alt, k, 1); ins_count++;
            if (t.del_after) {             // mpileup_compat = 1: `+<ins>-<del>`, the deletion is the read's next token on this column
                int64_t a = pos1 - ref_start1 + 1, e = a + (int64_t)t.del_after;
                a = std::max<int64_t>(0, std::min<int64_t>(a, (int64_t)ref.size()));
                e = std::max<int64_t>(a, std::min<int64_t>(e, (int64_t)ref.size()));
                alt_add(alt, "D" + ref.substr((size_t)a, (size_t)(e - a)), 1); del_count++;
            }
        } else if (t.indel < 0) {
            int64_t a = pos1 - ref_start1 + 1, e = a + (-t.indel);
            a = std::max<int64_t>(0, std::min<int64_t>(a, (int64_t)ref.size()));
            e = std::max<int64_t>(a, std::min<int64_t>(e, (int64_t)ref.size()));
            alt_add(alt, "D" + ref.substr((size_t)a, (size_t)(e - a)), 1); del_count++;
        }
    }
    if (site_depth >= 0) depth = site_depth;
    const int ref_count = std::max(0, depth - del_count - ins_count - alt_count);
    if (ref_count > 0) alt_add(alt, std::string("R") + rb, ref_count);
    depth_out = depth;
}
template <typename GetRead>
inline void alt_from_tokens(const c3r_token_t *tk, int n, GetRead get_read, const RefView &ref, int64_t ref_start1, int64_t pos1,
                            AltDict &alt, int &depth_out, int site_depth) {
    alt_from_stream(n, [tk](int i) { return TokView{tk[i].base, tk[i].indel, tk[i].read_idx, tk[i].qpos, tk[i].del_after, tk[i].rev != 0}; }, get_read, ref, ref_start1, pos1, alt, depth_out,
                    PadView{nullptr, 0}, site_depth);
}

// ---------------------------------------------------------------------------------------------- call_site
namespace dec {
static const char *GT21_LABELS[21] = {"AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT", "DelDel", "ADel", "CDel", "GDel",
                                      "TDel", "InsIns", "AIns", "CIns", "GIns", "TIns", "InsDel"};
static const int HOMO_SNP[4] = {0, 4, 7, 9};
static const int HETERO_SNP[6] = {1, 2, 3, 5, 6, 8};
enum { G_DELDEL = 10, G_ADEL = 11, G_INSINS = 15, G_AINS = 16, G_INSDEL = 20 };
static const char ACGT[] = "ACGT";

inline char iupac2acgt(char c) {
    static const char from[] = "ACGTURYSWKMBDHVN", to[] = "ACGTTACCAGACAAAA";
    const char *p = strchr(from, c);
    return (p && c) ? to[p - from] : 'A';
}
inline int gt21_index(char a, char b) {
    char lab[3] = {a, b, 0};
    for (int i = 0; i < 21; ++i) if (!strcmp(GT21_LABELS[i], lab)) return i;
    return 0;
}
inline AltDict indel_candidates(const AltDict &alt, char tag) {
    AltDict out;
    for (auto &kv : alt) {
        if (kv.first.empty() || kv.first[0] != tag) continue;
        const std::string key = kv.first.substr(1);
        if (key.size() >= 1 && key.size() <= 50) alt_set(out, key, kv.second);
    }
    return out;
}
inline std::string first_max_key(const AltDict &d) {
    std::string bk; bool have = false; int best = 0;
    for (auto &kv : d) if (!have || kv.second > best) { best = kv.second; bk = kv.first; have = true; }
    return bk;
}
inline std::vector<std::string> ranked_desc_reversed_ties(const AltDict &d) {   // sorted(key=count)[::-1]
    std::vector<std::pair<std::string, int>> v(d.begin(), d.end());
    std::stable_sort(v.begin(), v.end(), [](const std::pair<std::string, int> &a, const std::pair<std::string, int> &b) { return a.second < b.second; });
    std::vector<std::string> out;
    for (size_t i = v.size(); i-- > 0;) out.push_back(v[i].first);
    return out;
}
inline std::vector<std::string> two_insertions(const AltDict &alt) {
    std::vector<std::string> r = ranked_desc_reversed_ties(indel_candidates(alt, 'I'));
    if (r.size() > 2) r.resize(2);
    return r;
}
inline std::vector<std::string> two_deletions(const AltDict &alt) {
    std::vector<std::string> r = ranked_desc_reversed_ties(indel_candidates(alt, 'D'));
    if (r.size() <= 1) return {};
    if (r[0].size() > r[1].size()) return {r[0], r[1]};
    return {r[1], r[0]};
}
// find_alt_base: (ranked bases, chosen) ; chosen empty = None
inline void find_alt_base(const AltDict &alt, const std::string &proposed, bool have_proposed, std::vector<char> &ranked, std::string &chosen,
                          bool &chosen_none) {
    std::vector<std::pair<char, int>> v;
    for (auto &kv : alt) if (kv.first.size() >= 2 && kv.first[0] == 'X') v.emplace_back(kv.first[1], kv.second);
    std::stable_sort(v.begin(), v.end(), [](const std::pair<char, int> &a, const std::pair<char, int> &b) { return a.second > b.second; });
    ranked.clear();
    if (v.empty()) { chosen_none = true; chosen.clear(); return; }
    chosen = proposed; chosen_none = !have_proposed;
    bool found = false; int mine = 0;
    if (have_proposed && proposed.size() == 1)
        for (auto &x : v) if (x.first == proposed[0]) { found = true; mine = x.second; break; }
    if (!found || v[0].second - mine >= 9) { chosen = std::string(1, v[0].first); chosen_none = false; }
    for (auto &x : v) ranked.push_back(x.first);
}

struct Flags { bool ref, homo_snp, het_snp, homo_ins, het_base_ins, het_insins, homo_del, het_base_del, het_deldel, insdel; };

inline bool contains(const std::vector<float> &v, float x) { for (float y : v) if (y == x) return true; return false; }
inline int index_of(const std::vector<float> &v, float x) { for (size_t i = 0; i < v.size(); ++i) if (v[i] == x) return (int)i; return 0; }
inline int argmax(const std::vector<float> &v) { int b = 0; for (size_t i = 1; i < v.size(); ++i) if (v[i] > v[b]) b = (int)i; return b; }
inline float vmax(const std::vector<float> &v) { float m = v[0]; for (float y : v) if (y > m) m = y; return m; }

inline void call_site(const float *g, const float *z, const char *ref33, const AltDict &alt, Flags &f, std::string &ref_allele,
                      std::string &alt_allele, bool &alleles_none, float &prob) {
    const char center = strlen(ref33) > 1 ? ref33[16] : ref33[0];
    const char ra = iupac2acgt(center);
    const float z0 = z[0], z1 = z[1], z2 = z[2];
    const int rr = gt21_index(ra, ra);
    const float p_ref = z0 * g[rr];
    f = Flags{false, false, false, false, false, false, false, false, false, false};
    alleles_none = false;
    if (z0 >= 0.5f && g[rr] >= 0.5f) { f.ref = true; ref_allele = alt_allele = std::string(1, ra); prob = p_ref; return; }
    std::vector<float> homo_snp(4), het_snp(6), homo_ins(1), het_insins(1), het_base_ins(4), homo_del(1), het_deldel(1), het_base_del(4), insdel(1);
    for (int i = 0; i < 4; ++i) homo_snp[i] = z1 * g[HOMO_SNP[i]];
    for (int i = 0; i < 6; ++i) het_snp[i] = z2 * g[HETERO_SNP[i]];
    homo_ins[0] = z1 * g[G_INSINS]; het_insins[0] = z2 * g[G_INSINS];
    for (int i = 0; i < 4; ++i) het_base_ins[i] = g[G_AINS + i] * z2;
    homo_del[0] = z1 * g[G_DELDEL]; het_deldel[0] = z2 * g[G_DELDEL];
    for (int i = 0; i < 4; ++i) het_base_del[i] = g[G_ADEL + i] * z2;
    insdel[0] = z2 * g[G_INSDEL];
    const std::string C(1, center);
    bool have_ref = false, have_alt = false;
    float top = p_ref;
    std::vector<char> ranked; std::string chosen; bool none;
    // the reference loops `while reference_base is None or alternate_base is None`: a late `continue` after both were
    // assigned leaves the loop with that iteration's alleles and flags
    while (!have_ref || !have_alt) {
        top = std::max({p_ref, vmax(homo_snp), vmax(het_snp), vmax(homo_ins), vmax(homo_del), vmax(het_base_ins), vmax(het_insins),
                        vmax(het_base_del), vmax(het_deldel), vmax(insdel)});
        if (top == p_ref) { f = Flags{true, false, false, false, false, false, false, false, false, false}; ref_allele = alt_allele = std::string(1, ra); prob = top; return; }
        f.ref = false;
        f.homo_snp = contains(homo_snp, top); f.het_snp = contains(het_snp, top); f.homo_ins = contains(homo_ins, top);
        f.het_base_ins = contains(het_base_ins, top); f.het_insins = contains(het_insins, top); f.homo_del = contains(homo_del, top);
        f.het_base_del = contains(het_base_del, top); f.het_deldel = contains(het_deldel, top); f.insdel = contains(insdel, top);
        if (f.homo_snp) {
            ref_allele = C; have_ref = true;
            const int idx = index_of(homo_snp, top);
            const char *lab = GT21_LABELS[HOMO_SNP[argmax(homo_snp)]];
            std::string cand(1, lab[0] != center ? lab[0] : lab[1]);
            find_alt_base(alt, cand, true, ranked, chosen, none);
            alt_allele = chosen; have_alt = !none;
            if (none || alt_allele == ref_allele) { homo_snp[idx] = 0; continue; }
        } else if (f.het_snp) {
            const char *lab = GT21_LABELS[HETERO_SNP[argmax(het_snp)]];
            const int idx = index_of(het_snp, top);
            ref_allele = C; have_ref = true;
            if (lab[0] != center && lab[1] != center) {
                find_alt_base(alt, "", false, ranked, chosen, none);
                if (ranked.size() < 2) { het_snp[idx] = 0; continue; }
                alt_allele = std::string(1, ranked[0]) + "," + std::string(1, ranked[1]); have_alt = true;
            } else {
                std::string cand(1, lab[0] != center ? lab[0] : lab[1]);
                find_alt_base(alt, cand, true, ranked, chosen, none);
                alt_allele = chosen; have_alt = !none;
                if (none || alt_allele == ref_allele) { het_snp[idx] = 0; continue; }
            }
        } else if (f.homo_ins) {
            const std::string ins = first_max_key(indel_candidates(alt, 'I'));
            if (ins.empty()) { homo_ins[0] = 0; continue; }
            ref_allele = C; alt_allele = ins; have_ref = have_alt = true;
        } else if (f.het_base_ins) {
            const int idx = index_of(het_base_ins, top);
            const std::string ins = first_max_key(indel_candidates(alt, 'I'));
            if (ins.empty()) { het_base_ins[idx] = 0; continue; }
            ref_allele = C; alt_allele = ins; have_ref = have_alt = true;
            if (ACGT[idx] != center) {
                find_alt_base(alt, "", false, ranked, chosen, none);
                if (ranked.empty()) { het_base_ins[idx] = 0; continue; }
                alt_allele = std::string(1, ranked[0]) + "," + alt_allele;
            }
        } else if (f.het_insins) {
            const std::vector<std::string> two = two_insertions(alt);
            if (two.size() < 2) { het_insins[0] = 0; continue; }
            ref_allele = C; alt_allele = two[0]; have_ref = have_alt = true;
            if (two[1] != alt_allele) alt_allele = two[1] + "," + alt_allele;
            else { het_insins[0] = 0; continue; }
        } else if (f.homo_del) {
            const std::string d = first_max_key(indel_candidates(alt, 'D'));
            if (d.empty()) { homo_del[0] = 0; continue; }
            ref_allele = C + d; alt_allele = ref_allele.substr(0, 1); have_ref = have_alt = true;
        } else if (f.het_base_del) {
            const int idx = index_of(het_base_del, top);
            const std::string d = first_max_key(indel_candidates(alt, 'D'));
            if (d.empty()) { het_base_del[idx] = 0; continue; }
            ref_allele = C + d; alt_allele = ref_allele.substr(0, 1); have_ref = have_alt = true;
            if (ACGT[idx] != ref_allele[0]) alt_allele = alt_allele + "," + std::string(1, ACGT[idx]) + ref_allele.substr(1);
        } else if (f.het_deldel) {
            const std::vector<std::string> two = two_deletions(alt);
            if (two.size() < 2) { het_deldel[0] = 0; continue; }
            ref_allele = C + two[0]; alt_allele = ref_allele.substr(0, 1); have_ref = have_alt = true;
            const std::string a1 = alt_allele, a2 = ref_allele.substr(0, 1) + (two[1].size() + 1 <= ref_allele.size() ? ref_allele.substr(two[1].size() + 1) : std::string());
            if (a1 != a2 && ref_allele != a1 && ref_allele != a2) alt_allele = a1 + "," + a2;
            else { het_deldel[0] = 0; continue; }
        } else if (f.insdel) {
            const std::string ins = first_max_key(indel_candidates(alt, 'I')), d = first_max_key(indel_candidates(alt, 'D'));
            if (ins.empty() || d.empty()) { insdel[0] = 0; continue; }
            ref_allele = C + d; alt_allele = ref_allele.substr(0, 1) + "," + ins + ref_allele.substr(1); have_ref = have_alt = true;
        }
    }
    prob = top;
}

inline std::string iupac_to_n(const std::string &s) {
    if (s == ".") return s;
    std::string o = s;
    for (auto &c : o) { const char u = (char)toupper((unsigned char)c); if (!strchr("ACGTN,.", u) || !u) c = 'N'; }
    return o;
}
inline std::vector<std::string> split_comma(const std::string &s) {
    std::vector<std::string> o; size_t p = 0;
    while (true) { const size_t q = s.find(',', p); o.push_back(s.substr(p, q == std::string::npos ? std::string::npos : q - p)); if (q == std::string::npos) break; p = q + 1; }
    return o;
}
}  // namespace dec

// One candidate -> VCF row appended to `out` (with trailing '\n'); returns false when the reference prints nothing.
inline bool vcf_row(const char *ctg, int64_t pos, const char *ref33, int depth, const AltDict &alt, const float *probs24, int qual_for_pass,
                    bool show_ref, std::string &out) {
    using namespace dec;
    Flags f; std::string ra, aa; bool none; float p;
    call_site(probs24, probs24 + 21, ref33, alt, f, ra, aa, none, p);
    const bool is_ref = f.ref;
    if ((!show_ref && is_ref) || (!is_ref && ra == aa)) return false;
    const bool multi = aa.find(',') != std::string::npos;
    const char *gt = "0/0";
    if (is_ref) gt = "0/0";
    else if (f.homo_snp || f.homo_ins || f.homo_del) gt = "1/1";
    else if (f.het_snp || f.het_base_ins || f.het_insins || f.het_base_del || f.het_deldel) gt = "0/1";
    if (multi) gt = "1/2";
    AltDict snp, ins, dele; int ref_count = 0;
    for (auto &kv : alt) {
        const std::string &k = kv.first;
        if (k.empty()) continue;
        if (k[0] == 'X' && k.size() >= 2) alt_set(snp, k.substr(1, 1), kv.second);
        else if (k[0] == 'I') alt_set(ins, k.substr(1), kv.second);
        else if (k[0] == 'D') alt_set(dele, k.substr(1), kv.second);
        else if (k[0] == 'R') ref_count = kv.second;
    }
    ref_count = std::max(0, ref_count);
    int support = 0; std::vector<int> per_alt;
    if (is_ref) { support = ref_count; aa = "."; }
    else if (f.homo_snp || f.het_snp) {
        for (char b : aa) { if (b == ',') continue; support += alt_get(snp, std::string(1, b)); per_alt.push_back(support); }
    } else if (f.homo_ins || f.het_insins) {
        for (auto &s : split_comma(aa)) { const int n = alt_get(ins, s); support += n; per_alt.push_back(n); }
    } else if (f.het_base_ins) {
        const std::vector<std::string> parts = split_comma(aa);
        const std::string s = multi ? parts[1] : aa;
        const int n_snp = multi ? alt_get(snp, parts[0].substr(0, 1)) : 0, n_ins = alt_get(ins, s);
        support = n_ins + n_snp;
        if (multi) per_alt.push_back(n_snp);
        per_alt.push_back(n_ins);
    } else if (f.homo_del || f.het_deldel) {
        if (!dele.empty()) {
            if (f.homo_del) { support = ra.size() > 1 ? alt_get(dele, ra.substr(1)) : 0; per_alt.push_back(support); }
            else if (f.het_deldel && dele.size() > 1) {
                for (auto &a : split_comma(aa)) {
                    const long L = (long)ra.size() - (long)a.size();
                    int n = 0;
                    for (auto &kv : dele) if ((long)kv.first.size() == L) { n = kv.second; break; }
                    per_alt.push_back(n); support += n;
                }
            }
        }
    } else if (f.het_base_del) {
        const std::vector<std::string> parts = split_comma(aa);
        const bool has_snp = multi && parts.size() > 1;
        const int n_snp = has_snp ? alt_get(snp, parts[1].substr(0, 1)) : 0;
        const int n_del = ra.size() > 1 ? alt_get(dele, ra.substr(1)) : 0;
        support = n_del + n_snp;
        if (has_snp) per_alt.push_back(n_snp);
        per_alt.push_back(n_del);
    } else if (f.insdel) {
        for (auto &a : split_comma(aa)) {
            const long L = (long)ra.size() - (long)a.size();
            int n = 0;
            if (L < 0) {
                const std::string s = ra.size() > 1 ? a.substr(0, a.size() - (ra.size() - 1)) : a;
                n = alt_get(ins, s);
            } else {
                for (auto &kv : dele) if ((long)kv.first.size() == L) { n = kv.second; break; }
            }
            per_alt.push_back(n); support += n;
        }
    }
    double af = depth != 0 ? (support + 0.0) / depth : 0.0;
    if (af > 1) af = 1;
    const double pd = (double)p;
    double q = -10.0 * (std::log(M_E) / std::log(10.0)) * std::log(((1.0 - pd) + 1e-10) / (pd + 1e-10)) + 10.0;
    if (q < 0) q = 0;
    char qs[64]; snprintf(qs, sizeof qs, "%.2f", q);
    const double q2 = strtod(qs, nullptr);                 // float(round(tmp, 2))
    const char *filt = is_ref ? "RefCall" : ((qual_for_pass < 0 || q2 >= qual_for_pass) ? "PASS" : "LowQual");
    const std::string ra2 = iupac_to_n(ra), aa2 = iupac_to_n(aa);
    std::string ad = std::to_string(ref_count);
    for (int x : per_alt) { ad += ','; ad += std::to_string(x); }
    std::string afs;
    char buf[64];
    if (per_alt.size() <= 1) { snprintf(buf, sizeof buf, "%.4f", af); afs = buf; }
    else for (size_t i = 0; i < per_alt.size(); ++i) { snprintf(buf, sizeof buf, "%.4f", std::min(1.0, 1.0 * per_alt[i] / depth)); if (i) afs += ','; afs += buf; }
    char head[256];
    snprintf(head, sizeof head, "%s\t%lld\t.\t", ctg, (long long)pos);
    out += head; out += ra2; out += '\t'; out += aa2; out += '\t'; out += qs; out += '\t'; out += filt;
    snprintf(buf, sizeof buf, "\t.\tGT:GQ:DP:AD:AF\t%s:%d:%d:", gt, (int)q2, depth);
    out += buf; out += ad; out += ':'; out += afs; out += '\n';
    return true;
}

}  // namespace c3r
