// synth.cpp — deterministic chr20-scale synthetic long-read RNA alignments.
// Host-only helper for bench.py and the large-size parity tests; built with g++ into
// libc3r_synth.so.  Not part of the product path.
//
// Model: uniform random ACGT reference; transcripts = exons 150+-80 bp separated by log-uniform
// introns; reads sampled from transcripts with ONT dRNA004-like (or HiFi-like) errors, so CIGARs carry
// M/I/D/N/S/H ops; het/hom SNPs, short indels and A->G editing sites; MAPQ / flag mix that exercises
// the --excl-flags 2316 / --min-MQ 5 filters; optional HP tags.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/c3r_types.h"

namespace {

struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = splitmix(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
    int64_t range(int64_t lo, int64_t hi) { return lo + (int64_t)(uni() * (double)(hi - lo + 1)); }  // inclusive
    double gauss() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

struct Var { uint8_t kind; float af; uint8_t arg; };  // kind 1 snp, 2 ins, 3 del, 4 edit; arg = base code or length

const char ACGT[] = "ACGT";
inline uint8_t code_of(char c) { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15; }

struct Out {
    std::string ref;
    std::vector<c3r_read_t> reads;
    std::vector<uint32_t> cigar;
    std::vector<uint8_t> seq;
    int64_t n_exonic = 0, n_genes = 0;
};

}  // namespace

extern "C" {

typedef struct c3r_synth_params {
    int64_t contig_len;
    uint64_t seed;
    double depth;            // mean exonic depth
    double expressed_frac;   // fraction of the contig that is exonic (approx.)
    int32_t platform;        // 0 = ONT dRNA004-like, 1 = PacBio MAS-Seq-like
    int32_t phased;          // emit HP tags
    double intron_lo, intron_hi;
    int64_t region_start, region_end;   // place genes only inside [region_start, region_end) (0 = whole contig)
    double expr_sigma;       // sigma of the log-normal per-gene expression level; <= 0: the default 0.4 around `depth`.  Larger values keep the MEAN
                             // at `depth` (level = depth * exp(sigma * g - sigma^2 / 2)): 2.3 spreads the genes over four to five decades — a few loci at
                             // 1,000-10,000x, a long tail of one-to-three-read islands (what real RNA-seq looks like)
    double max_level;        // cap on a gene's level (<= 0: none)
} c3r_synth_params;

typedef struct c3r_synth_result {
    const char *ref; int64_t ref_len;
    const c3r_read_t *reads; int64_t n_reads;
    const uint32_t *cigar; int64_t n_cigar;
    const uint8_t *seq; int64_t n_seq;
    int64_t n_exonic, n_genes;
    void *owner;
} c3r_synth_result;

int c3r_synth_generate(const c3r_synth_params *P, c3r_synth_result *res) {
    Out *o = new Out();
    Rng rng(P->seed);
    const int64_t L = P->contig_len;
    o->ref.resize((size_t)L);
    for (int64_t i = 0; i < L; i += 32) {
        uint64_t r = rng.next();
        for (int k = 0; k < 32 && i + k < L; ++k) { o->ref[(size_t)(i + k)] = ACGT[r & 3]; r >>= 2; }
    }
    const double em = P->platform == 0 ? 0.03 : 0.003, ei = P->platform == 0 ? 0.015 : 0.001, ed = P->platform == 0 ? 0.025 : 0.001;
    const double mean_len = P->platform == 0 ? 900.0 : 2500.0;
    const int64_t g_lo = P->region_start > 0 ? P->region_start : 200;
    const int64_t g_hi = P->region_end > 0 ? P->region_end : L - 500;
    const double mean_exonic = 6.0 * 150.0;
    const double pitch = mean_exonic / std::max(1e-6, P->expressed_frac);

    struct Rec { int32_t pos; uint16_t flag; uint8_t mapq, hp; std::vector<uint32_t> cig; std::string seq; };
    std::vector<Rec> recs;
    std::vector<int32_t> tx;
    std::unordered_map<int32_t, Var> var;
    int64_t gstart = g_lo + (int64_t)(rng.uni() * pitch);
    while (gstart < g_hi) {
        // ---- one transcript
        tx.clear(); var.clear();
        const int n_ex = (int)rng.range(2, 10);
        int64_t pos = gstart;
        for (int e = 0; e < n_ex; ++e) {
            int64_t len = (int64_t)(150 + 80 * rng.gauss());
            if (len < 30) len = 30;
            if (pos + len >= L - 200) break;
            for (int64_t q = pos; q < pos + len; ++q) tx.push_back((int32_t)q);
            pos += len;
            if (e + 1 < n_ex) pos += (int64_t)std::exp(std::log(P->intron_lo) + rng.uni() * (std::log(P->intron_hi) - std::log(P->intron_lo)));
        }
        gstart += (int64_t)(-std::log(1.0 - rng.uni()) * pitch) + 1;
        const int tlen = (int)tx.size();
        if (tlen < 200) continue;
        o->n_genes++; o->n_exonic += tlen;
        for (int i = 0; i < tlen; ++i) {
            const double r = rng.uni();
            const char rb = o->ref[(size_t)tx[i]];
            if (r < 1 / 1000.0 + 1 / 3000.0) {
                char a; do { a = ACGT[rng.next() & 3]; } while (a == rb);
                var[i] = Var{1, r < 1 / 1000.0 ? 0.5f : 1.0f, (uint8_t)a};
            } else if (r < 1 / 1000.0 + 1 / 3000.0 + 1 / 8000.0) {
                var[i] = Var{(uint8_t)((rng.next() & 1) ? 2 : 3), 0.5f, (uint8_t)rng.range(1, 3)};
            } else if (r < 1 / 1000.0 + 1 / 3000.0 + 1 / 8000.0 + 1 / 5000.0 && rb == 'A') {
                var[i] = Var{4, (float)(0.1 + 0.2 * rng.uni()), (uint8_t)'G'};
            }
        }
        double level = P->expr_sigma > 0 ? P->depth * std::exp(P->expr_sigma * rng.gauss() - 0.5 * P->expr_sigma * P->expr_sigma) : P->depth * std::exp(0.4 * rng.gauss());
        if (P->max_level > 0 && level > P->max_level) level = P->max_level;
        const int n_reads = std::max(1, (int)(level * tlen / std::min(mean_len, (double)tlen)));
        for (int rix = 0; rix < n_reads; ++rix) {
            int Lr = P->platform == 0 ? (int)std::exp(std::log(900.0) + 0.6 * rng.gauss()) : (int)(2500 + 800 * rng.gauss());
            Lr = std::max(200, std::min(8000, Lr));
            Lr = std::min(Lr, tlen);
            const int s = (int)rng.range(0, tlen - Lr);
            const int hap = (int)(rng.next() & 1);
            const bool rev = rng.uni() < 0.5;
            Rec rec; rec.pos = tx[s];
            auto add = [&](uint32_t op, uint32_t n) {
                if (!n) return;
                if (!rec.cig.empty() && (rec.cig.back() & 15u) == op) rec.cig.back() += n << 4;
                else rec.cig.push_back((n << 4) | op);
            };
            int i = s; bool first = true;
            while (i < s + Lr) {
                if (!first && tx[i] != tx[i - 1] + 1) add(C3R_CIG_N, (uint32_t)(tx[i] - tx[i - 1] - 1));
                first = false;
                char base = o->ref[(size_t)tx[i]];
                const Var *v = nullptr;
                auto it = var.find(i);
                if (it != var.end()) v = &it->second;
                if (v) {
                    if (v->kind == 1 && (v->af >= 1.0f || hap == 1)) base = (char)v->arg;
                    else if (v->kind == 4 && rng.uni() < v->af) base = 'G';
                }
                const double r = rng.uni();
                if (r < ed && i > s && i + 1 < s + Lr && tx[i] == tx[i - 1] + 1) {
                    int n = 1;
                    while (rng.uni() < 0.4 && n < 10 && i + n + 1 < s + Lr && tx[i + n] == tx[i + n - 1] + 1) ++n;
                    add(C3R_CIG_D, (uint32_t)n);
                    i += n;
                    continue;
                }
                if (r < ed + em) { char a; do { a = ACGT[rng.next() & 3]; } while (a == base); base = a; }
                else if (r < ed + em + 0.001) base = 'N';
                rec.seq.push_back(base);
                add(C3R_CIG_M, 1);
                if (v && v->kind == 3 && hap == 1 && i + v->arg + 1 < s + Lr) {
                    bool contig = true;
                    for (int k = 0; k < v->arg; ++k) if (tx[i + k + 1] != tx[i + k] + 1) contig = false;
                    if (contig) { add(C3R_CIG_D, v->arg); i += v->arg + 1; continue; }
                }
                const bool vins = v && v->kind == 2 && hap == 1;
                if ((vins || rng.uni() < ei) && i + 1 < s + Lr) {
                    int n = vins ? v->arg : 1;
                    if (!vins) while (rng.uni() < 0.4 && n < 10) ++n;
                    for (int k = 0; k < n; ++k) rec.seq.push_back(vins ? ACGT[(tx[i] + k) & 3] : ACGT[rng.next() & 3]);
                    add(C3R_CIG_I, (uint32_t)n);
                }
                ++i;
            }
            if (rec.cig.empty()) continue;
            if (rng.uni() < 0.10) {
                const int n = (int)rng.range(5, 50);
                std::string cl; for (int k = 0; k < n; ++k) cl.push_back(ACGT[rng.next() & 3]);
                rec.seq = cl + rec.seq;
                rec.cig.insert(rec.cig.begin(), ((uint32_t)n << 4) | C3R_CIG_S);
            }
            if (rng.uni() < 0.10) {
                const int n = (int)rng.range(5, 50);
                for (int k = 0; k < n; ++k) rec.seq.push_back(ACGT[rng.next() & 3]);
                rec.cig.push_back(((uint32_t)n << 4) | C3R_CIG_S);
            }
            double r = rng.uni();
            rec.mapq = (uint8_t)(r < 0.92 ? 60 : (r < 0.97 ? rng.range(0, 4) : rng.range(5, 59)));
            rec.flag = rev ? 16 : 0;
            r = rng.uni();
            if (r < 0.02) rec.flag |= 256; else if (r < 0.04) rec.flag |= 2048; else if (r < 0.045) rec.flag |= 8;
            else if (r < 0.055) rec.flag |= 1024;
            rec.hp = 0;
            if (P->phased) rec.hp = (uint8_t)(rng.uni() < 0.9 ? hap + 1 : 0);
            recs.push_back(std::move(rec));
        }
    }
    std::vector<uint32_t> order(recs.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return recs[a].pos < recs[b].pos; });
    o->reads.reserve(recs.size());
    for (uint32_t idx : order) {
        const Rec &r = recs[idx];
        c3r_read_t h;
        memset(&h, 0, sizeof h);
        h.pos = r.pos; h.cigar_off = (uint32_t)o->cigar.size(); h.n_cigar = (uint32_t)r.cig.size(); h.l_seq = (uint32_t)r.seq.size();
        h.seq_off = o->seq.size(); h.flag = r.flag; h.mapq = r.mapq; h.hp = r.hp;
        o->cigar.insert(o->cigar.end(), r.cig.begin(), r.cig.end());
        for (size_t q = 0; q < r.seq.size(); q += 2) {
            const uint8_t hi = code_of(r.seq[q]), lo = q + 1 < r.seq.size() ? code_of(r.seq[q + 1]) : 0;
            o->seq.push_back((uint8_t)((hi << 4) | lo));
        }
        o->reads.push_back(h);
    }
    res->ref = o->ref.data(); res->ref_len = (int64_t)o->ref.size();
    res->reads = o->reads.data(); res->n_reads = (int64_t)o->reads.size();
    res->cigar = o->cigar.data(); res->n_cigar = (int64_t)o->cigar.size();
    res->seq = o->seq.data(); res->n_seq = (int64_t)o->seq.size();
    res->n_exonic = o->n_exonic; res->n_genes = o->n_genes;
    res->owner = o;
    return 0;
}

void c3r_synth_free(c3r_synth_result *res) {
    if (res && res->owner) { delete (Out *)res->owner; res->owner = nullptr; }
}

}  // extern "C"
