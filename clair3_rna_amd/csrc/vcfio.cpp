// vcfio.cpp — output side of libc3r_io.so: the merge step and the compressed, indexed VCF (include/c3r_io.h).
//
//   c3r_vcf_merge     rows of one contig, as its chunks produced them -> the records `sort_vcf` writes for that contig
//                     (src/sort_vcf.py:190-262: RefCall rows dropped unless --show_ref, QUAL <= --qual relabelled LowQual,
//                     REDIportal tagging, duplicate positions last-one-wins, sorted by position)
//   c3r_vcf_compress  `bgzip -f` + `tabix -f -p vcf` (src/sort_vcf.py:70-75): BGZF blocks deflated on threads + TBI v1 index
//
// Host-only C++ (zlib).  clair3_rna_amd/sort_vcf.py holds the same two steps in Python; they are the checkers: the merge is
// pinned on golden G6 (outputs of the reference's sort_vcf), and the compressor must reproduce the Python writer's bytes
// (tests/test_sort_vcf.py).
#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/c3r_io.h"
#include <sched.h>

// CPUs this process may run on (its affinity mask: one process per GPU is pinned to its share of the node, shard.host_budget), not the
// machine's: what default thread counts are taken from
static inline unsigned usable_cpus() {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int n = CPU_COUNT(&set); if (n > 0) return (unsigned)n; }
    return std::max(1u, std::thread::hardware_concurrency());
}

#define C3R_OK 0
#define C3R_EINVAL (-1)
#define C3R_EOVERFLOW (-6)

namespace {

struct Field { const char *p; size_t n; };

// Python str.split(None, maxsplit) on one line without its newline: runs of blanks separate, leading blanks skipped.
inline int split_ws(const char *s, size_t n, Field *f, int max_fields) {
    size_t i = 0; int k = 0;
    auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\f' || c == '\v' || c == '\n'; };
    while (k < max_fields) {
        while (i < n && blank(s[i])) ++i;
        if (i >= n) break;
        const size_t b = i;
        if (k == max_fields - 1) { f[k++] = Field{s + b, n - b}; break; }     // the rest (Python keeps it unsplit)
        while (i < n && !blank(s[i])) ++i;
        f[k++] = Field{s + b, i - b};
    }
    return k;
}

inline bool contains(const char *s, size_t n, const char *pat) {
    const size_t m = strlen(pat);
    if (m > n) return false;
    for (size_t i = 0; i + m <= n; ++i) if (!memcmp(s + i, pat, m)) return true;
    return false;
}

// row.split("\t")[k] spans; returns the number of tab-separated fields found (up to max)
inline int split_tabs(const char *s, size_t n, Field *f, int max_fields) {
    int k = 0; size_t b = 0;
    for (size_t i = 0; i <= n && k < max_fields; ++i) {
        if (i == n || s[i] == '\t') { f[k++] = Field{s + b, i - b}; b = i + 1; }
    }
    return k;
}

// One kept record: where its text lies (the caller's rows, or `arena` for a row that was relabelled / lacked its newline).
struct Kept { int32_t pos; uint32_t len; uint64_t off; bool in_arena; };

void replace_all(std::string &s, const char *from, const char *to) {
    const size_t lf = strlen(from), lt = strlen(to);
    for (size_t p = s.find(from); p != std::string::npos; p = s.find(from, p + lt)) s.replace(p, lf, to);
}

}  // namespace

extern "C" {

// No allocation per record: a contig of a whole-sample run is ~10^5-10^6 records (~100 MB of text), several contigs are merged at a
// time on worker threads, and a std::string per row made them queue up in the allocator (0.6-1.1 s for a large contig).  Records are
// indexed where they lie; only rows whose text changes (LowQual / RNAEditing relabel) are built, in one arena.
int c3r_vcf_merge(const char *rows, int64_t n_bytes, int qual, int show_ref, const int32_t *edit_pos, const char *const *edit_ref,
                  const char *const *edit_alt, int64_t n_edit, char *out, int64_t cap, int64_t *out_len, char *out_nt, int64_t cap_nt,
                  int64_t *out_nt_len, int64_t *counts) {
    if (n_bytes < 0 || (n_bytes && !rows) || !out_len || (n_edit && (!edit_pos || !edit_ref || !edit_alt))) return C3R_EINVAL;
    int64_t n_read = 0, n_kept = 0, n_tag = 0;
    std::vector<Kept> kept;
    kept.reserve((size_t)(n_bytes / 96) + 16);
    std::string arena, row, num;
    bool sorted = true;
    const char *p = rows, *end = rows + n_bytes;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;                 // the line without '\n'
        const size_t ln = (size_t)(le - p);
        const char *line = p;
        p = nl ? nl + 1 : end;
        if (ln == 0) continue;
        ++n_read;
        Field c[7];
        if (split_ws(line, ln, c, 7) < 6) return C3R_EINVAL;
        // (both numbers end at a blank or the line end; only a field that touches the end of the buffer needs a terminated copy)
        const int32_t pos = (int32_t)strtol(c[1].p, nullptr, 10);
        double q;
        if (c[5].p + c[5].n == end) { num.assign(c[5].p, c[5].n); q = strtod(num.c_str(), nullptr); }
        else q = strtod(c[5].p, nullptr);
        const bool is_ref = (c[4].n == 1 && c[4].p[0] == '.') || (c[3].n == c[4].n && !memcmp(c[3].p, c[4].p, c[3].n));
        if (is_ref && !show_ref) continue;
        const bool low = !is_ref && qual && q <= (double)qual;
        const int32_t *e = n_edit ? std::lower_bound(edit_pos, edit_pos + n_edit, pos) : nullptr;
        const bool edit = e && e != edit_pos + n_edit && *e == pos;
        if (!kept.empty() && pos < kept.back().pos) sorted = false;
        if (!low && !edit && nl) { kept.push_back(Kept{pos, (uint32_t)(ln + 1), (uint64_t)(line - rows), false}); ++n_kept; continue; }
        row.assign(line, ln);
        row += '\n';
        if (low) {                                      // _relabel: row.split("\t")[6] = "LowQual"
            Field f[8];
            if (split_tabs(row.data(), row.size(), f, 8) >= 7) row.replace((size_t)(f[6].p - row.data()), f[6].n, "LowQual");
        }
        if (edit && !contains(row.data(), row.size(), "Germline") && !contains(row.data(), row.size(), "RefCall")) {
            const int64_t k = e - edit_pos;
            Field f[9];
            if (split_tabs(row.data(), row.size(), f, 9) >= 7 && f[3].n == strlen(edit_ref[k]) && !memcmp(f[3].p, edit_ref[k], f[3].n) &&
                f[4].n == strlen(edit_alt[k]) && !memcmp(f[4].p, edit_alt[k], f[4].n)) {
                row.replace((size_t)(f[6].p - row.data()), f[6].n, "RNAEditing");
                ++n_tag;
            }
        }
        kept.push_back(Kept{pos, (uint32_t)row.size(), (uint64_t)arena.size(), true});
        arena += row;
        ++n_kept;                                       // (the reference counts overwritten duplicates as kept, too)
    }
    // by_pos[pos] = row: the last row of a position wins; output sorted by position (one whole-contig scan arrives sorted)
    if (!sorted) std::stable_sort(kept.begin(), kept.end(), [](const Kept &a, const Kept &b) { return a.pos < b.pos; });
    auto text = [&](const Kept &k) { return (k.in_arena ? arena.data() : rows) + k.off; };
    size_t total = 0, total_nt = 0, n_out = 0;
    for (size_t i = 0; i < kept.size(); ++i) {
        if (i + 1 < kept.size() && kept[i + 1].pos == kept[i].pos) continue;
        kept[n_out++] = kept[i];
        total += kept[i].len;
    }
    kept.resize(n_out);
    *out_len = (int64_t)total;
    if (counts) { counts[0] = n_read; counts[1] = n_kept; counts[2] = n_tag; }
    // the untagged twin: every "RNAEditing" of a row reads "PASS"
    std::vector<uint8_t> has_tag;
    if (out_nt_len) {
        has_tag.resize(n_out);
        for (size_t i = 0; i < n_out; ++i) {
            has_tag[i] = memmem(text(kept[i]), kept[i].len, "RNAEditing", 10) != nullptr;
            if (has_tag[i]) { row.assign(text(kept[i]), kept[i].len); replace_all(row, "RNAEditing", "PASS"); total_nt += row.size(); }
            else total_nt += kept[i].len;
        }
        *out_nt_len = (int64_t)total_nt;
    }
    if (!out || cap < (int64_t)total || (out_nt_len && (!out_nt || cap_nt < (int64_t)total_nt))) return C3R_EOVERFLOW;
    char *o = out;
    // (runs of untouched neighbouring input rows are one memcpy)
    for (size_t i = 0; i < n_out;) {
        size_t j = i + 1;
        uint64_t run = kept[i].len;
        while (!kept[i].in_arena && j < n_out && !kept[j].in_arena && kept[j].off == kept[i].off + run) run += kept[j++].len;
        memcpy(o, text(kept[i]), (size_t)run); o += run;
        i = j;
    }
    if (out_nt_len) {
        char *t = out_nt;
        for (size_t i = 0; i < n_out; ++i) {
            if (has_tag[i]) { row.assign(text(kept[i]), kept[i].len); replace_all(row, "RNAEditing", "PASS"); memcpy(t, row.data(), row.size()); t += row.size(); }
            else { memcpy(t, text(kept[i]), kept[i].len); t += kept[i].len; }
        }
    }
    return C3R_OK;
}

}  // extern "C"

namespace {
const uint8_t BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};
const size_t BLK = 0xff00;

bool deflate_block(const uint8_t *src, size_t n, std::vector<uint8_t> &dst) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    std::vector<uint8_t> body(deflateBound(&zs, (uLong)n) + 16);
    zs.next_in = const_cast<Bytef *>(src); zs.avail_in = (uInt)n;
    zs.next_out = body.data(); zs.avail_out = (uInt)body.size();
    const int rc = deflate(&zs, Z_FINISH);
    const size_t clen = body.size() - zs.avail_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END || clen + 26 > 0x10000) return false;
    static const uint8_t head[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
    dst.assign(head, head + 16);
    const uint16_t bsize = (uint16_t)(clen + 25);
    dst.push_back((uint8_t)(bsize & 0xff)); dst.push_back((uint8_t)(bsize >> 8));
    dst.insert(dst.end(), body.begin(), body.begin() + (long)clen);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n), isz = (uint32_t)n;
    for (int k = 0; k < 4; ++k) dst.push_back((uint8_t)(crc >> (8 * k)));
    for (int k = 0; k < 4; ++k) dst.push_back((uint8_t)(isz >> (8 * k)));
    return true;
}

// data -> BGZF (blocks of 0xff00 bytes, deflated on `threads` threads) in `out`; coffs[i] = file offset of block i
bool bgzf_compress(const uint8_t *data, size_t n, int threads, std::vector<uint8_t> &out, std::vector<uint64_t> &coffs) {
    const size_t nb = (n + BLK - 1) / BLK;
    std::vector<std::vector<uint8_t>> blocks(nb);
    std::atomic<size_t> next(0);
    std::atomic<bool> bad(false);
    auto work = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nb) break;
            if (!deflate_block(data + i * BLK, std::min(BLK, n - i * BLK), blocks[i])) bad = true;
        }
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), nb));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (bad) return false;
    coffs.resize(nb);
    size_t total = 0;
    for (size_t i = 0; i < nb; ++i) { coffs[i] = total; total += blocks[i].size(); }
    out.resize(total);
    for (size_t i = 0; i < nb; ++i) memcpy(out.data() + coffs[i], blocks[i].data(), blocks[i].size());
    return true;
}

inline int reg2bin_vcf(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

bool write_file(const std::string &path, const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = (na == 0 || fwrite(a, 1, na, f) == na) && (nb == 0 || fwrite(b, 1, nb, f) == nb);
    ok = (fclose(f) == 0) && ok;
    return ok;
}

template <typename T> void put(std::vector<uint8_t> &v, T x) { const uint8_t *p = (const uint8_t *)&x; v.insert(v.end(), p, p + sizeof(T)); }
}  // namespace

extern "C" {

}  // extern "C"

// The data lines of text[0, n) as tabix sees them: fn(offset of the line, its length + 1, CHROM field, 0-based begin, end).
// (Python: for line in data.split(b"\n"): n = len(line) + 1; line.split(b"\t", 5); end = beg + max(1, len(REF)).)
template <class F>
void for_each_record(const uint8_t *data, size_t n, F fn) {
    size_t u = 0;
    while (u <= n) {
        const uint8_t *nl = u < n ? (const uint8_t *)memchr(data + u, '\n', n - u) : nullptr;
        const size_t le = nl ? (size_t)(nl - data) : n;
        const size_t ln = le - u, step = ln + 1;
        if (ln > 0 && data[u] != '#') {
            Field c[6];
            const char *s = (const char *)data + u;
            int k = 0; size_t b = 0;
            for (size_t i = 0; i <= ln && k < 5; ++i) if (i == ln || s[i] == '\t') { c[k++] = Field{s + b, i - b}; b = i + 1; }
            if (k >= 4) {
                char num[24];                                          // POS: a terminated copy (strtoll must not run past the field)
                const size_t m = std::min(c[1].n, sizeof num - 1);
                memcpy(num, c[1].p, m); num[m] = 0;
                const int64_t beg = strtoll(num, nullptr, 10) - 1;
                fn(u, step, c[0], beg, beg + (int64_t)std::max<size_t>(1, c[3].n));
            }
        }
        u += step;
        if (!nl) break;
    }
}

// Streaming form of the compressor: text arrives in newline-terminated pieces (the header, then each contig's merged records), every
// full 0xff00-byte block is deflated on threads and written as soon as it is complete, index records are resolved once the blocks
// they point into have their file offsets.  The bytes are those of compressing the concatenated text in one go (BGZF blocks are cut
// at fixed offsets of the uncompressed stream), so c3r_vcf_compress is this object fed once.
struct c3r_vcfz {
    std::string gz_path;
    FILE *f = nullptr;
    int threads = 1;
    std::vector<uint8_t> pend;                 // bytes not yet in a block (< BLK after every write)
    uint64_t n_unc = 0;                        // uncompressed bytes already in blocks
    uint64_t n_seen = 0;                       // uncompressed bytes received
    uint64_t coff = 0;                         // compressed bytes written
    std::vector<uint64_t> coffs;               // file offset of every finished block
    std::vector<uint64_t> bstart;              // offset of its first byte in the uncompressed stream (blocks are BLK bytes unless a
                                               // piece was appended: the block before a piece, and a piece's last block, may be short)
    mutable size_t blk_cur = 0;                // block of the most recent lookup (lookups come in stream order)
    struct Rec { uint64_t u; uint32_t step; uint32_t ctg; int64_t beg, end; };
    std::vector<Rec> recs; size_t rec_head = 0; // index records waiting for their blocks
    struct Ctg { std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins; std::vector<uint64_t> lin; };
    std::vector<std::string> names; std::map<std::string, size_t> which; std::vector<Ctg> idx;
    bool bad = false;

    bool flush_blocks(bool all) {
        const size_t n = pend.size();
        const size_t nb = all ? (n + BLK - 1) / BLK : n / BLK;
        if (nb == 0) return true;
        const size_t take = std::min(n, nb * BLK);
        std::vector<uint8_t> gz; std::vector<uint64_t> co;
        if (!bgzf_compress(pend.data(), take, threads, gz, co)) return false;
        if (!gz.empty() && fwrite(gz.data(), 1, gz.size(), f) != gz.size()) return false;
        for (size_t i = 0; i < co.size(); ++i) { coffs.push_back(coff + co[i]); bstart.push_back(n_unc + i * BLK); }
        coff += gz.size();
        n_unc += take;
        pend.erase(pend.begin(), pend.begin() + (long)take);
        return true;
    }
    // a position at a block's end belongs to the next block; past the last byte: the EOF block
    bool voff(uint64_t u, bool closing, uint64_t &v) const {
        if (u < n_unc) {
            if (blk_cur >= bstart.size() || bstart[blk_cur] > u) blk_cur = 0;
            while (blk_cur + 1 < bstart.size() && bstart[blk_cur + 1] <= u) ++blk_cur;
            v = (coffs[blk_cur] << 16) | (u - bstart[blk_cur]);
            return true;
        }
        if (closing) { v = coff << 16; return true; }
        return false;
    }
    void resolve(bool closing) {
        for (; rec_head < recs.size(); ++rec_head) {
            const Rec &r = recs[rec_head];
            uint64_t v0, v1;
            if (!voff(r.u, closing, v0) || !voff(r.u + r.step, closing, v1)) break;
            Ctg &x = idx[r.ctg];
            auto &ch = x.bins[(uint32_t)reg2bin_vcf(r.beg, r.end)];
            if (!ch.empty() && ch.back().second == v0) ch.back().second = v1;
            else ch.emplace_back(v0, v1);
            const size_t w1 = (size_t)((r.end - 1) >> 14);
            if (x.lin.size() <= w1) x.lin.resize(w1 + 1, 0);
            for (size_t w = (size_t)(r.beg >> 14); w <= w1; ++w) if (x.lin[w] == 0) x.lin[w] = v0;
        }
        if (rec_head == recs.size()) { recs.clear(); rec_head = 0; }
    }
    // index records of the lines of text[0, n) (complete lines; the last one may lack its newline only in the final piece)
    void scan_lines(const uint8_t *data, size_t n) {
        std::string last; size_t last_id = 0;
        for_each_record(data, n, [&](size_t u, size_t step, const Field &ctg, int64_t beg, int64_t end) {
            if (last.size() != ctg.n || memcmp(last.data(), ctg.p, ctg.n) != 0 || names.empty()) {
                last.assign(ctg.p, ctg.n);
                auto it = which.find(last);
                if (it == which.end()) { it = which.emplace(last, names.size()).first; names.push_back(last); idx.emplace_back(); }
                last_id = it->second;
            }
            recs.push_back(Rec{n_seen + u, (uint32_t)step, (uint32_t)last_id, beg, end});
        });
    }
};

extern "C" {

int c3r_vcfz_open(const char *gz_path, int threads, c3r_vcfz **out) {
    if (!gz_path || !out) return C3R_EINVAL;
    *out = nullptr;
    c3r_vcfz *z = new c3r_vcfz();
    z->gz_path = gz_path;
    z->threads = threads > 0 ? threads : (int)std::min(32u, std::max(1u, usable_cpus()));
    z->f = fopen(gz_path, "wb");
    if (!z->f) { delete z; return C3R_EINVAL; }
    *out = z;
    return C3R_OK;
}

// text: whole lines (n == 0 or text[n - 1] == '\n'), in file order
int c3r_vcfz_write(c3r_vcfz *z, const char *text, int64_t n) {
    if (!z || n < 0 || (n && !text) || z->bad) return C3R_EINVAL;
    if (n == 0) return C3R_OK;
    if (text[n - 1] != '\n') return C3R_EINVAL;
    // (the trailing newline closes the last line: the split-based scan below would see one more, empty, line after it — nothing to index)
    z->scan_lines((const uint8_t *)text, (size_t)n - 1);
    z->n_seen += (uint64_t)n;
    z->pend.insert(z->pend.end(), (const uint8_t *)text, (const uint8_t *)text + n);
    if (!z->flush_blocks(false)) { z->bad = true; return C3R_EINVAL; }
    z->resolve(false);
    return C3R_OK;
}

// ---- pieces: a run of whole lines (one contig's merged records) compressed and indexed on its own, on any thread, and appended
// to the writer in file order.  A piece is its own series of BGZF blocks (the block before it is closed short), which changes where
// the blocks are cut but not what they hold; its index entries carry virtual offsets relative to the piece and are shifted by the
// file offset it lands on ((c + base) << 16 | u == v + (base << 16)).  Only the append — one fwrite and a fold of ~10^3 index entries
// — is left on the thread that keeps the file order.
struct c3r_vcfz_piece {
    std::vector<uint8_t> gz;
    uint64_t n = 0;                                                    // uncompressed bytes
    std::vector<std::string> names; std::vector<c3r_vcfz::Ctg> idx;    // lin: UINT64_MAX = no record starts in the window
};

int c3r_vcfz_piece_make(const char *text, int64_t n, int threads, c3r_vcfz_piece **out) {
    if (!out || n < 0 || (n && !text)) return C3R_EINVAL;
    *out = nullptr;
    if (n && text[n - 1] != '\n') return C3R_EINVAL;
    c3r_vcfz_piece *p = new c3r_vcfz_piece();
    p->n = (uint64_t)n;
    std::vector<uint64_t> co;
    const int nt = threads > 0 ? threads : (int)std::min(32u, std::max(1u, usable_cpus()));
    if (n && !bgzf_compress((const uint8_t *)text, (size_t)n, nt, p->gz, co)) { delete p; return C3R_EINVAL; }
    const uint64_t gz_end = (uint64_t)p->gz.size() << 16;
    auto rel = [&](uint64_t u) { return u < (uint64_t)n ? (co[(size_t)(u / BLK)] << 16) | (u % BLK) : gz_end; };
    std::string last; size_t id = 0;
    if (n) for_each_record((const uint8_t *)text, (size_t)n - 1, [&](size_t u, size_t step, const Field &ctg, int64_t beg, int64_t end) {
        if (p->names.empty() || last.size() != ctg.n || memcmp(last.data(), ctg.p, ctg.n) != 0) {
            last.assign(ctg.p, ctg.n);
            id = (size_t)(std::find(p->names.begin(), p->names.end(), last) - p->names.begin());
            if (id == p->names.size()) { p->names.push_back(last); p->idx.emplace_back(); }
        }
        c3r_vcfz::Ctg &x = p->idx[id];
        const uint64_t v0 = rel(u), v1 = rel(u + step);
        auto &ch = x.bins[(uint32_t)reg2bin_vcf(beg, end)];
        if (!ch.empty() && ch.back().second == v0) ch.back().second = v1;
        else ch.emplace_back(v0, v1);
        const size_t w1 = (size_t)((end - 1) >> 14);
        if (x.lin.size() <= w1) x.lin.resize(w1 + 1, UINT64_MAX);
        for (size_t w = (size_t)(beg >> 14); w <= w1; ++w) if (x.lin[w] == UINT64_MAX) x.lin[w] = v0;
    });
    *out = p;
    return C3R_OK;
}

void c3r_vcfz_piece_free(c3r_vcfz_piece *p) { delete p; }

int c3r_vcfz_append(c3r_vcfz *z, const c3r_vcfz_piece *p) {
    if (!z || !p || z->bad) return C3R_EINVAL;
    if (p->n == 0) return C3R_OK;
    // what was written before ends in a block of its own; a record that ended there ends at the piece's first block
    if (!z->flush_blocks(true)) { z->bad = true; return C3R_EINVAL; }
    z->resolve(true);
    const uint64_t base = z->coff;
    if (fwrite(p->gz.data(), 1, p->gz.size(), z->f) != p->gz.size()) { z->bad = true; return C3R_EINVAL; }
    z->coffs.push_back(base); z->bstart.push_back(z->n_unc);          // (one entry for the whole piece: nothing looks inside it again)
    z->coff += p->gz.size();
    z->n_unc += p->n; z->n_seen += p->n;
    const uint64_t shift = base << 16;
    for (size_t k = 0; k < p->names.size(); ++k) {
        auto it = z->which.find(p->names[k]);
        if (it == z->which.end()) { it = z->which.emplace(p->names[k], z->names.size()).first; z->names.push_back(p->names[k]); z->idx.emplace_back(); }
        c3r_vcfz::Ctg &x = z->idx[it->second];
        const c3r_vcfz::Ctg &y = p->idx[k];
        for (auto &kv : y.bins) {
            auto &ch = x.bins[kv.first];
            for (auto &c : kv.second) {
                if (!ch.empty() && ch.back().second == c.first + shift) ch.back().second = c.second + shift;
                else ch.emplace_back(c.first + shift, c.second + shift);
            }
        }
        if (x.lin.size() < y.lin.size()) x.lin.resize(y.lin.size(), 0);
        for (size_t w = 0; w < y.lin.size(); ++w) if (x.lin[w] == 0 && y.lin[w] != UINT64_MAX) x.lin[w] = y.lin[w] + shift;
    }
    return C3R_OK;
}

// keep != 0: finish <gz_path> and write <gz_path>.tbi; keep == 0: drop what was written (the caller decided on an empty output)
int c3r_vcfz_close(c3r_vcfz *z, int keep) {
    if (!z) return C3R_EINVAL;
    int rc = C3R_OK;
    if (!keep || z->bad) {
        if (z->f) fclose(z->f);
        remove(z->gz_path.c_str());
        rc = z->bad ? C3R_EINVAL : C3R_OK;
        delete z;
        return rc;
    }
    bool ok = z->flush_blocks(true);
    z->resolve(true);
    ok = ok && fwrite(BGZF_EOF, 1, sizeof BGZF_EOF, z->f) == sizeof BGZF_EOF;
    ok = (fclose(z->f) == 0) && ok;
    z->f = nullptr;
    // TBI v1, VCF preset (format 2, col_seq 1, col_beg 2, col_end 0, meta '#', skip 0): UCSC bins + 16 kb linear index
    std::vector<uint8_t> tbi = {'T', 'B', 'I', 1};
    std::string nm;
    for (auto &s : z->names) { nm += s; nm += '\0'; }
    const int32_t hdr[8] = {(int32_t)z->names.size(), 2, 1, 2, 0, '#', 0, (int32_t)nm.size()};
    for (int32_t h : hdr) put(tbi, h);
    tbi.insert(tbi.end(), nm.begin(), nm.end());
    for (c3r_vcfz::Ctg &x : z->idx) {
        for (size_t w = 1; w < x.lin.size(); ++w) if (x.lin[w] == 0) x.lin[w] = x.lin[w - 1];
        put(tbi, (int32_t)x.bins.size());
        for (auto &kv : x.bins) {
            put(tbi, (uint32_t)kv.first); put(tbi, (int32_t)kv.second.size());
            for (auto &c : kv.second) { put(tbi, c.first); put(tbi, c.second); }
        }
        put(tbi, (int32_t)x.lin.size());
        for (uint64_t v : x.lin) put(tbi, v);
    }
    std::vector<uint8_t> tgz; std::vector<uint64_t> tco;
    ok = ok && bgzf_compress(tbi.data(), tbi.size(), z->threads, tgz, tco);       // (a whole genome's index is a few MB: blocks on threads)
    ok = ok && write_file(z->gz_path + ".tbi", tgz.data(), tgz.size(), BGZF_EOF, sizeof BGZF_EOF);
    delete z;
    return ok ? C3R_OK : C3R_EINVAL;
}

int c3r_vcf_compress(const char *path, int threads) {
    if (!path) return C3R_EINVAL;
    FILE *f = fopen(path, "rb");
    if (!f) return C3R_EINVAL;
    std::vector<uint8_t> data;
    {
        uint8_t buf[1 << 16]; size_t k;
        while ((k = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + k);
        fclose(f);
    }
    c3r_vcfz *z = nullptr;
    int rc = c3r_vcfz_open((std::string(path) + ".gz").c_str(), threads, &z);
    if (rc) return rc;
    // one piece; a file that does not end in a newline is indexed like Python's data.split(b"\n") would
    if (!data.empty()) {
        if (data.back() == '\n') rc = c3r_vcfz_write(z, (const char *)data.data(), (int64_t)data.size());
        else {
            z->scan_lines(data.data(), data.size());
            z->n_seen += data.size();
            z->pend.insert(z->pend.end(), data.begin(), data.end());
        }
    }
    if (rc) { (void)c3r_vcfz_close(z, 0); return rc; }
    rc = c3r_vcfz_close(z, 1);
    if (rc == C3R_OK) remove(path);
    return rc;
}

}  // extern "C"

// ---- reference slice of a faidx-indexed FASTA (include/c3r_io.h)
extern "C" int c3r_fasta_fetch(const char *path, int64_t offset, int32_t linebases, int32_t linewidth, int64_t beg0, int64_t end0, int upper,
                               int threads, uint8_t *out) {
    if (!path || offset < 0 || linebases <= 0 || linewidth <= linebases || linewidth > linebases + 2 || beg0 < 0 || end0 < beg0) return C3R_EINVAL;
    if (end0 == beg0) return C3R_OK;
    if (!out) return C3R_EINVAL;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return C3R_EINVAL;
    // pieces of whole lines: piece k covers the lines [first_line + k * LINES, ...) clipped to [beg0, end0)
    const int64_t lb = linebases, lw = linewidth, term = lw - lb;
    const int64_t line0 = beg0 / lb, line1 = (end0 - 1) / lb + 1;
    const int64_t lines_per_piece = std::max<int64_t>(1, ((int64_t)4 << 20) / lw);
    const int64_t n_pieces = (line1 - line0 + lines_per_piece - 1) / lines_per_piece;
    std::atomic<int64_t> next(0);
    std::atomic<bool> bad(false);
    auto work = [&]() {
        std::vector<uint8_t> buf;
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_pieces || bad) break;
            const int64_t la = line0 + k * lines_per_piece, lz = std::min(line1, la + lines_per_piece);
            const int64_t b0 = std::max(beg0, la * lb), b1 = std::min(end0, lz * lb);                  // bases of this piece
            const int64_t f0 = offset + (b0 / lb) * lw + b0 % lb, f1 = offset + ((b1 - 1) / lb) * lw + (b1 - 1) % lb + 1;
            buf.resize((size_t)(f1 - f0));
            int64_t got = 0;
            while (got < f1 - f0) {
                const ssize_t r = pread(fd, buf.data() + got, (size_t)(f1 - f0 - got), (off_t)(f0 + got));
                if (r <= 0) break;
                got += r;
            }
            if (got != f1 - f0) { bad = true; break; }
            const uint8_t *src = buf.data();
            uint8_t *dst = out + (b0 - beg0);
            int64_t b = b0;
            while (b < b1) {
                const int64_t n = std::min(b1 - b, lb - b % lb);                                       // to the end of this line
                unsigned ends = 0;                                                                     // a line end inside a line: shorter than the index says
                if (upper) for (int64_t i = 0; i < n; ++i) { const uint8_t c = src[i]; ends |= (c == '\n') | (c == '\r'); dst[i] = (uint8_t)((uint8_t)(c - 'a') < 26 ? c - 32 : c); }
                else for (int64_t i = 0; i < n; ++i) { const uint8_t c = src[i]; ends |= (c == '\n') | (c == '\r'); dst[i] = c; }
                if (ends) { bad = true; break; }
                src += n; dst += n; b += n;
                if (b < b1) {                                                                          // the line end between two lines of the piece
                    if (src[term - 1] != '\n' || (term == 2 && src[0] != '\r')) { bad = true; break; }
                    src += term;
                }
            }
        }
    };
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : 8, n_pieces));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    close(fd);
    return bad ? C3R_EINVAL : C3R_OK;
}
