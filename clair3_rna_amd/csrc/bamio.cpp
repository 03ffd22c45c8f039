// bamio.cpp — libc3r_io.so: BAM / BGZF / BAI reader producing the flat read records of c3r_load_reads.
// C ABI in include/c3r_io.h.  Host-only (g++ -O2 -pthread -lz).
//
// What it replaces: the input side of `samtools mpileup <bam> -r ctg:beg-end --output-extra HP`
// (reference src/create_tensor_pileup.py:436-451).  Formats follow the SAM/BAM specification v1 (§4 BAM, §4.1 BGZF,
// §5.2 BAI: UCSC binning with 16 kb linear index); nothing here is derived from htslib sources.
//
//   indexed fetch ..... reg2bins -> chunk list (clipped by the linear index) -> inflate only those BGZF blocks
//   full load ......... block table from the BGZF headers, blocks inflated by a thread pool in batches, records parsed
//                       in file order with early exit once the contig is passed
//   index build ....... one pass over the records with their virtual offsets (`samtools index` equivalent)
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/c3r.h"
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

namespace {

struct Mapped {
    const uint8_t *p = nullptr; size_t n = 0; int fd = -1;
    bool open(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); fd = -1; return false; }
        n = (size_t)st.st_size;
        if (n == 0) { p = nullptr; return true; }
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); fd = -1; n = 0; return false; }
        p = (const uint8_t *)m;
        return true;
    }
    void close() {
        if (p) munmap((void *)p, n);
        if (fd >= 0) ::close(fd);
        p = nullptr; fd = -1; n = 0;
    }
};

inline uint16_t le16(const uint8_t *q) { return (uint16_t)(q[0] | (q[1] << 8)); }
inline uint32_t le32(const uint8_t *q) { return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24); }
inline int32_t le32s(const uint8_t *q) { return (int32_t)le32(q); }
inline uint64_t le64(const uint8_t *q) { return (uint64_t)le32(q) | ((uint64_t)le32(q + 4) << 32); }

// BGZF block at file offset `off`: total size and the position of the deflate payload.  0 on malformed input.
size_t bgzf_block_size(const uint8_t *f, size_t n, size_t off, size_t *payload_off) {
    if (off + 18 > n) return 0;
    const uint8_t *h = f + off;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
    const size_t xlen = le16(h + 10);
    if (off + 12 + xlen > n) return 0;
    size_t p = 12, bsize = 0;
    while (p + 4 <= 12 + xlen) {
        const uint8_t si1 = h[p], si2 = h[p + 1];
        const size_t slen = le16(h + p + 2);
        if (si1 == 66 && si2 == 67 && slen == 2) bsize = (size_t)le16(h + p + 4) + 1;
        p += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || off + bsize > n) return 0;
    *payload_off = 12 + xlen;
    return bsize;
}

struct Inflater {
    z_stream zs; bool ok;
    Inflater() { memset(&zs, 0, sizeof zs); ok = inflateInit2(&zs, -15) == Z_OK; }
    ~Inflater() { if (ok) inflateEnd(&zs); }
    // raw-deflate payload -> dst (isize bytes expected)
    bool run(const uint8_t *src, size_t n_src, uint8_t *dst, size_t isize) {
        if (!ok) return false;
        if (isize == 0) return true;
        inflateReset(&zs);
        zs.next_in = const_cast<Bytef *>(src); zs.avail_in = (uInt)n_src;
        zs.next_out = dst; zs.avail_out = (uInt)isize;
        const int rc = inflate(&zs, Z_FINISH);
        return rc == Z_STREAM_END && zs.avail_out == 0;
    }
};

// ---- sequential reader over virtual offsets (coffset << 16 | uoffset), used by the indexed fetch
struct BgzfCursor {
    const uint8_t *f; size_t n;
    Inflater inf;
    std::vector<uint8_t> buf;   // current block, inflated
    size_t coff = 0, bsize = 0, upos = 0;
    bool load(size_t off) {
        size_t pl;
        const size_t bs = bgzf_block_size(f, n, off, &pl);
        if (!bs) return false;
        const size_t isize = le32(f + off + bs - 4);
        buf.resize(isize);
        if (!inf.run(f + off + pl, bs - pl - 8, buf.data(), isize)) return false;
        coff = off; bsize = bs; upos = 0;
        return true;
    }
    bool seek(uint64_t voff) {
        if (!load((size_t)(voff >> 16))) return false;
        upos = (size_t)(voff & 0xffff);
        return upos <= buf.size();
    }
    uint64_t tell() const { return ((uint64_t)coff << 16) | (uint64_t)upos; }
    // 1 ok, 0 clean EOF before the first byte, -1 error / truncated
    int read(uint8_t *dst, size_t len) {
        size_t got = 0;
        while (got < len) {
            if (upos == buf.size()) {
                const size_t next = coff + bsize;
                if (next >= n) return got == 0 ? 0 : -1;
                if (!load(next)) return -1;
                continue;                      // (an empty block, e.g. the EOF marker, just moves on)
            }
            const size_t k = std::min(len - got, buf.size() - upos);
            memcpy(dst + got, buf.data() + upos, k);
            got += k; upos += k;
        }
        return 1;
    }
};

// ---- UCSC binning (SAM spec §5.3)
inline int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}
inline void reg2bins(int64_t beg, int64_t end, std::vector<uint32_t> &out) {
    --end;
    out.push_back(0);
    for (int64_t k = 1 + (beg >> 26); k <= 1 + (end >> 26); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 9 + (beg >> 23); k <= 9 + (end >> 23); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 73 + (beg >> 20); k <= 73 + (end >> 20); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 585 + (beg >> 17); k <= 585 + (end >> 17); ++k) out.push_back((uint32_t)k);
    for (int64_t k = 4681 + (beg >> 14); k <= 4681 + (end >> 14); ++k) out.push_back((uint32_t)k);
}

struct RefIndex {
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
    // the metadata pseudo-bin 37450 (SAM spec 5.2: two "chunks" — the file range of the contig's records, then the numbers of mapped and
    // unmapped reads), when the index has it: the weight a contig gets when the sample's contigs are dealt to the GPUs
    int64_t n_mapped = -1, n_unmapped = -1;
    uint64_t off_beg = 0, off_end = 0;
};

}  // namespace

// Large arrays (a contig's records, an inflated batch) in 2-MB aligned memory advised huge: a cold handle touches ~0.5 GB for the
// first time during its first fetch, and eight handles do so at once when a sample starts — 512 times fewer page faults where
// transparent huge pages are available ("madvise" or "always"), ordinary pages otherwise.
inline bool want_huge() {
    static const bool v = [] { const char *e = getenv("C3R_IO_HUGE"); return !(e && *e == '0'); }();      // C3R_IO_HUGE=0: ordinary pages
    return v;
}
template <class T>
struct HugeAlloc {
    typedef T value_type;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U> &) {}
    T *allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < ((size_t)4 << 20)) return static_cast<T *>(::operator new(bytes));
        void *p = nullptr;
        const size_t cap = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        if (posix_memalign(&p, (size_t)2 << 20, cap) != 0) throw std::bad_alloc();
        if (want_huge()) (void)madvise(p, cap, MADV_HUGEPAGE);
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n) { if (n * sizeof(T) < ((size_t)4 << 20)) ::operator delete(p); else free(p); }
    template <class U> bool operator==(const HugeAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U> &) const { return false; }
};
template <class T> using hvec = std::vector<T, HugeAlloc<T>>;

// a worker's share of a batch parsed on threads (kept with the handle: its capacity is reused batch after batch)
struct RecPart { hvec<c3r_read_t> reads; hvec<uint32_t> cigar; hvec<uint8_t> seq; };

struct c3r_bam {
    Mapped file;
    std::string path, err;
    std::vector<std::string> names;
    std::vector<int64_t> lens;
    uint64_t first_record_voff = 0;     // virtual offset of the first alignment
    size_t header_bytes = 0;            // uncompressed size of header + reference list
    bool has_bai = false;
    std::vector<RefIndex> bai;
    int n_threads = 1;
    bool malformed = false;             // the last fetch met a record whose fields do not fit its block (err holds the message)
    // result of the last fetch
    hvec<c3r_read_t> reads;
    hvec<uint32_t> cigar;
    hvec<uint8_t> seq;
    std::vector<RecPart> parts;
    double t_inflate = 0, t_visit = 0, t_parse = 0;      // seconds of the current fetch (reported under C3R_TIMING)
};

namespace {

int failb(c3r_bam *b, int code, const char *fmt, ...) {
    char tmp[512];
    va_list ap; va_start(ap, fmt); vsnprintf(tmp, sizeof tmp, fmt, ap); va_end(ap);
    if (b) b->err = tmp;
    return code;
}

// Where take_record puts what it keeps: the handle's own arrays, or a worker's part of a batch parsed on threads.
struct RecSink {
    hvec<c3r_read_t> *reads; hvec<uint32_t> *cigar; hvec<uint8_t> *seq;
    bool malformed = false; int32_t bad_pos = 0; const char *bad_what = nullptr;
};

// A record of the wanted contig whose declared fields run past its block: remember it (c3r_bam_fetch fails with C3R_EINVAL)
// and tell the caller to stop.
int malformed_record(c3r_bam *b, int32_t pos, const char *what) {
    if (!b->malformed) failb(b, C3R_EINVAL, "%s: malformed alignment record at position %d (%s)", b->path.c_str(), pos + 1, what);
    b->malformed = true;
    return 2;
}
inline int malformed_in(RecSink &o, int32_t pos, const char *what) {
    if (!o.malformed) { o.malformed = true; o.bad_pos = pos; o.bad_what = what; }
    return 2;
}

// Append one alignment (pointer to the 32 fixed bytes after block_size) if it belongs to (tid, [beg,end)).
// Returns 1 appended, 0 skipped, 2 = stop: past the region (sorted input) or a malformed record (o.malformed).
int take_record_into(RecSink &o, const uint8_t *r, size_t block_size, int tid, int64_t beg, int64_t end) {
    if (block_size < 32) return 0;
    const int32_t ref_id = le32s(r), pos = le32s(r + 4);
    const uint32_t l_read_name = r[8], mapq = r[9];
    uint32_t n_cig = le16(r + 12);
    const uint32_t flag = le16(r + 14);
    const uint32_t l_seq = le32(r + 16);
    if (ref_id != tid) return (ref_id > tid || ref_id < 0) ? 2 : 0;
    if (pos < 0) return 0;
    if (end > 0 && pos >= end) return 2;
    const size_t c0 = 32 + l_read_name, s0 = c0 + 4 * (size_t)n_cig, nb = ((size_t)l_seq + 1) / 2, a0 = s0 + nb + l_seq;
    if (a0 > block_size) return malformed_in(o, pos, "name / CIGAR / sequence longer than the record");
    const uint8_t *cig = r + c0;
    // aux: HP (any integer type) and CG:B,I (real CIGAR of reads with > 65535 ops, SAM spec §4.2.2)
    uint32_t hp = 0; const uint8_t *cg = nullptr; uint32_t cg_n = 0;
    size_t p = a0;
    while (p + 3 <= block_size) {
        const uint8_t t0 = r[p], t1 = r[p + 1], ty = r[p + 2];
        p += 3;
        size_t sz = 0;
        switch (ty) {
            case 'A': case 'c': case 'C': sz = 1; break;
            case 's': case 'S': sz = 2; break;
            case 'i': case 'I': case 'f': sz = 4; break;
            case 'Z': case 'H': { while (p < block_size && r[p]) ++p; ++p; continue; }
            case 'B': {
                if (p + 5 > block_size) { p = block_size; continue; }
                const uint8_t sub = r[p]; const uint32_t cnt = le32(r + p + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
                if ((size_t)cnt * es > block_size - (p + 5)) return malformed_in(o, pos, "B-array tag longer than the record");
                if (t0 == 'C' && t1 == 'G' && sub == 'I') { cg = r + p + 5; cg_n = cnt; }
                p += 5 + (size_t)cnt * es;
                continue;
            }
            default: p = block_size; continue;   // unknown type: stop scanning
        }
        if (p + sz > block_size) break;
        if (t0 == 'H' && t1 == 'P' && ty != 'A' && ty != 'f') {
            int64_t v = 0;
            switch (ty) {
                case 'c': v = (int8_t)r[p]; break;  case 'C': v = r[p]; break;
                case 's': v = (int16_t)le16(r + p); break;  case 'S': v = le16(r + p); break;
                case 'i': v = le32s(r + p); break;  case 'I': v = le32(r + p); break;
            }
            hp = (v > 0 && v < 256) ? (uint32_t)v : 0;
        }
        p += sz;
    }
    if (cg && n_cig == 2 && (le32(cig) & 15) == 4 && (le32(cig) >> 4) == l_seq && (le32(cig + 4) & 15) == 3) { cig = cg; n_cig = cg_n; }
    if (n_cig == 0) return 0;
    int64_t rlen = 0;
    for (uint32_t k = 0; k < n_cig; ++k) {
        const uint32_t c = le32(cig + 4 * k), op = c & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4;
    }
    if ((int64_t)pos + std::max<int64_t>(rlen, 1) <= beg) return 0;
    c3r_read_t rec;
    memset(&rec, 0, sizeof rec);
    rec.pos = pos; rec.cigar_off = (uint32_t)o.cigar->size(); rec.n_cigar = n_cig; rec.l_seq = l_seq; rec.seq_off = (uint64_t)o.seq->size();
    rec.flag = (uint16_t)flag; rec.mapq = (uint8_t)mapq; rec.hp = (uint8_t)hp;
    o.reads->push_back(rec);
    const size_t cb = o.cigar->size();
    o.cigar->resize(cb + n_cig);
    for (uint32_t k = 0; k < n_cig; ++k) (*o.cigar)[cb + k] = le32(cig + 4 * k);
    o.seq->insert(o.seq->end(), r + s0, r + s0 + nb);
    return 1;
}

int take_record(c3r_bam *b, const uint8_t *r, size_t block_size, int tid, int64_t beg, int64_t end) {
    RecSink o{&b->reads, &b->cigar, &b->seq};
    const int rc = take_record_into(o, r, block_size, tid, beg, end);
    if (o.malformed) return malformed_record(b, o.bad_pos, o.bad_what);
    return rc;
}

// The records of one inflated batch, parsed on the handle's threads: each thread fills its own arrays from a contiguous run of
// records, the runs are appended in file order with their offsets moved.  (One thread spent 0.14 s of a 250-Mb contig's fetch in
// take_record — as long as the inflate took on eight.)  Returns false when a record was malformed (b->malformed is set).
struct RecRef { const uint8_t *p; size_t bs; };
bool take_records(c3r_bam *b, const std::vector<RecRef> &recs, int tid, int64_t beg, int64_t end) {
    const size_t n = recs.size();
    size_t per = 1024;                                                 // records that make a thread worth starting
    if (const char *e = getenv("C3R_IO_PARSE_MIN")) per = (size_t)std::max(1LL, atoll(e));     // tests: threads on small files
    const size_t T = std::min<size_t>((size_t)std::max(1, b->n_threads), n / per + 1);
    if (T <= 1) {
        for (const RecRef &r : recs) if (take_record(b, r.p, r.bs, tid, beg, end) == 2 && b->malformed) return false;
        return true;
    }
    if (b->parts.size() < T) b->parts.resize(T);
    std::vector<RecSink> sinks(T);
    auto work = [&](size_t t) {
        RecPart &q = b->parts[t];
        q.reads.clear(); q.cigar.clear(); q.seq.clear();
        sinks[t] = RecSink{&q.reads, &q.cigar, &q.seq};
        for (size_t i = n * t / T; i < n * (t + 1) / T; ++i)
            if (take_record_into(sinks[t], recs[i].p, recs[i].bs, tid, beg, end) == 2 && sinks[t].malformed) break;
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    size_t nr = b->reads.size(), nc = b->cigar.size(), ns = b->seq.size();
    for (size_t t = 0; t < T; ++t) { const RecPart &q = b->parts[t]; nr += q.reads.size(); nc += q.cigar.size(); ns += q.seq.size(); }
    // (grow by doubling, not to the exact size: a contig is many batches)
    if (nr > b->reads.capacity()) b->reads.reserve(std::max(nr, 2 * b->reads.capacity()));
    if (nc > b->cigar.capacity()) b->cigar.reserve(std::max(nc, 2 * b->cigar.capacity()));
    if (ns > b->seq.capacity()) b->seq.reserve(std::max(ns, 2 * b->seq.capacity()));
    for (size_t t = 0; t < T; ++t) {
        RecPart &q = b->parts[t];
        const uint32_t base_c = (uint32_t)b->cigar.size();
        const uint64_t base_s = (uint64_t)b->seq.size();
        for (c3r_read_t &r : q.reads) { r.cigar_off += base_c; r.seq_off += base_s; }
        b->reads.insert(b->reads.end(), q.reads.begin(), q.reads.end());
        b->cigar.insert(b->cigar.end(), q.cigar.begin(), q.cigar.end());
        b->seq.insert(b->seq.end(), q.seq.begin(), q.seq.end());
        if (sinks[t].malformed) { malformed_record(b, sinks[t].bad_pos, sinks[t].bad_what); return false; }   // (what follows it is dropped, as in a serial pass)
    }
    return true;
}

// ---- header (through a cursor: it may span BGZF blocks)
int parse_header(c3r_bam *b) {
    BgzfCursor cur; cur.f = b->file.p; cur.n = b->file.n;
    if (!cur.load(0)) return failb(b, C3R_EINVAL, "%s: not a BGZF file", b->path.c_str());
    uint8_t h[12];
    if (cur.read(h, 8) != 1 || memcmp(h, "BAM\1", 4) != 0) return failb(b, C3R_EINVAL, "%s: bad BAM magic", b->path.c_str());
    const int32_t l_text = le32s(h + 4);
    std::vector<uint8_t> tmp((size_t)std::max(l_text, 0));
    if (l_text > 0 && cur.read(tmp.data(), (size_t)l_text) != 1) return failb(b, C3R_EINVAL, "%s: truncated header", b->path.c_str());
    if (cur.read(h, 4) != 1) return failb(b, C3R_EINVAL, "%s: truncated header", b->path.c_str());
    const int32_t n_ref = le32s(h);
    if (l_text < 0 || n_ref < 0) return failb(b, C3R_EINVAL, "%s: negative header length / reference count", b->path.c_str());
    size_t total = 12 + (size_t)std::max(l_text, 0);
    for (int i = 0; i < n_ref; ++i) {
        if (cur.read(h, 4) != 1) return failb(b, C3R_EINVAL, "%s: truncated reference list", b->path.c_str());
        const int32_t l_name = le32s(h);
        if (l_name <= 0 || l_name > (1 << 20)) return failb(b, C3R_EINVAL, "%s: bad reference name length %d in the header", b->path.c_str(), l_name);
        std::vector<uint8_t> nm((size_t)l_name + 4);
        if (cur.read(nm.data(), (size_t)l_name + 4) != 1) return failb(b, C3R_EINVAL, "%s: truncated reference list", b->path.c_str());
        b->names.emplace_back((const char *)nm.data(), (size_t)std::max(l_name - 1, 0));
        b->lens.push_back(le32s(nm.data() + l_name));
        total += 8 + (size_t)l_name;
    }
    b->header_bytes = total;
    if (cur.upos == cur.buf.size() && cur.coff + cur.bsize < cur.n) cur.load(cur.coff + cur.bsize);   // normalise like tell() after a read
    b->first_record_voff = cur.tell();
    return C3R_OK;
}

int load_bai(c3r_bam *b, const std::string &p) {
    Mapped m;
    if (!m.open(p.c_str())) return 1;
    int rc = 1;
    do {
        if (m.n < 8 || memcmp(m.p, "BAI\1", 4) != 0) break;
        size_t o = 4;
        const int32_t n_ref = le32s(m.p + o); o += 4;
        std::vector<RefIndex> idx((size_t)std::max(n_ref, 0));
        bool bad = false;
        for (int r = 0; r < n_ref && !bad; ++r) {
            if (o + 4 > m.n) { bad = true; break; }
            const int32_t n_bin = le32s(m.p + o); o += 4;
            for (int k = 0; k < n_bin; ++k) {
                if (o + 8 > m.n) { bad = true; break; }
                const uint32_t bin = le32(m.p + o); const int32_t n_chunk = le32s(m.p + o + 4); o += 8;
                if (o + 16 * (size_t)n_chunk > m.n) { bad = true; break; }
                if (bin != 37450) {   // (37450 = metadata pseudo-bin)
                    auto &v = idx[r].bins[bin];
                    for (int c = 0; c < n_chunk; ++c) v.emplace_back(le64(m.p + o + 16 * c), le64(m.p + o + 16 * c + 8));
                } else if (n_chunk >= 2) {
                    idx[r].off_beg = le64(m.p + o); idx[r].off_end = le64(m.p + o + 8);
                    idx[r].n_mapped = (int64_t)le64(m.p + o + 16); idx[r].n_unmapped = (int64_t)le64(m.p + o + 24);
                }
                o += 16 * (size_t)n_chunk;
            }
            if (bad || o + 4 > m.n) { bad = true; break; }
            const int32_t n_intv = le32s(m.p + o); o += 4;
            if (o + 8 * (size_t)n_intv > m.n) { bad = true; break; }
            idx[r].linear.resize((size_t)n_intv);
            for (int i = 0; i < n_intv; ++i) idx[r].linear[i] = le64(m.p + o + 8 * i);
            o += 8 * (size_t)n_intv;
        }
        if (bad || (size_t)n_ref != b->names.size()) break;
        b->bai.swap(idx); b->has_bai = true; rc = 0;
    } while (0);
    m.close();
    return rc;
}

struct BlockRef { size_t off, payload, bsize, isize; };

// BGZF blocks of the file range [off0, off1) (off1 = a block boundary or the file size)
int list_blocks(c3r_bam *b, size_t off0, size_t off1, std::vector<BlockRef> &blocks) {
    const uint8_t *f = b->file.p; const size_t n = b->file.n;
    for (size_t off = off0; off < off1 && off < n;) {
        size_t pl;
        const size_t bs = bgzf_block_size(f, n, off, &pl);
        if (!bs) return failb(b, C3R_EINVAL, "%s: corrupt BGZF block at %zu", b->path.c_str(), off);
        blocks.push_back({off, pl, bs, le32(f + off + bs - 4)});
        off += bs;
    }
    return C3R_OK;
}

// Pass over consecutive blocks: batches inflated in parallel, the first `skip` uncompressed bytes ignored (header, or the
// in-block offset of an index chunk), records handed to `visit(rec, block_size, voff_begin, voff_end)` in file order; visit
// returns false to stop.
// `batch_done()` runs whenever the record pointers handed to `visit` are about to go stale (end of a batch, after a record carried
// over from the previous batch): a visitor may queue pointers and parse them there.
template <class F, class G>
int scan_blocks(c3r_bam *b, const std::vector<BlockRef> &blocks, size_t skip, F visit, bool *truncated, G batch_done) {
    const uint8_t *f = b->file.p;
    size_t BATCH = 512;                   // blocks inflated per round (<= 32 MB of records in memory)
    if (const char *e = getenv("C3R_IO_BATCH")) BATCH = (size_t)std::max(1, atoi(e));   // tests: force records across rounds
    const int nt = std::max(1, b->n_threads);
    std::vector<uint8_t> carry;           // partial record bytes from the previous batch
    uint64_t carry_first_voff = 0;
    hvec<uint8_t> buf;
    std::vector<size_t> uoff;
    bool stop = false;
    for (size_t b0 = 0; b0 < blocks.size() && !stop; b0 += BATCH) {
        const size_t b1 = std::min(blocks.size(), b0 + BATCH);
        uoff.assign(b1 - b0 + 1, 0);
        for (size_t i = b0; i < b1; ++i) uoff[i - b0 + 1] = uoff[i - b0] + blocks[i].isize;
        buf.resize(uoff.back());
        std::atomic<size_t> next(b0);
        std::atomic<bool> bad(false);
        auto work = [&]() {
            Inflater inf;
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= b1) break;
                const BlockRef &k = blocks[i];
                if (!inf.run(f + k.off + k.payload, k.bsize - k.payload - 8, buf.data() + uoff[i - b0], k.isize)) bad = true;
            }
        };
        const auto ti0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto &t : th) t.join();
        const auto ti1 = std::chrono::steady_clock::now();
        b->t_inflate += std::chrono::duration<double>(ti1 - ti0).count();
        if (bad) return failb(b, C3R_EINVAL, "%s: inflate failed", b->path.c_str());
        // virtual offset of byte u of this batch, in the canonical form a reader's tell() has after consuming byte u-1:
        // inside a block (coffset, u - block start); exactly at a block's end the address of the block that follows
        // (empty blocks hold no byte and are never chosen, so the value does not depend on how blocks are batched)
        size_t jc = 0;                    // block of the most recent lookup: offsets are asked for in ascending order within a batch
        auto voff_of = [&](size_t u) -> uint64_t {
            if (u == 0) return (uint64_t)blocks[b0].off << 16;
            while (jc + 2 < uoff.size() && uoff[jc + 1] <= u - 1) ++jc;          // the last block that starts at or before byte u - 1
            const size_t j = jc;
            const BlockRef &k = blocks[b0 + j];
            if (u == uoff[j + 1]) return (uint64_t)(k.off + k.bsize) << 16;
            return ((uint64_t)k.off << 16) | (uint64_t)(u - uoff[j]);
        };
        size_t u = 0;
        if (skip) { const size_t k = std::min(skip, buf.size()); u = k; skip -= k; if (skip) continue; }
        // finish a carried record first
        if (!carry.empty()) {
            while (carry.size() < 4 && u < buf.size()) carry.push_back(buf[u++]);
            if (carry.size() < 4) continue;
            const size_t bs = le32(carry.data());
            while (carry.size() < 4 + bs && u < buf.size()) carry.push_back(buf[u++]);
            if (carry.size() < 4 + bs) continue;
            const bool go = visit(carry.data() + 4, bs, carry_first_voff, voff_of(u));
            batch_done();
            if (!go) { stop = true; break; }
            carry.clear();
        }
        while (u < buf.size()) {
            if (u + 4 > buf.size() || u + 4 + le32(buf.data() + u) > buf.size()) {
                carry.assign(buf.begin() + (long)u, buf.end());
                carry_first_voff = voff_of(u);
                break;
            }
            const size_t bs = le32(buf.data() + u);
            const uint64_t v0 = voff_of(u);                  // (in this order: the lookup keeps a cursor)
            const uint64_t v1 = voff_of(u + 4 + bs);
            if (!visit(buf.data() + u + 4, bs, v0, v1)) { stop = true; break; }
            u += 4 + bs;
        }
        const auto tv1 = std::chrono::steady_clock::now();
        b->t_visit += std::chrono::duration<double>(tv1 - ti1).count();
        batch_done();
        b->t_parse += std::chrono::duration<double>(std::chrono::steady_clock::now() - tv1).count();
    }
    if (truncated) *truncated = !stop && !carry.empty();      // the listed blocks end inside a record
    return C3R_OK;
}
template <class F>
int scan_blocks(c3r_bam *b, const std::vector<BlockRef> &blocks, size_t skip, F visit, bool *truncated = nullptr) {
    return scan_blocks(b, blocks, skip, visit, truncated, [] {});
}

// Whole-file pass (no index, or building one).
template <class F>
int scan_all(c3r_bam *b, F visit) {
    std::vector<BlockRef> blocks;
    const int rc = list_blocks(b, 0, b->file.n, blocks);
    if (rc) return rc;
    return scan_blocks(b, blocks, b->header_bytes, visit);
}

int fetch_indexed(c3r_bam *b, int tid, int64_t beg, int64_t end) {
    const RefIndex &ri = b->bai[tid];
    std::vector<uint32_t> bins;
    reg2bins(beg, end, bins);
    uint64_t min_off = 0;
    const size_t w = (size_t)(beg >> 14);
    if (!ri.linear.empty()) min_off = ri.linear[std::min(w, ri.linear.size() - 1)];
    std::vector<std::pair<uint64_t, uint64_t>> chunks;
    for (uint32_t bin : bins) {
        auto it = ri.bins.find(bin);
        if (it == ri.bins.end()) continue;
        for (auto &c : it->second) if (c.second > min_off) chunks.emplace_back(std::max(c.first, min_off), c.second);
    }
    std::sort(chunks.begin(), chunks.end());
    // merge overlapping / adjacent chunks so that no record is read twice
    std::vector<std::pair<uint64_t, uint64_t>> merged;
    for (auto &c : chunks) {
        if (!merged.empty() && c.first <= merged.back().second) merged.back().second = std::max(merged.back().second, c.second);
        else merged.push_back(c);
    }
    BgzfCursor cur; cur.f = b->file.p; cur.n = b->file.n;
    std::vector<uint8_t> rec;
    for (auto &c : merged) {
        // a long chunk (a whole contig, typically): inflate its blocks on all threads instead of one by one.  A record that
        // begins before c.second may end blocks later, so blocks are listed a margin past it, and further if that was short.
        const size_t c0 = (size_t)(c.first >> 16), c1 = (size_t)(c.second >> 16);
        size_t par_min = (size_t)1 << 20, margin0 = (size_t)4 << 20;
        if (const char *e = getenv("C3R_IO_PAR_MIN")) par_min = (size_t)atoll(e);       // tests: force either path
        if (const char *e = getenv("C3R_IO_MARGIN")) margin0 = (size_t)std::max(1LL, atoll(e));
        if (b->n_threads > 1 && c1 - c0 >= par_min) {
            bool done = false, past = false;
            for (size_t margin = margin0; !done; margin *= 4) {
                std::vector<BlockRef> blocks;
                const size_t lim = std::min(b->file.n, c1 + margin);
                int rc = list_blocks(b, c0, lim, blocks);
                if (rc) return rc;
                const size_t n_reads0 = b->reads.size(), n_cig0 = b->cigar.size(), n_seq0 = b->seq.size();
                bool truncated = false;
                past = false;
                // the visitor only looks at the fixed fields (which records belong to the region, where the region ends); the
                // records it queues are parsed on threads when their batch is complete
                std::vector<RecRef> queued;
                bool bad = false;
                rc = scan_blocks(b, blocks, (size_t)(c.first & 0xffff), [&](const uint8_t *r, size_t bs, uint64_t v0, uint64_t) {
                    if (v0 >= c.second || bad) return false;
                    if (bs >= 32) {
                        const int32_t ref_id = le32s(r), pos = le32s(r + 4);
                        if (ref_id != tid) { if (ref_id > tid || ref_id < 0) { past = true; return false; } return true; }
                        if (pos >= 0 && end > 0 && pos >= end) { past = true; return false; }
                        queued.push_back(RecRef{r, bs});
                    }
                    return true;
                }, &truncated, [&] { if (!queued.empty()) { if (!take_records(b, queued, tid, beg, end)) { bad = true; past = true; } queued.clear(); } });
                if (rc) return rc;
                if (truncated && lim < b->file.n) {        // start over with more blocks (rare: a record longer than the margin)
                    b->reads.resize(n_reads0); b->cigar.resize(n_cig0); b->seq.resize(n_seq0);
                    continue;
                }
                done = true;
            }
            if (past) return C3R_OK;                       // sorted: nothing further can overlap
            continue;
        }
        if (!cur.seek(c.first)) return failb(b, C3R_EINVAL, "%s: index points outside the file", b->path.c_str());
        for (;;) {
            if (cur.upos == cur.buf.size() && cur.coff + cur.bsize < cur.n && !cur.load(cur.coff + cur.bsize))
                return failb(b, C3R_EINVAL, "%s: corrupt BGZF block", b->path.c_str());
            if (cur.tell() >= c.second) break;
            uint8_t h[4];
            const int g = cur.read(h, 4);
            if (g == 0) break;
            if (g < 0) return failb(b, C3R_EINVAL, "%s: truncated record", b->path.c_str());
            const size_t bs = le32(h);
            rec.resize(bs);
            if (cur.read(rec.data(), bs) != 1) return failb(b, C3R_EINVAL, "%s: truncated record", b->path.c_str());
            if (take_record(b, rec.data(), bs, tid, beg, end) == 2) return C3R_OK;   // sorted: nothing further can overlap
        }
    }
    return C3R_OK;
}

}  // namespace

extern "C" {

void *c3r_io_alloc(size_t bytes) {
    void *p = nullptr;
    const size_t cap = (std::max<size_t>(bytes, 1) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    if (posix_memalign(&p, (size_t)2 << 20, cap) != 0) return nullptr;
    if (want_huge()) (void)madvise(p, cap, MADV_HUGEPAGE);
    return p;
}
void c3r_io_free(void *p) { free(p); }


int c3r_bam_open(const char *path, int n_threads, c3r_bam **out) {
    if (!path || !out) return C3R_EINVAL;
    c3r_bam *b = new c3r_bam();
    b->path = path;
    b->n_threads = n_threads > 0 ? n_threads : (int)std::max(1u, usable_cpus());
    *out = b;                      // handed back even on failure so that the caller can read the message
    if (!b->file.open(path)) return failb(b, C3R_EINVAL, "%s: cannot open", path);
    int rc = parse_header(b);
    if (rc) return rc;
    std::string p1 = b->path + ".bai", p2;
    if (b->path.size() > 4 && b->path.compare(b->path.size() - 4, 4, ".bam") == 0) p2 = b->path.substr(0, b->path.size() - 4) + ".bai";
    if (load_bai(b, p1) != 0 && !p2.empty()) load_bai(b, p2);
    return C3R_OK;
}

void c3r_bam_close(c3r_bam *b) {
    if (!b) return;
    b->file.close();
    delete b;
}

const char *c3r_bam_last_error(c3r_bam *b) { return b ? b->err.c_str() : "null handle"; }
int c3r_bam_n_contigs(c3r_bam *b) { return b ? (int)b->names.size() : C3R_EINVAL; }
int c3r_bam_has_index(c3r_bam *b) { return b && b->has_bai ? 1 : 0; }

int c3r_bam_contig(c3r_bam *b, int i, const char **name, int64_t *length) {
    if (!b || i < 0 || i >= (int)b->names.size()) return C3R_EINVAL;
    if (name) *name = b->names[(size_t)i].c_str();
    if (length) *length = b->lens[(size_t)i];
    return C3R_OK;
}

int c3r_bam_fetch(c3r_bam *b, const char *contig, int64_t beg0, int64_t end0, int64_t *n_reads, int64_t *n_cigar, int64_t *n_seq_bytes) {
    if (!b || !contig) return C3R_EINVAL;
    b->reads.clear(); b->cigar.clear(); b->seq.clear();
    b->malformed = false;
    b->t_inflate = b->t_visit = b->t_parse = 0;
    const auto t_fetch0 = std::chrono::steady_clock::now();
    int tid = -1;
    for (size_t i = 0; i < b->names.size(); ++i) if (b->names[i] == contig) { tid = (int)i; break; }
    int rc = C3R_OK;
    if (tid >= 0) {
        if (beg0 < 0) beg0 = 0;
        if (end0 <= 0) end0 = std::max<int64_t>(b->lens[(size_t)tid], (int64_t)1 << 29);
        if (b->has_bai) {
            rc = fetch_indexed(b, tid, beg0, end0);
        } else {
            rc = scan_all(b, [&](const uint8_t *r, size_t bs, uint64_t, uint64_t) { return take_record(b, r, bs, tid, beg0, end0) != 2; });
        }
    }
    if (rc == C3R_OK && b->malformed) { rc = C3R_EINVAL; b->reads.clear(); b->cigar.clear(); b->seq.clear(); }
    if (getenv("C3R_TIMING") && b->reads.size() > 10000)
        fprintf(stderr, "[c3r_bam_fetch %s] %zu reads in %.0f ms: inflate %.0f ms, record walk %.0f ms, parse %.0f ms\n", contig, b->reads.size(),
                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fetch0).count(), 1e3 * b->t_inflate, 1e3 * b->t_visit, 1e3 * b->t_parse);
    if (n_reads) *n_reads = (int64_t)b->reads.size();
    if (n_cigar) *n_cigar = (int64_t)b->cigar.size();
    if (n_seq_bytes) *n_seq_bytes = (int64_t)b->seq.size();
    return rc;
}

int c3r_bam_copy(c3r_bam *b, c3r_read_t *reads, uint32_t *cigar, uint8_t *seq) {
    if (!b) return C3R_EINVAL;
    if (reads && !b->reads.empty()) memcpy(reads, b->reads.data(), b->reads.size() * sizeof(c3r_read_t));
    if (cigar && !b->cigar.empty()) memcpy(cigar, b->cigar.data(), b->cigar.size() * 4);
    if (seq && !b->seq.empty()) memcpy(seq, b->seq.data(), b->seq.size());
    return C3R_OK;
}

int c3r_bam_contig_weight(c3r_bam *b, int i, int64_t *n_mapped, int64_t *file_bytes) {
    if (!b || i < 0 || (size_t)i >= b->names.size()) return C3R_EINVAL;
    int64_t nm = -1, fb = -1;
    if (b->has_bai && (size_t)i < b->bai.size()) {
        const RefIndex &ri = b->bai[(size_t)i];
        nm = ri.n_mapped;
        uint64_t lo = ri.off_beg, hi = ri.off_end;
        if (ri.n_mapped < 0) {             // no pseudo-bin: the span of the contig's chunks
            lo = ~0ull; hi = 0;
            for (auto &kv : ri.bins) for (auto &c : kv.second) { lo = std::min(lo, c.first); hi = std::max(hi, c.second); }
            // `samtools index` writes the pseudo-bin only for references that HAVE records: a contig without a single bin (chrY, chrM or a
            // decoy under --include_all_ctgs) holds no read — 0 mapped reads, 0 bytes, not "unknown" (which would push the whole deal of a
            // sample from mapped reads down to compressed bytes)
            if (ri.bins.empty()) nm = 0;
        }
        fb = hi > lo ? (int64_t)((hi >> 16) - (lo >> 16)) : 0;       // (virtual offsets: compressed file offset << 16 | offset in the block)
    }
    if (n_mapped) *n_mapped = nm;
    if (file_bytes) *file_bytes = fb;
    return C3R_OK;
}

int c3r_bam_index_build(const char *bam_path, const char *bai_path) {
    if (!bam_path || !bai_path) return C3R_EINVAL;
    c3r_bam *b = nullptr;
    int rc = c3r_bam_open(bam_path, 0, &b);
    if (rc) { c3r_bam_close(b); return rc; }
    const size_t n_ref = b->names.size();
    std::vector<RefIndex> idx(n_ref);
    int32_t last_tid = -1, last_pos = -1;
    bool unsorted = false;
    rc = scan_all(b, [&](const uint8_t *r, size_t bs, uint64_t v0, uint64_t v1) {
        if (bs < 32) return true;
        const int32_t tid = le32s(r), pos = le32s(r + 4);
        if (tid < 0 || (size_t)tid >= n_ref || pos < 0) return true;        // unplaced reads are not indexed
        if (tid < last_tid || (tid == last_tid && pos < last_pos)) { unsorted = true; return false; }
        last_tid = tid; last_pos = pos;
        {   // the pseudo-bin's numbers: file range of the contig's records, mapped / unmapped (FLAG 0x4) reads
            RefIndex &ri0 = idx[(size_t)tid];
            if (ri0.n_mapped < 0) { ri0.n_mapped = 0; ri0.n_unmapped = 0; ri0.off_beg = v0; }
            ri0.off_end = v1;
            if (le16(r + 14) & 4u) ri0.n_unmapped++; else ri0.n_mapped++;
        }
        const uint32_t l_read_name = r[8], n_cig = le16(r + 12);
        int64_t rlen = 0;
        const uint8_t *cig = r + 32 + l_read_name;
        if (32 + l_read_name + 4 * (size_t)n_cig <= bs)
            for (uint32_t k = 0; k < n_cig; ++k) { const uint32_t c = le32(cig + 4 * k), op = c & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rlen += c >> 4; }
        const int64_t beg = pos, end = pos + std::max<int64_t>(rlen, 1);
        RefIndex &ri = idx[(size_t)tid];
        auto &ch = ri.bins[(uint32_t)reg2bin(beg, end)];
        if (!ch.empty() && ch.back().second == v0) ch.back().second = v1; else ch.emplace_back(v0, v1);
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; ++w) if (ri.linear[w] == 0) ri.linear[w] = v0;
        return true;
    });
    if (rc == C3R_OK && unsorted) rc = failb(b, C3R_EINVAL, "%s: not coordinate-sorted", bam_path);
    if (rc == C3R_OK) {
        // windows no read starts in inherit the offset of the next populated window's predecessor (spec: fill from the left)
        for (auto &ri : idx) for (size_t w = 1; w < ri.linear.size(); ++w) if (ri.linear[w] == 0) ri.linear[w] = ri.linear[w - 1];
        FILE *fo = fopen(bai_path, "wb");
        if (!fo) rc = failb(b, C3R_EINVAL, "%s: cannot write", bai_path);
        else {
            auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, fo); };
            auto w64 = [&](uint64_t v) { fwrite(&v, 8, 1, fo); };
            fwrite("BAI\1", 1, 4, fo); w32((uint32_t)n_ref);
            for (auto &ri : idx) {
                w32((uint32_t)(ri.bins.size() + (ri.n_mapped >= 0 ? 1 : 0)));
                for (auto &kv : ri.bins) {
                    w32(kv.first); w32((uint32_t)kv.second.size());
                    for (auto &c : kv.second) { w64(c.first); w64(c.second); }
                }
                if (ri.n_mapped >= 0) {            // metadata pseudo-bin, as samtools index writes it
                    w32(37450u); w32(2u);
                    w64(ri.off_beg); w64(ri.off_end); w64((uint64_t)ri.n_mapped); w64((uint64_t)ri.n_unmapped);
                }
                w32((uint32_t)ri.linear.size());
                for (uint64_t v : ri.linear) w64(v);
            }
            fclose(fo);
        }
    }
    c3r_bam_close(b);
    return rc;
}

}  // extern "C"
