// prep_probe.hip — the read-preparation kernels (reads_kernels.hpp) on a read set dumped by tools/dump_reads.py: time per kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DC3R_PREP_GRP=g] [-DC3R_BIN_SHIFT=b] tools/prep_probe.hip -o tools/prep_probe
//   python tools/dump_reads.py /tmp/rs && tools/prep_probe /tmp/rs [count]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../clair3_rna_amd/csrc/reads_kernels.hpp"
using namespace c3r;
template <class T> static std::vector<T> slurp(const std::string &fn) {
    FILE *f = fopen(fn.c_str(), "rb"); if (!f) { perror(fn.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v((size_t)n / sizeof(T)); if (fread(v.data(), 1, (size_t)n, f) != (size_t)n) exit(1); fclose(f); return v;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char **argv) {
    const std::string base = argc > 1 ? argv[1] : "/tmp/rs";
    const bool count_only = argc > 2 && !strcmp(argv[2], "count");
    auto reads = slurp<c3r_read_t>(base + ".reads"); auto cig = slurp<uint32_t>(base + ".cigar");
    const int n = (int)reads.size();
    long long seq_bytes = 0; for (auto &r : reads) seq_bytes = std::max<long long>(seq_bytes, (long long)r.seq_off + (r.l_seq + 1) / 2);
    BinGeo g; g.base = reads[0].pos >> BIN_SHIFT; g.nb = (int)(((long long)reads[n - 1].pos + (1 << 21)) >> BIN_SHIFT) - g.base + 2; g.nbc = (g.nb >> CBIN_SHIFT) + 2; g.pad = 0;
    const size_t cnt_bytes = ((size_t)g.nb + 3 * (size_t)g.nbc) * 4;
    c3r_read_t *d_reads; uint32_t *d_cig, *d_cnt, *d_off; int4 *d_rtab; DevRead *d_out; uint8_t *d_serial; int32_t *d_nind, *d_pm; char *d_lbk; LoadStats *d_st; PileRec *d_recs = nullptr;
    CK(hipMalloc(&d_reads, (size_t)n * 32)); CK(hipMalloc(&d_cig, cig.size() * 4 + 16)); CK(hipMalloc(&d_cnt, cnt_bytes)); CK(hipMalloc(&d_off, (size_t)(g.nb + 1) * 4 + 16)); CK(hipMalloc(&d_rtab, (size_t)(g.nbc + 1) * 16));
    CK(hipMalloc(&d_out, (size_t)n * 32)); CK(hipMalloc(&d_serial, n + 16)); CK(hipMalloc(&d_nind, (size_t)n * 4)); CK(hipMalloc(&d_pm, (size_t)n * 4));
    const int nb_pm = (n + PM_BLK - 1) / PM_BLK, nb_bs = (g.nb + BS_BLK - 1) / BS_BLK, nb_bc = (g.nbc + BS_BLK - 1) / BS_BLK;
    const size_t lbk_bytes = 64 + (size_t)(nb_pm + nb_bs + 2 * nb_bc) * 8;
    CK(hipMalloc(&d_lbk, lbk_bytes)); CK(hipMalloc(&d_st, sizeof(LoadStats)));
    CK(hipMemcpy(d_reads, reads.data(), (size_t)n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(d_cig, cig.data(), cig.size() * 4, hipMemcpyHostToDevice));
    PrepArgs a; memset(&a, 0, sizeof a);
    a.reads = d_reads; a.n_reads = n; a.cigars = d_cig; a.n_cigar_ops = (long long)cig.size(); a.n_seq_bytes = seq_bytes; a.min_mq = 5; a.excl_flags = 2316; a.geo = g;
    a.cnt = d_cnt; a.sc = d_cnt + g.nb; a.ec = a.sc + g.nbc; uint32_t *pc = a.ec + g.nbc; a.rec_off = d_off; a.out = d_out; a.serial = d_serial; a.nind = d_nind; a.st = d_st;
    const unsigned grid = (unsigned)((n + PREP_READS - 1) / PREP_READS);
    hipEvent_t ev[6]; for (auto &e : ev) CK(hipEventCreate(&e));
    double t[5] = {0, 0, 0, 0, 0};
    LoadStats hs; size_t recs_cap = 0;
    const int R = 6;
    for (int rep = 0; rep < R; ++rep) {
        LoadStats init; memset(&init, 0, sizeof init); init.err = ~0ull;
        CK(hipMemcpy(d_st, &init, sizeof init, hipMemcpyHostToDevice));
        CK(hipMemset(d_lbk, 0, lbk_bytes));
        if (count_only || rep == 0) CK(hipMemset(d_cnt, 0, cnt_bytes));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(ev[0]));
        hipLaunchKernelGGL(k_prep<false>, dim3(grid), dim3(PREP_THREADS), 0, 0, a);
        CK(hipEventRecord(ev[1]));
        if (!count_only) {
            hipLaunchKernelGGL(k_prefmax_bins, dim3(nb_pm), dim3(1024), 0, 0, (const DevRead *)d_out, n, 5, 2316, g, (const int32_t *)d_nind, d_pm, pc, d_st, (int32_t *)d_lbk, (unsigned long long *)(d_lbk + 64));
            CK(hipEventRecord(ev[2]));
            hipLaunchKernelGGL(k_bin_scan, dim3(nb_bs + nb_bc), dim3(1024), 0, 0, (const uint32_t *)a.cnt, a.sc, pc, a.ec, g, nb_bs, d_off, d_rtab, d_st, (int32_t *)(d_lbk + 4),
                               (unsigned long long *)(d_lbk + 64 + (size_t)nb_pm * 8), (unsigned long long *)(d_lbk + 64 + (size_t)(nb_pm + nb_bs) * 8),
                               (unsigned long long *)(d_lbk + 64 + (size_t)(nb_pm + nb_bs + nb_bc) * 8));
            CK(hipEventRecord(ev[3]));
            CK(hipMemcpy(&hs, d_st, sizeof hs, hipMemcpyDeviceToHost));
            if (hs.err != ~0ull || hs.n_rec <= 0) { fprintf(stderr, "load error %llx, %d records\n", hs.err, hs.n_rec); return 1; }
            if ((size_t)hs.n_rec > recs_cap) { if (d_recs) CK(hipFree(d_recs)); recs_cap = (size_t)hs.n_rec + 1024; CK(hipMalloc(&d_recs, recs_cap * sizeof(PileRec))); }
            PrepArgs b = a; b.recs = d_recs;
            CK(hipEventRecord(ev[4]));
            hipLaunchKernelGGL(k_prep<true>, dim3(grid), dim3(PREP_THREADS), 0, 0, b);
            CK(hipEventRecord(ev[5]));
        }
        CK(hipDeviceSynchronize());
        if (rep == 0) continue;
        float ms;
        CK(hipEventElapsedTime(&ms, ev[0], ev[1])); t[0] += ms;
        if (!count_only) {
            CK(hipEventElapsedTime(&ms, ev[1], ev[2])); t[1] += ms; CK(hipEventElapsedTime(&ms, ev[2], ev[3])); t[2] += ms; CK(hipEventElapsedTime(&ms, ev[4], ev[5])); t[3] += ms;
        }
    }
    printf("grp %d: %d reads, %zu ops, %d bins", PREP_GRP, n, cig.size(), g.nb);
    if (!count_only) printf(", %d records, max_end %d, cover bound %d", hs.n_rec, hs.max_end, hs.max_cover);
    printf(" | k_prep<count> %.4f ms", t[0] / (R - 1));
    if (!count_only) printf("  k_prefmax_bins %.4f  k_bin_scan %.4f  k_prep<write> %.4f", t[1] / (R - 1), t[2] / (R - 1), t[3] / (R - 1));
    printf("\n");
    return 0;
}
