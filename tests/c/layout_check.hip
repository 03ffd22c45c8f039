// Host-side checks of K0's counter layout (clair3_rna_amd/csrc/reads_kernels.hpp, cnt_at) and of the giant spans' slice arithmetic
// (pileup_kernels.hpp, giant_slices / giant_slice): compiled by hipcc, runs without a GPU.
//   * a permutation of every block of 1024 bins (no two bins share a counter, none leaves its block: the host sizes the array in whole blocks)
//   * a group of four bins stays four consecutive, 16-byte-aligned words in bin order (k_bin_scan reads it with one load per thread)
//   * neighbouring groups lie at least a cache line (128 bytes) apart — the point of the layout: the bins of one locus are neighbours
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../clair3_rna_amd/csrc/reads_kernels.hpp"

int main() {
    const uint32_t N = 8 * 1024;
    std::vector<int> seen(N, 0);
    for (uint32_t b = 0; b < N; ++b) {
        const uint32_t i = c3r::cnt_at(b);
        if (i >= N || (i & ~1023u) != (b & ~1023u)) { printf("bin %u leaves its block: %u\n", b, i); return 1; }
        if (seen[i]++) { printf("two bins on counter %u\n", i); return 1; }
        if ((i & 3u) != (b & 3u) || c3r::cnt_at(b & ~3u) + (b & 3u) != i) { printf("group of bin %u is not contiguous\n", b); return 1; }
    }
    if (C3R_CNT_SWZ)
        for (uint32_t b = 0; b + 4 < N; b += 4) {
            if ((b & 1023u) == 1020u) continue;                       // (the next group opens the next block)
            const long d = (long)c3r::cnt_at(b + 4) - (long)c3r::cnt_at(b);
            if ((d < 0 ? -d : d) * 4 < 128) { printf("groups %u and %u share a cache line\n", b / 4, b / 4 + 1); return 1; }
        }
    // the slices of a giant span's records (k_deep_walk) and events (k_deep_alleles): 1 .. GIANT_MAX_HELP of them, a partition of [0, n) in order
    const int ns[] = {1, 2, 63, 64, 65, 4095, 4096, 4097, 8192, 131071, 131072, 131073, 264960, 655904, 12582912, 2147483000, 2147483646};
    const int slices[] = {1, 64, 4096, 16384};
    for (int n : ns)
        for (int sl : slices) {
            const int G = c3r::giant_slices(n, sl);
            if (G < 1 || G > c3r::GIANT_MAX_HELP || (G < c3r::GIANT_MAX_HELP && (long long)G * sl < n)) { printf("n %d slice %d: %d slices\n", n, sl, G); return 1; }
            int at = 0;
            for (int k = 0; k < G; ++k) {
                int lo, hi;
                c3r::giant_slice(n, k, G, lo, hi);
                if (lo < hi) { if (lo != at) { printf("n %d G %d: slice %d starts at %d, not %d\n", n, G, k, lo, at); return 1; } at = hi; }
                else if (lo < at) { printf("n %d G %d: empty slice %d before the end\n", n, G, k); return 1; }
            }
            if (at != n) { printf("n %d G %d: the slices end at %d\n", n, G, at); return 1; }
        }
    printf("cnt_at ok (swizzle %d), giant slices ok\n", (int)C3R_CNT_SWZ);
    return 0;
}
