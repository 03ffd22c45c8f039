// Host-side check of K0's counter layout (clair3_rna_amd/csrc/reads_kernels.hpp, cnt_at): compiled by hipcc, runs without a GPU.
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
    printf("cnt_at ok (swizzle %d)\n", (int)C3R_CNT_SWZ);
    return 0;
}
