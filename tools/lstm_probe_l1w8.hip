// lstm_probe_l1w8.hip — timing-only ablation of k_lstm1_w8 (layer 1, two wavefronts per SIMD) on random operands.
//   ABL bits: 2 = no gate math, 4 = no y1 store, 16 = weights from one L1-hot k-group
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;

template <int ABL, int TEAMS = 1>
static float run(const int32_t *x, const half8 *w, _Float16 *y, int n, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned ng = (n + 64 * TEAMS - 1) / (64 * TEAMS);
    dim3 grid = C3R_DIR_ILV ? dim3(2, ng) : dim3(ng, 2);
    const int ns = (n + 127) / 128 * 128;
    hipLaunchKernelGGL((k_lstm1_w8<18, ABL, TEAMS>), grid, dim3(512 * TEAMS), 0, 0, x, w, y, n, ns);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_lstm1_w8<18, ABL, TEAMS>), grid, dim3(512 * TEAMS), 0, 0, x, w, y, n, ns);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const int ns = (n + 127) / 128 * 128;
    const size_t ny = (size_t)ns * 33 * 256 * 2, nw = (size_t)2 * 16 * 10 * 2 * 64;
    int32_t *x; _Float16 *y; half8 *w;
    hipMalloc(&x, (size_t)n * 33 * 18 * 4); hipMalloc(&y, ny * 2 + 4096); hipMalloc(&w, nw * 16);
    {
        std::vector<int32_t> h((size_t)n * 33 * 18);
        unsigned long long s = 12345;
        for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % 41) - 20; }
        hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        std::vector<_Float16> hw(nw * 8);
        for (auto &v : hw) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (_Float16)(40.f * ((float)(s & 0xffff) / 32768.f - 1.f)); }
        hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    }
    const double flop = 2.0 * (32 * 2 + 128 * 3) / 3.0 * 512 * 33 * 2 * (double)n;   // executed-equivalent / 3
    struct { const char *name; float ms; } r[] = {
        {"l1w8 full", run<0>(x, w, y, n, 3)},
        {"l1w8 no gate math", run<2>(x, w, y, n, 3)},
        {"l1w8 no y1 store", run<4>(x, w, y, n, 3)},
        {"l1w8 weights L1-hot", run<16>(x, w, y, n, 3)},
        {"l1w8 no gate, no store", run<6>(x, w, y, n, 3)},
        {"l1w8 all three", run<22>(x, w, y, n, 3)},
        {"l1w8 full (again)", run<0>(x, w, y, n, 3)},
        {"l1w8 two teams", run<0, 2>(x, w, y, n, 3)},
        {"l1w8 two teams no gate", run<2, 2>(x, w, y, n, 3)},
        {"l1w8 two teams no store", run<4, 2>(x, w, y, n, 3)},
        {"l1w8 two teams (again)", run<0, 2>(x, w, y, n, 3)},
    };
    for (auto &e : r) printf("%-26s %8.3f ms  executed %6.1f TFLOP/s\n", e.name, e.ms, 3 * flop / e.ms / 1e9);
#ifdef C3R_L1_TIMING
    {   // where one workgroup's eight wavefronts spend a step (the last launch was the full kernel, TEAMS = 1 ... re-run it)
        (void)run<0>(x, w, y, n, 1);
        long long t[48];
        hipMemcpy(t, (char *)y + ny * 2, sizeof t, hipMemcpyDeviceToHost);
        const char *nm[6] = {"top (x fetch)", "x part", "h part", "cell + y1", "x store", "barrier"};
        printf("clocks per step (workgroup (dir 0, group 7)), by wavefront:\n%-14s", "");
        for (int wv = 0; wv < 8; ++wv) printf("  wave %d", wv);
        printf("\n");
        for (int ph = 0; ph < 6; ++ph) { printf("%-14s", nm[ph]); for (int wv = 0; wv < 8; ++wv) printf(" %7lld", t[wv * 6 + ph] / 33); printf("\n"); }
        printf("%-14s", "total"); for (int wv = 0; wv < 8; ++wv) { long long sm = 0; for (int ph = 0; ph < 6; ++ph) sm += t[wv * 6 + ph]; printf(" %7lld", sm / 33); } printf("\n");
    }
#endif
    return 0;
}
