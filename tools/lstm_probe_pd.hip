// lstm_probe_pd.hip — does a deeper weight prefetch ring hide the L2 latency?  Layer-2-shaped k_lstm_h (FC4, x via LDS-DMA) with
// fewer gate-row tiles per wavefront (H = 96: NT = 3, H = 64: NT = 2), which frees registers for prefetch distances PD = 2..5.
// Timing only (random data); reports the fraction of the MFMA-bound time each variant reaches at an assumed clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;
static const half8 *g_w4; static float *g_a4;
template <int H, int PD, int ABL = 0>
static float run(const _Float16 *x, const half8 *w, const float *b, _Float16 *y, int n, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((n + 63) / 64, 2);
    auto go = [&] { hipLaunchKernelGGL((k_lstm_h<256, 256, H, false, 2, ABL, true, PD, true>), grid, dim3(256), 0, 0, (const void *)x, w, b, y, n, g_w4, g_a4, (n + 127) / 128 * 128); };
    go(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) go();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const int ns = (n + 127) / 128 * 128;
    const size_t nx = (size_t)ns * 33 * 256 * 2, nw = (size_t)2 * 20 * 26 * 2 * 64, nb = 2 * 20 * 32;
    _Float16 *x, *y; float *b; half8 *w;
    hipMalloc(&x, nx * 2); hipMalloc(&y, 16); hipMalloc(&w, nw * 16); hipMalloc(&b, nb * 4);
    auto fill = [](void *d, size_t nhalf, float scale, unsigned seed) {
        std::vector<_Float16> h(nhalf);
        unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 1;
        for (size_t i = 0; i < nhalf; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (_Float16)(scale * ((float)(s & 0xffff) / 32768.f - 1.f)); }
        hipMemcpy(d, h.data(), nhalf * 2, hipMemcpyHostToDevice);
    };
    fill(x, nx, 1.0f, 1); fill(w, nw * 8, 400.f, 2); hipMemset(b, 0, nb * 4);
    const size_t nw4 = (size_t)2 * 33 * 4 * 10 * 2 * 64;
    half8 *w4; hipMalloc(&w4, nw4 * 16); fill(w4, nw4 * 8, 100.f, 3); g_w4 = w4;
    hipMalloc(&g_a4, (size_t)n * 2 * 128 * 4);
    struct R { const char *name; int H; float ms; };
    R r[] = {
        {"H160 NT5 PD2 (prod)", 160, run<160, 2>(x, w, b, y, n, 3)},
        {"H160 PD2 lo weights 8-bit", 160, run<160, 2, 128>(x, w, b, y, n, 3)},
        {"H160 PD2 no weight loads", 160, run<160, 2, 16>(x, w, b, y, n, 3)},
        {"H96  NT3 PD2", 96, run<96, 2>(x, w, b, y, n, 3)},
        {"H96  NT3 PD3", 96, run<96, 3>(x, w, b, y, n, 3)},
        {"H96  NT3 PD4", 96, run<96, 4>(x, w, b, y, n, 3)},
        {"H96  NT3 PD5", 96, run<96, 5>(x, w, b, y, n, 3)},
        {"H64  NT2 PD2", 64, run<64, 2>(x, w, b, y, n, 3)},
        {"H64  NT2 PD4", 64, run<64, 4>(x, w, b, y, n, 3)},
        {"H64  NT2 PD6", 64, run<64, 6>(x, w, b, y, n, 3)},
    };
    const double ghz = 1.8;
    for (auto &v : r) {
        const int NT = v.H / 32, NG = (256 + v.H) / 16;
        const double mfma_per_wave_step = NT * 2 * NG * 3 + (v.H / 16) * 2 * 3 * 1.0 /* L4: H/16 k-groups x 2 site blocks x 3, one 32-col block per wave */;
        const double rounds = (double)((n + 63) / 64) * 2 / 256.0;
        const double ideal_ms = mfma_per_wave_step * 32 * 33 * rounds / (ghz * 1e6);
        printf("%-22s %8.3f ms   MFMA-bound %.3f ms at %.1f GHz  -> %.0f %%\n", v.name, v.ms, ideal_ms, ghz, 100 * ideal_ms / v.ms);
    }
    return 0;
}
