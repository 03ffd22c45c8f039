// lstm_probe_w8.hip — timing-only ablation of k_lstm2_w8 (layer 2 + fused L4, two wavefronts per SIMD) on random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_probe_w8.hip -o tools/lstm_probe_w8
//   ABL bits: 1 = every weight load hits one L1-hot k-group (L1 -> register traffic kept, L2 -> L1 traffic gone), 2 = no gate math (cell update replaced by 3 adds), 16 = weights loaded for the first k-group only
//   (register-stationary afterwards: no L2 -> L1 weight stream), 64 = no x DMA after the first step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;

static const half8 *g_w4 = nullptr;
static float *g_a4 = nullptr;
template <int ABL>
static float run(const _Float16 *x, const half8 *w, const float *b, int n, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid = C3R_DIR_ILV ? dim3(2, (n + 63) / 64) : dim3((n + 63) / 64, 2);      // (k_lstm2_w8 reads the direction from blockIdx.x then)
    const int ns = (n + 127) / 128 * 128;
    hipLaunchKernelGGL((k_lstm2_w8<ABL>), grid, dim3(512), 0, 0, x, w, b, n, g_w4, g_a4, ns);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_lstm2_w8<ABL>), grid, dim3(512), 0, 0, x, w, b, n, g_w4, g_a4, ns);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const int ns = (n + 127) / 128 * 128;
    const size_t nx = (size_t)ns * 33 * 256 * 2, nw = (size_t)2 * 20 * 26 * 2 * 64, nb = 2 * 20 * 32;
    _Float16 *x; float *b; half8 *w;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 16); hipMalloc(&b, nb * 4);
    const bool fp8_safe = argc > 2 && atoi(argv[2]) != 0;      // every operand byte a finite fp8 number (bit 6 clear): for the MX-pipe probes
    auto fill = [fp8_safe](void *d, size_t nhalf, float scale, unsigned seed) {
        std::vector<_Float16> h(nhalf);
        unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 1;
        for (size_t i = 0; i < nhalf; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            h[i] = (_Float16)(scale * ((float)(s & 0xffff) / 32768.f - 1.f));
        }
        if (fp8_safe) { unsigned char *q = (unsigned char *)h.data(); for (size_t i = 0; i < nhalf * 2; ++i) q[i] &= 0xBF; }
        hipMemcpy(d, h.data(), nhalf * 2, hipMemcpyHostToDevice);
    };
    fill(x, nx, 1.0f, 1); fill(w, nw * 8, 400.f, 2);
    hipMemset(b, 0, nb * 4);
    const size_t nw4 = (size_t)2 * 33 * 4 * 10 * 2 * 64;
    half8 *w4; hipMalloc(&w4, nw4 * 16); fill(w4, nw4 * 8, 100.f, 3); g_w4 = w4;
    hipMalloc(&g_a4, (size_t)n * 2 * 128 * 4);
    const double flop = 2.0 * (416 * 640 + 160 * 128) * 33 * 2 * (double)n;
    struct { const char *name; float ms; } r[] = {
        {"w8 full", run<0>(x, w, b, n, 3)},
        {"w8 no gate math", run<2>(x, w, b, n, 3)},
        {"w8 no weight stream", run<16>(x, w, b, n, 3)},
        {"w8 no gate, no weights", run<18>(x, w, b, n, 3)},
        {"w8 weights L1-hot", run<1>(x, w, b, n, 3)},
        {"w8 no LDS operand reads", run<32>(x, w, b, n, 3)},
        {"w8 no x DMA", run<64>(x, w, b, n, 3)},
        {"w8 full (again)", run<0>(x, w, b, n, 3)},
        {"w8 full (3rd)", run<0>(x, w, b, n, 3)},
    };
    for (auto &e : r) printf("%-26s %8.3f ms  %7.1f algorithmic TFLOP/s (x3 executed = %6.1f = %4.1f %% of 2500)\n", e.name, e.ms, flop / e.ms / 1e9,
                             3 * flop / e.ms / 1e9, 3 * flop / e.ms / 1e9 / 2500 * 100);
    return 0;
}
