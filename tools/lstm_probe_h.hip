// lstm_probe_h.hip — timing-only ablation of the split-f16 LSTM kernel k_lstm_h (layer-2 shape).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;

template <int SB, int ABL>
static float run(const _Float16 *x, const half8 *w, const float *b, _Float16 *y, int n, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((n + 32 * SB - 1) / (32 * SB), 2);
    hipLaunchKernelGGL((k_lstm_h<256, 256, 160, false, SB, ABL>), grid, dim3(256), 0, 0, (const void *)x, w, b, y, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_lstm_h<256, 256, 160, false, SB, ABL>), grid, dim3(256), 0, 0, (const void *)x, w, b, y, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const size_t nx = (size_t)n * 33 * 256 * 2, ny = (size_t)n * 33 * 320 * 2, nw = (size_t)2 * 20 * 26 * 2 * 64, nb = 2 * 20 * 32;
    _Float16 *x, *y; float *b; half8 *w;
    hipMalloc(&x, nx * 2); hipMalloc(&y, ny * 2); hipMalloc(&w, nw * 16); hipMalloc(&b, nb * 4);
    hipMemset(x, 0x2c, nx * 2); hipMemset(w, 0x21, nw * 16); hipMemset(b, 0, nb * 4);
    const double flop = 2.0 * 416 * 640 * 33 * 2 * n;
    struct { const char *name; float ms; } r[] = {
        {"SB2 full", run<2, 0>(x, w, b, y, n, 3)},
        {"SB2 weights L1-hot", run<2, 1>(x, w, b, y, n, 3)},
        {"SB2 no gate math", run<2, 2>(x, w, b, y, n, 3)},
        {"SB2 no y store", run<2, 4>(x, w, b, y, n, 3)},
        {"SB2 no barrier", run<2, 8>(x, w, b, y, n, 3)},
        {"SB2 1+2+4+8", run<2, 15>(x, w, b, y, n, 3)},
    };
    for (auto &e : r) printf("%-22s %8.3f ms  %7.1f algorithmic TFLOP/s (x3 executed = %6.1f = %4.1f %% of 2500)\n", e.name, e.ms, flop / e.ms / 1e9,
                             3 * flop / e.ms / 1e9, 3 * flop / e.ms / 1e9 / 2500 * 100);
    return 0;
}
