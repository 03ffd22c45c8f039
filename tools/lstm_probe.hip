// lstm_probe.hip — timing-only ablation of k_lstm: which phase keeps the matrix pipe idle?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_probe.hip -o tools/lstm_probe && ./tools/lstm_probe [n_sites]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;

template <int SB, int ABL>
static float run(const float *x, const float4 *w, const float *b, float *y, int n, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((n + 32 * SB - 1) / (32 * SB), 2);
    hipLaunchKernelGGL((k_lstm<256, 256, 160, false, SB, ABL>), grid, dim3(256), 0, 0, (const void *)x, w, b, y, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_lstm<256, 256, 160, false, SB, ABL>), grid, dim3(256), 0, 0, (const void *)x, w, b, y, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 15520;
    const size_t nx = (size_t)n * 33 * 256, ny = (size_t)n * 33 * 320, nw = (size_t)2 * 20 * 52 * 64, nb = 2 * 20 * 32;
    float *x, *y, *b; float4 *w;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, nw * 16); hipMalloc(&b, nb * 4);
    std::vector<float> hx(nx), hw(nw * 4), hb(nb, 0.1f);
    for (size_t i = 0; i < nx; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    for (size_t i = 0; i < nw * 4; ++i) hw[i] = ((float)((i * 40503u) % 2000) / 1000.f - 1.f) * 0.05f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 16, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), nb * 4, hipMemcpyHostToDevice);
    const double flop = 2.0 * 416 * 640 * 33 * 2 * n;
    struct { const char *name; float ms; } r[] = {
        {"SB1 full", run<1, 0>(x, w, b, y, n, 5)},
        {"SB1 weights L1-hot", run<1, 1>(x, w, b, y, n, 5)},
        {"SB1 all ablated", run<1, 31>(x, w, b, y, n, 5)},
        {"SB2 full", run<2, 0>(x, w, b, y, n, 5)},
        {"SB2 weights L1-hot", run<2, 1>(x, w, b, y, n, 5)},
        {"SB2 no gate math", run<2, 2>(x, w, b, y, n, 5)},
        {"SB2 no y store", run<2, 4>(x, w, b, y, n, 5)},
        {"SB2 no barrier", run<2, 8>(x, w, b, y, n, 5)},
        {"SB2 const x", run<2, 16>(x, w, b, y, n, 5)},
        {"SB2 all ablated", run<2, 31>(x, w, b, y, n, 5)},
    };
    for (auto &e : r) printf("%-22s %8.3f ms  %7.1f TFLOP/s  %5.1f %% of 157.3\n", e.name, e.ms, flop / e.ms / 1e9, flop / e.ms / 1e9 / 157.3 * 100);
    return 0;
}
