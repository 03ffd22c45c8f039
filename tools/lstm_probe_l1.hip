// lstm_probe_l1.hip — where the layer-1 kernel (k_lstm1_skew) spends its time: timing-only ablations on random data.
// (Round 1: the int32 operand path measures 1.6 ms here, but reading pre-converted f16 planes instead — one 16-byte load per
// lane, converter 0.18 ms — moved the whole pass by < 0.1 ms in an A/B on one box, so the simpler int32 path stayed.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;
template <int ABL>
static float run(const int32_t *x, const half8 *w, const float *b, _Float16 *y, int n, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((n + 127) / 128, 2);
    const int ns = (n + 127) / 128 * 128;
    auto go = [&] { hipLaunchKernelGGL((k_lstm1_skew<18, ABL>), grid, dim3(256), 0, 0, x, w, b, y, n, ns); };
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
    int32_t *xi; hipMalloc(&xi, (size_t)n * 33 * 18 * 4);
    {
        std::vector<int32_t> h((size_t)n * 33 * 18);
        unsigned long long s = 12345;
        for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % 41) - 20; }
        hipMemcpy(xi, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    const size_t nw = (size_t)2 * 16 * 10 * 2 * 64, nb = 2 * 16 * 32;
    half8 *w; float *b; _Float16 *y;
    hipMalloc(&w, nw * 16); hipMalloc(&b, nb * 4); hipMalloc(&y, (size_t)ns * 33 * 256 * 2 * 2);
    {
        std::vector<_Float16> h(nw * 8);
        unsigned long long s = 777;
        for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (_Float16)(400.f * ((float)(s & 0xffff) / 32768.f - 1.f)); }
        hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemset(b, 0, nb * 4);
    }
    struct { const char *name; float ms; } r[] = {
        {"full (product)", run<0>(xi, w, b, y, n, 3)},
        {"no gate math", run<2>(xi, w, b, y, n, 3)},
        {"no y store", run<4>(xi, w, b, y, n, 3)},
        {"no x loads", run<8>(xi, w, b, y, n, 3)},
        {"weights L1-hot", run<16>(xi, w, b, y, n, 3)},
        {"no x, hot weights", run<24>(xi, w, b, y, n, 3)},
        {"no x, hot w, no y", run<28>(xi, w, b, y, n, 3)},
        {"all of the above", run<30>(xi, w, b, y, n, 3)},
    };
    const double rounds = (double)((n + 127) / 128) * 2 / 256.0;
    const double mfma_ms = 448.0 * 32 * 33 * rounds / 1.8e6;
    for (auto &e : r) printf("%-20s %7.3f ms   (MFMA-bound at 1.8 GHz: %.3f ms = %.0f %%)\n", e.name, e.ms, mfma_ms, 100 * mfma_ms / e.ms);
    return 0;
}
