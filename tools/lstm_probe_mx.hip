// lstm_probe_mx.hip — k_lstm2_mx (precision 2, layer 2 + fused L4) on random operands: kernel time.
// (The per-wavefront phase clocks of round 2 needed instrumentation inside the production kernel: removed in round 3, git history 92c9aeb.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_probe_mx.hip -o tools/lstm_probe_mx
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const int ns = (n + 127) / 128 * 128;
    const size_t nx = (size_t)ns * 33 * 256 * 2 * 2, nw = (size_t)2 * 20 * 26 * 2 * 64 * 16, nq = (size_t)2 * 4 * 13 * 5 * 64 * 32, nsc = (size_t)2 * 4 * 4 * 5 * 64 * 4;
    const size_t nw4 = (size_t)2 * 33 * 4 * 10 * 2 * 64 * 16, nq4 = (size_t)2 * 33 * 4 * 5 * 64 * 32, ns4 = (size_t)2 * 33 * 4 * 2 * 64 * 4;
    void *x, *w, *q, *sc, *w4, *q4, *s4; float *b, *a4;
    hipMalloc(&x, nx); hipMalloc(&w, nw); hipMalloc(&q, nq); hipMalloc(&sc, nsc); hipMalloc(&w4, nw4); hipMalloc(&q4, nq4); hipMalloc(&s4, ns4);
    hipMalloc(&b, 2 * 20 * 32 * 4); hipMalloc(&a4, (size_t)n * 2 * 128 * 4 + 4096);
    auto fill = [](void *d, size_t bytes, unsigned char mask, unsigned seed) {
        std::vector<unsigned char> h(bytes);
        unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 1;
        for (size_t i = 0; i < bytes; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (unsigned char)(s & mask); }
        hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    };
    fill(x, nx, 0x3f, 1); fill(w, nw, 0x3f, 2); fill(q, nq, 0x3f, 3); fill(w4, nw4, 0x3f, 4); fill(q4, nq4, 0x3f, 5);
    hipMemset(sc, 0x7f, nsc); hipMemset(s4, 0x7f, ns4); hipMemset(b, 0, 2 * 20 * 32 * 4); hipMemset(a4, 0, (size_t)n * 2 * 128 * 4 + 4096);
    dim3 grid(2, (n + 63) / 64);
    auto go = [&] { hipLaunchKernelGGL(k_lstm2_mx, grid, dim3(512), 0, 0, (const _Float16 *)x, (const half8 *)w, (const uint32_t *)q, (const uint32_t *)sc, (const float *)b, n,
                                       (const half8 *)w4, (const uint32_t *)q4, (const uint32_t *)s4, a4, ns); };
    go(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int r = 0; r < 3; ++r) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("k_lstm2_mx: %.3f ms per launch (%d sites)\n", ms / 3, n);
    return 0;
}
