// lstm_probe_l1rs.hip — k_lstm1_rs (layer 1 with register-stationary weights) on random operands: kernel time.
// (The per-wavefront phase clocks of round 2 needed instrumentation inside the production kernel: removed in round 3, git history 92c9aeb.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_probe_l1rs.hip -o tools/lstm_probe_l1rs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;
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
    dim3 grid(2, (n + 63) / 64);
    auto go = [&] { hipLaunchKernelGGL((k_lstm1_rs<18, false>), grid, dim3(1024), 0, 0, x, w, y, n, ns, (const int32_t *)nullptr); };
    go(); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int r = 0; r < 3; ++r) go(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("k_lstm1_rs: %.3f ms per launch (%d sites)\n", ms / 3, n);
    return 0;
}
