// lds_atomic_probe.hip — what an LDS atomic add costs per wavefront-instruction by address pattern (the question behind walk_records'
// 30 atomics per round): cycles per ds_add_u32 (no return) with 1 .. 8 wavefronts per SIMD issuing them back to back.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_atomic_probe.hip -o tools/lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int N = 18 * 256 + 64;
template <int PAT>
__global__ __launch_bounds__(256) void k(const int *idx, int iters, long long *cycles, int *sink) {
    __shared__ int cnt[N];
    for (int i = threadIdx.x; i < N; i += 256) cnt[i] = 0;
    __syncthreads();
    int a[30];
#pragma unroll
    for (int u = 0; u < 30; ++u) a[u] = idx[(blockIdx.x * 256 + threadIdx.x) * 30 + u];
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 30; ++u) {
            if (PAT == 0) atomicAdd(&cnt[a[u]], 1);
            else if (PAT == 1) cnt[a[u]] += 1;          // plain read-modify-write (wrong under conflicts: timing only)
            else { int v = atomicAdd(&cnt[a[u]], 1); if (v == 0x7fffffff) sink[1] = v; }
        }
    }
    __syncthreads();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (cnt[threadIdx.x] == 0x7fffffff) sink[0] = 1;
}

int main() {
    const int blocks = 256 * 5, iters = 200;
    std::vector<int> h((size_t)blocks * 256 * 30);
    int *d_idx, *sink; long long *d_cyc;
    hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_cyc, blocks * 8); hipMalloc(&sink, 64);
    auto run = [&](const char *name, auto fill, int pat) {
        for (int b = 0; b < blocks; ++b) for (int t = 0; t < 256; ++t) for (int u = 0; u < 30; ++u) h[((size_t)b * 256 + t) * 30 + u] = fill(b, t, u);
        hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (pat == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_idx, iters, d_cyc, sink);
            else if (pat == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_idx, iters, d_cyc, sink);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d_idx, iters, d_cyc, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        // 5 workgroups per CU x 4 wavefronts x 30 x iters wave-instructions per CU
        const double instr_per_cu = 5.0 * 4 * 30 * iters;
        printf("%-58s %8.3f ms   %6.1f ns per wave-instruction per CU  (~%.1f cycles at 2.1 GHz)\n", name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
    };
    srand(1);
    run("atomic, distinct banks (lane l -> word l)", [](int, int t, int u) { return (t & 63) + 64 * (u % 4); }, 0);
    run("atomic, stride 18 (pos = lane, one channel)", [](int, int t, int u) { return 18 * ((t & 63) + u) + 1; }, 0);
    run("atomic, walk-like: pos = rand32 + u, ch = rand4, stride 18", [](int, int t, int u) { static int r[256], c[256 * 30]; if (u == 0) r[t] = rand() % 32; return 18 * (r[t] + u) + (rand() % 4) + (t & 1 ? 9 : 0); }, 0);
    run("atomic, walk-like, channel-major [ch][pos]", [](int, int t, int u) { static int r[256]; if (u == 0) r[t] = rand() % 32; return 256 * ((rand() % 4) + (t & 1 ? 9 : 0)) + r[t] + u; }, 0);
    run("atomic, walk-like, stride 19", [](int, int t, int u) { static int r[256]; if (u == 0) r[t] = rand() % 32; return 19 * (r[t] + u) + (rand() % 4) + (t & 1 ? 9 : 0); }, 0);
    run("atomic, all lanes one address", [](int, int, int u) { return u; }, 0);
    run("atomic, 8 lanes per address, distinct banks", [](int, int t, int u) { return ((t & 63) >> 3) + 8 * (u % 8); }, 0);
    run("atomic, 2 lanes per address, distinct banks", [](int, int t, int u) { return ((t & 63) >> 1) + 32 * (u % 8); }, 0);
    run("plain rmw, distinct banks", [](int, int t, int u) { return (t & 63) + 64 * (u % 4); }, 1);
    run("plain rmw, walk-like stride 18", [](int, int t, int u) { static int r[256]; if (u == 0) r[t] = rand() % 32; return 18 * (r[t] + u) + (rand() % 4) + (t & 1 ? 9 : 0); }, 1);
    run("returning atomic, distinct banks", [](int, int t, int u) { return (t & 63) + 64 * (u % 4); }, 2);
    run("returning atomic, walk-like stride 18", [](int, int t, int u) { static int r[256]; if (u == 0) r[t] = rand() % 32; return 18 * (r[t] + u) + (rand() % 4) + (t & 1 ? 9 : 0); }, 2);
    return 0;
}
