// atomic_probe.hip — throughput of global (device-scope) atomicAdd on spread addresses, the question behind binning op records by
// position at load time: N threads each add 1 to a counter, returning or not, under three address patterns:
//   unique      thread t -> counter t
//   pairs       thread t -> counter t / 2                      (two records of one read per 32-bp bin)
//   contended   the 16 sixteen-lane groups of a workgroup walk the SAME run of counters (reads of one workgroup overlap)
// and, for scale, plain 32-byte record stores to scattered slots (what the scatter pass writes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/atomic_probe.hip -o tools/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int PAT, bool RET>
__global__ __launch_bounds__(256) void k_atomic(int *cnt, int *out, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    int idx;
    if (PAT == 0) idx = t;
    else if (PAT == 1) idx = t >> 1;
    else idx = (blockIdx.x * 40) + ((threadIdx.x & 15) >> 1) + ((threadIdx.x >> 4) & 15) * 2;
    if (RET) { const int v = atomicAdd(&cnt[idx], 1); if (v == 0x7fffffff) out[0] = v; }
    else __hip_atomic_fetch_add(&cnt[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void k_scatter(int4 *dst, int n, unsigned mul) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    // groups of 4 consecutive records land together (128 B), the groups scattered
    const unsigned g = ((unsigned)(t >> 2) * mul) % (unsigned)(n >> 2);
    int4 *d = dst + 2 * ((size_t)g * 4 + (t & 3));
    d[0] = make_int4(t, 1, 2, 3); d[1] = make_int4(4, 5, 6, t);
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4000000;
    int *cnt, *out; int4 *dst;
    hipMalloc(&cnt, (size_t)n * 4 + 4096); hipMalloc(&out, 64); hipMalloc(&dst, (size_t)n * 32);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto go) {
        hipMemset(cnt, 0, (size_t)n * 4); go(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int r = 0; r < 5; ++r) go(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.3f ms per %d   (%.1f G/s)\n", name, ms / 5, n, n / (ms / 5) * 1e-6);
    };
    const dim3 g((n + 255) / 256), b(256);
    time("unique, no return", [&] { hipLaunchKernelGGL((k_atomic<0, false>), g, b, 0, 0, cnt, out, n); });
    time("unique, returning", [&] { hipLaunchKernelGGL((k_atomic<0, true>), g, b, 0, 0, cnt, out, n); });
    time("pairs, no return", [&] { hipLaunchKernelGGL((k_atomic<1, false>), g, b, 0, 0, cnt, out, n); });
    time("pairs, returning", [&] { hipLaunchKernelGGL((k_atomic<1, true>), g, b, 0, 0, cnt, out, n); });
    time("contended, no return", [&] { hipLaunchKernelGGL((k_atomic<2, false>), g, b, 0, 0, cnt, out, n); });
    time("contended, returning", [&] { hipLaunchKernelGGL((k_atomic<2, true>), g, b, 0, 0, cnt, out, n); });
    time("scatter 32-B records", [&] { hipLaunchKernelGGL(k_scatter, g, b, 0, 0, dst, n, 2654435761u); });
    time("memset of the counters", [&] { hipMemsetAsync(cnt, 0, (size_t)n * 4, 0); });
    return 0;
}
