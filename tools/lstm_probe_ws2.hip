// lstm_probe_ws2.hip — the weight-stationary layer-2 member CU once more (see lstm_probe_ws.hip), now with TWO wavefronts per SIMD:
// the two wavefronts of a SIMD share one gate-row tile and each holds HALF of its K in registers (13 of the 26 k-groups x (hi, lo) =
// 104 weight registers), so that one can issue MFMAs while the other waits for its LDS operands; the second half's partial sums go
// through LDS to the first, which runs the cell update and publishes h_t.  Everything else as in the first probe: x_t and h_{t-1} of a
// 32-site group arrive by LDS-DMA, two groups interleaved over a double-buffered tile, NO cross-CU wait and no fused L4 rows — a
// lower bound on a member CU.  Break-even against k_lstm2_w8: 2.3 us per (32-site group, step) per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_probe_ws2.hip -o tools/lstm_probe_ws2 && tools/lstm_probe_ws2 [groups per cluster]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;

constexpr int WS_SITES = 32, WS_NG = 26, WS_HALF = 13, WS_KCX = 32, WS_KCH = 20;

// ABL: 1 = no cell update, 2 = no LDS-DMA (tiles stay as they are), 4 = no h publish, 8 = no exchange of the partial sums
template <int ABL>
__global__ __launch_bounds__(512, 2) void k_lstm2_ws2_probe(const _Float16 *__restrict__ xin, const half8 *__restrict__ Wp, _Float16 *__restrict__ hx,
                                                             int n_groups, int ns) {
    __shared__ __attribute__((aligned(16))) _Float16 tile[2][2][WS_KCX + WS_KCH][WS_SITES][8];      // [buffer][plane][k/8][site][8]
    __shared__ __attribute__((aligned(16))) float part[4][32][64];                                 // partial sums of the second half, [tile][reg][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hh = lane >> 5;
    const int tl = wave & 3, half = wave >> 2;                   // wavefronts w and w + 4 sit on one SIMD and share tile w
    const int member = blockIdx.x % 5, cluster = blockIdx.x / 5;
    const int my_tile = member * 4 + tl;
    const size_t plane_x = (size_t)ns * NET_T * 256, plane_h = (size_t)ns * 160;
    half8 wh[WS_HALF], wl[WS_HALF];
    {
        const half8 *wb = Wp + ((size_t)my_tile * WS_NG + half * WS_HALF) * 2 * 64 + lane;
#pragma unroll
        for (int g = 0; g < WS_HALF; ++g) { wh[g] = wb[(size_t)(g * 2 + 0) * 64]; wl[g] = wb[(size_t)(g * 2 + 1) * 64]; }
    }
    float cst[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    typedef const _Float16 __attribute__((address_space(1))) *gp_t;
    typedef _Float16 __attribute__((address_space(3))) *lp_t;
    auto dma = [&](int buf, int site0, int t) {
        if (ABL & 2) return;
        const int site = site0 + j;
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const int it = wave * 7 + q;                         // 52 instructions over eight wavefronts: 32 for x, 20 for h
            if (it >= 52) break;
            if (it < 32) {
                const int pl = it >> 4, kc = 2 * (it & 15) + hh;
                const _Float16 *src = xin + (size_t)pl * plane_x + (((size_t)t * WS_KCX + kc) * ns + site) * 8;
                __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)&tile[buf][pl][2 * (it & 15)][0][0], 16, 0, 0);
            } else {
                const int ih = it - 32, pl = ih / 10, kc = 2 * (ih % 10) + hh;
                const _Float16 *src = hx + (size_t)pl * plane_h + ((size_t)kc * ns + site) * 8;
                __builtin_amdgcn_global_load_lds((gp_t)src, (lp_t)&tile[buf][pl][WS_KCX + 2 * (ih % 10)][0][0], 16, 0, 0);
            }
        }
    };
    for (int gp = 0; gp < n_groups; gp += 2) {
        const int s0[2] = {(cluster * n_groups + gp) * WS_SITES, (cluster * n_groups + gp + 1) * WS_SITES};
        dma(0, s0[0], 0);
        for (int it = 0; it < 2 * NET_T; ++it) {
            const int gi = it & 1, buf = it & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // this step's tile has landed; the other buffer and `part` are free
            if (it + 1 < 2 * NET_T) dma(buf ^ 1, s0[gi ^ 1], (it + 1) >> 1);
            floatx16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            half8 bh[2], bl[2];
            const int g0 = half * WS_HALF;
            auto ldb = [&](int g, half8 &h, half8 &l) {
                h = *(const half8 *)&tile[buf][0][2 * (g0 + g) + hh][j][0];
                l = *(const half8 *)&tile[buf][1][2 * (g0 + g) + hh][j][0];
            };
            ldb(0, bh[0], bl[0]);
            static_for<0, WS_HALF>([&](auto gc) {
                constexpr int G = decltype(gc)::value;
                if constexpr (G + 1 < WS_HALF) ldb(G + 1, bh[(G + 1) & 1], bl[(G + 1) & 1]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[G], bh[G & 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[G], bh[G & 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[G], bl[G & 1], acc0, 0, 0, 0);
            });
            // the second half hands its sums over; the first half adds them, updates the cells and publishes h_t
            if (!(ABL & 8)) {
                if (half == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[tl][r][lane] = acc0[r] + acc1[r];
                }
                __syncthreads();
            }
            if (half == 0) {
                float z[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = acc0[r] + acc1[r] + ((ABL & 8) ? 0.f : part[tl][r][lane]);
                constexpr float K1 = -1.4426950408889634f * WUNSCALE, K2 = -2.8853900817779268f * WUNSCALE;
                float hval[4];
                if (ABL & 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) hval[u] = (z[4 * u] + z[4 * u + 1]) * 1e-9f;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float ei = fminf(__builtin_amdgcn_exp2f(K1 * z[4 * u + 0]), 1e18f);
                        const float ef = __builtin_amdgcn_exp2f(K1 * z[4 * u + 1]);
                        const float eg = fminf(__builtin_amdgcn_exp2f(K2 * z[4 * u + 2]), 1e18f);
                        const float eo = fminf(__builtin_amdgcn_exp2f(K1 * z[4 * u + 3]), 1e18f);
                        const float c = fmaf(__builtin_amdgcn_rcpf(1.0f + ef), cst[gi][u], gate_frac(ei, eg));
                        cst[gi][u] = c;
                        hval[u] = gate_frac(eo, fminf(__builtin_amdgcn_exp2f(-2.8853900817779268f * c), 1e18f));
                    }
                }
                if (!(ABL & 4)) {
                    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
                    half4 vh, vl;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { vh[q] = (_Float16)hval[q]; float d = hval[q] - (float)vh[q]; asm volatile("" : "+v"(d)); vl[q] = (_Float16)d; }
                    _Float16 *dst = hx + ((size_t)my_tile * ns + s0[gi] + j) * 8 + 4 * hh;
                    *(half4 *)dst = vh;
                    *(half4 *)(dst + plane_h) = vl;
                }
            }
        }
        __syncthreads();
    }
}

template <int ABL>
static float run(const _Float16 *x, const half8 *w, _Float16 *hx, int groups, int ns, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_lstm2_ws2_probe<ABL>), dim3(255), dim3(512), 0, 0, x, w, hx, groups, ns);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_lstm2_ws2_probe<ABL>), dim3(255), dim3(512), 0, 0, x, w, hx, groups, ns);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int groups = argc > 1 ? atoi(argv[1]) : 64;            // 32-site groups per cluster (even)
    const int n = 51 * groups * WS_SITES, ns = (n + 127) / 128 * 128;
    const size_t nx = (size_t)ns * 33 * 256 * 2, nw = (size_t)20 * 26 * 2 * 64, nh = (size_t)ns * 160 * 2;
    _Float16 *x, *hx; half8 *w;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 16); hipMalloc(&hx, nh * 2);
    auto fill = [](void *d, size_t nhalf, float scale, unsigned seed) {
        std::vector<_Float16> h(nhalf);
        unsigned long long s = seed * 0x9E3779B97F4A7C15ull + 1;
        for (size_t i = 0; i < nhalf; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (_Float16)(scale * ((float)(s & 0xffff) / 32768.f - 1.f)); }
        hipMemcpy(d, h.data(), nhalf * 2, hipMemcpyHostToDevice);
    };
    fill(x, nx, 1.0f, 1); fill(w, nw * 8, 400.f, 2); hipMemset(hx, 0, nh * 2);
    struct { const char *name; float ms; } r[] = {
        {"ws member CU, two half-K wavefronts per SIMD", run<0>(x, w, hx, groups, ns, 3)},
        {"  no cell update", run<1>(x, w, hx, groups, ns, 3)},
        {"  no LDS-DMA", run<2>(x, w, hx, groups, ns, 3)},
        {"  no h publish", run<4>(x, w, hx, groups, ns, 3)},
        {"  no exchange of partial sums", run<8>(x, w, hx, groups, ns, 3)},
        {"  no LDS-DMA, no cell update", run<3>(x, w, hx, groups, ns, 3)},
        {"ws member CU, full (again)", run<0>(x, w, hx, groups, ns, 3)},
    };
    printf("weight-stationary layer-2 member CU with two half-K wavefronts per SIMD, %d groups of 32 sites per 5-CU cluster (51 clusters), 33 steps each\n", groups);
    for (auto &e : r) {
        const double us_step = e.ms * 1e3 / ((double)groups * 33);
        printf("%-46s %8.3f ms   %.2f us per (32-site group, step) per CU   -> layer 2 of a 201,945-site chr20 batch on 255 CUs: %.1f ms  (k_lstm2_w8: 19.2; break-even 2.3 us)\n",
               e.name, e.ms, us_step, us_step * 33 * (2.0 * 201945 / 32) / 51 * 1e-3);
    }
    return 0;
}
