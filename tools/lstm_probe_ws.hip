// lstm_probe_ws.hip — what would ONE CU of a weight-stationary layer-2 cluster cost?  (timing only, random operands, no parity)
//
// The cluster design ("weight-stationary layer 2"): five CUs share a direction's 640 gate rows, four 32-row tiles each,
// one wavefront per SIMD holding its tile's split-f16 weights for the whole launch (26 k-groups x (hi, lo) = 208 registers); site
// groups stream through: per (group of 32 sites, step) a CU takes x_t (32 KB) and h_{t-1} (20.5 KB, all 160 units: its own 32 plus
// the 128 the four other CUs produced) into LDS by LDS-DMA, runs its 4 x 78 MFMAs, updates its 32 units' cells and publishes their
// h_t (2 KB hi + 2 KB lo) for the other four CUs.  Two groups are interleaved so that one group's tile lands while the other computes.
// This probe runs exactly that per-CU work — WITHOUT the cross-CU wait (h comes from a buffer nobody synchronises) and without the
// fused L4 rows — so its time per (group, step) is a LOWER bound on what a member CU of the real cluster would need.
// Break-even against k_lstm2_w8 (one CU: 64 sites x one step in ~38 k clocks at 1.65 GHz = 23 us): five CUs per direction must
// finish a 32-site step in 23 / 64 * 32 / 5 = 2.3 us each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_probe_ws.hip -o tools/lstm_probe_ws && tools/lstm_probe_ws [groups per CU]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;

constexpr int WS_SITES = 32, WS_NGX = 16, WS_NGH = 10, WS_NG = 26, WS_KCX = 32, WS_KCH = 20;

// ABL: 1 = no cell update, 2 = no LDS-DMA (tiles stay as they are), 4 = no h publish
template <int ABL>
__global__ __launch_bounds__(256, 1) void k_lstm2_ws_probe(const _Float16 *__restrict__ xin, const half8 *__restrict__ Wp, _Float16 *__restrict__ hx,
                                                            int n_groups, int ns) {
    // [buffer][plane][k/8][site][8]: x rows then h rows
    __shared__ __attribute__((aligned(16))) _Float16 tile[2][2][WS_KCX + WS_KCH][WS_SITES][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, hh = lane >> 5;
    const int member = blockIdx.x % 5, cluster = blockIdx.x / 5;
    const int my_tile = member * 4 + wave;                       // 0..19 of the direction
    const size_t plane_x = (size_t)ns * NET_T * 256, plane_h = (size_t)ns * 160;
    half8 wh[WS_NG], wl[WS_NG];
    {
        const half8 *wb = Wp + ((size_t)my_tile * WS_NG) * 2 * 64 + lane;
#pragma unroll
        for (int g = 0; g < WS_NG; ++g) { wh[g] = wb[(size_t)(g * 2 + 0) * 64]; wl[g] = wb[(size_t)(g * 2 + 1) * 64]; }
    }
    float cst[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    typedef const _Float16 __attribute__((address_space(1))) *gp_t;
    typedef _Float16 __attribute__((address_space(3))) *lp_t;
    // one instruction = two 512-byte rows (k/8 = 2r, 2r + 1) of one plane: lanes 0-31 the first row's sites, lanes 32-63 the second's
    auto dma = [&](int buf, int site0, int t) {
        if (ABL & 2) return;
        const int site = site0 + j;
#pragma unroll
        for (int q = 0; q < 13; ++q) {
            const int it = wave * 13 + q;                        // 52 instructions: 32 for x (2 planes x 16), 20 for h (2 planes x 10)
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
    // the cluster's site groups, two at a time
    for (int gp = 0; gp < n_groups; gp += 2) {
        const int s0[2] = {(cluster * n_groups + gp) * WS_SITES, (cluster * n_groups + gp + 1) * WS_SITES};
        dma(0, s0[0], 0);
        for (int it = 0; it < 2 * NET_T; ++it) {
            const int gi = it & 1, t = it >> 1, buf = it & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // this step's tile has landed; the other buffer is free
            if (it + 1 < 2 * NET_T) dma(buf ^ 1, s0[gi ^ 1], (it + 1) >> 1);
            floatx16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            half8 bh[2], bl[2];
            auto ldb = [&](int g, half8 &h, half8 &l) {
                h = *(const half8 *)&tile[buf][0][2 * g + hh][j][0];
                l = *(const half8 *)&tile[buf][1][2 * g + hh][j][0];
            };
            ldb(0, bh[0], bl[0]);
            static_for<0, WS_NG>([&](auto gc) {
                constexpr int G = decltype(gc)::value;
                if constexpr (G + 1 < WS_NG) ldb(G + 1, bh[(G + 1) & 1], bl[(G + 1) & 1]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[G], bh[G & 1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[G], bh[G & 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[G], bl[G & 1], acc0, 0, 0, 0);
            });
            constexpr float K1 = -1.4426950408889634f * WUNSCALE, K2 = -2.8853900817779268f * WUNSCALE;
            float hval[4];
            if (ABL & 1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) hval[u] = (acc0[4 * u] + acc1[4 * u + 1]) * 1e-9f;
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ei = fminf(__builtin_amdgcn_exp2f(K1 * (acc0[4 * u + 0] + acc1[4 * u + 0])), 1e18f);
                    const float ef = __builtin_amdgcn_exp2f(K1 * (acc0[4 * u + 1] + acc1[4 * u + 1]));
                    const float eg = fminf(__builtin_amdgcn_exp2f(K2 * (acc0[4 * u + 2] + acc1[4 * u + 2])), 1e18f);
                    const float eo = fminf(__builtin_amdgcn_exp2f(K1 * (acc0[4 * u + 3] + acc1[4 * u + 3])), 1e18f);
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
            (void)t;
        }
        __syncthreads();
    }
}

template <int ABL>
static float run(const _Float16 *x, const half8 *w, _Float16 *hx, int groups, int ns, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_lstm2_ws_probe<ABL>), dim3(255), dim3(256), 0, 0, x, w, hx, groups, ns);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_lstm2_ws_probe<ABL>), dim3(255), dim3(256), 0, 0, x, w, hx, groups, ns);
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
        {"ws member CU, full", run<0>(x, w, hx, groups, ns, 3)},
        {"  no cell update", run<1>(x, w, hx, groups, ns, 3)},
        {"  no LDS-DMA", run<2>(x, w, hx, groups, ns, 3)},
        {"  no h publish", run<4>(x, w, hx, groups, ns, 3)},
        {"ws member CU, full (again)", run<0>(x, w, hx, groups, ns, 3)},
    };
    printf("weight-stationary layer-2 member CU, %d groups of 32 sites per 5-CU cluster (51 clusters), 33 steps each\n", groups);
    for (auto &e : r) {
        const double us_step = e.ms * 1e3 / ((double)groups * 33);
        // a cluster serves 32 sites x one direction per group; both directions and all sites: 2 x n / 32 group-passes over 51 clusters
        printf("%-36s %8.3f ms   %.2f us per (32-site group, step) per CU   -> layer 2 of a 201,945-site chr20 batch on 255 CUs: %.1f ms  (k_lstm2_w8: 19.2)\n",
               e.name, e.ms, us_step, us_step * 33 * (2.0 * 201945 / 32) / 51 * 1e-3);
    }
    return 0;
}
