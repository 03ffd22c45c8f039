// lstm_probe_l1rs.hip — timing-only probe of a REGISTER-STATIONARY layer 1: 1024 threads (16 wavefronts, four per SIMD, 128 registers),
// one gate tile per wavefront with all its split-f16 weights (10 k-groups x hi/lo = 80 registers) loaded ONCE, the two 32-site blocks
// of the workgroup processed one after the other on one accumulator.  No weight stream at all; B operands from LDS; the cell
// update's arithmetic and the y1 stores are those of k_lstm1_w8.  Random operands, wrong numbers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_probe_l1rs.hip -o tools/lstm_probe_l1rs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../clair3_rna_amd/csrc/net_kernels.hpp"
using namespace c3r;

template <int SYNC>
__global__ __launch_bounds__(1024) void k_l1rs(const int32_t *__restrict__ xin, const half8 *__restrict__ Wp, _Float16 *__restrict__ y, int n, int nstride) {
    constexpr int H = 128, NGX = 2, NGH = 8, NG = 10, HP = H + 8, XP = 40, CIN = 18, NPC = 9, WG_SITES = 64, HV = 16;
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) _Float16 hb_hi[2][WG_SITES][HP];
    __shared__ __attribute__((aligned(16))) _Float16 hb_lo[2][WG_SITES][HP];
    __shared__ __attribute__((aligned(16))) _Float16 xs[2][WG_SITES][XP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;       // wave = tile of the direction (0..15)
    const int j = lane & 31, hh = lane >> 5;
    const int dir = blockIdx.x, site0 = blockIdx.y * WG_SITES;
    const size_t plane_out = (size_t)nstride * NET_T * 2 * H;
    for (int i = tid; i < 2 * WG_SITES * HP; i += 1024) { (&hb_hi[0][0][0])[i] = (_Float16)0.01f; (&hb_lo[0][0][0])[i] = (_Float16)0.f; }
    for (int i = tid; i < 2 * WG_SITES * XP; i += 1024) (&xs[0][0][0])[i] = (_Float16)1.f;
    // the tile's weights, once: [g][hi|lo]
    half8 wh[NG], wl[NG];
    const half8 *wbase = Wp + ((size_t)(dir * 16 + wave) * NG) * 2 * 64 + lane;
#pragma unroll
    for (int g = 0; g < NG; ++g) { wh[g] = wbase[(g * 2 + 0) * 64]; wl[g] = wbase[(g * 2 + 1) * 64]; }
    float cst[2][4];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int q = 0; q < 4; ++q) cst[sb][q] = 0.f;
    // x staging (as k_lstm1_w8, 1024 threads)
    typedef int int2v __attribute__((ext_vector_type(2)));
    int2v xr = {0, 0};
    const int pc = tid;
    auto x_fetch = [&](int tt_) { if (pc < WG_SITES * NPC) { int sj = site0 + pc / NPC; if (sj >= n) sj = n - 1; xr = *(const int2v *)(xin + ((size_t)sj * NET_T + tt_) * CIN + 2 * (pc % NPC)); } };
    auto x_store = [&](int buf) { if (pc < WG_SITES * NPC) { typedef _Float16 half2v __attribute__((ext_vector_type(2))); half2v v; v[0] = (_Float16)(float)xr[0]; v[1] = (_Float16)(float)xr[1]; *(half2v *)&xs[buf][pc / NPC][2 * (pc % NPC)] = v; } };
    __syncthreads();
    for (int step = 0; step < NET_T; ++step) {
        const int t = dir ? NET_T - 1 - step : step, cur = step & 1, nxt = cur ^ 1;
        if (step + 1 < NET_T) x_fetch(dir ? NET_T - 2 - step : step + 1);
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            floatx16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                half8 bh, bl;
                if (g < NGX) { bh = *(const half8 *)&xs[cur][32 * sb + j][16 * g + 8 * hh]; }
                else { bh = *(const half8 *)&hb_hi[cur][32 * sb + j][16 * (g - NGX) + 8 * hh]; bl = *(const half8 *)&hb_lo[cur][32 * sb + j][16 * (g - NGX) + 8 * hh]; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[g], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[g], bh, acc, 0, 0, 0);
                if (g >= NGX) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[g], bl, acc, 0, 0, 0);
            }
            constexpr float K1 = -1.4426950408889634f * WUNSCALE, K2 = -2.8853900817779268f * WUNSCALE;
            float hval[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float ei = fminf(__builtin_amdgcn_exp2f(K1 * acc[4 * u + 0]), 1e18f), ef = __builtin_amdgcn_exp2f(K1 * acc[4 * u + 1]);
                float eg = fminf(__builtin_amdgcn_exp2f(K2 * acc[4 * u + 2]), 1e18f), eo = fminf(__builtin_amdgcn_exp2f(K1 * acc[4 * u + 3]), 1e18f);
                ei = gate_frac(ei, eg);
                ef = __builtin_amdgcn_rcpf(1.0f + ef);
                const float cq = fmaf(ef, cst[sb][u], ei);
                cst[sb][u] = cq;
                eg = fminf(__builtin_amdgcn_exp2f(-2.8853900817779268f * cq), 1e18f);
                hval[u] = gate_frac(eo, eg);
            }
            half4 vh, vl;
#pragma unroll
            for (int q = 0; q < 4; ++q) { vh[q] = (_Float16)hval[q]; float d = hval[q] - (float)vh[q]; asm volatile("" : "+v"(d)); vl[q] = (_Float16)d; }
            *(half4 *)&hb_hi[nxt][32 * sb + j][8 * wave + 4 * hh] = vh;
            *(half4 *)&hb_lo[nxt][32 * sb + j][8 * wave + 4 * hh] = vl;
            _Float16 *yp = y + ((size_t)t * (2 * HV) + dir * HV + wave) * nstride * 8 + (uint32_t)(site0 + 32 * sb + j) * 8 + 4 * hh;
            *(half4 *)yp = vh;
            *(half4 *)(yp + plane_out) = vl;
        }
        if (step + 1 < NET_T) x_store(nxt);
        __syncthreads();
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 200000;
    const int ns = (n + 127) / 128 * 128;
    const size_t ny = (size_t)ns * 33 * 256 * 2, nw = (size_t)2 * 16 * 10 * 2 * 64;
    int32_t *x; _Float16 *y; half8 *w;
    hipMalloc(&x, (size_t)n * 33 * 18 * 4); hipMalloc(&y, ny * 2); hipMalloc(&w, nw * 16);
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
    hipLaunchKernelGGL(k_l1rs<0>, grid, dim3(1024), 0, 0, x, w, y, n, ns);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_l1rs<0>, grid, dim3(1024), 0, 0, x, w, y, n, ns);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("register-stationary layer 1 (timing probe): %.3f ms per launch (%d sites); k_lstm1_w8 takes 6.5-6.9 ms\n", ms / 3, n);
    return 0;
}
