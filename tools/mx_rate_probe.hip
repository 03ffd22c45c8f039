// mx_rate_probe.hip — issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8, VGPR scales) against v_mfma_f32_32x32x16_f16: cycles per
// instruction on one SIMD, one wavefront per SIMD, 4 independent accumulators, and a mix (2 f16 + 1 scaled, the precision-2 pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const intx8 *A, const int *S, float *out, long long *cyc, int iters) {
    const int lane = threadIdx.x & 63;
    intx8 a = A[lane], b = A[64 + lane];
    int sa = S[lane], sb = S[64 + lane];
    floatx16 c[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) c[q][i] = 0.f;
    half8 h0, h1;
    __builtin_memcpy(&h0, &a, 16); __builtin_memcpy(&h1, &b, 16);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (MODE == 0) { c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, c[q], 0, 0, 0); }
            if (MODE == 1) { c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[q], 0, 0, 0, sa, 0, sb); }
            if (MODE == 2) { c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, h1, c[q], 0, 0, 0); c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, h0, c[q], 0, 0, 0);
                             c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[q], 0, 0, 0, sa, 0, sb); }
            if (MODE == 3) { c[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[q], 2, 2, 0, sa, 0, sb); }      // fp6
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += c[q][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main() {
    intx8 *dA; int *dS; float *dout; long long *dc;
    hipMalloc(&dA, 192 * 32); hipMalloc(&dS, 512); hipMalloc(&dout, 1024 * 256 * 4); hipMalloc(&dc, 64);
    unsigned char h[192 * 32]; for (int i = 0; i < 192 * 32; ++i) h[i] = (unsigned char)(0x30 + (i * 7) % 0x30);
    int s[128]; for (int i = 0; i < 128; ++i) s[i] = 0x7f7f7f7f;
    hipMemcpy(dA, h, sizeof h, hipMemcpyHostToDevice); hipMemcpy(dS, s, sizeof s, hipMemcpyHostToDevice);
    const int iters = 2000;
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, dA, dS, dout, dc, iters);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(256), 0, 0, dA, dS, dout, dc, iters);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(256), 0, 0, dA, dS, dout, dc, iters);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(256), 0, 0, dA, dS, dout, dc, iters);
    long long c[4]; hipMemcpy(c, dc, 32, hipMemcpyDeviceToHost);
    printf("f16 32x32x16       : %.1f clocks per MFMA\n", (double)c[0] / (iters * 4));
    printf("scaled fp8 32x32x64: %.1f clocks per MFMA\n", (double)c[1] / (iters * 4));
    printf("2 f16 + 1 scaled   : %.1f clocks per triple\n", (double)c[2] / (iters * 4));
    printf("scaled fp6 32x32x64: %.1f clocks per MFMA\n", (double)c[3] / (iters * 4));
    return 0;
}
