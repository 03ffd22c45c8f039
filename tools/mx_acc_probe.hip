// mx_acc_probe.hip — does v_mfma_scale_f32_32x32x64_f8f6f4 keep the f32 accumulator's low bits when the dot product is small next to it?
// Every operand comes from memory (see mx_layout_probe.hip).  Row 0 x col 0: 64 products 1.0 * 1.5 = 96, scaled by 2^sexp, added to C.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const intx8 *A, const intx8 *B, const int *SA, const int *SB, const float *Cin, float *C) {
    const int lane = threadIdx.x, e = blockIdx.x;
    floatx16 c;
    for (int i = 0; i < 16; ++i) c[i] = Cin[(e * 64 + lane) * 16 + i];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[lane], B[lane], c, 0, 0, 0, SA[e * 64 + lane], 0, SB[lane]);
    for (int i = 0; i < 16; ++i) C[(e * 64 + lane) * 16 + i] = c[i];
}
int main() {
    const float hc[8] = {40000.123f, 12345.678f, 1.2345678f, 33333.333f, 65535.99f, 100.001f, 7.7777777f, 50000.5f};
    const int sexps[5] = {0, -3, -6, -9, -12};
    const int n = 40;
    unsigned char a[64][32], b[64][32];
    memset(a, 0, sizeof a); memset(b, 0, sizeof b);
    for (int g = 0; g < 2; ++g) for (int i = 0; i < 32; ++i) { a[32 * g][i] = 0x38; b[32 * g][i] = 0x3c; }
    std::vector<int> sa(n * 64), sb(64, 0x7f7f7f7f);
    std::vector<float> cin((size_t)n * 64 * 16), cout(cin.size());
    for (int e = 0; e < n; ++e) { for (int l = 0; l < 64; ++l) { sa[e * 64 + l] = (127 + sexps[e / 8]) | 0x7f7f7f00; for (int i = 0; i < 16; ++i) cin[(e * 64 + l) * 16 + i] = hc[e % 8]; } }
    intx8 *dA, *dB; int *dSA, *dSB; float *dCi, *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dSA, n * 256); hipMalloc(&dSB, 256); hipMalloc(&dCi, cin.size() * 4); hipMalloc(&dC, cin.size() * 4);
    hipMemcpy(dA, a, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, b, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dSA, sa.data(), n * 256, hipMemcpyHostToDevice); hipMemcpy(dSB, sb.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dCi, cin.data(), cin.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, dA, dB, dSA, dSB, dCi, dC);
    hipMemcpy(cout.data(), dC, cout.size() * 4, hipMemcpyDeviceToHost);
    for (int s = 0; s < 5; ++s) {
        const double add = 96.0 * std::ldexp(1.0, sexps[s]);
        printf("dot = %-10g:", add);
        for (int i = 0; i < 8; ++i) {
            const int e = s * 8 + i;
            const float got = cout[(size_t)e * 64 * 16], expv = (float)((double)hc[i] + add);
            printf("  %.8g (%+.2f ulp)", got, (got - expv) / (std::nextafterf(std::fabs(expv), 1e30f) - std::fabs(expv)));
        }
        printf("\n");
    }
    return 0;
}
