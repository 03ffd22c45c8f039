// mx_scale_probe.hip — WHICH 32 of a row's 64 K elements does a lane's E8M0 scale byte cover (v_mfma_scale_f32_32x32x64_f8f6f4, fp8)?
// A = all 1.0; B of column 0 = 1, 2, 4, 8 on the four quarters (lane group g, byte half h) = (0,0) (0,1) (1,0) (1,1); one scale -> 2^1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const intx8 *A, const intx8 *B, const int *SA, const int *SB, float *C) {
    const int lane = threadIdx.x, e = blockIdx.x;
    floatx16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[lane], B[lane], c, 0, 0, 0, SA[e * 64 + lane], 0, SB[e * 64 + lane]);
    if (lane == 0) C[e] = c[0];
}
int main() {
    unsigned char a[64][32], b[64][32];
    memset(a, 0, sizeof a); memset(b, 0, sizeof b);
    const unsigned char val[4] = {0x38, 0x40, 0x48, 0x50};      // 1, 2, 4, 8
    for (int g = 0; g < 2; ++g) for (int i = 0; i < 32; ++i) { a[32 * g][i] = 0x38; b[32 * g][i] = val[2 * g + i / 16]; }
    int sa[5][64], sb[5][64];
    for (int e = 0; e < 5; ++e) for (int l = 0; l < 64; ++l) { sa[e][l] = 0x7f7f7f7f; sb[e][l] = 0x7f7f7f7f; }
    sa[1][0] = 0x7f7f7f80; sa[2][32] = 0x7f7f7f80; sb[3][0] = 0x7f7f7f80; sb[4][32] = 0x7f7f7f80;
    intx8 *dA, *dB; int *dSA, *dSB; float *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dSA, sizeof sa); hipMalloc(&dSB, sizeof sb); hipMalloc(&dC, 64);
    hipMemcpy(dA, a, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, b, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dSA, sa, sizeof sa, hipMemcpyHostToDevice); hipMemcpy(dSB, sb, sizeof sb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(5), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    float c[5]; hipMemcpy(c, dC, 20, hipMemcpyDeviceToHost);
    const char *nm[5] = {"no scaling", "A lane 0", "A lane 32", "B lane 0", "B lane 32"};
    for (int e = 0; e < 5; ++e) {
        printf("%-10s: C[0][0] = %g", nm[e], c[e]);
        if (e) { const int d = (int)((c[e] - c[0]) / 16 + 0.5f); printf("  -> doubled quarters (g,h):%s%s%s%s", (d & 1) ? " (0,0)" : "", (d & 2) ? " (0,1)" : "", (d & 4) ? " (1,0)" : "", (d & 8) ? " (1,1)" : ""); }
        printf("\n");
    }
    return 0;
}
