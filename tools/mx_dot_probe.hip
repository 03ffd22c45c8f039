// mx_dot_probe.hip — the block-scaled fp8 MFMA against a float64 evaluation of the same bytes and scales, on operands shaped like the
// kernel's (lanes 0-31: fp8(w) x fp8(x_lo 2^18), lanes 32-63: fp8(w_lo) x fp8(x 2^6); per-lane E8M0 scales), row r x column c.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <random>
#include <vector>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(const intx8 *A, const intx8 *B, const int *SA, const int *SB, const float *Cin, float *C) {
    const int lane = threadIdx.x;
    floatx16 c;
    for (int i = 0; i < 16; ++i) c[i] = Cin[lane * 16 + i];
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[lane], B[lane], c, 0, 0, 0, SA[lane], 0, SB[lane]);
    for (int i = 0; i < 16; ++i) C[lane * 16 + i] = c[i];
}
static double dec(unsigned char x) { int e = (x >> 3) & 15, m = x & 7; double v = e ? (1.0 + m / 8.0) * std::ldexp(1.0, e - 7) : (m / 8.0) * std::ldexp(1.0, -6); return (x & 128) ? -v : v; }
int main() {
    std::mt19937 rng(5);
    unsigned char a[64][32], b[64][32]; int sa[64], sb[64]; std::vector<float> cin(64 * 16), cout(64 * 16);
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 32; ++i) { a[l][i] = (unsigned char)((rng() % 0x70) | ((rng() & 1) << 7)); b[l][i] = (unsigned char)((rng() % 0x70) | ((rng() & 1) << 7)); }
        sa[l] = (120 + (int)(rng() % 12)) | 0x7f7f7f00;
        sb[l] = (l >= 32 ? 121 : 109) | 0x7f7f7f00;
    }
    for (auto &v : cin) v = 0.f;
    intx8 *dA, *dB; int *dSA, *dSB; float *dCi, *dC;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dSA, 256); hipMalloc(&dSB, 256); hipMalloc(&dCi, 4096); hipMalloc(&dC, 4096);
    hipMemcpy(dA, a, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, b, 2048, hipMemcpyHostToDevice); hipMemcpy(dSA, sa, 256, hipMemcpyHostToDevice);
    hipMemcpy(dSB, sb, 256, hipMemcpyHostToDevice); hipMemcpy(dCi, cin.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dCi, dC);
    hipMemcpy(cout.data(), dC, 4096, hipMemcpyDeviceToHost);
    double worst = 0, worst_rel = 0;
    for (int col = 0; col < 32; ++col) for (int row = 0; row < 32; ++row) {
        double dot = 0;
        for (int g = 0; g < 2; ++g) {
            const double s = std::ldexp(1.0, (sa[32 * g + row] & 255) - 127) * std::ldexp(1.0, (sb[32 * g + col] & 255) - 127);
            for (int i = 0; i < 32; ++i) dot += dec(a[32 * g + row][i]) * dec(b[32 * g + col][i]) * s;
        }
        const int lane = col + 32 * ((row >> 2) & 1), reg = (row & 3) + 4 * (row >> 3);
        const double got = cout[lane * 16 + reg], exp = (double)cin[lane * 16 + reg] + dot;
        worst = std::fmax(worst, std::fabs(got - exp));
        worst_rel = std::fmax(worst_rel, std::fabs(got - exp) / (std::fabs(dot) + 1e-30));
        if (row < 2 && col < 3) {
            double D[2] = {0, 0};
            for (int g = 0; g < 2; ++g) for (int i = 0; i < 32; ++i) D[g] += dec(a[32 * g + row][i]) * dec(b[32 * g + col][i]);
            printf("row %d col %d: D0 %.6g D1 %.6g | sa(row) %d %d  sb(col) %d %d | hardware delta %.6g\n", row, col, D[0], D[1], sa[row] & 255, sa[32 + row] & 255,
                   sb[col] & 255, sb[32 + col] & 255, (double)cout[lane * 16 + reg] - (double)cin[lane * 16 + reg]);
        }
        if (row == 0 && col < 4) printf("C[0][%d]: c_in %.4f dot %.6g -> got %.6f expected %.6f\n", col, cin[lane * 16 + reg], dot, got, exp);
    }
    printf("worst |got - expected| = %.3g, worst relative to the dot product = %.3g\n", worst, worst_rel);
    return 0;
}
