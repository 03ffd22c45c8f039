// mx_layout_probe.hip — operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) operands, found by experiment
// (the ISA document is not in the image).  One wavefront per experiment; every operand comes from memory (operands or scales built
// with VALU moves right before the instruction gave non-deterministic results: a hazard the compiler does not cover).
// Findings: byte b of lane l of A meets byte b of lane l of B (row / column = l % 32); K = 64; a lane's scale byte (op_sel = byte
// index) covers 32 of its row's K elements — WHICH 32 is answered by mx_scale_probe.hip: bytes 0-15 of lanes r and 32 + r form the
// first scale block (scale from lane r), bytes 16-31 of both the second (scale from lane 32 + r); C/D layout as the f16 32x32 forms.
//   hipcc --offload-arch=gfx950 -O2 tools/mx_layout_probe.hip -o tools/mx_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int OPA, int OPB>
__global__ void k_mx(const intx8 *A, const intx8 *B, const int *SA, const int *SB, float *C) {
    const int lane = threadIdx.x, e = blockIdx.x;
    floatx16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[e * 64 + lane], B[e * 64 + lane], c, 0, 0, OPA, SA[e * 64 + lane], OPB, SB[e * 64 + lane]);
    for (int i = 0; i < 16; ++i) C[(e * 64 + lane) * 16 + i] = c[i];
}
struct Exp { unsigned char a[64][32], b[64][32]; int sa[64], sb[64]; };
static float dec(unsigned char x) { int e = (x >> 3) & 15, m = x & 7; float v = e ? (1.0f + m / 8.0f) * __builtin_exp2f((float)(e - 7)) : (m / 8.0f) * __builtin_exp2f(-6.f); return (x & 128) ? -v : v; }

int main() {
    std::vector<Exp> ex;
    auto blank = [] { Exp x; memset(&x, 0, sizeof x); for (int l = 0; l < 64; ++l) { x.sa[l] = 0x7f7f7f7f; x.sb[l] = 0x7f7f7f7f; } return x; };
    // 0..63: A one-hot 1.0 at row 0 (lane 32*(e/32), byte e%32); B col 0 = distinct values per (lane group, byte)
    for (int e = 0; e < 64; ++e) {
        Exp x = blank();
        x.a[32 * (e / 32)][e % 32] = 0x38;
        for (int g = 0; g < 2; ++g) for (int i = 0; i < 32; ++i) x.b[32 * g][i] = (unsigned char)(0x20 + g * 32 + i);
        ex.push_back(x);
    }
    // 64: all ones on row 0 / col 0, no scaling -> K
    { Exp x = blank(); for (int g = 0; g < 2; ++g) for (int i = 0; i < 32; ++i) { x.a[32 * g][i] = 0x38; x.b[32 * g][i] = 0x38; } ex.push_back(x); }
    // 65..: the same with one scale byte of one lane set to 2^1 (128): operand A/B, lane in {0, 32, 1, 33}, byte 0..3
    for (int op = 0; op < 2; ++op) for (int li = 0; li < 4; ++li) for (int by = 0; by < 4; ++by) {
        Exp x = ex[64];
        const int lane = (li == 0) ? 0 : (li == 1) ? 32 : (li == 2) ? 1 : 33;
        int &s = op ? x.sb[lane] : x.sa[lane];
        s = (s & ~(0xff << (8 * by))) | (0x80 << (8 * by));
        ex.push_back(x);
    }
    // row / column identity: A one-hot at (lane 5, byte 0) = 1.0, B one-hot at (lane 9, byte 0) = 2.0 -> where does C land
    { Exp x = blank(); x.a[5][0] = 0x38; x.b[9][0] = 0x40; ex.push_back(x); }
    const int n = (int)ex.size();
    intx8 *dA, *dB; int *dSA, *dSB; float *dC;
    hipMalloc(&dA, n * 64 * 32); hipMalloc(&dB, n * 64 * 32); hipMalloc(&dSA, n * 64 * 4); hipMalloc(&dSB, n * 64 * 4); hipMalloc(&dC, n * 64 * 16 * 4);
    for (int e = 0; e < n; ++e) {
        hipMemcpy((char *)dA + e * 2048, ex[e].a, 2048, hipMemcpyHostToDevice); hipMemcpy((char *)dB + e * 2048, ex[e].b, 2048, hipMemcpyHostToDevice);
        hipMemcpy(dSA + e * 64, ex[e].sa, 256, hipMemcpyHostToDevice); hipMemcpy(dSB + e * 64, ex[e].sb, 256, hipMemcpyHostToDevice);
    }
    std::vector<float> C(n * 64 * 16);
    hipLaunchKernelGGL((k_mx<0, 0>), dim3(n), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    auto c00 = [&](int e) { return C[(e * 64 + 0) * 16 + 0]; };
    printf("pairing (A one-hot position -> B position it meets; '--' = none):\n");
    for (int e = 0; e < 64; ++e) {
        int m = -1;
        for (int i = 0; i < 64; ++i) if (dec((unsigned char)(0x20 + i)) == c00(e)) m = i;
        if (m >= 0) printf(" A(g%d,b%02d)->B(g%d,b%02d)", e / 32, e % 32, m / 32, m % 32); else printf(" A(g%d,b%02d)->--(%g)    ", e / 32, e % 32, c00(e));
        if (e % 4 == 3) printf("\n");
    }
    printf("all ones, scales 2^0: C[0][0] = %g\n", c00(64));
    int e = 65;
    for (int op = 0; op < 2; ++op) for (int li = 0; li < 4; ++li) { printf("scale byte -> 2^1, operand %c lane %2d:", op ? 'B' : 'A', (li == 0) ? 0 : (li == 1) ? 32 : (li == 2) ? 1 : 33);
        for (int by = 0; by < 4; ++by) printf("  byte%d: %g", by, c00(e++)); printf("\n"); }
    {   // op_sel: which byte of the scale register is used
        std::vector<float> C1(C.size()), C2(C.size());
        hipLaunchKernelGGL((k_mx<1, 0>), dim3(n), dim3(64), 0, 0, dA, dB, dSA, dSB, dC); hipMemcpy(C1.data(), dC, C1.size() * 4, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL((k_mx<0, 3>), dim3(n), dim3(64), 0, 0, dA, dB, dSA, dSB, dC); hipMemcpy(C2.data(), dC, C2.size() * 4, hipMemcpyDeviceToHost);
        printf("op_sel_a = 1, A lane 0 scale bytes 0..3 -> 2^1:"); for (int by = 0; by < 4; ++by) printf(" %g", C1[((65 + by) * 64) * 16]); printf("\n");
        printf("op_sel_b = 3, B lane 0 scale bytes 0..3 -> 2^1:"); for (int by = 0; by < 4; ++by) printf(" %g", C2[((65 + 16 + by) * 64) * 16]); printf("\n");
    }
    printf("A one-hot lane 5, B one-hot lane 9: nonzero C at");
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (C[(e * 64 + l) * 16 + i] != 0.f) printf(" (lane %d, reg %d) = %g", l, i, C[(e * 64 + l) * 16 + i]);
    printf("\n");
    return 0;
}
