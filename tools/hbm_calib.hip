// Known byte counts for the HBM counters (MI355X_MICROARCH.md: "WRITE_SIZE is uncalibrated: calibrate on a known byte count in your own
// access pattern"): a 512 MB fill, a 512 MB -> 512 MB copy, both 16 bytes per lane, and a 16-byte-per-lane SCATTERED store of 128 MB
// (every lane writes one 16-byte record at an unrelated address), a 512 MB read as 32-byte records (two 16-byte loads per lane: the
// pile records' pattern) and a gather of lone 8-byte words (a read's bases).
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o tools/hbm_calib;  rocprofv3 --pmc WRITE_SIZE --kernel-trace -- tools/hbm_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void calib_fill(int4 *p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_int4(1, 2, 3, 4); }
__global__ void calib_copy(const int4 *s, int4 *d, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i]; }
__global__ void calib_scatter(int4 *p, size_t n, size_t space) {          // n records of 16 bytes into a space of `space` records, one in four used
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[(i * 2654435761ull) % space] = make_int4((int)i, 2, 3, 4);
}
__global__ void calib_rec32(const int4 *s, int *sink, size_t nrec) {     // one lane per 32-byte record, two 16-byte loads (the pile records' pattern)
    int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) { const int4 a = s[2 * i], b = s[2 * i + 1]; acc += a.x + b.w; }
    if (acc == 12345) *sink = acc;
}
__global__ void calib_gather8(const uint2 *s, int *sink, size_t n, size_t space) {      // lone 8-byte loads at unrelated addresses (a read's bases)
    int acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += s[(i * 2654435761ull) % space].x;
    if (acc == 12345) *sink = acc;
}
int main() {
    const size_t n = (512u << 20) / 16;
    int4 *a, *b;
    if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess) return 1;
    for (int r = 0; r < 3; ++r) {
        calib_fill<<<4096, 256>>>(a, n);
        calib_copy<<<4096, 256>>>(a, b, n);
        calib_scatter<<<4096, 256>>>(b, n / 4, n);
        calib_rec32<<<4096, 256>>>(a, (int *)b, n / 2);
        calib_gather8<<<4096, 256>>>((const uint2 *)a, (int *)b, n / 4, 2 * n);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("fill 512 MB written; copy 512 MB read + 512 MB written; scatter 128 MB written as lone 16-byte records; rec32 512 MB read as 32-byte records; gather8 64 MB read as lone 8-byte words\n");
    return 0;
}
