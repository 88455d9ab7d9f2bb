// Issue-rate probe: v_fma_f64 vs v_mad_u64_u32 vs v_mul_lo_u32/v_mul_hi_u32 (whole chip, 8 independent chains per lane).
// build: hipcc -O3 --offload-arch=gfx950 tools/fmabench.hip -o tools/fmabench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE> __global__ void k(double *out, uint64_t *iout, int iters, double x, uint32_t y) {
    double a[8]; uint64_t u[8]; uint32_t w[8];
    for (int i = 0; i < 8; i++) { a[i] = x + i + threadIdx.x; u[i] = y + i + threadIdx.x; w[i] = y * (i + 3) + threadIdx.x; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = __builtin_fma(a[i], x, a[(i + 1) & 7]);
            else if (MODE == 1) u[i] = (uint64_t)(uint32_t)u[i] * y + u[(i + 1) & 7];
            else { w[i] = w[i] * y + __umulhi(w[(i + 1) & 7], y); }
        }
    }
    double s = 0; uint64_t t = 0;
    for (int i = 0; i < 8; i++) { s += a[i]; t += u[i] + w[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s; iout[blockIdx.x * blockDim.x + threadIdx.x] = t;
}
int main() {
    double *d; uint64_t *di; const int blocks = 256 * 16, threads = 256, iters = 20000;
    hipMalloc(&d, blocks * threads * 8); hipMalloc(&di, blocks * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"v_fma_f64", "v_mad_u64_u32", "v_mul_lo+v_mul_hi (2 instr)"};
    for (int m = 0; m < 3; m++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<blocks, threads>>>(d, di, iters, 1.0000001, 12345u);
            else if (m == 1) k<1><<<blocks, threads>>>(d, di, iters, 1.0000001, 12345u);
            else k<2><<<blocks, threads>>>(d, di, iters, 1.0000001, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ops = (double)blocks * threads * iters * 8;
        printf("%-30s %8.2f ms  %8.2f G lane-ops/s  (%.2f T/s)\n", names[m], ms, ops / ms / 1e6, ops / ms / 1e9);
    }
    return 0;
}
