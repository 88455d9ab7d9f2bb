// Where does the time of the quad sum-check evaluation kernel go?  Variants: 0 loads+xor, 1 +field arithmetic, 2 +wave/block reduction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "field.h"
using namespace otti;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ Fr shfl_xor_fr(const Fr &x, int mask) { Fr r; for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_xor((int)x.v[i], mask, 64); return r; }
template <int MODE> __global__ __launch_bounds__(256) void k(const Fr *A, const Fr *B, size_t half, Fr *out) {
    Fr acc0 = fr_zero(), acc1 = fr_zero();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        Fr a0 = A[i], a1 = A[i + half], b0 = B[i], b1 = B[i + half];
        if (MODE == 0) { for (int q = 0; q < 8; q++) { acc0.v[q] ^= a0.v[q] ^ b0.v[q]; acc1.v[q] ^= a1.v[q] ^ b1.v[q]; } }
        else {
            acc0 = fr_add(acc0, fr_mul(a0, b0));
            Fr a2 = fr_sub(fr_add(a1, a1), a0), b2 = fr_sub(fr_add(b1, b1), b0);
            acc1 = fr_add(acc1, fr_mul(a2, b2));
        }
    }
    if (MODE >= 2) {
        for (int off = 32; off >= 1; off >>= 1) { acc0 = fr_add(acc0, shfl_xor_fr(acc0, off)); acc1 = fr_add(acc1, shfl_xor_fr(acc1, off)); }
        if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = acc0; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = acc1; }
    } else { out[(blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2] = acc0; out[(blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 2 + 1] = acc1; }
}
template <int MODE> int run(const char *name, Fr *A, Fr *B, size_t half, Fr *out, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, blocks, 256, 0, 0, A, B, half, out); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; r++) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, blocks, 256, 0, 0, A, B, half, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-34s blocks=%5d : %8.1f us  %7.1f GB/s\n", name, blocks, best * 1e3, 4.0 * half * 32 / best / 1e6);
    return 0;
}
int main() {
    size_t n = (size_t)1 << 22, half = n / 2;
    std::vector<Fr> h(n); for (size_t i = 0; i < n; i++) for (int q = 0; q < 8; q++) h[i].v[q] = (uint32_t)(0x9e3779b9u * (i * 8 + q + 1)) & (q == 7 ? 0x0fffffffu : 0xffffffffu);
    Fr *A, *B, *out; CK(hipMalloc((void **)&A, n * 32)); CK(hipMalloc((void **)&B, n * 32)); CK(hipMalloc((void **)&out, (size_t)8192 * 256 * 64));
    CK(hipMemcpy(A, h.data(), n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), n * 32, hipMemcpyHostToDevice));
    for (int blocks : {512, 1024, 2048, 8192}) {
        run<0>("loads + xor", A, B, half, out, blocks);
        run<1>("loads + field arithmetic", A, B, half, out, blocks);
        run<2>("loads + arithmetic + wave reduce", A, B, half, out, blocks);
    }
    return 0;
}
