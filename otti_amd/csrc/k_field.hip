// Element-wise field kernels, conversions, strided gather.
#include "kernels_common.h"

namespace otti {

// ------------------------------------------------------------------------------------------------ element-wise
__global__ void k_fr_op(int op, const Fr *a, const Fr *b, Fr *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr x = a[i], y = b[i];
        out[i] = op == 0 ? fr_mul(x, y) : op == 1 ? fr_add(x, y) : fr_sub(x, y);
    }
}
__global__ void k_fr_scale(const Fr *in, Fr k, Fr *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = fr_mul(in[i], k);
}
__global__ void k_fr_fill(Fr *p, Fr v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
void dev_fr_op(DevCtx &c, int op, const Fr *a, const Fr *b, Fr *out, size_t n) { if (n) hipLaunchKernelGGL(k_fr_op, grid_for(n), kBlock, 0, c.stream, op, a, b, out, n); }
void dev_scale(DevCtx &c, const Fr *in, const Fr &k, Fr *out, size_t n) { if (n) hipLaunchKernelGGL(k_fr_scale, grid_for(n), kBlock, 0, c.stream, in, k, out, n); }
void dev_from_canonical(DevCtx &c, const Fr *in, Fr *out, size_t n) { dev_scale(c, in, fr_R2(), out, n); }
void dev_to_canonical(DevCtx &c, const Fr *in, Fr *out, size_t n) { Fr one = fr_zero(); one.v[0] = 1; dev_scale(c, in, one, out, n); }
void dev_fill_zero(DevCtx &c, Fr *p, size_t n) { if (n) OTTI_HIP(hipMemsetAsync(p, 0, n * sizeof(Fr), c.stream)); }
void dev_fill_one(DevCtx &c, Fr *p, size_t n) { if (n) hipLaunchKernelGGL(k_fr_fill, grid_for(n), kBlock, 0, c.stream, p, fr_one(), n); }
void dev_fetch(DevCtx &c, const Fr *src, int slot, size_t n) { OTTI_HIP(hipMemcpyAsync(c.h_results + slot, src, n * sizeof(Fr), hipMemcpyDeviceToHost, c.stream)); }

// VarsAssignment::new on the device: canonical little-endian scalars (the caller's bytes, uploaded as they are) -> Montgomery form in
// place; values >= l are counted (upstream: R1CSError::InvalidScalar) and left as zero.
// counts[0]: non-canonical scalars; counts[1]: scalars below 2^128 (the share of small values picks the commitment's MSM variant)
__global__ __launch_bounds__(kBlock) void k_witness_ingest(Fr *z, size_t n, unsigned long long *counts) {
    unsigned bad = 0, small = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr raw = z[i];
        if (fr_raw_is_canonical(raw.v)) { z[i] = fr_mul(raw, fr_R2()); small += (raw.v[4] | raw.v[5] | raw.v[6] | raw.v[7]) == 0 ? 1u : 0u; }
        else { z[i] = fr_zero(); bad++; small++; }
    }
    for (int o = 32; o >= 1; o >>= 1) { bad += __shfl_down(bad, o); small += __shfl_down(small, o); }
    if ((threadIdx.x & 63) == 0) { if (bad) atomicAdd(&counts[0], (unsigned long long)bad); if (small) atomicAdd(&counts[1], (unsigned long long)small); }
}
size_t dev_witness_ingest(DevCtx &c, Fr *z, size_t n, size_t *n_small) {
    if (n_small) *n_small = 0;
    if (!n) return 0;
    OTTI_HIP(hipMemsetAsync(c.d_counts.p, 0, 2 * sizeof(unsigned long long), c.stream));
    hipLaunchKernelGGL(k_witness_ingest, grid_for(n), kBlock, 0, c.stream, z, n, c.d_counts.p);
    unsigned long long h[2] = {0, 0};
    OTTI_HIP(hipMemcpyAsync(h, c.d_counts.p, sizeof h, hipMemcpyDeviceToHost, c.stream));
    OTTI_HIP(hipStreamSynchronize(c.stream));
    if (n_small) *n_small = (size_t)h[1];
    return (size_t)h[0];
}
// Whole-chip throughput of the Montgomery product in GF(l) (operands in registers, every CU busy): what the sum-check, sparse-product
// and eq kernels are priced against beside the HBM roof — at 7-13 products per 192 bytes they are bounded by the multiplier first.
__global__ __launch_bounds__(kBlock) void k_fr_mul_peak(Fr *io, int iters) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    // the form the streaming kernels use since round 4: nine 29-bit limbs kept unpacked (fr9.h), two chains per lane; x <- x * y in the memory
    // format is mont261(x, 32 y) (every product below 1.1 l: no normalisation between them)
    const Fr9 y5 = fr9_unpack5(io[2 * i + 1]); Fr9 x = fr9_unpack(io[2 * i]), z = fr9_unpack(io[2 * i + 1]);
    for (int k = 0; k < iters; k++) { x = fr9_mul(x, y5); z = fr9_mul(z, y5); }
    io[2 * i] = fr_add(fr9_pack_lt2l(x), fr9_pack_lt2l(z));
}
double dev_fr_mul_peak(DevCtx &c) {
    const int blocks = 2048, iters = 250; const size_t n = (size_t)blocks * kBlock;
    DevBuf<Fr> io(2 * n);
    dev_fill_one(c, io.p, 2 * n);
    double best = 0;
    for (int rep = 0; rep < 4; rep++) {                                  // the first launch warms up; best of the rest
        OTTI_HIP(hipEventRecord(c.ev0, c.stream));
        hipLaunchKernelGGL(k_fr_mul_peak, blocks, kBlock, 0, c.stream, io.p, iters);
        OTTI_HIP(hipEventRecord(c.ev1, c.stream)); OTTI_HIP(hipEventSynchronize(c.ev1));
        float ms = 0; OTTI_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
        if (rep && ms > 0) best = std::max(best, 2.0 * (double)n * iters / (ms * 1e-3));
    }
    return best;
}
__global__ __launch_bounds__(kBlock) void k_gather_strided(const Fr *in, size_t stride, size_t offset, Fr *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i * stride + offset];
}
void dev_gather_strided(DevCtx &c, const Fr *in, size_t stride, size_t offset, Fr *out, size_t n) {
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_gather_strided, grid_for(n), kBlock, 0, c.stream, in, stride, offset, out, n);
}

}  // namespace otti
