// Kernels of SNARK mode's R1CSEvalProof (snark.h): dereferencing the eq tables by row / column address, the memory-checking hash
// layer, product-circuit layers, and the rounds of the batched cubic sum-check over many (A, B, C) table triples at once.
// All HBM-streaming work on 32-byte field elements (no MFMA applies: 256-bit modular integers); one launch handles every instance of a
// batch (grid.y = instance), so a round of SumcheckInstanceProof::prove_cubic_batched is two launches whatever the batch size.
//
// Kernel <-> upstream loop [RECALL; the reference's Spartan/ submodule is empty]:
//   k_gather                sparse_mlpoly.rs AddrTimestamps::deref_mem
//   k_hash_mem / k_hash_ops sparse_mlpoly.rs Layers::build_hash_layer (init / audit, read / write)
//   k_prod_layer            product_tree.rs ProductCircuit::compute_layer
//   k_abc_evals             sumcheck.rs SumcheckInstanceProof::prove_cubic_batched, the evaluation loop (comb = A * B * C at 0, 2, 3)
//   k_fold_many             dense_mlpoly.rs DensePolynomial::bound_poly_var_top over every table of the batch
//   k_dot_many / k_sum3     DensePolynomial::evaluate (against a shared eq table) / DotProductCircuit::evaluate
#include "kernels_common.h"
#include "snark_dev.h"

namespace otti {

__global__ __launch_bounds__(kBlock) void k_gather(const Fr *table, const uint32_t *idx, Fr *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = table[idx[i]];
}
void dev_gather(DevCtx &c, const Fr *table, const uint32_t *idx, Fr *out, size_t n) {
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_gather, grid_for(n), kBlock, 0, c.stream, table, idx, out, n);
}

// hash(addr, val, ts) - gamma = ts * r^2 + val * r + addr - gamma
__device__ __forceinline__ Fr hash3(const Fr &addr, const Fr &val, const Fr &ts, const Fr &r, const Fr &r2, const Fr &gamma) {
    return fr_sub(fr_add(fr_add(fr_mul(ts, r2), fr_mul(val, r)), addr), gamma);
}
// memory cells: addr = the cell index, val = the eq table, ts = 0 (init) or the audit timestamp
__global__ __launch_bounds__(kBlock) void k_hash_mem(const Fr *eval_table, const Fr *audit_ts, Fr *out_init, Fr *out_audit, size_t M, Fr r, Fr r2, Fr gamma) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < M; i += (size_t)gridDim.x * blockDim.x) {
        const Fr base = fr_sub(fr_add(fr_mul(eval_table[i], r), fr_from_u64((uint64_t)i)), gamma);
        out_init[i] = base;
        out_audit[i] = fr_add(base, fr_mul(audit_ts[i], r2));
    }
}
// operations: read uses the read timestamp, write the same plus one (so write = read + r^2)
__global__ __launch_bounds__(kBlock) void k_hash_ops(const Fr *addr_f, const Fr *deref, const Fr *read_ts, Fr *out_read, Fr *out_write, size_t N, Fr r, Fr r2, Fr gamma) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
        const Fr rd = hash3(addr_f[i], deref[i], read_ts[i], r, r2, gamma);
        out_read[i] = rd; out_write[i] = fr_add(rd, r2);
    }
}
void dev_hash_mem(DevCtx &c, const Fr *eval_table, const Fr *audit_ts, Fr *out_init, Fr *out_audit, size_t M, const Fr &r, const Fr &gamma) {
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_hash_mem, grid_for(M), kBlock, 0, c.stream, eval_table, audit_ts, out_init, out_audit, M, r, fr_mul(r, r), gamma);
}
void dev_hash_ops(DevCtx &c, const Fr *addr_f, const Fr *deref, const Fr *read_ts, Fr *out_read, Fr *out_write, size_t N, const Fr &r, const Fr &gamma) {
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_hash_ops, grid_for(N), kBlock, 0, c.stream, addr_f, deref, read_ts, out_read, out_write, N, r, fr_mul(r, r), gamma);
}

// one layer of every product circuit of a batch: out_left[i] = in_left[i] * in_right[i], out_right[i] = in_left[q + i] * in_right[q + i]
__global__ __launch_bounds__(kBlock) void k_prod_layer(LayerList L, size_t q) {
    const Fr *il = L.in_left[blockIdx.y], *ir = L.in_right[blockIdx.y]; Fr *ol = L.out_left[blockIdx.y], *orr = L.out_right[blockIdx.y];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        ol[i] = fr_mul(il[i], ir[i]); orr[i] = fr_mul(il[q + i], ir[q + i]);
    }
}
void dev_prod_layer(DevCtx &c, const LayerList &L, size_t q) {
    if (!q || !L.n) return;
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_prod_layer, dim3((unsigned)grid_for(q), (unsigned)L.n), kBlock, 0, c.stream, L, q);
}

// ---- a round of the batched cubic sum-check.  Instance y: sums over i < half of A*B*C at the points 0, 2, 3 of the variable being bound.
__global__ __launch_bounds__(kBlock) void k_abc_evals(AbcList L, size_t half, Fr *partials) {
    const Fr *A = L.A[blockIdx.y], *B = L.B[blockIdx.y], *C = L.C[blockIdx.y];
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        const Fr a0 = A[i], a1 = A[i + half], b0 = B[i], b1 = B[i + half], c0 = C[i], c1 = C[i + half];
        acc[0] = fr_add(acc[0], fr_mul(fr_mul(a0, b0), c0));
        const Fr da = fr_sub(a1, a0), db = fr_sub(b1, b0), dc = fr_sub(c1, c0);
        Fr a = fr_add(a1, da), b = fr_add(b1, db), cc = fr_add(c1, dc);
        acc[1] = fr_add(acc[1], fr_mul(fr_mul(a, b), cc));
        a = fr_add(a, da); b = fr_add(b, db); cc = fr_add(cc, dc);
        acc[2] = fr_add(acc[2], fr_mul(fr_mul(a, b), cc));
    }
    block_reduce<3>(acc);
    if (threadIdx.x == 0) for (int k = 0; k < 3; k++) partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + k] = acc[k];
}
// out[y * K + k] = sum over the nblk partials of instance y (out may be pinned host memory)
template <int K> __global__ __launch_bounds__(kBlock) void k_reduce_many(const Fr *partials, int nblk, Fr *out) {
    Fr acc[K];
    for (int k = 0; k < K; k++) acc[k] = fr_zero();
    for (int b = threadIdx.x; b < nblk; b += blockDim.x)
        for (int k = 0; k < K; k++) acc[k] = fr_add(acc[k], partials[((size_t)blockIdx.x * nblk + b) * K + k]);
    block_reduce<K>(acc);
    if (threadIdx.x == 0) for (int k = 0; k < K; k++) out[(size_t)blockIdx.x * K + k] = acc[k];
}
static inline int many_grid(size_t n, int ninst) { return (int)std::max<size_t>(1, std::min<size_t>((n + kBlock - 1) / kBlock, std::max<size_t>(1, (size_t)kMaxBlocks / (size_t)ninst))); }
// results land in c.h_results[slot .. slot + 3 n) after c.sync()
void dev_abc_evals(DevCtx &c, const AbcList &L, size_t half, Fr *partials, int slot) {
    const int g = many_grid(half, L.n);
    KScope ks(c, KC_SC_CUBIC);
    hipLaunchKernelGGL(k_abc_evals, dim3((unsigned)g, (unsigned)L.n), kBlock, 0, c.stream, L, half, partials);
    hipLaunchKernelGGL(k_reduce_many<3>, L.n, kBlock, 0, c.stream, (const Fr *)partials, g, c.d_results_alias + slot);
}
__global__ __launch_bounds__(kBlock) void k_fold_many(PtrList L, size_t half, Fr r) {
    Fr *Z = L.p[blockIdx.y];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) { const Fr a = Z[i], b = Z[i + half]; Z[i] = fr_add(a, fr_mul(r, fr_sub(b, a))); }
}
void dev_fold_many(DevCtx &c, const PtrList &L, size_t half, const Fr &r) {
    if (!half || !L.n) return;
    KScope ks(c, KC_SC_CUBIC);
    hipLaunchKernelGGL(k_fold_many, dim3((unsigned)many_grid(half, L.n), (unsigned)L.n), kBlock, 0, c.stream, L, half, r);
}
// element 0 of every table of the list -> c.h_results[slot ..)
__global__ void k_pick0(PtrList L, Fr *out) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < L.n) out[i] = L.p[i][0]; }
void dev_pick0(DevCtx &c, const PtrList &L, int slot) { if (L.n) hipLaunchKernelGGL(k_pick0, 1, 64, 0, c.stream, L, c.d_results_alias + slot); }

// ---- evaluations: out[y] = <E, P_y> for every polynomial of the list (E: the eq table of the point), and sum l * r * w
__global__ __launch_bounds__(kBlock) void k_dot_many(const Fr *E, PtrList L, size_t n, Fr *partials) {
    const Fr *P = L.p[blockIdx.y];
    Fr acc[1] = {fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[0] = fr_add(acc[0], fr_mul(E[i], P[i]));
    block_reduce<1>(acc);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc[0];
}
void dev_dot_many(DevCtx &c, const Fr *E, const PtrList &L, size_t n, Fr *partials, int slot) {
    const int g = many_grid(n, L.n);
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_dot_many, dim3((unsigned)g, (unsigned)L.n), kBlock, 0, c.stream, E, L, n, partials);
    hipLaunchKernelGGL(k_reduce_many<1>, L.n, kBlock, 0, c.stream, (const Fr *)partials, g, c.d_results_alias + slot);
}
__global__ __launch_bounds__(kBlock) void k_sum3(AbcList L, size_t n, Fr *partials) {
    const Fr *A = L.A[blockIdx.y], *B = L.B[blockIdx.y], *C = L.C[blockIdx.y];
    Fr acc[1] = {fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[0] = fr_add(acc[0], fr_mul(fr_mul(A[i], B[i]), C[i]));
    block_reduce<1>(acc);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc[0];
}
void dev_sum3(DevCtx &c, const AbcList &L, size_t n, Fr *partials, int slot) {
    const int g = many_grid(n, L.n);
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_sum3, dim3((unsigned)g, (unsigned)L.n), kBlock, 0, c.stream, L, n, partials);
    hipLaunchKernelGGL(k_reduce_many<1>, L.n, kBlock, 0, c.stream, (const Fr *)partials, g, c.d_results_alias + slot);
}

}  // namespace otti
