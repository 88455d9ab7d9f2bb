// K1 / K6: the sparse matrix-vector products (multiply_vec, compute_eval_table_sparse) and the CSR uploads, whole and sharded.
#include "kernels_common.h"

namespace otti {

// ------------------------------------------------------------------------------------------------ K1 / K6 sparse products
__device__ __forceinline__ Fr row_dot(const uint32_t *ptr, const uint32_t *idx, const Fr *val, const Fr *x, size_t r) {
    Fr acc = fr_zero();
    uint32_t p0 = ptr[r], p1 = ptr[r + 1];
    for (uint32_t p = p0; p < p1; p++) acc = fr_add(acc, fr_mul(val[p], x[idx[p]]));
    return acc;
}
__global__ __launch_bounds__(kBlock) void k_spmv3_light(DCsr3 m, size_t rows, const Fr *x, Fr *o0, Fr *o1, Fr *o2, int combine, Fr c0, Fr c1, Fr c2) {
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        uint32_t l0 = m.ptr[0][r + 1] - m.ptr[0][r], l1 = m.ptr[1][r + 1] - m.ptr[1][r], l2 = m.ptr[2][r + 1] - m.ptr[2][r];
        if (max(l0, max(l1, l2)) > (uint32_t)kHeavyRow) continue;
        Fr a0 = row_dot(m.ptr[0], m.idx[0], m.val[0], x, r);
        Fr a1 = row_dot(m.ptr[1], m.idx[1], m.val[1], x, r);
        Fr a2 = row_dot(m.ptr[2], m.idx[2], m.val[2], x, r);
        if (combine) o0[r] = fr_add(fr_add(fr_mul(c0, a0), fr_mul(c1, a1)), fr_mul(c2, a2));
        else { o0[r] = a0; o1[r] = a1; o2[r] = a2; }
    }
}
// Long lists (a linear combination over thousands of variables; the constant-1 column of a compiled circuit, which can hold O(N)
// entries in the transposed copy) are cut into segments of kHeavySeg entries: one workgroup per segment writes the three raw partial
// sums, then one thread per long list adds its segments up and applies the combination.
constexpr uint32_t kHeavySeg = 2048;
__global__ __launch_bounds__(kBlock) void k_spmv3_heavy_seg(DCsr3 m, const uint32_t *seg_row, const uint32_t *seg_no, const Fr *x, Fr *partial) {
    const size_t r = seg_row[blockIdx.x]; const uint32_t sn = seg_no[blockIdx.x];
    Fr acc[3];
    for (int k = 0; k < 3; k++) {
        acc[k] = fr_zero();
        const uint32_t p0 = m.ptr[k][r], p1 = m.ptr[k][r + 1];
        const uint64_t lo = (uint64_t)p0 + (uint64_t)sn * kHeavySeg;
        if (lo >= p1) continue;
        const uint32_t hi = (uint32_t)min((uint64_t)p1, lo + kHeavySeg);
        for (uint32_t p = (uint32_t)lo + threadIdx.x; p < hi; p += blockDim.x) acc[k] = fr_add(acc[k], fr_mul(m.val[k][p], x[m.idx[k][p]]));
    }
    block_reduce<3>(acc);
    if (threadIdx.x == 0) for (int k = 0; k < 3; k++) partial[(size_t)blockIdx.x * 3 + k] = acc[k];
}
__global__ __launch_bounds__(64) void k_spmv3_heavy_combine(const uint32_t *heavy, const uint32_t *seg_begin, size_t n_heavy, const Fr *partial, Fr *o0, Fr *o1, Fr *o2,
                                                            int combine, Fr c0, Fr c1, Fr c2) {
    size_t h = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (h >= n_heavy) return;
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (uint32_t s = seg_begin[h]; s < seg_begin[h + 1]; s++)
        for (int k = 0; k < 3; k++) acc[k] = fr_add(acc[k], partial[(size_t)s * 3 + k]);
    const size_t r = heavy[h];
    if (combine) o0[r] = fr_add(fr_add(fr_mul(c0, acc[0]), fr_mul(c1, acc[1])), fr_mul(c2, acc[2]));
    else { o0[r] = acc[0]; o1[r] = acc[1]; o2[r] = acc[2]; }
}
void dev_spmv3(DevCtx &c, const DeviceCsrSet &m, const Fr *x, Fr *o0, Fr *o1, Fr *o2, bool combine, const Fr coef[3]) {
    Fr z = fr_zero();
    Fr c0 = coef ? coef[0] : z, c1 = coef ? coef[1] : z, c2 = coef ? coef[2] : z;
    KScope ks(c, KC_SPMV);
    hipLaunchKernelGGL(k_spmv3_light, grid_for(m.rows), kBlock, 0, c.stream, m.view(), m.rows, x, o0, o1, o2, (int)combine, c0, c1, c2);
    if (m.n_heavy) {
        // segment partials are scratch of the CALLER's context: the matrix object itself is shared by concurrent provers
        if (c.spmv_partial.n < 3 * m.n_seg) { OTTI_HIP(hipStreamSynchronize(c.stream)); c.spmv_partial.alloc(3 * m.n_seg); }
        hipLaunchKernelGGL(k_spmv3_heavy_seg, (unsigned)m.n_seg, kBlock, 0, c.stream, m.view(), (const uint32_t *)m.seg_row.p, (const uint32_t *)m.seg_no.p, x, c.spmv_partial.p);
        hipLaunchKernelGGL(k_spmv3_heavy_combine, (unsigned)((m.n_heavy + 63) / 64), 64, 0, c.stream, (const uint32_t *)m.heavy.p, (const uint32_t *)m.seg_begin.p, m.n_heavy,
                           (const Fr *)c.spmv_partial.p, o0, o1, o2, (int)combine, c0, c1, c2);
    }
}

static void upload_csr_set(DeviceCsrSet &d, const SparseMat M[3], bool by_col) {
    size_t rows = by_col ? M[0].by_col.rows : M[0].by_row.rows;
    d.rows = rows;
    std::vector<uint32_t> heavy, seg_row, seg_no, seg_begin;
    for (int k = 0; k < 3; k++) {
        const Csr &s = by_col ? M[k].by_col : M[k].by_row;
        d.ptr[k].alloc(s.ptr.size()); d.idx[k].alloc(std::max<size_t>(1, s.idx.size())); d.val[k].alloc(std::max<size_t>(1, s.val.size()));
        OTTI_HIP(hipMemcpy(d.ptr[k].p, s.ptr.data(), s.ptr.size() * 4, hipMemcpyHostToDevice));
        if (!s.idx.empty()) {
            OTTI_HIP(hipMemcpy(d.idx[k].p, s.idx.data(), s.idx.size() * 4, hipMemcpyHostToDevice));
            OTTI_HIP(hipMemcpy(d.val[k].p, s.val.data(), s.val.size() * sizeof(Fr), hipMemcpyHostToDevice));
        }
    }
    for (size_t r = 0; r < rows; r++) {
        uint32_t mx = 0;
        for (int k = 0; k < 3; k++) { const Csr &s = by_col ? M[k].by_col : M[k].by_row; mx = std::max(mx, s.ptr[r + 1] - s.ptr[r]); }
        if (mx > (uint32_t)kHeavyRow) {
            heavy.push_back((uint32_t)r); seg_begin.push_back((uint32_t)seg_row.size());
            for (uint32_t sn = 0; sn * kHeavySeg < mx; sn++) { seg_row.push_back((uint32_t)r); seg_no.push_back(sn); }
        }
    }
    seg_begin.push_back((uint32_t)seg_row.size());
    d.n_heavy = heavy.size(); d.n_seg = seg_row.size();
    if (!heavy.empty()) {
        auto up = [](DevBuf<uint32_t> &b, const std::vector<uint32_t> &v) { b.alloc(v.size()); OTTI_HIP(hipMemcpy(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); };
        up(d.heavy, heavy); up(d.seg_row, seg_row); up(d.seg_no, seg_no); up(d.seg_begin, seg_begin);
    }
}
std::shared_ptr<DeviceInstance> upload_instance(const Instance &I) {
    DevCtx::get();
    auto d = std::make_shared<DeviceInstance>();
    upload_csr_set(d->by_row, I.M, false); upload_csr_set(d->by_col, I.M, true);
    d->nnz = I.M[0].val.size() + I.M[1].val.size() + I.M[2].val.size();
    return d;
}

std::shared_ptr<DeviceShard> upload_instance_shard(const Instance &I, int rank, int world) {
    DevCtx::get();
    if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world || (size_t)world > I.num_cons || (size_t)world > 2 * I.num_vars)
        throw Error(OTTI_ERR_BAD_ARG, "shard: world must be a power of two not larger than the instance");
    auto d = std::make_shared<DeviceShard>();
    d->rank = rank; d->world = world;
    const uint32_t g = (uint32_t)world, k = (uint32_t)rank;
    SparseMat rows[3], cols[3];
    for (int m = 0; m < 3; m++) {
        const SparseMat &M = I.M[m];
        for (size_t e = 0; e < M.val.size(); e++) {
            if (M.row[e] % g == k) { rows[m].row.push_back(M.row[e] / g); rows[m].col.push_back(M.col[e]); rows[m].val.push_back(M.val[e]); }
            if (M.col[e] % g == k) { cols[m].row.push_back(M.row[e]); cols[m].col.push_back(M.col[e] / g); cols[m].val.push_back(M.val[e]); }
        }
        build_csr(rows[m].by_row, rows[m].row, rows[m].col, rows[m].val, I.num_cons / g);
        build_csr(cols[m].by_col, cols[m].col, cols[m].row, cols[m].val, 2 * I.num_vars / g);
    }
    upload_csr_set(d->by_row, rows, false); upload_csr_set(d->by_col, cols, true);
    return d;
}

}  // namespace otti
