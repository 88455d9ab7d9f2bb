// K1 / K6: the sparse matrix-vector products (multiply_vec, compute_eval_table_sparse) and the CSR copies they read, built on the device
// from the uploaded entry lists (whole instance, or one rank's shard).
#include "kernels_common.h"

namespace otti {

// ------------------------------------------------------------------------------------------------ K1 / K6 sparse products
// one entry's product.  kSmall: the coefficient is read as a 4-byte code first; only codes that are not small integers touch the 32-byte value
template <bool kSmall> __device__ __forceinline__ Fr entry_term(const DCsr3 &m, int k, uint32_t p, const Fr *x) {
    const Fr xv = x[m.idx[k][p]];
    if (kSmall) { const int32_t c = m.small[k][p]; if (c != kNotSmall) return fr_mul_small(xv, c); }
    return fr9_pack_lt2l(fr9_mul(fr9_unpack5(m.val[k][p]), fr9_unpack(xv)));           // nine limbs (fr9.h): 32 v x / 2^261 + l < 1.1 l
}
// c0 a0 + c1 a1 + c2 a2 in nine limbs (fr9.h): three products summed limb by limb, one reduction (each product < 1.1 l)
__device__ __forceinline__ Fr combine3(const Fr &c0, const Fr &c1, const Fr &c2, const Fr &a0, const Fr &a1, const Fr &a2) {
    const Fr9 t = fr9_add(fr9_add(fr9_mul(fr9_unpack5(c0), fr9_unpack(a0)), fr9_mul(fr9_unpack5(c1), fr9_unpack(a1))), fr9_mul(fr9_unpack5(c2), fr9_unpack(a2)));
    return fr9_canon(fr9_norm(t));
}
template <bool kSmall> __device__ __forceinline__ Fr row_dot(const DCsr3 &m, int k, const Fr *x, size_t r) {
    Fr acc = fr_zero();
    const uint32_t p0 = m.ptr[k][r], p1 = m.ptr[k][r + 1];
    for (uint32_t p = p0; p < p1; p++) acc = fr_add(acc, entry_term<kSmall>(m, k, p, x));
    return acc;
}
template <bool kSmall> __global__ __launch_bounds__(kBlock) void k_spmv3_light(DCsr3 m, size_t rows, const Fr *x, Fr *o0, Fr *o1, Fr *o2, int combine, Fr c0, Fr c1, Fr c2) {
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
        uint32_t l0 = m.ptr[0][r + 1] - m.ptr[0][r], l1 = m.ptr[1][r + 1] - m.ptr[1][r], l2 = m.ptr[2][r + 1] - m.ptr[2][r];
        if (max(l0, max(l1, l2)) > (uint32_t)kHeavyRow) continue;
        Fr a0 = row_dot<kSmall>(m, 0, x, r);
        Fr a1 = row_dot<kSmall>(m, 1, x, r);
        Fr a2 = row_dot<kSmall>(m, 2, x, r);
        if (combine) o0[r] = combine3(c0, c1, c2, a0, a1, a2);
        else { o0[r] = a0; o1[r] = a1; o2[r] = a2; }
    }
}
// The same with FOUR lanes per row (a quad walks its row four entries at a time and adds up by quad shuffles): for matrices with several
// entries per row the value / index loads of a wave are then 128-byte runs instead of one entry per lane at row-length strides.
__device__ __forceinline__ Fr quad_sum(Fr a) {
    a = fr_add(a, shfl_xor_fr(a, 1)); a = fr_add(a, shfl_xor_fr(a, 2)); return a;
}
template <bool kSmall> __global__ __launch_bounds__(kBlock) void k_spmv3_quad(DCsr3 m, size_t rows, const Fr *x, Fr *o0, Fr *o1, Fr *o2, int combine, Fr c0, Fr c1, Fr c2) {
    const int q = threadIdx.x & 3;
    for (size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 2; r < ((rows + 63) & ~(size_t)63); r += ((size_t)gridDim.x * blockDim.x) >> 2) {
        Fr a[3] = {fr_zero(), fr_zero(), fr_zero()};
        bool heavy = false;
        if (r < rows) {
            heavy = max(m.ptr[0][r + 1] - m.ptr[0][r], max(m.ptr[1][r + 1] - m.ptr[1][r], m.ptr[2][r + 1] - m.ptr[2][r])) > (uint32_t)kHeavyRow;
            if (!heavy)
                for (int k = 0; k < 3; k++) {
                    const uint32_t p1 = m.ptr[k][r + 1];
                    for (uint32_t p = m.ptr[k][r] + (uint32_t)q; p < p1; p += 4) a[k] = fr_add(a[k], entry_term<kSmall>(m, k, p, x));
                }
        }
        for (int k = 0; k < 3; k++) a[k] = quad_sum(a[k]);         // every lane of the wave takes part (rows are padded to whole waves above)
        if (r < rows && !heavy && q == 0) {
            if (combine) o0[r] = combine3(c0, c1, c2, a[0], a[1], a[2]);
            else { o0[r] = a[0]; o1[r] = a[1]; o2[r] = a[2]; }
        }
    }
}
// Long lists (a linear combination over thousands of variables; the constant-1 column of a compiled circuit, which can hold O(N)
// entries in the transposed copy) are cut into segments of kHeavySeg entries: one workgroup per segment writes the three raw partial
// sums, then one thread per long list adds its segments up and applies the combination.
constexpr uint32_t kHeavySeg = 2048;
template <bool kSmall> __global__ __launch_bounds__(kBlock) void k_spmv3_heavy_seg(DCsr3 m, const uint32_t *seg_row, const uint32_t *seg_no, const Fr *x, Fr *partial) {
    const size_t r = seg_row[blockIdx.x]; const uint32_t sn = seg_no[blockIdx.x];
    Fr acc[3];
    for (int k = 0; k < 3; k++) {
        acc[k] = fr_zero();
        const uint32_t p0 = m.ptr[k][r], p1 = m.ptr[k][r + 1];
        const uint64_t lo = (uint64_t)p0 + (uint64_t)sn * kHeavySeg;
        if (lo >= p1) continue;
        const uint32_t hi = (uint32_t)min((uint64_t)p1, lo + kHeavySeg);
        for (uint32_t p = (uint32_t)lo + threadIdx.x; p < hi; p += blockDim.x) acc[k] = fr_add(acc[k], entry_term<kSmall>(m, k, p, x));
    }
    block_reduce<3>(acc);
    if (threadIdx.x == 0) for (int k = 0; k < 3; k++) partial[(size_t)blockIdx.x * 3 + k] = acc[k];
}
// one workgroup per long list: its segments' partial sums are added up across the workgroup (the constant-1 column of a compiled circuit
// can have hundreds of segments: one thread walking them alone took ~90 us)
__global__ __launch_bounds__(kBlock) void k_spmv3_heavy_combine(const uint32_t *heavy, const uint32_t *seg_begin, size_t n_heavy, const Fr *partial, Fr *o0, Fr *o1, Fr *o2,
                                                                int combine, Fr c0, Fr c1, Fr c2) {
    const size_t h = blockIdx.x;
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (uint32_t s = seg_begin[h] + threadIdx.x; s < seg_begin[h + 1]; s += blockDim.x)
        for (int k = 0; k < 3; k++) acc[k] = fr_add(acc[k], partial[(size_t)s * 3 + k]);
    block_reduce<3>(acc);
    if (threadIdx.x != 0) return;
    const size_t r = heavy[h];
    if (combine) o0[r] = fr_add(fr_add(fr_mul(c0, acc[0]), fr_mul(c1, acc[1])), fr_mul(c2, acc[2]));
    else { o0[r] = acc[0]; o1[r] = acc[1]; o2[r] = acc[2]; }
}
void dev_spmv3(DevCtx &c, const DeviceCsrSet &m, const Fr *x, Fr *o0, Fr *o1, Fr *o2, bool combine, const Fr coef[3]) {
    Fr z = fr_zero();
    Fr c0 = coef ? coef[0] : z, c1 = coef ? coef[1] : z, c2 = coef ? coef[2] : z;
    KScope ks(c, KC_SPMV);
    // measured on the compiler-like 2^20 instance (4.6 entries per row and matrix): 0.54 -> 0.37 ms; on the uniform one (1 entry) the quad
    // kernel would idle three lanes in four (0.16 -> 0.54 ms)
    const bool quad = m.avg_row >= 3.0;
    const bool sm = m.use_small;
    if (quad && sm) hipLaunchKernelGGL(k_spmv3_quad<true>, grid_for(4 * m.rows), kBlock, 0, c.stream, m.view(), m.rows, x, o0, o1, o2, (int)combine, c0, c1, c2);
    else if (quad) hipLaunchKernelGGL(k_spmv3_quad<false>, grid_for(4 * m.rows), kBlock, 0, c.stream, m.view(), m.rows, x, o0, o1, o2, (int)combine, c0, c1, c2);
    else if (sm) hipLaunchKernelGGL(k_spmv3_light<true>, grid_for(m.rows), kBlock, 0, c.stream, m.view(), m.rows, x, o0, o1, o2, (int)combine, c0, c1, c2);
    else hipLaunchKernelGGL(k_spmv3_light<false>, grid_for(m.rows), kBlock, 0, c.stream, m.view(), m.rows, x, o0, o1, o2, (int)combine, c0, c1, c2);
    if (m.n_heavy) {
        // segment partials are scratch of the CALLER's context: the matrix object itself is shared by concurrent provers
        if (c.spmv_partial.n < 3 * m.n_seg) { OTTI_HIP(hipStreamSynchronize(c.stream)); c.spmv_partial.alloc(3 * m.n_seg); }
        if (sm) hipLaunchKernelGGL(k_spmv3_heavy_seg<true>, (unsigned)m.n_seg, kBlock, 0, c.stream, m.view(), (const uint32_t *)m.seg_row.p, (const uint32_t *)m.seg_no.p, x, c.spmv_partial.p);
        else hipLaunchKernelGGL(k_spmv3_heavy_seg<false>, (unsigned)m.n_seg, kBlock, 0, c.stream, m.view(), (const uint32_t *)m.seg_row.p, (const uint32_t *)m.seg_no.p, x, c.spmv_partial.p);
        hipLaunchKernelGGL(k_spmv3_heavy_combine, (unsigned)m.n_heavy, kBlock, 0, c.stream, (const uint32_t *)m.heavy.p, (const uint32_t *)m.seg_begin.p, m.n_heavy,
                           (const Fr *)c.spmv_partial.p, o0, o1, o2, (int)combine, c0, c1, c2);
    }
}

// ------------------------------------------------------------------------------------------------ CSR copies, built on the device
// The host keeps the matrices as entry lists in caller order (upstream's Vec<SparseMatEntry>); the two access paths the kernels want
// (by row for multiply_vec, by column for compute_eval_table_sparse) are counting sorts made here from ONE upload of the lists:
// count per major index (atomics), exclusive scan -> ptr, scatter through per-row cursors.  The order of the entries inside a row
// depends on the scatter's arrival order; every consumer adds the row's products in GF(l), where the sum does not depend on it.
__global__ __launch_bounds__(kBlock) void k_csr_count(const uint32_t *major, size_t n, uint32_t *ptr) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(&ptr[major[i] + 1], 1u);
}
__global__ __launch_bounds__(kBlock) void k_csr_fill(const uint32_t *major, const uint32_t *minor, const Fr *val, const int32_t *code, size_t n, uint32_t *cursor,
                                                     uint32_t *idx, Fr *out, int32_t *out_code) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t p = atomicAdd(&cursor[major[i]], 1u);
        idx[p] = minor[i]; out[p] = val[i];
        if (out_code) out_code[p] = code[i];
    }
}
// the small-integer codes of an entry list (fr_small_code) and how many of them there are
__global__ __launch_bounds__(kBlock) void k_coef_codes(const Fr *val, size_t n, int32_t *code, unsigned long long *n_small) {
    unsigned mine = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const int32_t c = fr_small_code(val[i]); code[i] = c; mine += c != kNotSmall; }
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor((int)mine, off, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_small, (unsigned long long)mine);
}
// in-place inclusive scan of u32 counts: every workgroup scans 4096 elements and reports its total; the totals are scanned the same way
// (recursively: two levels reach 2^24 elements, three 2^36), then added back
constexpr int kScanItems = 16;
__global__ __launch_bounds__(kBlock) void k_scan_blocks(uint32_t *a, size_t n, uint32_t *totals) {
    __shared__ uint32_t s_wave[kBlock / 64];
    const size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kScanItems;
    uint32_t v[kScanItems], run = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) { run += base + i < n ? a[base + i] : 0u; v[i] = run; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = run;
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, off, 64); if (lane >= off) x += y; }
    if (lane == 63) s_wave[wave] = x;
    __syncthreads();
    uint32_t before = 0;
    for (int k = 0; k < wave; k++) before += s_wave[k];
    const uint32_t excl = before + x - run;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) if (base + i < n) a[base + i] = v[i] + excl;
    if (threadIdx.x == kBlock - 1) totals[blockIdx.x] = before + x;
}
__global__ __launch_bounds__(kBlock) void k_scan_add(uint32_t *a, size_t n, const uint32_t *totals_scanned) {
    if (blockIdx.x == 0) return;
    const uint32_t add = totals_scanned[blockIdx.x - 1];
    const size_t base = ((size_t)blockIdx.x * kBlock + threadIdx.x) * kScanItems;
#pragma unroll
    for (int i = 0; i < kScanItems; i++) if (base + i < n) a[base + i] += add;
}
static void scan_inplace(DevCtx &c, uint32_t *a, size_t n, std::vector<DevBuf<uint32_t>> &levels, size_t depth = 0) {
    const size_t per = (size_t)kBlock * kScanItems, nb = (n + per - 1) / per;
    if (levels.size() <= depth) levels.emplace_back();
    if (levels[depth].n < nb) { OTTI_HIP(hipStreamSynchronize(c.stream)); levels[depth].alloc(nb); }
    hipLaunchKernelGGL(k_scan_blocks, (unsigned)nb, kBlock, 0, c.stream, a, n, levels[depth].p);
    if (nb == 1) return;
    uint32_t *totals = levels[depth].p;                       // (a deeper level may grow `levels`: take the pointer first)
    scan_inplace(c, totals, nb, levels, depth + 1);
    hipLaunchKernelGGL(k_scan_add, (unsigned)nb, kBlock, 0, c.stream, a, n, (const uint32_t *)totals);
}
// rows whose longest list (over the three matrices) exceeds kHeavyRow
__global__ __launch_bounds__(kBlock) void k_csr_count_heavy(const uint32_t *p0, const uint32_t *p1, const uint32_t *p2, size_t rows, unsigned *count) {
    for (size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x)
        if (max(p0[r + 1] - p0[r], max(p1[r + 1] - p1[r], p2[r + 1] - p2[r])) > (uint32_t)kHeavyRow) atomicAdd(count, 1u);
}
struct DeviceCoo { DevBuf<uint32_t> row, col; DevBuf<Fr> val; DevBuf<int32_t> code; size_t n = 0; };
static void upload_coo(DevCtx &c, DeviceCoo &d, const std::vector<uint32_t> &row, const std::vector<uint32_t> &col, const std::vector<Fr> &val) {
    d.n = val.size();
    d.row.alloc(std::max<size_t>(1, d.n)); d.col.alloc(std::max<size_t>(1, d.n)); d.val.alloc(std::max<size_t>(1, d.n));
    if (!d.n) return;
    OTTI_HIP(hipMemcpyAsync(d.row.p, row.data(), d.n * 4, hipMemcpyHostToDevice, c.stream));
    OTTI_HIP(hipMemcpyAsync(d.col.p, col.data(), d.n * 4, hipMemcpyHostToDevice, c.stream));
    OTTI_HIP(hipMemcpyAsync(d.val.p, val.data(), d.n * sizeof(Fr), hipMemcpyHostToDevice, c.stream));
}
// decide once per instance whether the kernels read coefficient codes: worth it when most entries are small integers
static bool classify_coefficients(DevCtx &c, DeviceCoo coo[3]) {
    DevBuf<unsigned long long> count(1); unsigned long long n_small = 0; size_t total = 0;
    OTTI_HIP(hipMemsetAsync(count.p, 0, sizeof(unsigned long long), c.stream));
    for (int k = 0; k < 3; k++) {
        total += coo[k].n;
        coo[k].code.alloc(std::max<size_t>(1, coo[k].n));
        if (coo[k].n) hipLaunchKernelGGL(k_coef_codes, grid_for(coo[k].n), kBlock, 0, c.stream, (const Fr *)coo[k].val.p, coo[k].n, coo[k].code.p, count.p);
    }
    OTTI_HIP(hipMemcpyAsync(&n_small, count.p, sizeof n_small, hipMemcpyDeviceToHost, c.stream));
    OTTI_HIP(hipStreamSynchronize(c.stream));
    return total >= 1024 && 2 * n_small >= total;
}
static void build_csr_set(DevCtx &c, DeviceCsrSet &d, const DeviceCoo coo[3], bool by_col, size_t rows, bool use_small) {
    if (rows + 1 > ((size_t)1 << 31)) throw Error(OTTI_ERR_BAD_ARG, "instance too large for 32-bit indices");
    d.rows = rows; d.use_small = use_small;
    d.avg_row = rows ? (double)(coo[0].n + coo[1].n + coo[2].n) / (3.0 * (double)rows) : 0.0;
    DevBuf<uint32_t> cursor(rows);
    std::vector<DevBuf<uint32_t>> scan_levels; scan_levels.reserve(4);
    for (int k = 0; k < 3; k++) {
        const DeviceCoo &m = coo[k];
        const uint32_t *major = by_col ? m.col.p : m.row.p, *minor = by_col ? m.row.p : m.col.p;
        d.ptr[k].alloc(rows + 1); d.idx[k].alloc(std::max<size_t>(1, m.n)); d.val[k].alloc(std::max<size_t>(1, m.n));
        if (use_small) d.small[k].alloc(std::max<size_t>(1, m.n));
        OTTI_HIP(hipMemsetAsync(d.ptr[k].p, 0, (rows + 1) * 4, c.stream));
        if (m.n) hipLaunchKernelGGL(k_csr_count, grid_for(m.n), kBlock, 0, c.stream, major, m.n, d.ptr[k].p);
        scan_inplace(c, d.ptr[k].p, rows + 1, scan_levels);   // ptr[0] = 0: the inclusive sum over [0, rows] is the exclusive one shifted
        if (m.n) {
            OTTI_HIP(hipMemcpyAsync(cursor.p, d.ptr[k].p, rows * 4, hipMemcpyDeviceToDevice, c.stream));
            hipLaunchKernelGGL(k_csr_fill, grid_for(m.n), kBlock, 0, c.stream, major, minor, (const Fr *)m.val.p, (const int32_t *)m.code.p, m.n, cursor.p, d.idx[k].p, d.val[k].p,
                               use_small ? d.small[k].p : (int32_t *)nullptr);
        }
    }
    // long lists: found on the device; only when there are any do the row pointers come back for the segment lists
    DevBuf<unsigned> n_heavy_dev(1); unsigned n_heavy = 0;
    OTTI_HIP(hipMemsetAsync(n_heavy_dev.p, 0, sizeof(unsigned), c.stream));
    hipLaunchKernelGGL(k_csr_count_heavy, grid_for(rows), kBlock, 0, c.stream, (const uint32_t *)d.ptr[0].p, (const uint32_t *)d.ptr[1].p, (const uint32_t *)d.ptr[2].p, rows, n_heavy_dev.p);
    OTTI_HIP(hipMemcpyAsync(&n_heavy, n_heavy_dev.p, sizeof(unsigned), hipMemcpyDeviceToHost, c.stream));
    OTTI_HIP(hipStreamSynchronize(c.stream));                 // also: cursor / scan scratch / the entry lists may go out of scope after this
    d.n_heavy = 0; d.n_seg = 0;
    if (!n_heavy) return;
    std::vector<uint32_t> ptr[3], heavy, seg_row, seg_no, seg_begin;
    for (int k = 0; k < 3; k++) { ptr[k].resize(rows + 1); OTTI_HIP(hipMemcpy(ptr[k].data(), d.ptr[k].p, (rows + 1) * 4, hipMemcpyDeviceToHost)); }
    for (size_t r = 0; r < rows; r++) {
        uint32_t mx = 0;
        for (int k = 0; k < 3; k++) mx = std::max(mx, ptr[k][r + 1] - ptr[k][r]);
        if (mx > (uint32_t)kHeavyRow) {
            heavy.push_back((uint32_t)r); seg_begin.push_back((uint32_t)seg_row.size());
            for (uint32_t sn = 0; sn * kHeavySeg < mx; sn++) { seg_row.push_back((uint32_t)r); seg_no.push_back(sn); }
        }
    }
    seg_begin.push_back((uint32_t)seg_row.size());
    d.n_heavy = heavy.size(); d.n_seg = seg_row.size();
    auto up = [](DevBuf<uint32_t> &b, const std::vector<uint32_t> &v) { b.alloc(v.size()); OTTI_HIP(hipMemcpy(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice)); };
    up(d.heavy, heavy); up(d.seg_row, seg_row); up(d.seg_no, seg_no); up(d.seg_begin, seg_begin);
}
std::shared_ptr<DeviceInstance> upload_instance(const Instance &I) {
    DevCtx &c = DevCtx::get();
    auto d = std::make_shared<DeviceInstance>();
    DeviceCoo coo[3];
    for (int k = 0; k < 3; k++) upload_coo(c, coo[k], I.M[k].row, I.M[k].col, I.M[k].val);
    const bool use_small = classify_coefficients(c, coo);
    build_csr_set(c, d->by_row, coo, false, I.num_cons, use_small);
    build_csr_set(c, d->by_col, coo, true, 2 * I.num_vars, use_small);
    d->nnz = I.M[0].val.size() + I.M[1].val.size() + I.M[2].val.size();
    return d;
}

std::shared_ptr<DeviceShard> upload_instance_shard(const Instance &I, int rank, int world) {
    DevCtx &c = DevCtx::get();
    if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world || (size_t)world > I.num_cons || (size_t)world > 2 * I.num_vars)
        throw Error(OTTI_ERR_BAD_ARG, "shard: world must be a power of two not larger than the instance");
    auto d = std::make_shared<DeviceShard>();
    d->rank = rank; d->world = world;
    const uint32_t g = (uint32_t)world, k = (uint32_t)rank;
    // rows r = k (mod g) renumbered r / g, columns untouched; and the entries of columns c = k (mod g) renumbered c / g, rows untouched
    DeviceCoo rows[3], cols[3];
    for (int m = 0; m < 3; m++) {
        const SparseMat &M = I.M[m];
        std::vector<uint32_t> rr, rc, cr, cc; std::vector<Fr> rv, cv;
        for (size_t e = 0; e < M.val.size(); e++) {
            if (M.row[e] % g == k) { rr.push_back(M.row[e] / g); rc.push_back(M.col[e]); rv.push_back(M.val[e]); }
            if (M.col[e] % g == k) { cr.push_back(M.row[e]); cc.push_back(M.col[e] / g); cv.push_back(M.val[e]); }
        }
        upload_coo(c, rows[m], rr, rc, rv); upload_coo(c, cols[m], cr, cc, cv);
        OTTI_HIP(hipStreamSynchronize(c.stream));             // the host lists above go out of scope
    }
    const bool small_rows = classify_coefficients(c, rows), small_cols = classify_coefficients(c, cols);
    build_csr_set(c, d->by_row, rows, false, I.num_cons / g, small_rows);
    build_csr_set(c, d->by_col, cols, true, 2 * I.num_vars / g, small_cols);
    return d;
}

}  // namespace otti
