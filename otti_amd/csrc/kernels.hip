// gfx950 (MI355X, CDNA4) kernels of the Spartan NIZK proving path.  Wave64; 256 CUs in 8 XCDs; every kernel here is integer
// work on 256-bit field elements (8 x u32 limbs, v_mad_u64_u32 chains) — no MFMA applies.  Streaming kernels (K1-K7, K9)
// move 32-byte elements with two 16-byte accesses per lane and are HBM-bound; the fixed-base MSM (K8) is integer-ALU-bound.
//
// Kernel <-> upstream hot loop [RECALL: the reference's Spartan/ submodule is empty]:
//   k_spmv3_*            K1 sparse_mlpoly.rs SparseMatPolynomial::multiply_vec, K6 compute_eval_table_sparse (transposed copy)
//   k_eq_small/_expand   K2 dense_mlpoly.rs EqPolynomial::evals
//   k_sc_*               K3/K7 sumcheck.rs prove_cubic_with_additive_term / prove_quad inner loops, fused with
//                        K4 dense_mlpoly.rs DensePolynomial::bound_poly_var_top of the previous round
//   k_fold_top/_bot      K4/K5 bound_poly_var_top / bound_poly_var_bot
//   k_msm_rows/_finish   K8 DensePolynomial::commit_inner -> Commitments::commit (dalek vartime_multiscalar_mul), also K10's L/R
//   k_poly_bound_*       K9 DensePolynomial::bound
//   k_bullet_step        K10 nizk/bullet.rs BulletReductionProof::prove scalar bookkeeping
#include "device.h"
#include "fp10.h"
#include "pool.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <mutex>

namespace otti {

void hip_check(hipError_t e, const char *what, const char *file, int line) {
    if (e == hipSuccess) return;
    char buf[512]; snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    throw Error(OTTI_ERR_NO_DEVICE, buf);
}

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;           // 8 workgroups per CU; grid-stride beyond that
constexpr int kHeavyRow = 64;              // sparse rows longer than this go to the one-workgroup-per-row kernel

static inline int grid_for(size_t n) { size_t b = (n + kBlock - 1) / kBlock; return (int)std::max<size_t>(1, std::min<size_t>(b, kMaxBlocks)); }

// ------------------------------------------------------------------------------------------------ context
DevCtx &DevCtx::get() {
    static std::mutex mu; static DevCtx *ctx = nullptr; static bool failed = false; static std::string why;
    std::lock_guard<std::mutex> lk(mu);
    if (ctx) return *ctx;
    if (failed) throw Error(OTTI_ERR_NO_DEVICE, why);
    try {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count == 0) throw Error(OTTI_ERR_NO_DEVICE, "no HIP device visible: the MI355X proving path has no CPU fallback");
        int dev = 0;
        const char *env = getenv("OTTI_DEVICE"); if (!env) env = getenv("LOCAL_RANK");
        if (env) dev = atoi(env) % count;
        OTTI_HIP(hipSetDevice(dev));
        hipDeviceProp_t prop; OTTI_HIP(hipGetDeviceProperties(&prop, dev));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            char buf[256]; snprintf(buf, sizeof buf, "device %d is %s; this library carries gfx950 code objects only", dev, prop.gcnArchName);
            throw Error(OTTI_ERR_NO_DEVICE, buf);
        }
        DevCtx *c = new DevCtx();
        c->device = dev; c->num_cu = prop.multiProcessorCount;
        OTTI_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->partials.alloc((size_t)kMaxBlocks * 4);
        c->results.alloc(kResultSlots);
        OTTI_HIP(hipHostMalloc((void **)&c->h_results, kResultSlots * sizeof(Fr), hipHostMallocDefault));
        OTTI_HIP(hipHostGetDevicePointer((void **)&c->d_results_alias, c->h_results, 0));
        OTTI_HIP(hipHostMalloc((void **)&c->h_flag, 64, hipHostMallocDefault));
        *c->h_flag = 0;
        OTTI_HIP(hipHostGetDevicePointer((void **)&c->d_flag_alias, c->h_flag, 0));
        c->d_counter.alloc(1);
        OTTI_HIP(hipMemset(c->d_counter.p, 0, sizeof(unsigned)));
        OTTI_HIP(hipEventCreate(&c->ev0)); OTTI_HIP(hipEventCreate(&c->ev1));
        ctx = c;
        return *ctx;
    } catch (const Error &e) { failed = true; why = e.what(); throw; }
}

void DevCtx::ensure_points(size_t rows, size_t splits) {
    // Growing these buffers invalidates results of launches still in flight (h_points in particular is read by the host after an event
    // wait), so capacities start generous and the prover sizes them for the whole proof before its first launch; a later growth
    // drains the stream first.
    const size_t want_partial = std::max<size_t>(rows * splits, 8192), want_rows = std::max<size_t>(rows, 2 * kHostPtsCap);
    if (want_partial > msm_partial_cap) { if (stream) OTTI_HIP(hipStreamSynchronize(stream)); msm_partial.alloc(want_partial); msm_partial_cap = want_partial; }
    if (want_rows > points_cap) {
        if (stream) OTTI_HIP(hipStreamSynchronize(stream));
        if (h_points) (void)hipHostFree(h_points);
        OTTI_HIP(hipHostMalloc((void **)&h_points, want_rows * 32, hipHostMallocDefault));
        d_points.alloc(want_rows * 32); msm_final.alloc(want_rows); points_cap = want_rows;
    }
    if (!h_pts) {
        OTTI_HIP(hipHostMalloc((void **)&h_pts, kHostPtsCap * sizeof(Pt), hipHostMallocDefault));
        OTTI_HIP(hipHostGetDevicePointer((void **)&d_pts_alias, h_pts, 0));
        d_counter2.alloc(1); OTTI_HIP(hipMemset(d_counter2.p, 0, sizeof(unsigned)));
    }
}

// ------------------------------------------------------------------------------------------------ kernel timing
KStats &KStats::get() { static KStats s; return s; }
int KStats::begin(DevCtx &c, int k) {
    if (!on || !((mask >> k) & 1u)) return -1;
    if (pool.empty()) { pool.resize(16384); for (auto &e : pool) OTTI_HIP(hipEventCreate(&e)); cls.resize(8192); }
    if (used + 1 > cls.size()) return -1;                     // pool exhausted until the next flush
    int rec = (int)used++;
    cls[rec] = k;
    OTTI_HIP(hipEventRecord(pool[2 * rec], c.stream));
    return rec;
}
void KStats::end(DevCtx &c, int rec) { if (rec >= 0) (void)hipEventRecord(pool[2 * rec + 1], c.stream); }
void KStats::flush() {
    for (size_t r = 0; r < used; r++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, pool[2 * r], pool[2 * r + 1]) == hipSuccess) { total_ms[cls[r]] += ms; count[cls[r]]++; }
    }
    used = 0;
}
void KStats::reset() { used = 0; for (int k = 0; k < KC_COUNT; k++) { total_ms[k] = 0; count[k] = 0; } }

// ------------------------------------------------------------------------------------------------ inter-workgroup hand-off
// Publishing a workgroup's partial result to the LAST workgroup of the same launch.  A per-workgroup agent release fence
// (buffer_wbl2) serialises on the XCD's L2 and cost ~100 us over a 2048-workgroup grid; instead every handed-off byte is written
// with write-through (sc1) stores and read with sc1 loads (8-byte relaxed agent-scope atomics lower to exactly those), the storing
// lane drains its stores (s_waitcnt vmcnt(0)) before its agent-scope counter add, and the last arriver's other waves read only
// after a workgroup barrier behind the lane whose add returned last (MI355X guide, "valid forms", sc1 row).
__device__ __forceinline__ void store_words_sc1(void *dst, const uint32_t *w, int nwords) {
    unsigned long long *p = reinterpret_cast<unsigned long long *>(dst);
    for (int i = 0; i < nwords / 2; i++)
        __hip_atomic_store(p + i, (unsigned long long)w[2 * i] | ((unsigned long long)w[2 * i + 1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void load_words_sc1(uint32_t *w, const void *src, int nwords) {
    const unsigned long long *p = reinterpret_cast<const unsigned long long *>(src);
    for (int i = 0; i < nwords / 2; i++) {
        unsigned long long v = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w[2 * i] = (uint32_t)v; w[2 * i + 1] = (uint32_t)(v >> 32);
    }
}
// true in every thread of exactly one workgroup: the last one to call it in this launch (counter is left at zero for the next launch)
__device__ __forceinline__ bool arrive_and_check_last(unsigned *counter, unsigned total) {
    __shared__ int s_is_last;
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this lane's sc1 stores have left
        unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_is_last = (old == total - 1) ? 1 : 0;
        if (s_is_last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    return s_is_last != 0;
}

// ------------------------------------------------------------------------------------------------ wave / block reductions of Fr
__device__ __forceinline__ Fr shfl_xor_fr(const Fr &x, int mask) {
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = (uint32_t)__shfl_xor((int)x.v[i], mask, 64);
    return r;
}
// sums acc[0..K) over the workgroup (blockDim.x a multiple of 64, <= 1024); thread 0 ends up with the totals
template <int K> __device__ __forceinline__ void block_reduce(Fr (&acc)[K]) {
    __shared__ Fr sm[K][16];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = fr_add(acc[k], shfl_xor_fr(acc[k], off));
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();                         // protects sm against a previous use
    if (lane == 0) for (int k = 0; k < K; k++) sm[k][wave] = acc[k];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < K; k++) { Fr t = sm[k][0]; for (int w = 1; w < nw; w++) t = fr_add(t, sm[k][w]); acc[k] = t; }
}
template <int K> __global__ __launch_bounds__(kBlock) void k_reduce_partials(const Fr *partials, int nblocks, Fr *out) {
    Fr acc[K];
    for (int k = 0; k < K; k++) acc[k] = fr_zero();
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x)
        for (int k = 0; k < K; k++) acc[k] = fr_add(acc[k], partials[(size_t)b * K + k]);
    block_reduce<K>(acc);
    if (threadIdx.x == 0) for (int k = 0; k < K; k++) out[k] = acc[k];
}

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
        hipLaunchKernelGGL(k_spmv3_heavy_seg, (unsigned)m.n_seg, kBlock, 0, c.stream, m.view(), (const uint32_t *)m.seg_row.p, (const uint32_t *)m.seg_no.p, x, m.seg_partial.p);
        hipLaunchKernelGGL(k_spmv3_heavy_combine, (unsigned)((m.n_heavy + 63) / 64), 64, 0, c.stream, (const uint32_t *)m.heavy.p, (const uint32_t *)m.seg_begin.p, m.n_heavy,
                           (const Fr *)m.seg_partial.p, o0, o1, o2, (int)combine, c0, c1, c2);
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
        d.seg_partial.alloc(3 * seg_row.size());
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
__global__ __launch_bounds__(kBlock) void k_gather_strided(const Fr *in, size_t stride, size_t offset, Fr *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i * stride + offset];
}
void dev_gather_strided(DevCtx &c, const Fr *in, size_t stride, size_t offset, Fr *out, size_t n) {
    KScope ks(c, KC_OTHER);
    hipLaunchKernelGGL(k_gather_strided, grid_for(n), kBlock, 0, c.stream, in, stride, offset, out, n);
}

// ------------------------------------------------------------------------------------------------ K2 eq tables
struct FrArgs { Fr v[13]; };
// one workgroup builds eq(r, .) for ell <= 12 by doubling, ping-ponging between two global buffers
__global__ __launch_bounds__(1024) void k_eq_small(FrArgs r, int ell, Fr *out, Fr *tmp) {
    Fr *cur = (ell & 1) ? tmp : out, *nxt = (ell & 1) ? out : tmp;   // after ell swaps the result sits in `out`
    if (threadIdx.x == 0) cur[0] = fr_one();
    __syncthreads();
    size_t size = 1;
    for (int j = 0; j < ell; j++) {
        Fr rj = r.v[j];
        for (size_t k = threadIdx.x; k < size; k += blockDim.x) { Fr v = cur[k], hi = fr_mul(v, rj); nxt[2 * k + 1] = hi; nxt[2 * k] = fr_sub(v, hi); }
        __syncthreads();
        Fr *t = cur; cur = nxt; nxt = t; size *= 2;
    }
}
// out[i] = hi[i >> lo_bits] * lo[i & (2^lo_bits - 1)]  (index bits are MSB-first over r, so the product of two sub-tables is the table)
__global__ __launch_bounds__(kBlock) void k_eq_expand(const Fr *hi, const Fr *lo, int lo_bits, Fr *out, size_t n) {
    size_t mask = ((size_t)1 << lo_bits) - 1;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = fr_mul(hi[i >> lo_bits], lo[i & mask]);
}
void dev_eq_evals(DevCtx &c, const Fr *r, size_t ell, Fr *out, Fr *scratch) {
    auto small = [&](const Fr *rr, int e, Fr *dst, Fr *tmp) {
        FrArgs a; for (int i = 0; i < 13; i++) a.v[i] = i < e ? rr[i] : fr_zero();
        hipLaunchKernelGGL(k_eq_small, 1, 1024, 0, c.stream, a, e, dst, tmp);
    };
    KScope ks(c, KC_EQ);
    if (ell <= 12) { small(r, (int)ell, out, scratch); return; }
    int lo_bits = 12, hi_bits = (int)ell - 12;
    if (hi_bits > 13) throw Error(OTTI_ERR_BAD_ARG, "eq table larger than 2^25");
    Fr *lo = scratch, *hi = scratch + 4096, *tmp = hi + (hi_bits > 12 ? 8192 : 4096);   // scratch >= 3 * 4096 (5 * 4096 for ell = 25)
    small(r + hi_bits, lo_bits, lo, tmp);
    small(r, hi_bits, hi, tmp);                                          // same stream: ordered after the first use of tmp
    size_t n = (size_t)1 << ell;
    hipLaunchKernelGGL(k_eq_expand, grid_for(n), kBlock, 0, c.stream, hi, lo, lo_bits, out, n);
}

// ------------------------------------------------------------------------------------------------ K3/K4/K7 sum-check rounds
struct Pair { Fr lo, hi; };
__device__ __forceinline__ void cubic_accum(Fr (&acc)[3], const Pair &a, const Pair &b, const Pair &c, const Pair &d) {
    // comb = A * (B * C - D) at t = 0, 2, 3 with X(2) = 2 X[hi] - X[lo], X(3) = X(2) + X[hi] - X[lo]
    acc[0] = fr_add(acc[0], fr_mul(a.lo, fr_sub(fr_mul(b.lo, c.lo), d.lo)));
    Fr da = fr_sub(a.hi, a.lo), db = fr_sub(b.hi, b.lo), dc = fr_sub(c.hi, c.lo), dd = fr_sub(d.hi, d.lo);
    Fr a2 = fr_add(a.hi, da), b2 = fr_add(b.hi, db), c2 = fr_add(c.hi, dc), d2 = fr_add(d.hi, dd);
    acc[1] = fr_add(acc[1], fr_mul(a2, fr_sub(fr_mul(b2, c2), d2)));
    Fr a3 = fr_add(a2, da), b3 = fr_add(b2, db), c3 = fr_add(c2, dc), d3 = fr_add(d2, dd);
    acc[2] = fr_add(acc[2], fr_mul(a3, fr_sub(fr_mul(b3, c3), d3)));
}
__device__ __forceinline__ void quad_accum(Fr (&acc)[2], const Pair &a, const Pair &b) {
    acc[0] = fr_add(acc[0], fr_mul(a.lo, b.lo));
    Fr a2 = fr_sub(fr_add(a.hi, a.hi), a.lo), b2 = fr_sub(fr_add(b.hi, b.hi), b.lo);
    acc[1] = fr_add(acc[1], fr_mul(a2, b2));
}
__device__ __forceinline__ Pair load_pair(const Fr *T, size_t i, size_t half) { Pair p; p.lo = T[i]; p.hi = T[i + half]; return p; }
// fold the table of length 4q by r (bound_poly_var_top) for the two entries that form pair i of the folded table
__device__ __forceinline__ Pair fold_pair(Fr *T, size_t i, size_t q, const Fr &r) {
    Fr a = T[i], b = T[i + q], c = T[i + 2 * q], d = T[i + 3 * q];
    Pair p; p.lo = fr_add(a, fr_mul(r, fr_sub(c, a))); p.hi = fr_add(b, fr_mul(r, fr_sub(d, b)));
    T[i] = p.lo; T[i + q] = p.hi;
    return p;
}
template <int K> __device__ __forceinline__ void store_partials(Fr (&acc)[K], Fr *partials) {
    block_reduce<K>(acc);
    if (threadIdx.x == 0) for (int k = 0; k < K; k++) partials[(size_t)blockIdx.x * K + k] = acc[k];
}
// Round sums without a second launch or a stream synchronise: every workgroup publishes its partial sums, the last one to arrive
// (agent-scope counter; release/acquire per the gfx950 inter-workgroup recipe) adds them up, writes the K totals straight into
// pinned host memory and then stores the launch's sequence number into a host-visible flag the prover thread is spinning on.
template <int K> __device__ __forceinline__ void finish_in_kernel(Fr (&acc)[K], const Mailbox &mb) {
    block_reduce<K>(acc);
    if (gridDim.x > 1) {
        if (threadIdx.x == 0) for (int k = 0; k < K; k++) store_words_sc1(&mb.partials[(size_t)blockIdx.x * K + k], acc[k].v, 8);
        if (!arrive_and_check_last(mb.counter, gridDim.x)) return;
        for (int k = 0; k < K; k++) acc[k] = fr_zero();
        for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
            for (int k = 0; k < K; k++) { Fr t; load_words_sc1(t.v, &mb.partials[(size_t)b * K + k], 8); acc[k] = fr_add(acc[k], t); }
        block_reduce<K>(acc);
    }
    if (threadIdx.x == 0) {
        for (int k = 0; k < K; k++) mb.host_results[mb.slot + k] = acc[k];
        __threadfence_system();
        __hip_atomic_store(mb.host_flag, mb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// All of an item's loads are issued before any arithmetic so that their HBM latency is paid once per item, not once per table.
__global__ __launch_bounds__(kBlock) void k_sc_cubic_eval(const Fr *A, const Fr *B, const Fr *C, const Fr *D, size_t half, Mailbox mb) {
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        Pair a, b, c, d;
        a.lo = A[i]; a.hi = A[i + half]; b.lo = B[i]; b.hi = B[i + half]; c.lo = C[i]; c.hi = C[i + half]; d.lo = D[i]; d.hi = D[i + half];
        __builtin_amdgcn_sched_barrier(0);                    // keep the scheduler from sinking the loads next to their uses
        cubic_accum(acc, a, b, c, d);
    }
    finish_in_kernel<3>(acc, mb);
}
__device__ __forceinline__ Pair fold_regs(const Fr &x0, const Fr &x1, const Fr &x2, const Fr &x3, const Fr &r) {
    Pair p; p.lo = fr_add(x0, fr_mul(r, fr_sub(x2, x0))); p.hi = fr_add(x1, fr_mul(r, fr_sub(x3, x1))); return p;
}
__global__ __launch_bounds__(kBlock) void k_sc_cubic_fold_eval(Fr *A, Fr *B, Fr *C, Fr *D, size_t q, Fr r, Mailbox mb) {
    Fr acc[3] = {fr_zero(), fr_zero(), fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        // two load groups of eight elements: all sixteen in flight at once would need the whole register file (one wave per SIMD)
        Fr a0 = A[i], a1 = A[i + q], a2 = A[i + 2 * q], a3 = A[i + 3 * q];
        Fr b0 = B[i], b1 = B[i + q], b2 = B[i + 2 * q], b3 = B[i + 3 * q];
        __builtin_amdgcn_sched_barrier(0);
        Pair a = fold_regs(a0, a1, a2, a3, r); A[i] = a.lo; A[i + q] = a.hi;
        Fr c0 = C[i], c1 = C[i + q], c2 = C[i + 2 * q], c3 = C[i + 3 * q];
        Fr d0 = D[i], d1 = D[i + q], d2 = D[i + 2 * q], d3 = D[i + 3 * q];
        __builtin_amdgcn_sched_barrier(0);
        Pair b = fold_regs(b0, b1, b2, b3, r); B[i] = b.lo; B[i + q] = b.hi;
        Pair c = fold_regs(c0, c1, c2, c3, r); C[i] = c.lo; C[i + q] = c.hi;
        Pair d = fold_regs(d0, d1, d2, d3, r); D[i] = d.lo; D[i + q] = d.hi;
        cubic_accum(acc, a, b, c, d);
    }
    finish_in_kernel<3>(acc, mb);
}
__global__ __launch_bounds__(kBlock) void k_sc_quad_eval(const Fr *A, const Fr *B, size_t half, Mailbox mb) {
    Fr acc[2] = {fr_zero(), fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        Pair a, b;
        a.lo = A[i]; a.hi = A[i + half]; b.lo = B[i]; b.hi = B[i + half];
        __builtin_amdgcn_sched_barrier(0);
        quad_accum(acc, a, b);
    }
    finish_in_kernel<2>(acc, mb);
}
__global__ __launch_bounds__(kBlock) void k_sc_quad_fold_eval(Fr *A, Fr *B, size_t q, Fr r, Mailbox mb) {
    Fr acc[2] = {fr_zero(), fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        Fr a0 = A[i], a1 = A[i + q], a2 = A[i + 2 * q], a3 = A[i + 3 * q];
        Fr b0 = B[i], b1 = B[i + q], b2 = B[i + 2 * q], b3 = B[i + 3 * q];
        __builtin_amdgcn_sched_barrier(0);
        Pair a = fold_regs(a0, a1, a2, a3, r); A[i] = a.lo; A[i + q] = a.hi;
        Pair b = fold_regs(b0, b1, b2, b3, r); B[i] = b.lo; B[i + q] = b.hi;
        quad_accum(acc, a, b);
    }
    finish_in_kernel<2>(acc, mb);
}
__global__ __launch_bounds__(kBlock) void k_fold_top(Fr *Z, size_t half, Fr r) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        Fr a = Z[i], b = Z[i + half]; Z[i] = fr_add(a, fr_mul(r, fr_sub(b, a)));
    }
}
__global__ __launch_bounds__(kBlock) void k_fold_bot(const Fr *Z, Fr *out, size_t half, Fr r) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        Fr a = Z[2 * i], b = Z[2 * i + 1]; out[i] = fr_add(a, fr_mul(r, fr_sub(b, a)));
    }
}
template <int K> static void finish_round(DevCtx &c, int nblocks, int slot) {
    { KScope ks(c, KC_REDUCE); hipLaunchKernelGGL(k_reduce_partials<K>, 1, kBlock, 0, c.stream, (const Fr *)c.partials.p, nblocks, c.results.p + slot); }
    dev_fetch(c, c.results.p + slot, slot, K);
}
Mailbox DevCtx::next_mailbox(int slot) {
    Mailbox mb; mb.partials = partials.p; mb.counter = d_counter.p; mb.host_results = d_results_alias; mb.host_flag = d_flag_alias;
    mb.seq = ++seq; mb.slot = slot; return mb;
}
void DevCtx::wait_ticket(unsigned long long ticket) {
    volatile unsigned long long *f = h_flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        if (*f >= ticket) return;
#if defined(__x86_64__)
        _mm_pause();
#endif
        if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            OTTI_HIP(hipStreamSynchronize(stream));          // surfaces a device fault as an error instead of spinning forever
            if (*f >= ticket) return;
            throw Error(OTTI_ERR_INTERNAL, "sum-check round result never arrived");
        }
    }
}
unsigned long long dev_sc_cubic_eval(DevCtx &c, const Fr *A, const Fr *B, const Fr *C, const Fr *D, size_t len, int slot) {
    size_t half = len / 2; int g = grid_for(half); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_CUBIC); hipLaunchKernelGGL(k_sc_cubic_eval, g, kBlock, 0, c.stream, A, B, C, D, half, mb);
    return mb.seq;
}
unsigned long long dev_sc_cubic_fold_eval(DevCtx &c, Fr *A, Fr *B, Fr *C, Fr *D, size_t len, const Fr &r, int slot) {
    if (len < 4) throw Error(OTTI_ERR_INTERNAL, "fold_eval needs len >= 4");
    size_t q = len / 4; int g = grid_for(q); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_CUBIC); hipLaunchKernelGGL(k_sc_cubic_fold_eval, g, kBlock, 0, c.stream, A, B, C, D, q, r, mb);
    return mb.seq;
}
unsigned long long dev_sc_quad_eval(DevCtx &c, const Fr *A, const Fr *B, size_t len, int slot) {
    size_t half = len / 2; int g = grid_for(half); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_QUAD); hipLaunchKernelGGL(k_sc_quad_eval, g, kBlock, 0, c.stream, A, B, half, mb);
    return mb.seq;
}
unsigned long long dev_sc_quad_fold_eval(DevCtx &c, Fr *A, Fr *B, size_t len, const Fr &r, int slot) {
    if (len < 4) throw Error(OTTI_ERR_INTERNAL, "fold_eval needs len >= 4");
    size_t q = len / 4; int g = grid_for(q); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_QUAD); hipLaunchKernelGGL(k_sc_quad_fold_eval, g, kBlock, 0, c.stream, A, B, q, r, mb);
    return mb.seq;
}
void dev_fold_top(DevCtx &c, Fr *Z, size_t len, const Fr &r) { size_t h = len / 2; if (h) hipLaunchKernelGGL(k_fold_top, grid_for(h), kBlock, 0, c.stream, Z, h, r); }
void dev_fold_bot(DevCtx &c, const Fr *Z, Fr *out, size_t len, const Fr &r) { size_t h = len / 2; if (h) hipLaunchKernelGGL(k_fold_bot, grid_for(h), kBlock, 0, c.stream, Z, out, h, r); }

__global__ __launch_bounds__(kBlock) void k_dot(const Fr *a, const Fr *b, size_t n, Fr *partials) {
    Fr acc[1] = {fr_zero()};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc[0] = fr_add(acc[0], fr_mul(a[i], b[i]));
    store_partials<1>(acc, partials);
}
void dev_dot(DevCtx &c, const Fr *a, const Fr *b, size_t n, int slot) {
    int g = grid_for(n);
    hipLaunchKernelGGL(k_dot, g, kBlock, 0, c.stream, a, b, n, c.partials.p);
    finish_round<1>(c, g, slot);
}

// ------------------------------------------------------------------------------------------------ K9 DensePolynomial::bound
__global__ __launch_bounds__(kBlock) void k_poly_bound_slab(const Fr *Z, size_t L, size_t R, const Fr *Lv, size_t rows_per_slab, Fr *scratch) {
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= R) return;
    size_t i0 = blockIdx.y * rows_per_slab, i1 = min(L, i0 + rows_per_slab);
    Fr acc = fr_zero();
    for (size_t i = i0; i < i1; i++) acc = fr_add(acc, fr_mul(Lv[i], Z[i * R + j]));
    scratch[(size_t)blockIdx.y * R + j] = acc;
}
__global__ __launch_bounds__(kBlock) void k_colsum(const Fr *scratch, size_t slabs, size_t R, Fr *out) {
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= R) return;
    Fr acc = scratch[j];
    for (size_t s = 1; s < slabs; s++) acc = fr_add(acc, scratch[s * R + j]);
    out[j] = acc;
}
void dev_poly_bound(DevCtx &c, const Fr *Z, size_t L, size_t R, const Fr *Lv, Fr *out, Fr *scratch) {
    size_t slabs = std::min<size_t>(L, 64), rps = (L + slabs - 1) / slabs;
    KScope ks(c, KC_BOUND);
    dim3 grid((unsigned)((R + kBlock - 1) / kBlock), (unsigned)slabs);
    hipLaunchKernelGGL(k_poly_bound_slab, grid, kBlock, 0, c.stream, Z, L, R, Lv, rps, scratch);
    hipLaunchKernelGGL(k_colsum, (unsigned)((R + kBlock - 1) / kBlock), kBlock, 0, c.stream, (const Fr *)scratch, slabs, R, out);
}

// ------------------------------------------------------------------------------------------------ K8 fixed-base MSM
// One workgroup sums a chunk of one row's terms.  Phase 1 stages the chunk's scalars in LDS as s' = raw(s) + K with
// K = sum_w 2^(c-1+cw): the signed radix-2^c digit of window w is then (window w of s') - 2^(c-1), with no carry chain between
// windows, so any thread can take any (term, window) pair.  Phase 2: thread (term lane tl, window w) walks the terms tl, tl+lanes, ..
// and adds the table entry |d| * 2^(cw) * P[base] (affine Niels, 96-byte gather from HBM/L2; negated in registers when d < 0) into
// its own accumulator with one 7-multiply mixed addition per pair.  Phase 3: LDS tree over the 256 accumulators.
constexpr int kMsmMaxChunk = 1024;             // terms per workgroup (LDS: 36 B each)
constexpr int kMsmBulkChunk = 512;             // bulk launches: terms per workgroup, and
constexpr int kMsmListCap = (kMsmBulkChunk + 8) * 16;   // their (term, window) work-list entries (2 B each): (chunk + extras) * W must fit
static_assert(kMsmMaxChunk + 8 <= 2048, "work-list entries pack the term index in 11 bits");
// Bullet-reduction round fused into the MSM launch: the scalars of rows L (0) and R (1) are not read from memory but derived in phase 1
// from the round state (a, b: the two folded vectors; s: coefficients of the original generators), after applying the previous
// round's challenge.  State is ping-ponged (read *_in, write *_out) so that no workgroup of the launch reads what another one writes.
struct BulletArgs { int on, fold; uint32_t n; const Fr *a_in, *b_in, *s_in; Fr *a_out, *b_out, *s_out; Fr u, uinv; };
struct MsmArgs {
    const Niels *table; int c, W; uint32_t E; int lanes;            // lanes = 256 / W term lanes
    const Fr *dense; size_t stride, n_dense; uint32_t chunk, nchunks;
    const Fr *extra_s; uint32_t extra_base[8]; int n_extra;
    uint32_t K[9];                                                  // recoding constant (288 bits)
    Pt *partial;
    // fused finish (rows <= 2, 1 < nchunks <= 128): the last workgroup to arrive sums every row's partials and mails the extended
    // row sums to pinned host memory, then raises the host flag — no finish launch, no copy, no stream synchronise
    int fuse; uint32_t rows; unsigned *counter; Pt *host_pts; unsigned long long *host_flag; unsigned long long seq;
    BulletArgs bul;
};
__device__ __forceinline__ Fr bullet_fold_a(const BulletArgs &U, size_t x) { return U.fold ? fr_add(fr_mul(U.a_in[x], U.u), fr_mul(U.uinv, U.a_in[U.n + x])) : U.a_in[x]; }
__device__ __forceinline__ Fr bullet_fold_b(const BulletArgs &U, size_t x) { return U.fold ? fr_add(fr_mul(U.b_in[x], U.uinv), fr_mul(U.u, U.b_in[U.n + x])) : U.b_in[x]; }
__device__ __forceinline__ Fr bullet_fold_s(const BulletArgs &U, size_t j) { return U.fold ? fr_mul(U.s_in[j], ((j & (2 * (size_t)U.n - 1)) < U.n) ? U.uinv : U.u) : U.s_in[j]; }
// kKind = MSM_BULK: the bulk launches (a commitment: many rows, every workgroup a full chunk) — no bullet bookkeeping, no fused finish.
// kKind = MSM_BULK_SPARSE: the same for scalars that are mostly small numbers (compacted work list, see phase 2).
// kKind = MSM_SMALL: the one/two-row launches of the evaluation proof, latency-bound, with both.  Separate instantiations also keep them
// apart in profiles (k_msm_rows<false> is the kernel bench.py's roofline object is about).
enum { MSM_BULK = 0, MSM_SMALL = 1, MSM_BULK_SPARSE = 2 };
template <int kKind> __global__ __launch_bounds__(kBlock) void k_msm_rows(MsmArgs A) {
    constexpr bool kSmall = kKind == MSM_SMALL;
    // the recoded scalars (phases 1-2) and the reduction tree (phase 3) never live at the same time: one LDS region for both
    constexpr size_t kRawBytes = ((kKind == MSM_BULK_SPARSE ? kMsmBulkChunk : kMsmMaxChunk) + 8) * 9 * sizeof(uint32_t), kTreeBytes = (kBlock / 2) * sizeof(P10);
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[kRawBytes > kTreeBytes ? kRawBytes : kTreeBytes];
    __shared__ uint32_t s_base[8];
    uint32_t *s_raw = reinterpret_cast<uint32_t *>(s_mem);
    P10 *sm = reinterpret_cast<P10 *>(s_mem);
    const size_t row = blockIdx.y; const uint32_t chunk_id = blockIdx.x;
    const size_t j0 = (size_t)chunk_id * A.chunk;
    const uint32_t n_here = (uint32_t)min((size_t)A.chunk, A.n_dense - j0);
    const uint32_t n_ex = chunk_id == 0 ? (uint32_t)A.n_extra : 0u;
    // ---- phase 1: recoded scalars into LDS
    __shared__ Fr s_dot;                                       // bullet mode: c_L / c_R of this row (chunk 0 only)
    const bool bullet = kSmall && A.bul.on;
    if (bullet) {
        const BulletArgs &U = A.bul; const size_t n = U.n, h = n / 2;
        if (row == 0) {                                        // persist the folded state for the next round (each element written once)
            const size_t cx = (n + gridDim.x - 1) / gridDim.x, x0 = (size_t)chunk_id * cx;
            for (size_t x = x0 + threadIdx.x; x < min(n, x0 + cx); x += blockDim.x) { U.a_out[x] = bullet_fold_a(U, x); U.b_out[x] = bullet_fold_b(U, x); }
            for (uint32_t t = threadIdx.x; t < n_here; t += blockDim.x) U.s_out[j0 + t] = bullet_fold_s(U, j0 + t);
        }
        if (chunk_id == 0) {                                   // c_L = <a_L, b_R> (row 0), c_R = <a_R, b_L> (row 1)
            Fr acc[1] = {fr_zero()};
            for (size_t x = threadIdx.x; x < h; x += blockDim.x)
                acc[0] = fr_add(acc[0], row == 0 ? fr_mul(bullet_fold_a(U, x), bullet_fold_b(U, h + x)) : fr_mul(bullet_fold_a(U, h + x), bullet_fold_b(U, x)));
            block_reduce<1>(acc);
            if (threadIdx.x == 0) s_dot = acc[0];
            __syncthreads();
        }
    }
    for (uint32_t t = threadIdx.x; t < n_here + n_ex; t += blockDim.x) {
        Fr sc;
        if (t >= n_here) sc = (bullet && t == n_here) ? s_dot : A.extra_s[row * A.n_extra + (t - n_here)];
        else if (bullet) {
            const BulletArgs &U = A.bul; const size_t j = j0 + t, n = U.n, h = n / 2, i = j & (n - 1);
            // L = <a_L, G_R>: generator slots of the upper half, paired with a[i - h];  R = <a_R, G_L>: lower half with a[i + h]
            if (row == 0) sc = i >= h ? fr_mul(bullet_fold_a(U, i - h), bullet_fold_s(U, j)) : fr_zero();
            else sc = i < h ? fr_mul(bullet_fold_a(U, i + h), bullet_fold_s(U, j)) : fr_zero();
        } else sc = A.dense[row * A.stride + j0 + t];
        Fr raw = fr_to_raw(sc);
        uint64_t cy = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { cy += (uint64_t)raw.v[i] + A.K[i]; s_raw[t * 9 + i] = (uint32_t)cy; cy >>= 32; }
        s_raw[t * 9 + 8] = (uint32_t)cy + A.K[8];
    }
    if (threadIdx.x < 8) s_base[threadIdx.x] = A.extra_base[threadIdx.x];
    __syncthreads();
    // ---- phase 2: one mixed addition per (term, window) pair, in 26/25-bit limbs (fp10.h)
    P10 acc = p10_identity();
    const int w = threadIdx.x % A.W, tl = threadIdx.x / A.W;
    if constexpr (kKind == MSM_BULK_SPARSE) {
        // First compact the pairs whose digit is non-zero into an LDS work list (wave-aggregated append), then every thread takes
        // list entries round-robin.  A witness produced by a compiler is mostly small numbers (bits, counters, fixed-point values):
        // with c = 16 a 64-bit scalar has 4-5 non-zero digits out of 16, and with the fixed (term, window) mapping below the lanes
        // of its empty windows would idle while the others work.  (For uniform scalars the list would simply be all pairs, at the
        // price of half-size chunks; the host picks this variant from the witness's share of small values.)
        __shared__ uint16_t s_list[kMsmListCap];
        __shared__ uint32_t s_count;
        if (threadIdx.x == 0) s_count = 0;
        __syncthreads();
        const uint32_t n_tot = n_here + n_ex, iters = (n_tot + A.lanes - 1) / A.lanes;
        const int pos = w * A.c, limb = pos >> 5, off = pos & 31;
        const uint32_t mask = (1u << A.c) - 1u; const int half = 1 << (A.c - 1);
        const unsigned lane = threadIdx.x & 63;
        for (uint32_t it = 0; it < iters; it++) {
            const uint32_t t = (uint32_t)tl + it * (uint32_t)A.lanes;
            bool nz = false;
            if (tl < A.lanes && t < n_tot) {
                uint64_t x = s_raw[t * 9 + limb];
                if (limb < 8) x |= (uint64_t)s_raw[t * 9 + limb + 1] << 32;
                nz = ((uint32_t)(x >> off) & mask) != (uint32_t)half;
            }
            const unsigned long long bal = __ballot(nz);
            uint32_t base_pos = 0;
            if (lane == 0 && bal) base_pos = atomicAdd(&s_count, (uint32_t)__popcll(bal));
            base_pos = __shfl(base_pos, 0);
            if (nz) s_list[base_pos + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)((t << 5) | (uint32_t)w);
        }
        __syncthreads();
        const uint32_t count = s_count;
        const size_t WE = (size_t)A.W * A.E;
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
            const uint32_t e16 = s_list[i], t = e16 >> 5, ww = e16 & 31u;
            const int p2 = (int)ww * A.c, l2 = p2 >> 5, o2 = p2 & 31;
            uint64_t x = s_raw[t * 9 + l2];
            if (l2 < 8) x |= (uint64_t)s_raw[t * 9 + l2 + 1] << 32;
            const int d = (int)((uint32_t)(x >> o2) & mask) - half;
            const size_t base = t < n_here ? j0 + t : (size_t)s_base[t - n_here];
            const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            N10 e = n10_unpack(A.table[base * WE + (size_t)ww * A.E + (mag - 1)]);
            if (d < 0) e = n10_negate(e);
            acc = p10_madd(acc, e);
        }
    } else if (tl < A.lanes) {
        const int pos = w * A.c, limb = pos >> 5, off = pos & 31;
        const uint32_t mask = (1u << A.c) - 1u; const int half = 1 << (A.c - 1);
        const size_t WE = (size_t)A.W * A.E;
        for (uint32_t t = tl; t < n_here + n_ex; t += A.lanes) {
            uint64_t x = s_raw[t * 9 + limb];
            if (limb < 8) x |= (uint64_t)s_raw[t * 9 + limb + 1] << 32;
            int d = (int)((uint32_t)(x >> off) & mask) - half;
            if (d == 0) continue;
            size_t base = t < n_here ? j0 + t : (size_t)s_base[t - n_here];
            uint32_t mag = (uint32_t)(d < 0 ? -d : d);
            N10 e = n10_unpack(A.table[base * WE + (size_t)w * A.E + (mag - 1)]);
            if (d < 0) e = n10_negate(e);
            acc = p10_madd(acc, e);
        }
    }
    // ---- phase 3: LDS tree (reuses the scalar region: everyone must be done reading it)
    __syncthreads();
    const F10 d2 = f10_const(fp_2D());
    for (int sft = kBlock / 2; sft >= 1; sft >>= 1) {
        if ((int)threadIdx.x >= sft && (int)threadIdx.x < 2 * sft) sm[threadIdx.x - sft] = acc;
        __syncthreads();
        if ((int)threadIdx.x < sft) acc = p10_add(acc, sm[threadIdx.x], d2);
        __syncthreads();
    }
    if (!kSmall || !A.fuse) { if (threadIdx.x == 0) A.partial[row * A.nchunks + chunk_id] = p10_pack(acc); return; }
    if (threadIdx.x == 0) { Pt pk = p10_pack(acc); store_words_sc1(&A.partial[row * A.nchunks + chunk_id], pk.X.v, 32); }
    if (!arrive_and_check_last(A.counter, gridDim.x * gridDim.y)) return;
    {   // 128 threads per row: one partial each, 7-level tree
        const uint32_t r = threadIdx.x >> 7, idx = threadIdx.x & 127;
        P10 sum = p10_identity();
        if (r < A.rows && idx < A.nchunks) { Pt pk; load_words_sc1(pk.X.v, &A.partial[(size_t)r * A.nchunks + idx], 32); sum = p10_unpack(pk); }
        for (int sft = 64; sft >= 1; sft >>= 1) {
            if ((int)idx >= sft && (int)idx < 2 * sft) sm[r * 64 + idx - sft] = sum;
            __syncthreads();
            if ((int)idx < sft) sum = p10_add(sum, sm[r * 64 + idx], d2);
            __syncthreads();
        }
        if (idx == 0 && r < A.rows) { A.host_pts[r] = p10_pack(sum); __threadfence_system(); }
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            __hip_atomic_store(A.host_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// one wave per row: sum the row's chunk partials into one extended point
__global__ __launch_bounds__(64) void k_msm_finish(const Pt *partial, uint32_t nchunks, size_t rows, Pt *final_pts) {
    __shared__ P10 sm[32];
    const size_t row = blockIdx.x;
    const F10 d2 = f10_const(fp_2D());
    P10 acc = p10_identity();
    for (uint32_t k = threadIdx.x; k < nchunks; k += 64) acc = p10_add(acc, p10_unpack(partial[row * nchunks + k]), d2);
    for (int sft = 32; sft >= 1; sft >>= 1) {
        if ((int)threadIdx.x >= sft && (int)threadIdx.x < 2 * sft) sm[threadIdx.x - sft] = acc;
        __syncthreads();
        if ((int)threadIdx.x < sft) acc = p10_add(acc, sm[threadIdx.x], d2);
        __syncthreads();
    }
    if (threadIdx.x == 0) final_pts[row] = p10_pack(acc);
}
// RFC 9496 encode, one lane per point (the inverse square root is a ~265-multiplication dependent chain: pack 64 rows per wave)
__global__ __launch_bounds__(64) void k_encode_points(const Pt *pts, const Pt *addend, size_t n, uint8_t *out32) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    P10 p = p10_unpack(pts[i]);
    if (addend) p = p10_add(p, p10_unpack(addend[i]), f10_const(fp_2D()));
    uint8_t enc[32]; p10_encode(enc, p);
    uint32_t *o = (uint32_t *)(out32 + 32 * i);
    for (int k = 0; k < 8; k++) o[k] = (uint32_t)enc[4 * k] | ((uint32_t)enc[4 * k + 1] << 8) | ((uint32_t)enc[4 * k + 2] << 16) | ((uint32_t)enc[4 * k + 3] << 24);
}
static unsigned long long msm_launch(DevCtx &c, const DeviceGens &g, const Fr *dense, size_t stride, size_t n_dense, size_t rows, const Fr *extra_s,
                                     const uint32_t *extra_base, size_t n_extra, int mode, const Pt *addend, const BulletArgs *bul, bool sparse_hint);
unsigned long long dev_msm_rows(DevCtx &c, const DeviceGens &g, const Fr *dense, size_t stride, size_t n_dense, size_t rows, const Fr *extra_s,
                                const uint32_t *extra_base, size_t n_extra, int mode, const Pt *addend, bool sparse_hint) {
    return msm_launch(c, g, dense, stride, n_dense, rows, extra_s, extra_base, n_extra, mode, addend, nullptr, sparse_hint);
}
// share of the n scalars whose canonical value is below 2^128 (what a compiled circuit's witness is mostly made of)
__global__ __launch_bounds__(kBlock) void k_count_small(const Fr *z, size_t n, unsigned long long *count) {
    unsigned mine = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr r = fr_to_raw(z[i]);
        mine += (r.v[4] | r.v[5] | r.v[6] | r.v[7]) == 0 ? 1u : 0u;
    }
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_down(mine, o);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, (unsigned long long)mine);
}
double dev_small_fraction(DevCtx &c, const Fr *z, size_t n) {
    if (!n) return 0.0;
    DevBuf<unsigned long long> cnt(1);
    OTTI_HIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), c.stream));
    hipLaunchKernelGGL(k_count_small, grid_for(n), kBlock, 0, c.stream, z, n, cnt.p);
    unsigned long long h = 0;
    OTTI_HIP(hipMemcpyAsync(&h, cnt.p, sizeof h, hipMemcpyDeviceToHost, c.stream));
    OTTI_HIP(hipStreamSynchronize(c.stream));
    return (double)h / (double)n;
}
unsigned long long dev_bullet_round(DevCtx &c, const DeviceGens &g, size_t R, size_t n_cur, bool fold, const Fr &u, const Fr &u_inv, const Fr *a_in,
                                    const Fr *b_in, const Fr *s_in, Fr *a_out, Fr *b_out, Fr *s_out, const Fr *extra_s, const uint32_t *extra_base) {
    BulletArgs U; U.on = 1; U.fold = fold ? 1 : 0; U.n = (uint32_t)n_cur; U.a_in = a_in; U.b_in = b_in; U.s_in = s_in;
    U.a_out = a_out; U.b_out = b_out; U.s_out = s_out; U.u = u; U.uinv = u_inv;
    return msm_launch(c, g, nullptr, 0, R, 2, extra_s, extra_base, 2, MSM_COMPRESSED, nullptr, &U, false);
}
static unsigned long long msm_launch(DevCtx &c, const DeviceGens &g, const Fr *dense, size_t stride, size_t n_dense, size_t rows, const Fr *extra_s,
                                     const uint32_t *extra_base, size_t n_extra, int mode, const Pt *addend, const BulletArgs *bul, bool sparse_hint) {
    const bool raw_points = mode == MSM_RAW;
    if (n_extra > 8) throw Error(OTTI_ERR_INTERNAL, "msm: too many extra terms");
    if (!rows) return 0;
    MsmArgs A;
    A.table = g.table.p; A.c = g.c; A.W = g.W; A.E = (uint32_t)g.E; A.lanes = kBlock / g.W;
    A.dense = dense; A.stride = stride; A.n_dense = n_dense; A.extra_s = extra_s; A.n_extra = (int)n_extra;
    for (int i = 0; i < 8; i++) A.extra_base[i] = i < (int)n_extra ? extra_base[i] : 0;
    // chunking: aim for >= 1024 workgroups (4 per CU) but keep at least one term per term lane and at most kMsmMaxChunk per workgroup
    size_t nchunks = std::max<size_t>(1, (1024 + rows - 1) / rows);
    nchunks = std::min(nchunks, std::max<size_t>(1, n_dense / (size_t)A.lanes));
    if (!n_dense) nchunks = 1;
    // bulk launches keep a (term, window) work list in LDS: (chunk + extras) * W <= kMsmListCap;  W <= 32 there (5-bit window field)
    const bool bulk = rows * n_dense >= ((size_t)1 << 16) && !bul;
    const bool sparse = bulk && sparse_hint && g.W <= 32;
    const size_t max_chunk = sparse ? std::min<size_t>(kMsmBulkChunk, (size_t)kMsmListCap / (size_t)g.W - n_extra) : (size_t)kMsmMaxChunk;
    nchunks = std::max(nchunks, (n_dense + max_chunk - 1) / max_chunk);
    size_t chunk = n_dense ? (n_dense + nchunks - 1) / nchunks : 1;
    nchunks = n_dense ? (n_dense + chunk - 1) / chunk : 1;
    A.chunk = (uint32_t)chunk; A.nchunks = (uint32_t)nchunks;
    for (int i = 0; i < 9; i++) A.K[i] = 0;
    for (int w = 0; w < g.W; w++) { int bit = g.c - 1 + g.c * w; A.K[bit >> 5] |= 1u << (bit & 31); }
    c.ensure_points(rows, nchunks);
    A.partial = c.msm_partial.p;
    A.fuse = (!bulk && mode == MSM_COMPRESSED && !addend && rows <= 2 && nchunks > 1 && nchunks <= 128) ? 1 : 0;
    if (bul) A.bul = *bul; else { memset(&A.bul, 0, sizeof A.bul); }
    A.rows = (uint32_t)rows; A.counter = c.d_counter2.p; A.host_pts = c.d_pts_alias; A.host_flag = c.d_flag_alias; A.seq = A.fuse ? ++c.seq : 0;
    dim3 grid((unsigned)nchunks, (unsigned)rows);
    {
        KScope ks(c, bulk ? KC_MSM_ROWS : KC_MSM_SMALL);
        if (sparse) hipLaunchKernelGGL(k_msm_rows<MSM_BULK_SPARSE>, grid, kBlock, 0, c.stream, A);
        else if (bulk) hipLaunchKernelGGL(k_msm_rows<MSM_BULK>, grid, kBlock, 0, c.stream, A);
        else hipLaunchKernelGGL(k_msm_rows<MSM_SMALL>, grid, kBlock, 0, c.stream, A);
    }
    if (A.fuse) { c.pending_host_encode = rows; return A.seq; }
    // rows with a single chunk need no finish pass: their partial IS the row sum
    const Pt *finals = c.msm_partial.p;
    if (nchunks > 1) {
        KScope ks(c, KC_MSM_FINISH);
        hipLaunchKernelGGL(k_msm_finish, (unsigned)rows, 64, 0, c.stream, (const Pt *)c.msm_partial.p, (uint32_t)nchunks, rows, c.msm_final.p);
        finals = c.msm_final.p;
    }
    if (mode == MSM_KEEP) {
        if (c.msm_keep.n < rows) c.msm_keep.alloc(rows);
        OTTI_HIP(hipMemcpyAsync(c.msm_keep.p, finals, rows * sizeof(Pt), hipMemcpyDeviceToDevice, c.stream));
        c.pending_host_encode = 0;
    } else if (raw_points) {
        if (rows > kHostPtsCap) throw Error(OTTI_ERR_INTERNAL, "msm: too many raw rows");
        OTTI_HIP(hipMemcpyAsync(c.h_pts, finals, rows * sizeof(Pt), hipMemcpyDeviceToHost, c.stream));
        c.pending_host_encode = 0;
    } else if (rows > kHostEncodeRows || addend) {
        KScope ks(c, KC_MSM_FINISH);
        hipLaunchKernelGGL(k_encode_points, (unsigned)((rows + 63) / 64), 64, 0, c.stream, finals, addend, rows, c.d_points.p);
        OTTI_HIP(hipMemcpyAsync(c.h_points, c.d_points.p, rows * 32, hipMemcpyDeviceToHost, c.stream));
        c.pending_host_encode = 0;
    } else {
        // a handful of points: the dependent inverse-square-root chain runs ~30x faster on a host core than on one GPU lane
        OTTI_HIP(hipMemcpyAsync(c.h_pts, finals, rows * sizeof(Pt), hipMemcpyDeviceToHost, c.stream));
        c.pending_host_encode = rows;
    }
    return 0;
}
void DevCtx::wait_points(unsigned long long ticket) {
    if (!ticket) { sync(); return; }
    wait_ticket(ticket);
    encode_pending();
}
void DevCtx::sync() {
    OTTI_HIP(hipStreamSynchronize(stream));
    encode_pending();
}
void DevCtx::encode_pending() {
    if (pending_host_encode >= 2) {
        SpinPool &pool = SpinPool::get(); const int nt = std::min<int>(pool.workers() + 1, (int)pending_host_encode);
        const size_t n = pending_host_encode;
        std::vector<std::function<void()>> tasks(nt);
        for (int t = 0; t < nt; t++) tasks[t] = [this, t, nt, n] { for (size_t i = t; i < n; i += nt) pt_encode(h_points + 32 * i, h_pts[i]); };
        pool.parallel(tasks.data(), nt);
    } else if (pending_host_encode == 1) pt_encode(h_points, h_pts[0]);
    pending_host_encode = 0;
}

// table build.  Row (base b, window w) holds d * B for d = 1..E with B = 2^(cw) * P[b].  Rows are cut into blocks of T entries:
// k_table_starts (one thread per row) walks the block starts (kT+1) * B; k_table_fill (one thread per block) fills its T extended
// points by repeated addition of B, then turns them into affine Niels form with one batch inversion over the block.
__global__ __launch_bounds__(kBlock) void k_table_starts(const Pt *bases, size_t nb, int c, int W, size_t nblk, int lgT, Pt *starts) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nb * W) return;
    size_t b = t / W; int w = (int)(t % W);
    Pt B = bases[b];
    for (int k = 0; k < c * w; k++) B = pt_dbl(B);
    Pt TB = B;
    for (int k = 0; k < lgT; k++) TB = pt_dbl(TB);
    Pt acc = B; Pt *row = starts + t * nblk;
    row[0] = acc;
    for (size_t k = 1; k < nblk; k++) { acc = pt_add(acc, TB); row[k] = acc; }
}
__global__ __launch_bounds__(kBlock) void k_table_fill(const Pt *starts, size_t nrows, size_t nblk, size_t T, Pt *tmp, Niels *out) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nrows * nblk) return;
    const size_t r = t / nblk;
    const Pt B = starts[r * nblk];
    Pt acc = starts[t];
    Pt *blk = tmp + t * T; Niels *oblk = out + t * T;                                      // row r, block k: entries r*E + k*T ..
    Fp prod = fp_one();
    for (size_t d = 0; d < T; d++) {
        if (d) acc = pt_add(acc, B);
        Pt p = acc; p.T = prod;                                                             // T is not needed for the affine form: park the prefix product there
        blk[d] = p; prod = fp_mul(prod, acc.Z);
    }
    Fp inv = fp_inv(prod);
    for (size_t d = T; d-- > 0;) { Pt p = blk[d]; Fp zinv = fp_mul(inv, p.T); inv = fp_mul(inv, p.Z); oblk[d] = pt_to_niels(p, zinv); }
}
std::shared_ptr<DeviceGens> build_device_gens(const Gens &g, int c) {
    DevCtx &ctx = DevCtx::get();
    auto d = std::make_shared<DeviceGens>();
    d->c = c; d->W = 253 / c + 1; d->E = (size_t)1 << (c - 1); d->nbases = g.P.size();
    const size_t per_base = (size_t)d->W * d->E;
    const int lgT = std::min(6, c - 1); const size_t T = (size_t)1 << lgT, nblk = d->E / T;
    d->table.alloc(d->nbases * per_base);
    DevBuf<Pt> bases(d->nbases), starts(d->nbases * d->W * nblk);
    OTTI_HIP(hipMemcpy(bases.p, g.P.data(), d->nbases * sizeof(Pt), hipMemcpyHostToDevice));
    { size_t n = d->nbases * d->W; hipLaunchKernelGGL(k_table_starts, (unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, ctx.stream, (const Pt *)bases.p, d->nbases, c, d->W, nblk, lgT, starts.p); }
    size_t chunk = std::max<size_t>(1, ((size_t)4 << 30) / (per_base * sizeof(Pt)));            // <= 4 GiB of extended temporaries
    chunk = std::min(chunk, d->nbases);
    DevBuf<Pt> tmp(chunk * per_base);
    for (size_t b0 = 0; b0 < d->nbases; b0 += chunk) {
        size_t nb = std::min(chunk, d->nbases - b0), nrows = nb * d->W, nthreads = nrows * nblk;
        hipLaunchKernelGGL(k_table_fill, (unsigned)((nthreads + kBlock - 1) / kBlock), kBlock, 0, ctx.stream, (const Pt *)(starts.p + b0 * d->W * nblk), nrows, nblk, T,
                           tmp.p, d->table.p + b0 * per_base);
    }
    ctx.sync();
    return d;
}

// ------------------------------------------------------------------------------------------------ K10 bullet reduction bookkeeping
// Instead of folding the generator vector (n/2 two-scalar multiplications per round upstream), keep the ORIGINAL generators and a
// coefficient vector s with G^(k)_i = sum_{j = i mod n} s[j] * P[j]; L and R of each round are then fixed-base MSM rows over P.
__global__ __launch_bounds__(1024) void k_bullet_step(Fr *a, Fr *b, Fr *s, size_t R, size_t n, int fold_first, Fr u, Fr uinv, Fr *rows, Fr *extra_out) {
    if (fold_first) {
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
            a[i] = fr_add(fr_mul(a[i], u), fr_mul(uinv, a[n + i]));
            b[i] = fr_add(fr_mul(b[i], uinv), fr_mul(u, b[n + i]));
        }
        for (size_t j = threadIdx.x; j < R; j += blockDim.x) s[j] = fr_mul(s[j], ((j & (2 * n - 1)) < n) ? uinv : u);
        __syncthreads();
    }
    if (n < 2) return;
    size_t h = n / 2;
    Fr acc[2] = {fr_zero(), fr_zero()};
    for (size_t i = threadIdx.x; i < h; i += blockDim.x) {
        acc[0] = fr_add(acc[0], fr_mul(a[i], b[h + i]));       // c_L = <a_L, b_R>
        acc[1] = fr_add(acc[1], fr_mul(a[h + i], b[i]));       // c_R = <a_R, b_L>
    }
    block_reduce<2>(acc);
    if (threadIdx.x == 0) { extra_out[0] = acc[0]; extra_out[2] = acc[1]; }
    for (size_t j = threadIdx.x; j < R; j += blockDim.x) {
        size_t i = j & (n - 1);
        Fr sj = s[j];
        rows[j] = i >= h ? fr_mul(a[i - h], sj) : fr_zero();       // L = <a_L, G_R> : generator slots in the upper half
        rows[R + j] = i < h ? fr_mul(a[i + h], sj) : fr_zero();    // R = <a_R, G_L>
    }
}
void dev_bullet_step(DevCtx &c, Fr *a, Fr *b, Fr *s, size_t R, size_t n_cur, bool fold_first, const Fr &u, const Fr &u_inv, Fr *rows, Fr *extra_out) {
    KScope ks(c, KC_BULLET);
    hipLaunchKernelGGL(k_bullet_step, 1, 1024, 0, c.stream, a, b, s, R, n_cur, (int)fold_first, u, u_inv, rows, extra_out);
}

}  // namespace otti
