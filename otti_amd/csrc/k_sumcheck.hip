// K2 eq tables, K3/K4/K5/K7 sum-check rounds and folds, K9 DensePolynomial::bound.
#include "kernels_common.h"

namespace otti {

// ------------------------------------------------------------------------------------------------ K2 eq tables
struct FrArgs { Fr v[13]; };
// eq(r, .) over n <= 13 variables, index bits MSB-first over r: either the full table (out[0 .. 2^n)) or the "pyramid" of the tables over the
// LAST k variables, k = 0 .. n (level k, 2^k entries, at out + 2^k - 1; level k prepends r[n-k] as the new most significant index bit) that
// the sum-checks read their eq factors from.  One level at a time in one workgroup was 12-13 dependent steps through global memory and, at
// 8191 products, the throughput of ONE CU: 16-26 us per launch, 41 of them on a SNARK proof's sequential path.  Here every workgroup builds
// the two SMALL pyramids — over the last six variables and over the ones above them (at most seven) — side by side in LDS (six or seven
// dependent steps, nine-limb products), and the large levels are outer products of the two, dealt out over all workgroups:
//     level k > 6:  T_k[i] = U_(k-6)[i >> 6] * T_6[i & 63].
// Up to two jobs per launch (blockIdx.y): the pair of pyramids every layer needs, or the two factors of a table over more than 13 variables.
struct EqJob { FrArgs r; int n; int pyramid; Fr *out; };
constexpr int kEqLow = 6, kEqTreeBlock = 256, kEqTreeGroups = 16;
__device__ __forceinline__ Fr eq_mul9(const Fr &x, const Fr &y) { return fr9_pack_lt2l(fr9_mul(fr9_unpack5(x), fr9_unpack(y))); }   // 32 x y / 2^261 + l < 1.1 l
__global__ __launch_bounds__(kEqTreeBlock) void k_eq_tree(EqJob j0, EqJob j1) {
    const EqJob &J = blockIdx.y ? j1 : j0;
    const int n = J.n, a = n < kEqLow ? n : kEqLow, b = n - a;
    __shared__ Fr sT[2 << kEqLow], sU[256];                   // pyramids over the last a variables and over the b before them (level k at 2^k - 1)
    if (threadIdx.x == 0) { sT[0] = fr_one(); sU[0] = fr_one(); }
    __syncthreads();
    for (int s = 1; s <= (a > b ? a : b); s++) {
        const int half = 1 << (s - 1), t = (int)threadIdx.x;
        if (s <= a && t < half) { const Fr o = sT[half - 1 + t], hi = eq_mul9(o, J.r.v[n - s]); sT[2 * half - 1 + half + t] = hi; sT[2 * half - 1 + t] = fr_sub(o, hi); }
        if (s <= b && t >= 64 && t < 64 + half) { const int i = t - 64; const Fr o = sU[half - 1 + i], hi = eq_mul9(o, J.r.v[n - a - s]); sU[2 * half - 1 + half + i] = hi; sU[2 * half - 1 + i] = fr_sub(o, hi); }
        __syncthreads();
    }
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nthr = (size_t)gridDim.x * blockDim.x, lowmask = ((size_t)1 << a) - 1;
    const Fr *Ta = sT + lowmask;                              // level a of T
    if (J.pyramid) {
        const size_t small = ((size_t)2 << a) - 1, total = ((size_t)2 << n) - 1;       // levels 0 .. a as they stand; the rest by products
        if (blockIdx.x == 0) for (size_t g = threadIdx.x; g < small; g += blockDim.x) J.out[g] = sT[g];
        for (size_t g = small + tid; g < total; g += nthr) {
            const int k = 63 - __builtin_clzll((unsigned long long)(g + 1));           // level of pyramid entry g
            const size_t i = g + 1 - ((size_t)1 << k);
            J.out[g] = eq_mul9(sU[((size_t)1 << (k - a)) - 1 + (i >> a)], Ta[i & lowmask]);
        }
    } else {
        const size_t total = (size_t)1 << n;
        const Fr *Ub = sU + (((size_t)1 << b) - 1);
        for (size_t i = tid; i < total; i += nthr) J.out[i] = b ? eq_mul9(Ub[i >> a], Ta[i & lowmask]) : Ta[i];
    }
}
static void launch_eq_tree(DevCtx &c, const Fr *r0, size_t n0, bool pyr0, Fr *out0, const Fr *r1, size_t n1, bool pyr1, Fr *out1) {
    if (n0 > 13 || n1 > 13) throw Error(OTTI_ERR_BAD_ARG, "eq table job over more than 13 variables");
    EqJob a, b;
    for (size_t i = 0; i < 13; i++) { a.r.v[i] = i < n0 ? r0[i] : fr_zero(); b.r.v[i] = (out1 && i < n1) ? r1[i] : fr_zero(); }
    a.n = (int)n0; a.pyramid = pyr0 ? 1 : 0; a.out = out0; b.n = out1 ? (int)n1 : 0; b.pyramid = pyr1 ? 1 : 0; b.out = out1;
    const size_t big = std::max(n0, out1 ? n1 : (size_t)0);
    const unsigned groups = big > (size_t)kEqLow ? (unsigned)std::min<size_t>(kEqTreeGroups, (((size_t)2 << big) + kEqTreeBlock - 1) / kEqTreeBlock) : 1u;
    hipLaunchKernelGGL(k_eq_tree, dim3(groups, out1 ? 2u : 1u), kEqTreeBlock, 0, c.stream, a, b);
}
// out[i] = hi[i >> lo_bits] * lo[i & (2^lo_bits - 1)]  (index bits are MSB-first over r, so the product of two sub-tables is the table)
// A thread keeps ONE lo entry unpacked and walks `per` hi entries (the same for the whole workgroup: scalar loads): per element one
// nine-limb product and one pack, stores of 8 KB per workgroup and step.  per = 1 .. 16, so that small tables still fill the chip.
__global__ __launch_bounds__(kBlock) void k_eq_expand(const Fr *__restrict__ hi, const Fr *__restrict__ lo, int lo_bits, Fr *__restrict__ out, size_t n_hi, size_t per) {
    const size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;                   // < 2^lo_bits (the grid's x is exact)
    const Fr9 l9 = fr9_unpack(lo[j]);
    const size_t h0 = (size_t)blockIdx.y * per, h1 = min(n_hi, h0 + per);
    for (size_t h = h0; h < h1; h++)
        out[(h << lo_bits) | j] = fr9_pack_lt2l(fr9_mul(fr9_unpack5(hi[h]), l9));     // nine limbs (fr9.h): 32 hi * lo / 2^261 + l < 1.1 l
}
// two tables of at most 13 variables each in ONE launch (the L and R halves of an evaluation point: every polynomial-evaluation proof starts with them)
void dev_eq_evals2(DevCtx &c, const Fr *r0, size_t ell0, Fr *out0, const Fr *r1, size_t ell1, Fr *out1, Fr *scratch) {
    if (ell0 > 13 || ell1 > 13) { dev_eq_evals(c, r0, ell0, out0, scratch); dev_eq_evals(c, r1, ell1, out1, scratch); return; }
    KScope ks(c, KC_EQ);
    launch_eq_tree(c, r0, ell0, false, out0, r1, ell1, false, out1);
}
void dev_eq_evals(DevCtx &c, const Fr *r, size_t ell, Fr *out, Fr *scratch) {
    KScope ks(c, KC_EQ);
    if (ell <= 13) { launch_eq_tree(c, r, ell, false, out, nullptr, 0, false, nullptr); return; }
    const int lo_bits = 12, hi_bits = (int)ell - 12;
    if (hi_bits > 13) throw Error(OTTI_ERR_BAD_ARG, "eq table larger than 2^25");
    Fr *lo = scratch, *hi = scratch + 4096;                               // scratch >= 3 * 4096 elements (both factors in one launch, then their product)
    launch_eq_tree(c, r + hi_bits, lo_bits, false, lo, r, (size_t)hi_bits, false, hi);
    const size_t n_hi = (size_t)1 << hi_bits, per = std::min<size_t>(16, std::max<size_t>(1, n_hi / 128));       // >= 2048 workgroups where the table allows
    hipLaunchKernelGGL(k_eq_expand, dim3((unsigned)(((size_t)1 << lo_bits) / kBlock), (unsigned)((n_hi + per - 1) / per)), kBlock, 0, c.stream, hi, lo, lo_bits, out, n_hi, per);
}

// ------------------------------------------------------------------------------------------------ K3/K4/K7 sum-check rounds
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
// What the LAST thread does to the launch's totals before they go to the host: POST_CUBIC3 turns (Q(0), Q(1), leading coefficient) of a
// quadratic into its values at 0, 2, 3 (the nine-limb kernels below sum the former: linear, so applied once per launch, not per item).
enum { POST_NONE = 0, POST_CUBIC3 = 1 };
template <int K, int kPost = POST_NONE> __device__ __forceinline__ void finish_in_kernel(Fr (&acc)[K], const Mailbox &mb) {
    block_reduce<K>(acc);
    if (gridDim.x > 1) {
        if (threadIdx.x == 0) for (int k = 0; k < K; k++) store_words_sc1(&mb.partials[(size_t)blockIdx.x * K + k], acc[k].v, 8);
        if (!arrive_and_check_last(mb.counter, gridDim.x)) return;
        for (int k = 0; k < K; k++) acc[k] = fr_zero();
        for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
            for (int k = 0; k < K; k++) { Fr t; load_words_sc1(t.v, &mb.partials[(size_t)b * K + k], 8); acc[k] = fr_add(acc[k], t); }
        block_reduce<K>(acc);
    }
    if (mb.line_mail && K <= 3) {
        // the totals, the number and the tag as ONE 128-byte line in one store instruction, no fence (device.h kLineMark; 0.7 us of a round's 15)
        __shared__ Fr s_line[3];
        if (threadIdx.x == 0) {
            if constexpr (kPost == POST_CUBIC3) quadratic_to_023(acc);
            for (int k = 0; k < 3; k++) { s_line[k] = k < K ? acc[k] : fr_zero(); if (k < K) mb.dev_results[mb.slot + k] = acc[k]; }
        }
        __syncthreads();
        if (threadIdx.x < 8) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const unsigned lane = threadIdx.x;
            u32x4 q = {0u, 0u, 0u, 0u};
            if (lane < 6) { const Fr &x = s_line[lane >> 1]; for (int i = 0; i < 4; i++) q[i] = x.v[4 * (lane & 1) + i]; }
            else if (lane == 6) { const unsigned long long tag = line_tag(mb.seq, s_line); q[0] = (uint32_t)mb.seq; q[1] = (uint32_t)(mb.seq >> 32); q[2] = (uint32_t)tag; q[3] = (uint32_t)(tag >> 32); }
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(reinterpret_cast<char *>(mb.host_results) + 16 * lane), "v"(q) : "memory");
        }
        return;
    }
    if (threadIdx.x == 0) {
        if constexpr (kPost == POST_CUBIC3) { static_assert(K == 3, "three totals"); quadratic_to_023(acc); }
        for (int k = 0; k < K; k++) { mb.dev_results[mb.slot + k] = acc[k]; mb.host_results[mb.slot + k] = acc[k]; }
        __threadfence_system();
        __hip_atomic_store(mb.host_flag, mb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ---- the four-table cubic round (kernel-level ABI; the prover itself uses the three-table form below), in nine limbs (fr9.h):
// comb = A (B C - D) at t = 0, 2, 3 as it stands — a cubic in t has no shorter set of sums — with X(2) = 2 X_hi - X_lo and
// X(3) = X(2) + X_hi - X_lo carried down where both sides of a product would otherwise have 31-bit limbs.  A and C go in times 32.
// a_*5, c_*5 normalised below 71 l; b_*, d_* normalised below 2.2 l.
__device__ __forceinline__ void cubic4_accum9(Fr9 (&acc)[3], const Fr9 &a_lo5, const Fr9 &a_hi5, const Fr9 &b_lo, const Fr9 &b_hi, const Fr9 &c_lo5, const Fr9 &c_hi5,
                                              const Fr9 &d_lo, const Fr9 &d_hi) {
    acc[0] = fr9_add(acc[0], fr9_mul(a_lo5, fr9_sub_kl<4>(fr9_mul(b_lo, c_lo5), d_lo)));
    const Fr9 da5 = fr9_sub_kl<128>(a_hi5, a_lo5), db = fr9_sub_kl<4>(b_hi, b_lo), dc5 = fr9_sub_kl<128>(c_hi5, c_lo5), dd = fr9_sub_kl<4>(d_hi, d_lo);
    const Fr9 a2 = fr9_norm(fr9_add(a_hi5, da5)), b2 = fr9_norm(fr9_add(b_hi, db)), c2 = fr9_add(c_hi5, dc5), d2 = fr9_norm(fr9_add(d_hi, dd));   // 270 l, 8.4 l, 270 l, 8.4 l
    acc[1] = fr9_add(acc[1], fr9_mul(a2, fr9_sub_kl<16>(fr9_mul(b2, c2), d2)));                 // B C < 5.5 l; X < 21.5 l; A X / 2^261 + l < 12.4 l
    const Fr9 a3 = fr9_norm(fr9_add(a2, da5)), b3 = fr9_norm(fr9_add(b2, db)), c3 = fr9_norm(fr9_add(c2, dc5)), d3 = fr9_norm(fr9_add(d2, dd));   // 469 l, 14.6 l, 469 l, 14.6 l
    acc[2] = fr9_add(acc[2], fr9_mul(a3, fr9_sub_kl<16>(fr9_mul(b3, c3), d3)));                 // B C < 14.4 l; X < 30.4 l; A X / 2^261 + l < 28.9 l
}
// All of an item's loads are issued before any arithmetic so that their HBM latency is paid once per item, not once per table.
__global__ __launch_bounds__(kBlock) void k_sc_cubic_eval(const Fr *A, const Fr *B, const Fr *C, const Fr *D, size_t half, Mailbox mb) {
    Fr9 acc[3] = {fr9_zero(), fr9_zero(), fr9_zero()}; unsigned n = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        const Fr a0 = A[i], a1 = A[i + half], b0 = B[i], b1 = B[i + half], c0 = C[i], c1 = C[i + half], d0 = D[i], d1 = D[i + half];
        __builtin_amdgcn_sched_barrier(0);                    // keep the scheduler from sinking the loads next to their uses
        cubic4_accum9(acc, fr9_unpack5(a0), fr9_unpack5(a1), fr9_unpack(b0), fr9_unpack(b1), fr9_unpack5(c0), fr9_unpack5(c1), fr9_unpack(d0), fr9_unpack(d1));
        if ((++n & 3u) == 0) acc9_carry(acc);
    }
    Fr tot[3]; acc9_canon<3>(tot, acc);
    finish_in_kernel<3>(tot, mb);
}
__global__ __launch_bounds__(kBlock) void k_sc_cubic_fold_eval(Fr *A, Fr *B, Fr *C, Fr *D, size_t q, Fr r, Mailbox mb) {
    const Fr9 r5 = fr9_unpack5(r);
    Fr9 acc[3] = {fr9_zero(), fr9_zero(), fr9_zero()}; unsigned n = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        // two load groups of eight elements: all sixteen in flight at once would need the whole register file
        const Fr a0 = A[i], a1 = A[i + q], a2 = A[i + 2 * q], a3 = A[i + 3 * q];
        const Fr b0 = B[i], b1 = B[i + q], b2 = B[i + 2 * q], b3 = B[i + 3 * q];
        __builtin_amdgcn_sched_barrier(0);
        Fr w0, w1;
        const Fr9 a_lo = fold9(a0, a2, r5, w0), a_hi = fold9(a1, a3, r5, w1); A[i] = w0; A[i + q] = w1;
        const Fr c0 = C[i], c1 = C[i + q], c2 = C[i + 2 * q], c3 = C[i + 3 * q];
        const Fr d0 = D[i], d1 = D[i + q], d2 = D[i + 2 * q], d3 = D[i + 3 * q];
        __builtin_amdgcn_sched_barrier(0);
        const Fr9 b_lo = fold9(b0, b2, r5, w0), b_hi = fold9(b1, b3, r5, w1); B[i] = w0; B[i + q] = w1;
        const Fr9 c_lo = fold9(c0, c2, r5, w0), c_hi = fold9(c1, c3, r5, w1); C[i] = w0; C[i + q] = w1;
        const Fr9 d_lo = fold9(d0, d2, r5, w0), d_hi = fold9(d1, d3, r5, w1); D[i] = w0; D[i + q] = w1;
        cubic4_accum9(acc, fr9_shl5(a_lo), fr9_shl5(a_hi), b_lo, b_hi, fr9_shl5(c_lo), fr9_shl5(c_hi), d_lo, d_hi);
        if ((++n & 3u) == 0) acc9_carry(acc);
    }
    Fr tot[3]; acc9_canon<3>(tot, acc);
    finish_in_kernel<3>(tot, mb);
}
// ---- phase one without the eq table.  eq(tau, .) is a tensor product, so after j rounds the fourth table of the cubic sum-check is
// D_j[(b, i)] = c_j * (b ? tau_j : 1 - tau_j) * E_j[i]  with  E_j = eq(tau_{j+1..}, .)  and  c_j = prod_{k<j} eq(tau_k, r_k):
// it never has to be stored, folded or streamed.  The round sums become  e_t = c_j * ((1 - tau_j) + t (2 tau_j - 1)) * S_t  with
// S_t = sum_i E_j[i] * (A_t[i] B_t[i] - C_t[i]); the kernels below return S_t (three tables instead of four: a quarter less HBM
// traffic and register pressure), the host applies the two scalar factors.  E_j[i] itself is hi[i >> lo_bits] * lo[i & mask] from the
// two small "pyramids" of k_eq_tree (L2-resident), or lo[i] once at most lo_bits variables are left.
// ---- the same two kernels in nine 29-bit limbs (fr9.h): operands stay unpacked from load to store.  Per item they sum
//   Q(0) += E (B_lo C_lo - D_lo),   Q(1) += E (B_hi C_hi - D_hi),   Q_inf += E (B_hi - B_lo)(C_hi - C_lo)
// — S_t's values at 0 and 1 and its leading coefficient: every operand is a table element or a single difference, so no product needs an
// operand carried down beyond one normalisation (the points 2 and 3 would want 2 X_hi - X_lo and 3 X_hi - 2 X_lo: limbs of 31 bits on both
// sides of a product) — and the launch's last thread turns the totals into S_0, S_2, S_3 (POST_CUBIC3).  Montgomery radix 2^261: C and E go
// in times 32 (shifted unpack / fr9_shl5), everything else as it is, and every product comes out in the memory format.
// b_*, d_* normalised below 2.2 l; c_*5 normalised below 71 l (32 times an element below 2.2 l); e5 normalised below 32 l.
// Products: (norm x norm), (norm x limbs < 2^30.6): column sums < 2^63; values: P < 1.4 l, X < 5.4 l, E X / 2^261 + l < 1.4 l.
__device__ __forceinline__ void cubic3_accum9(Fr9 (&acc)[3], const Fr9 &e5, const Fr9 &b_lo, const Fr9 &b_hi, const Fr9 &c_lo5, const Fr9 &c_hi5, const Fr9 &d_lo, const Fr9 &d_hi) {
    const Fr9 db = fr9_norm(fr9_sub_kl<4>(b_hi, b_lo));
    const Fr9 dc5 = fr9_sub_kl<128>(c_hi5, c_lo5);
    const Fr9 x0 = fr9_sub_kl<4>(fr9_mul(b_lo, c_lo5), d_lo);
    const Fr9 x1 = fr9_sub_kl<4>(fr9_mul(b_hi, c_hi5), d_hi);
    const Fr9 xi = fr9_mul(db, dc5);
    acc[0] = fr9_add(acc[0], fr9_mul(e5, x0));
    acc[1] = fr9_add(acc[1], fr9_mul(e5, x1));
    acc[2] = fr9_add(acc[2], fr9_mul(e5, xi));
}
__global__ __launch_bounds__(kBlock) void k_sc_cubic3_eval(const Fr *B, const Fr *C, const Fr *D, size_t half, EqSrc E, Mailbox mb) {
    Fr9 acc[3] = {fr9_zero(), fr9_zero(), fr9_zero()}; unsigned n = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        const Fr b0 = B[i], b1 = B[i + half], c0 = C[i], c1 = C[i + half], d0 = D[i], d1 = D[i + half];
        const Fr9 e5 = eq5_at(E, i);
        __builtin_amdgcn_sched_barrier(0);
        cubic3_accum9(acc, e5, fr9_unpack(b0), fr9_unpack(b1), fr9_unpack5(c0), fr9_unpack5(c1), fr9_unpack(d0), fr9_unpack(d1));
        if ((++n & 3u) == 0) acc9_carry(acc);
    }
    Fr tot[3]; acc9_canon<3>(tot, acc);
    finish_in_kernel<3, POST_CUBIC3>(tot, mb);
}
__global__ __launch_bounds__(kBlock) void k_sc_cubic3_fold_eval(Fr *B, Fr *C, Fr *D, size_t q, Fr r, EqSrc E, Mailbox mb, Armed go) {
    if (go.want) { Fr v[1]; if (!armed_fetch<1>(go, v)) return; r = v[0]; }
    const Fr9 r5 = fr9_unpack5(r);
    Fr9 acc[3] = {fr9_zero(), fr9_zero(), fr9_zero()}; unsigned n = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        const Fr b0 = B[i], b1 = B[i + q], b2 = B[i + 2 * q], b3 = B[i + 3 * q];
        const Fr c0 = C[i], c1 = C[i + q], c2 = C[i + 2 * q], c3 = C[i + 3 * q];
        __builtin_amdgcn_sched_barrier(0);
        Fr w0, w1;
        const Fr9 b_lo = fold9(b0, b2, r5, w0), b_hi = fold9(b1, b3, r5, w1); B[i] = w0; B[i + q] = w1;
        const Fr d0 = D[i], d1 = D[i + q], d2 = D[i + 2 * q], d3 = D[i + 3 * q];
        const Fr9 e5 = eq5_at(E, i);
        __builtin_amdgcn_sched_barrier(0);
        const Fr9 c_lo = fold9(c0, c2, r5, w0), c_hi = fold9(c1, c3, r5, w1); C[i] = w0; C[i + q] = w1;
        const Fr9 d_lo = fold9(d0, d2, r5, w0), d_hi = fold9(d1, d3, r5, w1); D[i] = w0; D[i + q] = w1;
        cubic3_accum9(acc, e5, b_lo, b_hi, fr9_shl5(c_lo), fr9_shl5(c_hi), d_lo, d_hi);
        if ((++n & 3u) == 0) acc9_carry(acc);
    }
    Fr tot[3]; acc9_canon<3>(tot, acc);
    finish_in_kernel<3, POST_CUBIC3>(tot, mb);
}
// ---- phase two in nine limbs: e_0 = sum A_lo B_lo and e_2 = sum (2 A_hi - A_lo)(2 B_hi - B_lo) as they stand (a third sum for the point 1
// would cost a product; one carried-down operand costs 24 light instructions).  B goes in times 32 (radix 2^261, fr9.h).
// a_* normalised below 2.2 l, b_*5 normalised below 71 l; values: u < 8.4 l, v5 < 270 l, u v5 / 2^261 + l < 5.5 l.
__device__ __forceinline__ void quad_accum9(Fr9 (&acc)[2], const Fr9 &a_lo, const Fr9 &a_hi, const Fr9 &b_lo5, const Fr9 &b_hi5) {
    acc[0] = fr9_add(acc[0], fr9_mul(a_lo, b_lo5));
    const Fr9 u = fr9_norm(fr9_sub_kl<4>(fr9_add(a_hi, a_hi), a_lo));             // limbs < 2^31 before the sweep
    const Fr9 v5 = fr9_sub_kl<128>(fr9_add(b_hi5, b_hi5), b_lo5);                 // limbs < 2^31: 9 * 2^29 * 2^31 + 6 * 2^58 < 2^63.4
    acc[1] = fr9_add(acc[1], fr9_mul(u, v5));
}
__global__ __launch_bounds__(kBlock) void k_sc_quad_eval(const Fr *A, const Fr *B, size_t half, Mailbox mb) {
    Fr9 acc[2] = {fr9_zero(), fr9_zero()}; unsigned n = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < half; i += (size_t)gridDim.x * blockDim.x) {
        const Fr a0 = A[i], a1 = A[i + half], b0 = B[i], b1 = B[i + half];
        __builtin_amdgcn_sched_barrier(0);
        quad_accum9(acc, fr9_unpack(a0), fr9_unpack(a1), fr9_unpack5(b0), fr9_unpack5(b1));
        if ((++n & 3u) == 0) { acc[0] = fr9_norm(acc[0]); acc[1] = fr9_norm(acc[1]); }
    }
    Fr tot[2]; acc9_canon<2>(tot, acc);
    finish_in_kernel<2>(tot, mb);
}
__global__ __launch_bounds__(kBlock) void k_sc_quad_fold_eval(Fr *A, Fr *B, size_t q, Fr r, Mailbox mb, Armed go) {
    if (go.want) { Fr v[1]; if (!armed_fetch<1>(go, v)) return; r = v[0]; }
    const Fr9 r5 = fr9_unpack5(r);
    Fr9 acc[2] = {fr9_zero(), fr9_zero()}; unsigned n = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        const Fr a0 = A[i], a1 = A[i + q], a2 = A[i + 2 * q], a3 = A[i + 3 * q];
        const Fr b0 = B[i], b1 = B[i + q], b2 = B[i + 2 * q], b3 = B[i + 3 * q];
        __builtin_amdgcn_sched_barrier(0);
        Fr w0, w1;
        const Fr9 a_lo = fold9(a0, a2, r5, w0), a_hi = fold9(a1, a3, r5, w1); A[i] = w0; A[i + q] = w1;
        const Fr9 b_lo = fold9(b0, b2, r5, w0), b_hi = fold9(b1, b3, r5, w1); B[i] = w0; B[i + q] = w1;
        quad_accum9(acc, a_lo, a_hi, fr9_shl5(b_lo), fr9_shl5(b_hi));
        if ((++n & 3u) == 0) { acc[0] = fr9_norm(acc[0]); acc[1] = fr9_norm(acc[1]); }
    }
    Fr tot[2]; acc9_canon<2>(tot, acc);
    finish_in_kernel<2>(tot, mb);
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
// 242-VGPR kernels: two workgroups per CU are resident, so 512 workgroups already fill the chip; below ~2^21 elements a wider grid only
// adds partials for the last workgroup to sum (measured: 2^20 evaluate 47 -> 32 us), above it the extra workgroups hide tail effects.
static inline int sc_grid(size_t n) { return std::min(grid_for(n), n <= ((size_t)1 << 21) ? 512 : kMaxBlocks); }
unsigned long long dev_sc_cubic_eval(DevCtx &c, const Fr *A, const Fr *B, const Fr *C, const Fr *D, size_t len, int slot) {
    size_t half = len / 2; int g = sc_grid(half); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_CUBIC); hipLaunchKernelGGL(k_sc_cubic_eval, g, kBlock, 0, c.stream, A, B, C, D, half, mb);
    return mb.seq;
}
unsigned long long dev_sc_cubic_fold_eval(DevCtx &c, Fr *A, Fr *B, Fr *C, Fr *D, size_t len, const Fr &r, int slot) {
    if (len < 4) throw Error(OTTI_ERR_INTERNAL, "fold_eval needs len >= 4");
    size_t q = len / 4; int g = sc_grid(q); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_CUBIC); hipLaunchKernelGGL(k_sc_cubic_fold_eval, g, kBlock, 0, c.stream, A, B, C, D, q, r, mb);
    return mb.seq;
}
void dev_eq_pyramid(DevCtx &c, const Fr *r_host, size_t n, Fr *out) { dev_eq_pyramid2(c, r_host, n, out, nullptr, 0, nullptr); }
void dev_eq_pyramid2(DevCtx &c, const Fr *r0_host, size_t n0, Fr *out0, const Fr *r1_host, size_t n1, Fr *out1) {
    KScope ks(c, KC_EQ);
    launch_eq_tree(c, r0_host, n0, true, out0, r1_host, n1, true, out1);
}
unsigned long long dev_sc_cubic3_eval(DevCtx &c, const Fr *B, const Fr *C, const Fr *D, size_t len, const EqSrc &E, int slot) {
    size_t half = len / 2; int g = sc_grid(half); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_CUBIC); hipLaunchKernelGGL(k_sc_cubic3_eval, g, kBlock, 0, c.stream, B, C, D, half, E, mb);
    return mb.seq;
}
static unsigned long long cubic3_fold_eval(DevCtx &c, Fr *B, Fr *C, Fr *D, size_t len, const Fr *r, const EqSrc &E, int slot) {
    if (len < 4) throw Error(OTTI_ERR_INTERNAL, "fold_eval needs len >= 4");
    size_t q = len / 4; int g = sc_grid(q); Mailbox mb = c.next_mailbox(slot);
    const Armed go = r ? Armed{nullptr, nullptr, 0} : c.arm();
    KScope ks(c, KC_SC_CUBIC); hipLaunchKernelGGL(k_sc_cubic3_fold_eval, g, kBlock, 0, c.stream, B, C, D, q, r ? *r : fr_zero(), E, mb, go);
    return mb.seq;
}
unsigned long long dev_sc_cubic3_fold_eval(DevCtx &c, Fr *B, Fr *C, Fr *D, size_t len, const Fr &r, const EqSrc &E, int slot) { return cubic3_fold_eval(c, B, C, D, len, &r, E, slot); }
unsigned long long dev_sc_cubic3_fold_eval_armed(DevCtx &c, Fr *B, Fr *C, Fr *D, size_t len, const EqSrc &E, int slot) { return cubic3_fold_eval(c, B, C, D, len, nullptr, E, slot); }
unsigned long long dev_sc_quad_eval(DevCtx &c, const Fr *A, const Fr *B, size_t len, int slot) {
    size_t half = len / 2; int g = sc_grid(half); Mailbox mb = c.next_mailbox(slot);
    KScope ks(c, KC_SC_QUAD); hipLaunchKernelGGL(k_sc_quad_eval, g, kBlock, 0, c.stream, A, B, half, mb);
    return mb.seq;
}
static unsigned long long quad_fold_eval(DevCtx &c, Fr *A, Fr *B, size_t len, const Fr *r, int slot) {
    if (len < 4) throw Error(OTTI_ERR_INTERNAL, "fold_eval needs len >= 4");
    size_t q = len / 4; int g = sc_grid(q); Mailbox mb = c.next_mailbox(slot);
    const Armed go = r ? Armed{nullptr, nullptr, 0} : c.arm();
    KScope ks(c, KC_SC_QUAD); hipLaunchKernelGGL(k_sc_quad_fold_eval, g, kBlock, 0, c.stream, A, B, q, r ? *r : fr_zero(), mb, go);
    return mb.seq;
}
unsigned long long dev_sc_quad_fold_eval(DevCtx &c, Fr *A, Fr *B, size_t len, const Fr &r, int slot) { return quad_fold_eval(c, A, B, len, &r, slot); }
unsigned long long dev_sc_quad_fold_eval_armed(DevCtx &c, Fr *A, Fr *B, size_t len, int slot) { return quad_fold_eval(c, A, B, len, nullptr, slot); }
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
// A workgroup is 2^lanes_log2 row lanes x (kBlock >> lanes_log2) columns (>= 64 columns: a wave has one lane): lane l takes rows l, l + lanes, ...
// of its slab, the lanes' sums meet in LDS.  More, shorter threads where the matrix is small (2^20: 16 dependent products per thread on one wave per
// SIMD became 4 on four) — sums of field elements, so the result does not depend on the split.
template <int K> __device__ __forceinline__ Fr lanes_sum(Fr acc, int lane, int lanes, int cols) {
    __shared__ Fr sm[K][kBlock];
    if (lanes > 1) {
        sm[0][threadIdx.x] = acc;
        __syncthreads();
        for (int s = lanes >> 1; s >= 1; s >>= 1) {
            if (lane < s) { acc = fr_add(acc, sm[0][threadIdx.x + s * cols]); sm[0][threadIdx.x] = acc; }
            __syncthreads();
        }
    }
    return acc;
}
__global__ __launch_bounds__(kBlock) void k_poly_bound_slab(const Fr *Z, size_t L, size_t R, const Fr *Lv, size_t rows_per_slab, Fr *scratch, size_t lv_mask, int lanes_log2) {
    const int lanes = 1 << lanes_log2, cols = kBlock >> lanes_log2, lane = (int)threadIdx.x / cols, cj = (int)threadIdx.x % cols;
    const size_t j = blockIdx.x * (size_t)cols + cj;
    const size_t i0 = blockIdx.y * rows_per_slab, i1 = min(L, i0 + rows_per_slab);
    Fr acc = fr_zero();
    if (j < R) for (size_t i = i0 + lane; i < i1; i += lanes) acc = fr_add(acc, fr_mul(Lv[i & lv_mask], Z[i * R + j]));
    acc = lanes_sum<1>(acc, lane, lanes, cols);
    if (lane == 0 && j < R) scratch[(size_t)blockIdx.y * R + j] = acc;
}
constexpr int kColsumLanesLog2 = 3;                          // 8 lanes x 32 columns: 8 + 3 dependent additions for 64 slabs instead of 63
__global__ __launch_bounds__(kBlock) void k_colsum(const Fr *scratch, size_t slabs, size_t R, Fr *out) {
    const int lanes = 1 << kColsumLanesLog2, cols = kBlock >> kColsumLanesLog2, lane = (int)threadIdx.x / cols, cj = (int)threadIdx.x % cols;
    const size_t j = blockIdx.x * (size_t)cols + cj;
    Fr acc = fr_zero();
    if (j < R) for (size_t s = lane; s < slabs; s += lanes) acc = fr_add(acc, scratch[s * R + j]);
    acc = lanes_sum<1>(acc, lane, lanes, cols);
    if (lane == 0 && j < R) out[j] = acc;
}
// row lanes per workgroup of the slab kernel: as many (up to 4) as keep every lane at least two rows and the launch below ~2048 workgroups
static int bound_lanes_log2(size_t R, size_t slabs, size_t rps) {
    int lg = 0;
    while (lg < 2 && ((size_t)2 << lg) * 2 <= rps && ((R + (kBlock >> (lg + 1)) - 1) / (kBlock >> (lg + 1))) * slabs <= 2048) lg++;
    return lg;
}
void dev_poly_bound(DevCtx &c, const Fr *Z, size_t L, size_t R, const Fr *Lv, Fr *out, Fr *scratch) {
    size_t slabs = std::min<size_t>(L, 64), rps = (L + slabs - 1) / slabs;
    KScope ks(c, KC_BOUND);
    const int ll = bound_lanes_log2(R, slabs, rps); const size_t cols = (size_t)kBlock >> ll, ccols = (size_t)kBlock >> kColsumLanesLog2;
    dim3 grid((unsigned)((R + cols - 1) / cols), (unsigned)slabs);
    hipLaunchKernelGGL(k_poly_bound_slab, grid, kBlock, 0, c.stream, Z, L, R, Lv, rps, scratch, ~(size_t)0, ll);
    hipLaunchKernelGGL(k_colsum, (unsigned)((R + ccols - 1) / ccols), kBlock, 0, c.stream, (const Fr *)scratch, slabs, R, out);
}
// The same bound in two steps, for an evaluation point whose FIRST variables are not known yet: with the left table L = eq(first a variables) x eq(rest),
//     (L^T Z)[j] = sum_c eq(first)[c] * P_c[j],   P_c[j] = sum_i' eq(rest)[i'] Z[(c m + i') R + j]     (m = 2^(variables of rest) rows per chunk).
// This computes the chunk sums P_c (chunks x R, out) from eq(rest) alone (Lv_rest, m entries); the caller finishes with dev_poly_bound(out, chunks, R, eq(first)).
__global__ __launch_bounds__(kBlock) void k_colsum_groups(const Fr *scratch, size_t group, size_t R, Fr *out) {
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= R) return;
    const Fr *base = scratch + (size_t)blockIdx.y * group * R;
    Fr acc = base[j];
    for (size_t s = 1; s < group; s++) acc = fr_add(acc, base[s * R + j]);
    out[(size_t)blockIdx.y * R + j] = acc;
}
bool dev_poly_bound_chunks(DevCtx &c, const Fr *Z, size_t L, size_t R, const Fr *Lv_rest, size_t m, Fr *out, Fr *scratch) {
    const size_t slabs = std::min<size_t>(L, 64), rps = (L + slabs - 1) / slabs;
    if (!m || (m & (m - 1)) || m > L || L % m || m % rps || slabs * rps != L) return false;        // (chunks must be whole groups of slabs)
    KScope ks(c, KC_BOUND);
    const int ll = bound_lanes_log2(R, slabs, rps); const size_t cols = (size_t)kBlock >> ll;
    dim3 grid((unsigned)((R + cols - 1) / cols), (unsigned)slabs);
    hipLaunchKernelGGL(k_poly_bound_slab, grid, kBlock, 0, c.stream, Z, L, R, Lv_rest, rps, scratch, m - 1, ll);
    hipLaunchKernelGGL(k_colsum_groups, dim3((unsigned)((R + kBlock - 1) / kBlock), (unsigned)(L / m)), kBlock, 0, c.stream, (const Fr *)scratch, m / rps, R, out);
    return true;
}

}  // namespace otti
