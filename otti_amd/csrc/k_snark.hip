// Kernels of SNARK mode's R1CSEvalProof (snark.h): dereferencing the eq tables by row / column address, the memory-checking hash
// layer, product-circuit layers, and the rounds of the batched cubic sum-check over many (A, B, C) table triples at once.
// All HBM-streaming work on 32-byte field elements (no MFMA applies: 256-bit modular integers); one launch handles every instance of a
// batch (grid.y = instance), so a round of SumcheckInstanceProof::prove_cubic_batched is one launch whatever the batch size.
//
// Kernel <-> upstream loop [RECALL; the reference's Spartan/ submodule is empty]:
//   k_gather                sparse_mlpoly.rs AddrTimestamps::deref_mem
//   k_hash_mem / k_hash_ops sparse_mlpoly.rs Layers::build_hash_layer (init / audit, read / write)
//   k_prod_layer            product_tree.rs ProductCircuit::compute_layer
//   k_pc_round              sumcheck.rs SumcheckInstanceProof::prove_cubic_batched: one round = bound_poly_var_top of every table of the
//                           batch by the previous challenge + the evaluation loop (comb = A * B * C at 0, 2, 3), fused
//   k_pc_export             the tables' last elements (and the host-played tail of every layer)
//   k_dot_many / k_sum3     DensePolynomial::evaluate (against a shared eq table) / DotProductCircuit::evaluate
#include "kernels_common.h"
#include "snark_dev.h"

namespace otti {

__global__ __launch_bounds__(kBlock) void k_gather(const Fr *table, const uint32_t *idx, Fr *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = table[idx[i]];
}
void dev_gather(DevCtx &c, const Fr *table, const uint32_t *idx, Fr *out, size_t n) {
    KScope ks(c, KC_GATHER);
    hipLaunchKernelGGL(k_gather, grid_for(n), kBlock, 0, c.stream, table, idx, out, n);
}

// addresses and timestamps of the dense representation: small integers -> Montgomery form
__global__ __launch_bounds__(kBlock) void k_u32_to_fr(const uint32_t *in, Fr *out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = fr_from_u64((uint64_t)in[i]);
}
void dev_u32_to_fr(DevCtx &c, const uint32_t *in, Fr *out, size_t n) {
    if (n) hipLaunchKernelGGL(k_u32_to_fr, grid_for(n), kBlock, 0, c.stream, in, out, n);
}

// hash(addr, val, ts) - gamma = ts * r^2 + val * r + addr - gamma
__device__ __forceinline__ Fr hash3(const Fr &addr, const Fr &val, const Fr &ts, const Fr &r, const Fr &r2, const Fr &gamma) {
    return fr_sub(fr_add(fr_add(fr_mul(ts, r2), fr_mul(val, r)), addr), gamma);
}
// A rank's residue class of a hashed vector (sharded SNARK::prove, snark_prover.cpp): local element j of a vector of 2 * half_loc elements is
// global element (j / half_loc) * half_glob + (j % half_loc) * G + rk — the circuit's input is its left half then its right half, and every
// layer pairs index i with i + (side length) / 2, so both halves keep the elements with i = rk (mod G).  G = 1: the identity.
struct ShardMap { size_t half_loc, half_glob; uint32_t G, rk; };
__device__ __forceinline__ size_t shard_global(const ShardMap &m, size_t j) { return m.G == 1 ? j : (j / m.half_loc) * m.half_glob + (j % m.half_loc) * m.G + m.rk; }
// memory cells: addr = the cell index, val = the eq table, ts = 0 (init) or the audit timestamp
__global__ __launch_bounds__(kBlock) void k_hash_mem(const Fr *eval_table, const Fr *audit_ts, Fr *out_init, Fr *out_audit, size_t M, Fr r, Fr r2, Fr gamma, ShardMap sm) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < M; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = shard_global(sm, j);
        const Fr base = fr_sub(fr_add(fr_mul(eval_table[i], r), fr_from_u64((uint64_t)i)), gamma);
        out_init[j] = base;
        out_audit[j] = fr_add(base, fr_mul(audit_ts[i], r2));
    }
}
// operations: read uses the read timestamp, write the same plus one (so write = read + r^2)
__global__ __launch_bounds__(kBlock) void k_hash_ops(const Fr *addr_f, const Fr *deref, const Fr *read_ts, Fr *out_read, Fr *out_write, size_t N, Fr r, Fr r2, Fr gamma, ShardMap sm) {
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < N; j += (size_t)gridDim.x * blockDim.x) {
        const size_t i = shard_global(sm, j);
        const Fr rd = hash3(addr_f[i], deref[i], read_ts[i], r, r2, gamma);
        out_read[j] = rd; out_write[j] = fr_add(rd, r2);
    }
}
// n = the GLOBAL vector length; G ranks: this rank writes its n / G elements
void dev_hash_mem(DevCtx &c, const Fr *eval_table, const Fr *audit_ts, Fr *out_init, Fr *out_audit, size_t M, const Fr &r, const Fr &gamma, int G, int rk) {
    KScope ks(c, KC_HASH_LAYER);
    const size_t Ml = M / (size_t)G; const ShardMap sm = {std::max<size_t>(1, Ml / 2), M / 2, (uint32_t)G, (uint32_t)rk};
    hipLaunchKernelGGL(k_hash_mem, grid_for(Ml), kBlock, 0, c.stream, eval_table, audit_ts, out_init, out_audit, Ml, r, fr_mul(r, r), gamma, sm);
}
void dev_hash_ops(DevCtx &c, const Fr *addr_f, const Fr *deref, const Fr *read_ts, Fr *out_read, Fr *out_write, size_t N, const Fr &r, const Fr &gamma, int G, int rk) {
    KScope ks(c, KC_HASH_LAYER);
    const size_t Nl = N / (size_t)G; const ShardMap sm = {std::max<size_t>(1, Nl / 2), N / 2, (uint32_t)G, (uint32_t)rk};
    hipLaunchKernelGGL(k_hash_ops, grid_for(Nl), kBlock, 0, c.stream, addr_f, deref, read_ts, out_read, out_write, Nl, r, fr_mul(r, r), gamma, sm);
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
    KScope ks(c, KC_PROD_LAYER);
    hipLaunchKernelGGL(k_prod_layer, dim3((unsigned)grid_for(q), (unsigned)L.n), kBlock, 0, c.stream, L, q);
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
// ---- a round of the batched cubic sum-check as ONE launch (fold by the previous challenge + the sums of this round at the points 0, 2, 3
// of the variable being bound), results mailed to the host by the last workgroup (finish_in_kernel of k_sumcheck.hip, here for every
// instance of the batch at once).
// An instance without a third table (a product circuit: L.C[y] == nullptr): its three totals are (Q(0), Q(1), leading coefficient) of a quadratic (product-circuit instances of the nine-limb
// round kernel) and leave as its values at 0, 2, 3; otherwise they are those values already (dot-product triples).
__device__ __forceinline__ void finish_many(Fr (&acc)[3], const Mailbox &mb, const PcList &L) {
    block_reduce<3>(acc);
    const unsigned total = gridDim.x * gridDim.y, ny = gridDim.y;
    if (total == 1) {
        if (threadIdx.x == 0) {
            if (!L.C[0]) quadratic_to_023(acc);
            for (int k = 0; k < 3; k++) mb.host_results[mb.slot + k] = acc[k];
            __threadfence_system();
            __hip_atomic_store(mb.host_flag, mb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    __shared__ Fr s_tot[3 * 64];
    if (threadIdx.x == 0) for (int k = 0; k < 3; k++) store_words_sc1(&mb.partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + k], acc[k].v, 8);
    if (!arrive_and_check_last(mb.counter, total)) return;
    // last workgroup: output o = (instance, t) is summed by four adjacent lanes
    const unsigned o = threadIdx.x >> 2, sub = threadIdx.x & 3;
    Fr s = fr_zero();
    if (o < 3 * ny) {
        const unsigned y = o / 3, k = o % 3;
        for (unsigned b = sub; b < gridDim.x; b += 4) { Fr t; load_words_sc1(t.v, &mb.partials[((size_t)y * gridDim.x + b) * 3 + k], 8); s = fr_add(s, t); }
    }
    s = fr_add(s, shfl_xor_fr(s, 1)); s = fr_add(s, shfl_xor_fr(s, 2));
    if (o < 3 * ny && sub == 0) s_tot[o] = s;
    __syncthreads();
    if (threadIdx.x < ny) {
        Fr q[3] = {s_tot[3 * threadIdx.x], s_tot[3 * threadIdx.x + 1], s_tot[3 * threadIdx.x + 2]};
        if (!L.C[threadIdx.x]) quadratic_to_023(q);
        for (int k = 0; k < 3; k++) mb.host_results[mb.slot + 3 * threadIdx.x + k] = q[k];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(mb.host_flag, mb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ---- nine-limb forms (fr9.h).  Product-circuit instance (third table = the shared eq table, never stored): per item
//   Q(0) += (E a_lo) b_lo,  Q(1) += (E a_hi) b_hi,  Q_inf += (E a_hi - E a_lo)(b_hi - b_lo)
// — the quadratic's values at 0, 1 and its leading coefficient, in FIVE products (E a_lo and E a_hi serve two sums each; six when E multiplied
// the three products a b); finish_many turns the totals into the values at 0, 2, 3.  E goes in times 2^10: E a_* then carries the factor 32
// the second product needs.  a_*, b_* normalised below 2.2 l; e10 normalised below 2^10 l.
__device__ __forceinline__ void abe_accum9(Fr9 (&acc)[3], const Fr9 &a_lo, const Fr9 &a_hi, const Fr9 &b_lo, const Fr9 &b_hi, const Fr9 &e10) {
    const Fr9 u = fr9_mul(e10, a_lo), v = fr9_mul(e10, a_hi);                               // 2^10 l * 2.2 l / 2^261 + l < 5.4 l
    const Fr9 dv = fr9_norm(fr9_sub_kl<8>(v, u)), db = fr9_sub_kl<4>(b_hi, b_lo);           // < 13.4 l; < 6.2 l (loose)
    acc[0] = fr9_add(acc[0], fr9_mul(u, b_lo));                                             // 5.4 * 2.2 / 2^9 + 1 < 1.1 l
    acc[1] = fr9_add(acc[1], fr9_mul(v, b_hi));
    acc[2] = fr9_add(acc[2], fr9_mul(dv, db));                                              // 13.4 * 6.2 / 2^9 + 1 < 1.2 l
}
// Dot-product triple (three real tables): the cubic's values at 0, 2, 3 as they stand.  a_* plain, b_*5 and c_*5 times 32.
__device__ __forceinline__ void abc_accum9(Fr9 (&acc)[3], const Fr9 &a_lo, const Fr9 &a_hi, const Fr9 &b_lo5, const Fr9 &b_hi5, const Fr9 &c_lo5, const Fr9 &c_hi5) {
    acc[0] = fr9_add(acc[0], fr9_mul(fr9_mul(a_lo, b_lo5), c_lo5));
    // point 2: x = 2 a_hi - a_lo + 4l (carried down), y5, z5 = 2 X_hi5 - X_lo5 + 128 l (limbs < 2^31): values 8.4 l, 270 l: products < 5.5 l, < 3.9 l
    const Fr9 x2 = fr9_norm(fr9_sub_kl<4>(fr9_add(a_hi, a_hi), a_lo));
    const Fr9 y2 = fr9_sub_kl<128>(fr9_add(b_hi5, b_hi5), b_lo5), z2 = fr9_sub_kl<128>(fr9_add(c_hi5, c_hi5), c_lo5);
    acc[1] = fr9_add(acc[1], fr9_mul(fr9_mul(x2, y2), z2));
    // point 3: X_3 = X_2 + (X_hi - X_lo): carried down on every side (their limbs would reach 3.5 * 2^30)
    const Fr9 x3 = fr9_norm(fr9_add(x2, fr9_sub_kl<4>(a_hi, a_lo)));
    const Fr9 y3 = fr9_norm(fr9_add(y2, fr9_sub_kl<128>(b_hi5, b_lo5))), z3 = fr9_norm(fr9_add(z2, fr9_sub_kl<128>(c_hi5, c_lo5)));
    acc[2] = fr9_add(acc[2], fr9_mul(fr9_mul(x3, y3), z3));          // values 14.6 l, 469 l: 14.6 * 469 / 512 + 1 = 14.4 l, then 14.4 * 469 / 512 + 1 = 14.2 l
}
// kFold: the tables have 4q entries and are folded by r to 2q (pairs (i, i + q)); otherwise they have 2q entries as they are
template <bool kFold> __global__ __launch_bounds__(kBlock) void k_pc_round(PcList L, size_t q, Fr r, EqSrc E, Mailbox mb, Armed go) {
    if (kFold && go.want) { Fr v[1]; if (!armed_fetch<1>(go, v)) return; r = v[0]; }
    Fr *A = L.A[blockIdx.y], *B = L.B[blockIdx.y], *C = L.C[blockIdx.y];
    const Fr9 r5 = fr9_unpack5(r);
    Fr9 acc[3] = {fr9_zero(), fr9_zero(), fr9_zero()}; unsigned n = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        Fr9 a_lo, a_hi, b_lo, b_hi;
        if (kFold) {
            const Fr a0 = A[i], a1 = A[i + q], a2 = A[i + 2 * q], a3 = A[i + 3 * q];
            const Fr b0 = B[i], b1 = B[i + q], b2 = B[i + 2 * q], b3 = B[i + 3 * q];
            __builtin_amdgcn_sched_barrier(0);
            Fr w0, w1;
            a_lo = fold9(a0, a2, r5, w0); a_hi = fold9(a1, a3, r5, w1); A[i] = w0; A[i + q] = w1;
            b_lo = fold9(b0, b2, r5, w0); b_hi = fold9(b1, b3, r5, w1); B[i] = w0; B[i + q] = w1;
        } else {
            const Fr a0 = A[i], a1 = A[i + q], b0 = B[i], b1 = B[i + q];
            __builtin_amdgcn_sched_barrier(0);
            a_lo = fr9_unpack(a0); a_hi = fr9_unpack(a1); b_lo = fr9_unpack(b0); b_hi = fr9_unpack(b1);
        }
        if (C) {
            Fr9 c_lo5, c_hi5;
            if (kFold) {
                const Fr c0 = C[i], c1 = C[i + q], c2 = C[i + 2 * q], c3 = C[i + 3 * q];
                Fr w0, w1;
                const Fr9 c_lo = fold9(c0, c2, r5, w0), c_hi = fold9(c1, c3, r5, w1); C[i] = w0; C[i + q] = w1;
                c_lo5 = fr9_shl5(c_lo); c_hi5 = fr9_shl5(c_hi);
            } else { c_lo5 = fr9_unpack5(C[i]); c_hi5 = fr9_unpack5(C[i + q]); }
            abc_accum9(acc, a_lo, a_hi, fr9_shl5(b_lo), fr9_shl5(b_hi), c_lo5, c_hi5);
        } else abe_accum9(acc, a_lo, a_hi, b_lo, b_hi, eq_s_at<10>(E, i));
        if ((++n & 3u) == 0) acc9_carry(acc);
    }
    Fr tot[3]; acc9_canon<3>(tot, acc);
    finish_many(tot, mb, L);
}
static Mailbox pc_mailbox(DevCtx &c, int slot, int nblocks) {
    if ((size_t)nblocks * 3 > c.partials.n) throw Error(OTTI_ERR_INTERNAL, "round partials exceed the context's buffer");
    return c.next_mailbox(slot);
}
unsigned long long dev_pc_eval(DevCtx &c, const PcList &L, size_t len, const EqSrc &E, int slot) {
    const size_t half = len / 2; const int g = many_grid(half, L.n); Mailbox mb = pc_mailbox(c, slot, g * L.n);
    KScope ks(c, KC_PC_ROUND);
    hipLaunchKernelGGL(k_pc_round<false>, dim3((unsigned)g, (unsigned)L.n), kBlock, 0, c.stream, L, half, fr_zero(), E, mb, Armed{nullptr, nullptr, 0});
    return mb.seq;
}
unsigned long long dev_pc_fold_eval(DevCtx &c, const PcList &L, size_t len, const Fr *r, const EqSrc &E, int slot) {
    if (len < 4) throw Error(OTTI_ERR_INTERNAL, "fold_eval needs len >= 4");
    const size_t q = len / 4; const int g = many_grid(q, L.n); Mailbox mb = pc_mailbox(c, slot, g * L.n);
    const Armed go = r ? Armed{nullptr, nullptr, 0} : c.arm();
    KScope ks(c, KC_PC_ROUND);
    hipLaunchKernelGGL(k_pc_round<true>, dim3((unsigned)g, (unsigned)L.n), kBlock, 0, c.stream, L, q, r ? *r : fr_zero(), E, mb, go);
    return mb.seq;
}
// tail hand-over: every table of the batch (folded by r when fold is set) into pinned host memory; the flag follows the last workgroup
__global__ __launch_bounds__(64) void k_pc_export(PcList L, size_t n_out, int fold, Fr r, Mailbox mb, Armed go) {
    if (go.want) { Fr v[1]; if (!armed_fetch<1>(go, v)) return; r = v[0]; }
    const Fr *T[3] = {L.A[blockIdx.y], L.B[blockIdx.y], L.C[blockIdx.y]};
    for (int t = 0; t < 3; t++) {
        if (!T[t]) continue;
        Fr *out = mb.host_results + mb.slot + ((size_t)3 * blockIdx.y + t) * n_out;
        for (size_t i = threadIdx.x; i < n_out; i += blockDim.x) {
            Fr v = T[t][i];
            if (fold) v = fr_add(v, fr_mul(r, fr_sub(T[t][i + n_out], v)));
            out[i] = v;
        }
    }
    __threadfence_system();
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        unsigned old = __hip_atomic_fetch_add(mb.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == gridDim.y - 1;
        if (s_last) {
            __hip_atomic_store(mb.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(mb.host_flag, mb.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
unsigned long long dev_pc_export(DevCtx &c, const PcList &L, size_t len, bool fold, const Fr *r, int slot) {
    const size_t n_out = fold ? len / 2 : len;
    if (slot + (size_t)3 * L.n * n_out > (size_t)kResultSlots) throw Error(OTTI_ERR_INTERNAL, "sum-check tail does not fit the pinned result buffer");
    Mailbox mb = c.next_mailbox(slot);
    const Armed go = (fold && !r) ? c.arm() : Armed{nullptr, nullptr, 0};
    KScope ks(c, KC_PC_ROUND);
    hipLaunchKernelGGL(k_pc_export, dim3(1, (unsigned)L.n), 64, 0, c.stream, L, n_out, fold ? 1 : 0, r ? *r : fr_zero(), mb, go);
    return mb.seq;
}
// ------------------------------------------------------------------------------------------------ the persistent tail (snark_dev.h)
struct TailArgs {
    PcList L; int W; uint32_t len0, t_out; int fold_on_load; Fr r; EqSrc E;
    TailMail *mail; Fr *host_out; unsigned long long seq0; Armed go;        // go.want: the go() number of the first round's challenge
    unsigned long long *stamps;                                             // OTTI_TAIL_STAMPS: wall_clock64 (100 MHz) of workgroup (0,0) at the phase boundaries, 8 per round
    int test_drop;                                                          // OTTI_TEST_TAIL_DROP (tests): every other workgroup returns at once, as if it had never become resident
};
// A wave works on ONE evaluation point (wave % 3) and group g = wave / 3 takes the pairs g*64 + lane, + 64*groups, ...: no divergence over
// the point inside a wave, and a wave's 64 partial sums of its one point fold by shuffles of ONE field element per level (the first
// version gave every lane its own point and reduced three elements, two of them zero, over all lanes: 11 us at 1024 elements per table).
// s_part[pt][g] <- the wave's sum; threads 0..2 add the groups that had pairs into s_tot[0..3).  Every thread of the workgroup calls it.
__device__ __forceinline__ void tail_reduce1(Fr acc, int pt, int grp, int ngrp, uint32_t half, Fr (*s_part)[16], Fr *s_tot) {
    const int lane = threadIdx.x & 63, groups = (int)min((uint32_t)ngrp, (half + 63) / 64);
    if (grp < groups) {
        int top = 64; if (half < 64) { top = 1; while ((uint32_t)top < half) top <<= 1; }          // lanes at and beyond `half` hold zero: only as many levels as there is data
#pragma unroll 1
        for (int off = top >> 1; off >= 1; off >>= 1) acc = fr_add(acc, shfl_xor_fr(acc, off));
        if (lane == 0) s_part[pt][grp] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 3) { Fr t = s_part[threadIdx.x][0]; for (int g = 1; g < groups; g++) t = fr_add(t, s_part[threadIdx.x][g]); s_tot[threadIdx.x] = t; }
    __syncthreads();
}
// s_tot: the round's sums in LDS (visible to the whole workgroup).  A round's mail is the line in one instruction (snark_dev.h); the hand-over
// after the last round follows plain stores of the tables into pinned memory and keeps its release fence.
__device__ __forceinline__ void tail_post(TailMail *m, const Fr *s_tot, bool with_sums, unsigned long long seq) {
    if (with_sums) {
        if (threadIdx.x < 8) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const unsigned lane = threadIdx.x;
            u32x4 q = {0u, 0u, 0u, 0u};
            if (lane < 6) { const Fr &x = s_tot[lane >> 1]; for (int i = 0; i < 4; i++) q[i] = x.v[4 * (lane & 1) + i]; }
            else if (lane == 6) { const unsigned long long tag = go_tag(seq, s_tot, 3); q[0] = (uint32_t)seq; q[1] = (uint32_t)(seq >> 32); q[2] = (uint32_t)tag; q[3] = (uint32_t)(tag >> 32); }
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(reinterpret_cast<char *>(m) + 16 * lane), "v"(q) : "memory");
        }
    } else if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(kTailThreads) void k_pc_tail(TailArgs a) {
    __shared__ Fr T0[3 * kTailCap];                                    // three tables of kTailCap elements: A, B, C (96 KB of the CU's 160: one workgroup per CU)
    __shared__ Fr s_part[3][16]; __shared__ Fr s_tot[3];
    const int w = blockIdx.x, y = blockIdx.y, W = a.W, tid = threadIdx.x, nthr = blockDim.x;
    if (a.test_drop && ((w + y) & 1)) return;
    const Fr *src[3] = {a.L.A[y], a.L.B[y], a.L.C[y]};
    TailMail *const mail = a.mail + (size_t)y * W + w;
    uint32_t L = a.len0 / (uint32_t)W;                                  // this workgroup's share of every table
    // ---- load (folding by the previous round's challenge when the tables in HBM are one round behind); element j here is element j W + w there
    const Fr9 r_load5 = fr9_unpack5(a.r);
    for (uint32_t idx = tid; idx < 3 * L; idx += nthr) {
        const uint32_t t = idx / L, j = idx - t * L; const size_t i = (size_t)j * W + w;
        Fr v;
        if (!src[t]) v = eq_at(a.E, i);                                 // product circuit: the shared eq table, materialised (t == 2 only)
        else if (a.fold_on_load) { const Fr lo = src[t][i], hi = src[t][i + a.len0]; (void)fold9(lo, hi, r_load5, v); }
        else v = src[t][i];
        T0[t * kTailCap + j] = v;
    }
    __syncthreads();
    unsigned long long seq = a.seq0, want = a.go.want;
    const uint32_t L_out = a.t_out / (uint32_t)W;
    unsigned long long *stamp = (a.stamps && tid == 0 && w == 0 && y == 0) ? a.stamps : nullptr; int rnd = 0;
    if (stamp) stamp[0] = wall_clock64();
    const int wave = tid >> 6, lane = tid & 63, ngrp = (nthr >> 6) / 3, pt = wave % 3, grp = wave / 3;   // wave -> evaluation point (0, 2, 3) and group of pairs; the 16th wave of 1024 threads idles
    while (L > L_out) {
        const uint32_t half = L / 2;
        // ---- this round's sums: S_t = sum over pairs of (A_t B_t C_t), t in {0, 2, 3}
        // nine limbs (fr9.h): the first table plain, the other two times 32 — each of the two products then has exactly one operand carrying the
        // radix correction (abc_accum9 above has the bounds; here every operand comes canonical out of the LDS)
        Fr9 acc9 = fr9_zero(); unsigned n_acc = 0;
        if (grp < ngrp)
            for (uint32_t p = (uint32_t)grp * 64 + lane; p < half; p += (uint32_t)ngrp * 64) {
                const Fr9 a_lo = fr9_unpack(T0[p]), b_lo5 = fr9_unpack5(T0[kTailCap + p]), c_lo5 = fr9_unpack5(T0[2 * kTailCap + p]);
                Fr9 term;
                if (pt == 0) term = fr9_mul(fr9_mul(a_lo, b_lo5), c_lo5);
                else {
                    const Fr9 a_hi = fr9_unpack(T0[p + half]), b_hi5 = fr9_unpack5(T0[kTailCap + p + half]), c_hi5 = fr9_unpack5(T0[2 * kTailCap + p + half]);
                    const Fr9 x2 = fr9_norm(fr9_sub_kl<4>(fr9_add(a_hi, a_hi), a_lo));
                    const Fr9 y2 = fr9_sub_kl<128>(fr9_add(b_hi5, b_hi5), b_lo5), z2 = fr9_sub_kl<128>(fr9_add(c_hi5, c_hi5), c_lo5);
                    if (pt == 1) term = fr9_mul(fr9_mul(x2, y2), z2);
                    else {
                        const Fr9 x3 = fr9_norm(fr9_add(x2, fr9_sub_kl<4>(a_hi, a_lo)));
                        const Fr9 y3 = fr9_norm(fr9_add(y2, fr9_sub_kl<128>(b_hi5, b_lo5))), z3 = fr9_norm(fr9_add(z2, fr9_sub_kl<128>(c_hi5, c_lo5)));
                        term = fr9_mul(fr9_mul(x3, y3), z3);
                    }
                }
                acc9 = fr9_add(acc9, term);
                if ((++n_acc & 3u) == 0) acc9 = fr9_norm(acc9);
            }
        Fr acc = fr9_canon(fr9_norm(acc9));
        if (stamp) stamp[8 * rnd + 1] = wall_clock64();               // sums done
        tail_reduce1(acc, pt, grp, ngrp, half, s_part, s_tot);
        if (stamp) stamp[8 * rnd + 2] = wall_clock64();               // reduced
        tail_post(mail, s_tot, true, seq++);
        if (stamp) stamp[8 * rnd + 3] = wall_clock64();               // mailed
        // ---- the round's challenge, then bound_poly_var_top of the three tables in LDS
        Fr rv[1]; Armed g = a.go; g.want = want++;
        if (!armed_fetch<1>(g, rv)) return;
        if (stamp) stamp[8 * rnd + 4] = wall_clock64();               // challenge here
        const Fr9 r5 = fr9_unpack5(rv[0]);
        for (uint32_t idx = tid; idx < 3 * half; idx += nthr) {
            const uint32_t t = idx / half, e = idx - t * half;
            Fr *X = T0 + t * kTailCap;
            Fr w; (void)fold9(X[e], X[e + half], r5, w);
            X[e] = w;
        }
        __syncthreads();
        L = half;
        if (stamp) { stamp[8 * rnd + 5] = wall_clock64(); rnd++; stamp[8 * rnd] = stamp[8 * (rnd - 1) + 5]; }   // folded = next round's start
    }
    // ---- hand the host-played tail over: table t of instance y at host_out[(3 y + t) * t_out ..), element j of this workgroup at j W + w
    for (uint32_t idx = tid; idx < 3 * L; idx += nthr) {
        const uint32_t t = idx / L, j = idx - t * L;
        if (!src[t]) continue;
        a.host_out[((size_t)3 * y + t) * a.t_out + (size_t)j * W + w] = T0[t * kTailCap + j];
    }
    __threadfence_system();
    __syncthreads();
    tail_post(mail, s_tot, false, seq);
}
unsigned long long dev_pc_tail(DevCtx &c, const PcList &L, int W, size_t len0, size_t t_out, const Fr *fold_r, const EqSrc &E, int slot) {
    int rounds = 0; for (size_t l = len0; l > t_out; l >>= 1) rounds++;
    if (W < 1 || (W & (W - 1)) || L.n < 1 || L.n * W > kTailMaxGroups || len0 / (size_t)W > (size_t)kTailCap || (len0 & (len0 - 1)) || (t_out & (t_out - 1)) ||
        t_out < (size_t)W || len0 <= t_out || rounds < 1)
        throw Error(OTTI_ERR_INTERNAL, "persistent sum-check tail: bad geometry");
    if (slot + (size_t)3 * L.n * t_out > (size_t)kResultSlots) throw Error(OTTI_ERR_INTERNAL, "sum-check tail does not fit the pinned result buffer");
    c.ensure_tail_mail();
    TailArgs a;
    a.L = L; a.W = W; a.len0 = (uint32_t)len0; a.t_out = (uint32_t)t_out; a.fold_on_load = fold_r ? 1 : 0; a.r = fold_r ? *fold_r : fr_zero(); a.E = E;
    a.mail = c.d_tail_alias; a.host_out = c.d_results_alias + slot;
    a.seq0 = c.seq + 1; c.seq += (unsigned long long)rounds + 1;
    a.go = c.arm_many(rounds);
    static const bool test_drop = getenv("OTTI_TEST_TAIL_DROP") != nullptr;
    a.test_drop = test_drop ? 1 : 0;

    static const bool want_stamps = getenv("OTTI_TAIL_STAMPS") != nullptr;
    static thread_local unsigned long long *h_stamps = nullptr, *d_stamps = nullptr;
    a.stamps = nullptr;
    if (want_stamps) {
        if (!h_stamps) { OTTI_HIP(hipHostMalloc((void **)&h_stamps, 8 * 32 * 8, hipHostMallocDefault)); OTTI_HIP(hipHostGetDevicePointer((void **)&d_stamps, h_stamps, 0)); }
        else {                                                    // the previous launch's stamps (its kernel has long finished: the host went through all its rounds)
            const unsigned long long *t = h_stamps; const int nr_ = (int)t[8 * 31];
            for (int r = 0; r < nr_ && r < 30; r++)
                fprintf(stderr, "[otti] k_pc_tail W=%d L=%llu round %d: sums %.2f | reduce %.2f | mail %.2f | wait for the challenge %.2f | fold %.2f us\n", (int)t[8 * 31 + 1], t[8 * 31 + 2] >> r, r,
                        0.01 * (double)(t[8 * r + 1] - t[8 * r]), 0.01 * (double)(t[8 * r + 2] - t[8 * r + 1]), 0.01 * (double)(t[8 * r + 3] - t[8 * r + 2]),
                        0.01 * (double)(t[8 * r + 4] - t[8 * r + 3]), 0.01 * (double)(t[8 * r + 5] - t[8 * r + 4]));
        }
        memset(h_stamps, 0, 8 * 32 * 8); h_stamps[8 * 31] = (unsigned long long)rounds; h_stamps[8 * 31 + 1] = (unsigned long long)W; h_stamps[8 * 31 + 2] = len0 / (size_t)W;
        a.stamps = d_stamps;
    }
    KScope ks(c, KC_PC_ROUND);
    hipLaunchKernelGGL(k_pc_tail, dim3((unsigned)W, (unsigned)L.n), kTailThreads, 0, c.stream, a);
    return a.seq0;
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
    KScope ks(c, KC_DOT_MANY);
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
    KScope ks(c, KC_DOT_MANY);
    hipLaunchKernelGGL(k_sum3, dim3((unsigned)g, (unsigned)L.n), kBlock, 0, c.stream, L, n, partials);
    hipLaunchKernelGGL(k_reduce_many<1>, L.n, kBlock, 0, c.stream, (const Fr *)partials, g, c.d_results_alias + slot);
}

}  // namespace otti
