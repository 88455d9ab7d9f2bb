// Shared by the k_*.hip translation units: launch geometry, the inter-workgroup hand-off used by every kernel whose last workgroup
// finishes the job (sum-check rounds, small MSMs), and the wave/block reductions of field elements.
#pragma once
#include "device.h"
#include "fp10.h"
#include "fp9.h"
#include "fr9.h"
#include "pool.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <mutex>

namespace otti {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;           // 8 workgroups per CU; grid-stride beyond that
constexpr int kHeavyRow = 64;              // sparse rows longer than this go to the one-workgroup-per-row kernel

static inline int grid_for(size_t n) { size_t b = (n + kBlock - 1) / kBlock; return (int)std::max<size_t>(1, std::min<size_t>(b, kMaxBlocks)); }

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

// ------------------------------------------------------------------------------------------------ armed launches (device.h)
// Every workgroup calls this first.  Workgroup (0,0) — and only it — waits for the host's value in pinned memory against the launch's
// deadline and republishes its DECISION in HBM: the value, or abort (host said so / deadline passed).  The others wait for that
// decision alone; their own backstop (the deadline plus a quarter: the leader never ran) only exists so that a grid always drains.
// Returns false in every thread of the workgroup when the launch was aborted or gave up: the kernel then returns at once, having
// touched nothing.
// One poll of a GoBox = the header (sequence number, tag) and the N values in ONE batch of 16-byte loads, all issued before the first is waited
// for.  kSystem: the box is the host's (system scope: one PCIe round trip from go() to the challenge); otherwise the leader's copy in HBM
// (agent scope).  The loads of a batch may be served at different moments; go_tag() tells a batch that mixes old and new words.
typedef uint32_t go_u32x4 __attribute__((ext_vector_type(4)));
template <int N, bool kSystem> __device__ __forceinline__ void go_batch_load(const GoBox *box, go_u32x4 &hd, go_u32x4 (&w)[8]) {
    if constexpr (N == 1 && kSystem)
        asm volatile("global_load_dwordx4 %0, %3, off sc0 sc1\n\tglobal_load_dwordx4 %1, %3, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:48 sc0 sc1\n\t"
                     "s_waitcnt vmcnt(0)" : "=&v"(hd), "=&v"(w[0]), "=&v"(w[1]) : "v"(box) : "memory");
    else if constexpr (N == 1)
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %3, off offset:32 sc1\n\tglobal_load_dwordx4 %2, %3, off offset:48 sc1\n\t"
                     "s_waitcnt vmcnt(0)" : "=&v"(hd), "=&v"(w[0]), "=&v"(w[1]) : "v"(box) : "memory");
    else if constexpr (kSystem)
        asm volatile("global_load_dwordx4 %0, %9, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off offset:32 sc0 sc1\n\tglobal_load_dwordx4 %2, %9, off offset:48 sc0 sc1\n\t"
                     "global_load_dwordx4 %3, %9, off offset:64 sc0 sc1\n\tglobal_load_dwordx4 %4, %9, off offset:80 sc0 sc1\n\tglobal_load_dwordx4 %5, %9, off offset:96 sc0 sc1\n\t"
                     "global_load_dwordx4 %6, %9, off offset:112 sc0 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:128 sc0 sc1\n\tglobal_load_dwordx4 %8, %9, off offset:144 sc0 sc1\n\t"
                     "s_waitcnt vmcnt(0)" : "=&v"(hd), "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]) : "v"(box) : "memory");
    else
        asm volatile("global_load_dwordx4 %0, %9, off sc1\n\tglobal_load_dwordx4 %1, %9, off offset:32 sc1\n\tglobal_load_dwordx4 %2, %9, off offset:48 sc1\n\t"
                     "global_load_dwordx4 %3, %9, off offset:64 sc1\n\tglobal_load_dwordx4 %4, %9, off offset:80 sc1\n\tglobal_load_dwordx4 %5, %9, off offset:96 sc1\n\t"
                     "global_load_dwordx4 %6, %9, off offset:112 sc1\n\tglobal_load_dwordx4 %7, %9, off offset:128 sc1\n\tglobal_load_dwordx4 %8, %9, off offset:144 sc1\n\t"
                     "s_waitcnt vmcnt(0)" : "=&v"(hd), "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]) : "v"(box) : "memory");
}
// The leader workgroup polls the host's line with up to kGoPollers of its waves, their polls staggered: one poll is a PCIe round trip — 1.2 us alone,
// 1.9 us measured inside the persistent launch — and with a single poller the host's go() waited 0.95 us on average (up to 1.9) just to be
// looked at; with four the line is looked at every ~0.5 us.  The first wave to see the number (or an abort, or the deadline) decides for all.
template <int N> __device__ __forceinline__ bool armed_fetch(const Armed &a, Fr (&v)[N]) {
    static_assert(N == 1 || N == 4, "a GoBox carries one value or four");
    __shared__ Fr s_v[N]; __shared__ int s_state;             // 0: undecided, 1: the values are in s_v, 2: aborted / gave up, 3: a wave is writing its decision
    const bool leader = (blockIdx.x | blockIdx.y | blockIdx.z) == 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, pollers = leader ? min(a.pollers, (int)(blockDim.x >> 6)) : 1;
    if (threadIdx.x == 0) s_state = 0;
    __syncthreads();
    if (lane == 0 && wave < pollers) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), abort_bit = 1ull << 63;   // 100 MHz
        int ok = -1;
        Fr t[N];
        for (int k = 0; k < N; k++) t[k] = fr_zero();
        go_u32x4 hd, w[8];
        if (leader) {
            for (int i = 0; i < wave; i++) __builtin_amdgcn_s_sleep(15);                          // ~0.45 us apart
            bool timed_out = false; unsigned polls = 0;
            while (ok < 0) {
                if (__hip_atomic_load(&s_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) break;      // another wave has decided
                go_batch_load<N, true>(a.host, hd, w);
                const unsigned long long s = (unsigned long long)hd[0] | ((unsigned long long)hd[1] << 32), tag = (unsigned long long)hd[2] | ((unsigned long long)hd[3] << 32);
                if (s == a.want) {
                    for (int k = 0; k < N; k++) for (int i = 0; i < 4; i++) { t[k].v[i] = w[2 * k][i]; t[k].v[4 + i] = w[2 * k + 1][i]; }
                    if (go_tag(a.want, t, N) == tag) { ok = 1; break; }
                }
                // (the clock is looked at every 64th poll only: s_memrealtime is a round trip of its own — 0.7 us, measured as the difference between a
                // poll loop with it, 1.9 us per iteration, and the load alone, 1.16 — and sat on the path of every poll)
                if (s == ~0ull) ok = 0;
                else if ((++polls & 63u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > a.deadline) { ok = 0; timed_out = true; }
            }
            if (ok >= 0 && atomicCAS(&s_state, 0, 3) == 0) {
                if (timed_out) __hip_atomic_store(&a.host->timed_out, a.want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                // the copies for the other workgroups (device.h: kGoCopies of them): values, then (number, tag) in one 16-byte store; a reader checks
                // the tag as this workgroup checked the host's (an abort carries no values: its number alone says so)
                if (a.relay == 0) {
                    if (ok) for (int k = 0; k < N; k++) store_words_sc1(&a.dev->v[k], t[k].v, 8);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&a.dev->seq, ok ? a.want : (a.want | abort_bit), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    const unsigned long long s_out = ok ? a.want : (a.want | abort_bit), tag_out = ok ? go_tag(a.want, t, N) : 0ull;
                    go_u32x4 h; h[0] = (uint32_t)s_out; h[1] = (uint32_t)(s_out >> 32); h[2] = (uint32_t)tag_out; h[3] = (uint32_t)(tag_out >> 32);
                    for (int c = 0; c < kGoCopies; c++) {
                        char *box = reinterpret_cast<char *>(a.dev) + (size_t)c * kGoCopyStride;
                        if (ok)
                            for (int k = 0; k < N; k++) {
                                go_u32x4 lo, hi; for (int i = 0; i < 4; i++) { lo[i] = t[k].v[i]; hi[i] = t[k].v[4 + i]; }
                                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" :: "v"(box + 32 + 32 * k), "v"(lo), "v"(hi) : "memory");
                            }
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(box), "v"(h) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (stores this wave does not wait for were seen to reach the readers microseconds later)
                }
                for (int k = 0; k < N; k++) s_v[k] = t[k];
                __hip_atomic_store(&s_state, ok ? 1 : 2, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            const unsigned long long backstop = a.deadline + (a.deadline >> 2);   // the leader's deadline and a margin (it reports; this only drains the grid)
            const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            const GoBox *mine = reinterpret_cast<const GoBox *>(reinterpret_cast<const char *>(a.dev) + (size_t)(wg % kGoCopies) * kGoCopyStride);
            if (a.relay == 0) {
                for (unsigned polls = 0; ok < 0; polls++) {
                    const unsigned long long s = __hip_atomic_load(&a.dev->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (s == a.want) ok = 1;
                    else if (s == (a.want | abort_bit) || ((polls & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > backstop)) ok = 0;
                    else if ((polls & 0xfffu) == 0xfffu && __hip_atomic_load(&a.host->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ~0ull) ok = 0;
                    else __builtin_amdgcn_s_sleep(1);
                }
                if (ok) for (int k = 0; k < N; k++) load_words_sc1(t[k].v, &a.dev->v[k], 8);
            } else
            for (unsigned polls = 0; ok < 0; polls++) {
                go_batch_load<N, false>(mine, hd, w);
                const unsigned long long s = (unsigned long long)hd[0] | ((unsigned long long)hd[1] << 32), tag = (unsigned long long)hd[2] | ((unsigned long long)hd[3] << 32);
                if (s == a.want) {
                    for (int k = 0; k < N; k++) for (int i = 0; i < 4; i++) { t[k].v[i] = w[2 * k][i]; t[k].v[4 + i] = w[2 * k + 1][i]; }
                    if (go_tag(a.want, t, N) == tag) { ok = 1; break; }
                }
                if (s == (a.want | abort_bit) || ((polls & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > backstop)) ok = 0;
                // a leader that is not resident (a grid larger than the CUs this process may use) cannot pass the host's abort on: look at the
                // host word itself once in a while (one PCIe read per few thousand polls)
                else if ((polls & 0xfffu) == 0xfffu && __hip_atomic_load(&a.host->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == ~0ull) ok = 0;
                else __builtin_amdgcn_s_sleep(1);
            }
            for (int k = 0; k < N; k++) s_v[k] = t[k];
            s_state = ok ? 1 : 2;
        }
    }
    __syncthreads();
    for (int k = 0; k < N; k++) v[k] = s_v[k];
    return s_state == 1;
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

// ------------------------------------------------------------------------------------------------ shared by the sum-check kernels
struct Pair { Fr lo, hi; };
// entries (i, i + q) of a table of length 4q folded by r (bound_poly_var_top): the pair i of the folded table
__device__ __forceinline__ Pair fold_regs(const Fr &x0, const Fr &x1, const Fr &x2, const Fr &x3, const Fr &r) {
    Pair p; p.lo = fr_add(x0, fr_mul(r, fr_sub(x2, x0))); p.hi = fr_add(x1, fr_mul(r, fr_sub(x3, x1))); return p;
}
// eq table given as the tensor product of two small tables (k_eq_pyramid levels), or as one table once few variables are left
__device__ __forceinline__ Fr eq_at(const EqSrc &e, size_t i) {
    i = i * e.stride + e.offset;
    const bool top = e.top_bit >= 0 && ((i >> e.top_bit) & 1);
    if (e.top_bit >= 0) i &= ((size_t)1 << e.top_bit) - 1;
    Fr v = !e.hi ? e.lo[i] : fr_mul(e.hi[i >> e.lo_bits], e.lo[i & (((size_t)1 << e.lo_bits) - 1)]);
    if (e.top_bit >= 0) v = fr_mul(v, top ? e.top : fr_sub(fr_one(), e.top));
    return v;
}


// ------------------------------------------------------------------------------------------------ nine-limb helpers (fr9.h) of the round kernels
// E[i] << S-ish: the eq factor carrying S bits of radix correction: 32 E (S = 5) or 1024 E (S = 10), below 3 l / 65 l from two factors,
// below 2^S l from one table
template <int S> __device__ __forceinline__ Fr9 eq_s_at(const EqSrc &e, size_t i) {
    i = i * e.stride + e.offset;
    if (!e.hi) return fr9_unpack_s<S>(e.lo[i]);
    return fr9_mul(fr9_unpack_s<S>(e.hi[i >> e.lo_bits]), fr9_unpack5(e.lo[i & (((size_t)1 << e.lo_bits) - 1)]));
}
__device__ __forceinline__ Fr9 eq5_at(const EqSrc &e, size_t i) { return eq_s_at<5>(e, i); }
// q = (Q(0), Q(1), Q_inf) of Q(t) = Q(0) + c t + Q_inf t^2  ->  (Q(0), Q(2), Q(3)):  Q(2) = 2 (Q(1) + Q_inf) - Q(0),  Q(3) = 3 Q(1) + 6 Q_inf - 2 Q(0)
__device__ __forceinline__ void quadratic_to_023(Fr (&q)[3]) {
    const Fr t = fr_add(q[1], q[2]), t2 = fr_dbl(t);
    const Fr s2 = fr_sub(t2, q[0]);
    const Fr s3 = fr_sub(fr_add(fr_add(t2, t), fr_add(fr_dbl(q[2]), q[2])), fr_dbl(q[0]));
    q[1] = s2; q[2] = s3;
}
// a thread's running sums (limbs grow by < 2^29 per item: carried down every fourth item) -> canonical words
__device__ __forceinline__ void acc9_carry(Fr9 (&acc)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) acc[k] = fr9_norm(acc[k]);
}
template <int K> __device__ __forceinline__ void acc9_canon(Fr (&out)[K], const Fr9 (&acc)[K]) {
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = fr9_canon(fr9_norm(acc[k]));
}
// bound_poly_var_top of (x0, x2) by r (r5 = 32 r): the folded element, normalised and below 2.2 l, and its canonical word for the table
__device__ __forceinline__ Fr9 fold9(const Fr &x0, const Fr &x2, const Fr9 &r5, Fr &word) {
    const Fr9 a = fr9_unpack(x0);
    const Fr9 s = fr9_norm(fr9_add(a, fr9_mul(r5, fr9_sub_kl<2>(fr9_unpack(x2), a))));   // r5 (x2 - x0 + 2l) / 2^261 + l < 1.2 l
    word = fr9_pack_lt3l(s);
    return s;
}

}  // namespace otti
