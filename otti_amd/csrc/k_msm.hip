// K8 fixed-base MSM over the generator window table (commitments, bullet-reduction rounds), the table build, K10 bookkeeping.
#include "kernels_common.h"

namespace otti {

// ------------------------------------------------------------------------------------------------ K8 fixed-base MSM
// One workgroup sums a chunk of one row's terms (a whole row of up to 4096 terms in the bulk launches, walked in sub-chunks of
// kMsmMaxChunk / kMsmBulkChunk terms that fit the LDS; the accumulators live across the sub-chunks).  Phase 1 stages a sub-chunk's
// scalars in LDS as s' = raw(s) + K with
// K = sum_w 2^(c-1+cw): the signed radix-2^c digit of window w is then (window w of s') - 2^(c-1), with no carry chain between
// windows, so any thread can take any (term, window) pair.  Phase 2: thread (term lane tl, window w) walks the terms tl, tl+lanes, ..
// and adds the table entry |d| * 2^(cw) * P[base] (affine Niels, 96-byte gather from HBM/L2; negated in registers when d < 0) into
// its own accumulator with one 7-multiply mixed addition per pair.  Phase 3: LDS tree over the 256 accumulators.
constexpr int kMsmMaxChunk = 1024;             // terms staged at a time (LDS: 36 B each)
constexpr int kMsmBulkChunk = 512;             // bulk launches: terms per workgroup, and
constexpr int kMsmListCap = (kMsmBulkChunk + 8) * 16;   // their (term, window) work-list entries (2 B each): (chunk + extras) * W must fit
static_assert(kMsmBulkChunk + 8 <= 2048, "work-list entries pack the term index (within a sub-chunk) in 11 bits");
// Bullet-reduction round fused into the MSM launch: the scalars of rows L (0) and R (1) are not read from memory but derived in phase 1
// from the round state (a, b: the two folded vectors; s: coefficients of the original generators), after applying the previous
// round's challenge.  State is ping-ponged (read *_in, write *_out) so that no workgroup of the launch reads what another one writes.
// The latency-bound kernel runs its scalar preparation exactly once per launch, from a cold instruction cache: two dozen inlined
// copies of the 250-instruction Montgomery product cost more in instruction fetch than in arithmetic.  One shared copy instead.
__device__ __attribute__((noinline)) Fr fr_mul_shared(Fr a, Fr b) { return fr_mul(a, b); }
struct BulletArgs { int on, fold; uint32_t n; const Fr *a_in, *b_in, *s_in; Fr *a_out, *b_out, *s_out; Fr u, uinv, u_raw, uinv_raw; };
struct MsmArgs {
    const TabEntry *table; int c, W; uint32_t E; int lanes;            // lanes = 256 / W term lanes
    const Fr *dense; size_t stride, n_dense; uint32_t chunk, nchunks;
    const Fr *extra_s; uint32_t extra_base[8]; int n_extra;
    uint32_t K[9];                                                  // recoding constant (288 bits)
    Pt *partial;
    // fused finish (rows <= 2, 1 < nchunks <= 128): the last workgroup to arrive sums every row's partials and mails the extended
    // row sums to pinned host memory, then raises the host flag — no finish launch, no copy, no stream synchronise
    int fuse; uint32_t rows; unsigned *counter; Pt *host_pts; unsigned long long *host_flag; unsigned long long seq;
    BulletArgs bul;
    Armed go;                                                       // armed bullet round: {u, u_inv, raw(u), raw(u_inv)} arrive through the GoBox
    unsigned long long *stamps;                                     // OTTI_MSM_STAMPS: s_memrealtime (100 MHz) at the phase boundaries of k_msm_small
};
__device__ __forceinline__ Fr bullet_fold_a(const BulletArgs &U, size_t x) { return U.fold ? fr_add(fr_mul_shared(U.a_in[x], U.u), fr_mul_shared(U.uinv, U.a_in[U.n + x])) : U.a_in[x]; }
__device__ __forceinline__ Fr bullet_fold_b(const BulletArgs &U, size_t x) { return U.fold ? fr_add(fr_mul_shared(U.b_in[x], U.uinv), fr_mul_shared(U.u, U.b_in[U.n + x])) : U.b_in[x]; }
// Row L only has non-zero scalars on the generator slots of the upper half of every length-n block, row R on the lower half: a
// bullet launch therefore walks R/2 "active" terms per row; term t of row `row` sits on generator j = (t / h) * n + (t mod h) + (row == 0 ? h : 0).
__device__ __forceinline__ size_t bullet_slot(const BulletArgs &U, size_t row, size_t t) {
    const size_t n = U.n, h = n / 2;
    return (t / h) * n + (t % h) + (row == 0 ? h : 0);
}
__device__ __forceinline__ Fr bullet_fold_s(const BulletArgs &U, size_t j) { return U.fold ? fr_mul_shared(U.s_in[j], ((j & (2 * (size_t)U.n - 1)) < U.n) ? U.uinv : U.u) : U.s_in[j]; }
// s' = raw(s) + K with K = sum_w 2^(c-1+cw): nine words per scalar; window w of s' minus 2^(c-1) is the signed digit of window w
__device__ __forceinline__ void recode_scalar(uint32_t *dst9, const Fr &sc, const uint32_t (&K)[9]) {
    const Fr raw = fr_to_raw(sc);
    uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { cy += (uint64_t)raw.v[i] + K[i]; dst9[i] = (uint32_t)cy; cy >>= 32; }
    dst9[8] = (uint32_t)cy + K[8];
}
__device__ __forceinline__ int recoded_digit(const uint32_t *src9, int w, int c) {
    const int pos = w * c, limb = pos >> 5, off = pos & 31;
    uint64_t x = src9[limb];
    if (limb < 8) x |= (uint64_t)src9[limb + 1] << 32;
    return (int)((uint32_t)(x >> off) & ((1u << c) - 1u)) - (1 << (c - 1));
}
// kKind = MSM_BULK: the bulk launches (a commitment: many rows, every workgroup a full chunk).
// kKind = MSM_BULK_SPARSE: the same for scalars that are mostly small numbers (compacted work list, see phase 2).
// The one/two-row and tiny-row launches (evaluation proof, blinding commitments) go to k_msm_small below: they are latency-bound and
// organised differently.  Separate kernels also keep them apart in profiles (k_msm_rows<0> is the witness commitment).
enum { MSM_BULK = 0, MSM_SMALL = 1, MSM_BULK_SPARSE = 2 };
template <int kKind> __global__ __launch_bounds__(kBlock) void k_msm_rows(MsmArgs A) {
    static_assert(kKind == MSM_BULK || kKind == MSM_BULK_SPARSE, "small launches use k_msm_small");
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
    // ---- phase 1 (recoded scalars into LDS) happens sub-chunk by sub-chunk below: a workgroup takes a whole chunk — a row, as a rule —
    // in pieces that fit the LDS and keeps its accumulators across them, so there is ONE reduction tree and one partial per chunk
    if (threadIdx.x < 8) s_base[threadIdx.x] = A.extra_base[threadIdx.x];
    __syncthreads();
    // ---- phase 2: one mixed addition per (term, window) pair, in nine 29-bit limbs (fp9.h: 90 multiply-adds per product instead of
    // fp10.h's 100 + the 19-folds, one asm block each; the tree below and everything latency-bound stay in fp10.h's form)
    P9 acc9 = p9_identity();
    const int w = threadIdx.x % A.W, tl = threadIdx.x / A.W;
    if constexpr (kKind == MSM_BULK_SPARSE) {
        // First compact the pairs whose digit is non-zero into an LDS work list (wave-aggregated append), then every thread takes
        // list entries round-robin.  A witness produced by a compiler is mostly small numbers (bits, counters, fixed-point values):
        // with c = 16 a 64-bit scalar has 4-5 non-zero digits out of 16, and with the fixed (term, window) mapping below the lanes
        // of its empty windows would idle while the others work.  (For uniform scalars the list would simply be all pairs, at the
        // price of half-size chunks; the host picks this variant from the witness's share of small values.)
        // A workgroup takes a whole chunk (a row, as a rule) in sub-chunks of kMsmBulkChunk terms and keeps its accumulators across them:
        // ONE reduction tree per chunk — with ~5 additions per scalar the tree of a 512-term chunk cost as much as its additions.
        __shared__ uint16_t s_list[kMsmListCap];
        __shared__ uint32_t s_count;
        const uint32_t n_all = n_here + n_ex;
        const int pos = w * A.c, limb = pos >> 5, off = pos & 31;
        const uint32_t mask = (1u << A.c) - 1u; const int half = 1 << (A.c - 1);
        const unsigned lane = threadIdx.x & 63;
        const size_t WE = (size_t)A.W * A.E;
        const uint32_t sub_cap = min((uint32_t)kMsmBulkChunk, (uint32_t)(kMsmListCap / A.W));      // terms whose pairs fit the work list
        for (uint32_t sub0 = 0; sub0 < n_all; sub0 += sub_cap) {
            const uint32_t n_tot = min(sub_cap, n_all - sub0), iters = (n_tot + A.lanes - 1) / A.lanes;
            __syncthreads();                                     // the previous sub-chunk's scalars and list have been consumed
            for (uint32_t t = threadIdx.x; t < n_tot; t += blockDim.x) {
                const uint32_t T = sub0 + t;
                const Fr sc = T >= n_here ? A.extra_s[row * A.n_extra + (T - n_here)] : A.dense[row * A.stride + j0 + T];
                recode_scalar(s_raw + t * 9, sc, A.K);
            }
            if (threadIdx.x == 0) s_count = 0;
            __syncthreads();
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
            for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
                const uint32_t e16 = s_list[i], t = e16 >> 5, ww = e16 & 31u, T = sub0 + t;
                const int d = recoded_digit(s_raw + t * 9, (int)ww, A.c);
                const size_t base = T < n_here ? j0 + T : (size_t)s_base[T - n_here];
                const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
                N9 e = n9_unpack(A.table[base * WE + (size_t)ww * A.E + (mag - 1)].n);
                if (d < 0) e = n9_negate(e);
                acc9 = p9_madd(acc9, e);
            }
        }
    } else {
        const size_t WE = (size_t)A.W * A.E;
        const uint32_t n_all = n_here + n_ex;
        for (uint32_t sub0 = 0; sub0 < n_all; sub0 += kMsmMaxChunk) {
            const uint32_t n_tot = min((uint32_t)kMsmMaxChunk, n_all - sub0);
            if (sub0) __syncthreads();                           // the previous sub-chunk's scalars have been consumed
            for (uint32_t t = threadIdx.x; t < n_tot; t += blockDim.x) {
                const uint32_t T = sub0 + t;
                const Fr sc = T >= n_here ? A.extra_s[row * A.n_extra + (T - n_here)] : A.dense[row * A.stride + j0 + T];
                recode_scalar(s_raw + t * 9, sc, A.K);
            }
            __syncthreads();
            if (tl < A.lanes)
                for (uint32_t t = tl; t < n_tot; t += A.lanes) {
                    const int d = recoded_digit(s_raw + t * 9, w, A.c);
                    if (d == 0) continue;
                    const uint32_t T = sub0 + t;
                    size_t base = T < n_here ? j0 + T : (size_t)s_base[T - n_here];
                    uint32_t mag = (uint32_t)(d < 0 ? -d : d);
                    N9 e = n9_unpack(A.table[base * WE + (size_t)w * A.E + (mag - 1)].n);
                    if (d < 0) e = n9_negate(e);
                    acc9 = p9_madd(acc9, e);
                }
        }
    }
    // ---- phase 3: LDS tree (reuses the scalar region: everyone must be done reading it)
    __syncthreads();
    P10 acc = p10_unpack(p9_pack(acc9));
    const F10 d2 = f10_const(fp_2D());
    for (int sft = kBlock / 2; sft >= 1; sft >>= 1) {
        if ((int)threadIdx.x >= sft && (int)threadIdx.x < 2 * sft) sm[threadIdx.x - sft] = acc;
        __syncthreads();
        if ((int)threadIdx.x < sft) acc = p10_add(acc, sm[threadIdx.x], d2);
        __syncthreads();
    }
    if (threadIdx.x == 0) A.partial[row * A.nchunks + chunk_id] = p10_pack(acc);
}

// ------------------------------------------------------------------------------------------------ the latency-bound launches
// One or two rows of up to R terms (Cx, the bullet-reduction rounds, delta), or many rows of a handful of terms (blind * h per row
// commitment, the tape-only points of every sum-check round).  Nothing here is throughput: a 2^20 proof makes 14 such launches one
// after the other, each a dependent chain  scalars -> one mixed addition per (term, window) pair -> a reduction tree over ~16 k
// points -> two points to the host.  The chain is what is shortened:
//   * every addition is quad-parallel (fp10.h): 4 lanes per point, 2-3 multiplication depths per addition instead of 7-9;
//   * a workgroup (64 quads) takes a SHORT chunk (about two pairs per quad), so the tree starts almost at once; its six levels go
//     through LDS with one barrier each (every level has its own slots);
//   * rows of at most two: the last workgroup to arrive (agent-scope counter, sc1 hand-off) sums the chunk results the same way
//     and mails the extended row sums to pinned host memory — no finish launch, no copy, no stream synchronise;
//   * a bullet round's c_L / c_R is not computed by one workgroup ahead of its MSM: every workgroup takes a slice of the dot product
//     and adds (its slice) * Q as one more term — the sum over workgroups is c_L * Q; the folded state (a, b, s) for the next
//     round is written after the workgroup has handed its point over, off the path to the host.
constexpr int kSmallChunk = 64;                // terms per workgroup at most (LDS: 36 B each)
constexpr int kSmallQuads = kBlock / 4;
// the folded a in RAW (non-Montgomery) form: a Montgomery product with one raw operand is raw, so the scalars of L / R and the slices of
// c_L / c_R come out ready for recoding without a conversion multiplication on the path to the first addition
__device__ __forceinline__ Fr bullet_fold_a_raw(const BulletArgs &U, size_t x) {
    Fr one = fr_zero(); one.v[0] = 1;
    return U.fold ? fr_add(fr_mul_shared(U.a_in[x], U.u_raw), fr_mul_shared(U.uinv_raw, U.a_in[U.n + x])) : fr_mul_shared(U.a_in[x], one);
}
__device__ __forceinline__ void recode_raw(uint32_t *dst9, const Fr &raw, const uint32_t (&K)[9]) {
    uint64_t cy = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { cy += (uint64_t)raw.v[i] + K[i]; dst9[i] = (uint32_t)cy; cy >>= 32; }
    dst9[8] = (uint32_t)cy + K[8];
}
__global__ __launch_bounds__(kBlock) void k_msm_small(MsmArgs A) {
    __shared__ uint32_t s_raw[(kSmallChunk + 8) * 9];
    __shared__ uint32_t s_base[8];
    __shared__ F10 s_tree[kSmallQuads - 1][4];                   // a level with `sft` writers per row segment uses slots [(sft - 1) * segs, (2 sft - 1) * segs)
    const size_t row = blockIdx.y; const uint32_t chunk_id = blockIdx.x;
    const size_t j0 = (size_t)chunk_id * A.chunk;
    const uint32_t n_here = j0 < A.n_dense ? (uint32_t)min((size_t)A.chunk, A.n_dense - j0) : 0u;
    const bool bullet = A.bul.on != 0;
    BulletArgs U = A.bul;
    if (A.go.want) { Fr v[4]; if (!armed_fetch<4>(A.go, v)) return; U.u = v[0]; U.uinv = v[1]; U.u_raw = v[2]; U.uinv_raw = v[3]; }
    // extras: plain launches carry them in chunk 0; a bullet round has {slice of <a, b> on Q} everywhere and {blind on H} in chunk 0
    const uint32_t n_ex = bullet ? (chunk_id == 0 ? 2u : 1u) : (chunk_id == 0 ? (uint32_t)A.n_extra : 0u);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool stamp0 = A.stamps && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0;
    if (stamp0) A.stamps[0] = wall_clock64();
    // ---- scalars, recoded, into LDS.  Wave 0: the chunk's terms; wave 1: the bullet dot-product slice; wave 2: the other extras
    Fr keep_s = fr_zero();                                       // bullet: the folded generator coefficient of this thread's term
    Fr raw_one = fr_zero(); raw_one.v[0] = 1;                    // Montgomery product with the integer 1 = conversion to the raw integer
    if (wave == 0 && (uint32_t)lane < n_here) {
        if (bullet) {
            const size_t j = bullet_slot(U, row, j0 + lane), n = U.n, h = n / 2, i = j & (n - 1);
            keep_s = bullet_fold_s(U, j);
            // L = <a_L, G_R>: generator slots of the upper half, paired with a[i - h];  R = <a_R, G_L>: lower half with a[i + h]
            recode_raw(s_raw + lane * 9, fr_mul_shared(bullet_fold_a_raw(U, row == 0 ? i - h : i + h), keep_s), A.K);
        } else recode_raw(s_raw + lane * 9, fr_mul_shared(A.dense[row * A.stride + j0 + lane], raw_one), A.K);
    } else if (wave == 1 && bullet) {                            // c_L = <a_L, b_R> (row 0), c_R = <a_R, b_L> (row 1): elements chunk_id, chunk_id + #chunks, ..
        const size_t h = U.n / 2;
        Fr acc = fr_zero();
        for (size_t x = chunk_id + (size_t)lane * gridDim.x; x < h; x += (size_t)64 * gridDim.x)
            acc = fr_add(acc, row == 0 ? fr_mul_shared(bullet_fold_a_raw(U, x), bullet_fold_b(U, h + x)) : fr_mul_shared(bullet_fold_a_raw(U, h + x), bullet_fold_b(U, x)));
        // lanes at and beyond ceil(h / #chunks) hold zero: reduce only as far as there is data
        const size_t per = (h + gridDim.x - 1) / gridDim.x;
        for (int off = 32; off >= 1; off >>= 1) { if ((size_t)off < per) acc = fr_add(acc, shfl_xor_fr(acc, off)); }
        if (lane == 0) recode_raw(s_raw + n_here * 9, acc, A.K);
    } else if (wave == 2 && (uint32_t)lane < n_ex && !(bullet && lane == 0)) {
        recode_raw(s_raw + (n_here + lane) * 9, fr_mul_shared(A.extra_s[row * A.n_extra + lane], raw_one), A.K);
    }
    if (threadIdx.x < 8) s_base[threadIdx.x] = A.extra_base[threadIdx.x];
    __syncthreads();
    if (stamp0) A.stamps[1] = wall_clock64();
    // ---- Two passes through ONE body (the instruction stream of an addition is long and every launch starts with a cold
    // instruction cache: each phase running its own inlined copy cost more than the arithmetic).  Pass 0: this workgroup's (term,
    // window) pairs, quad `qid` taking pairs qid, qid + 64, .., then a tree over the quads; the result goes to the chunk's slot in
    // cached form.  Pass 1 (last workgroup to arrive only, rows <= 2): the same with the chunk results of a row as operands, the
    // quads split between the rows.  A step of the body: fetch an operand in cached form (table entry / chunk result / LDS slot of
    // the tree level), add it.  Operand loads are issued one step ahead of the addition that hides them.
    const int q = threadIdx.x & 3, qid = threadIdx.x >> 2;
    const size_t WE = (size_t)A.W * A.E;
    const F10 d2 = f10_const(fp_2D());
    F10 acc = q10_identity(q);
    bool last = false;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t segs = pass ? A.rows : 1u, qpr = kSmallQuads / segs, r = (uint32_t)qid / qpr, idx = (uint32_t)qid % qpr;
        const uint32_t count = pass ? A.nchunks : (n_here + n_ex) * (uint32_t)A.W;       // operands of this segment
        const int n_op = (int)((count + qpr - 1) / qpr);
        int top = 1; while ((uint32_t)top < qpr && (uint32_t)top < count) top <<= 1;
        int n_lv = 0; while ((1 << n_lv) < top) n_lv++;
        // operand fetch (issue only: the value is unpacked when it is consumed)
        Fp nxt; bool nxt_have = false, nxt_neg = false;
        auto fetch = [&](int step) {
            const uint32_t p = idx + (uint32_t)step * qpr;
            nxt_have = false;
            if (p >= count) return;
            if (pass) { load_words_sc1(nxt.v, reinterpret_cast<const Fp *>(&A.partial[(size_t)r * A.nchunks + p]) + q, 8); nxt_have = true; nxt_neg = false; return; }
            const uint32_t t = p / (uint32_t)A.W, w = p - t * (uint32_t)A.W;
            const int d = recoded_digit(s_raw + t * 9, (int)w, A.c);
            if (d == 0) return;
            const size_t base = t < n_here ? (bullet ? bullet_slot(U, row, j0 + t) : j0 + t) : (size_t)s_base[t - n_here];
            const Niels *e = &A.table[base * WE + (size_t)w * A.E + ((uint32_t)(d < 0 ? -d : d) - 1)].n;
            nxt_neg = d < 0; nxt_have = true;
            if (q < 3) nxt = reinterpret_cast<const Fp *>(e)[q == 2 ? 2 : (((q == 0) != nxt_neg) ? 1 : 0)];   // Niels = {yplusx, yminusx, xy2d}
        };
        if (pass) acc = q10_identity(q);
        if (n_op) fetch(0);
        F10 *const seg_tree = &s_tree[0][0] + (size_t)r * 4 + q;               // slot (level base + r * sft + k) of this lane: seg_tree[((sft - 1) * segs + r * (sft - 1) + k) * 4]
#pragma unroll 1
        for (int step = 0; step <= n_op + n_lv; step++) {
            const F10 ua = q10_u(acc, q);                                         // needed by both roles of a tree level; formed before any wait
            F10 v; bool have = false;
            if (step < n_op) {
                const Fp cur = nxt; const bool cur_neg = nxt_neg; have = nxt_have;
                if (step + 1 < n_op) fetch(step + 1);
                if (have) {
                    if (!pass && q == 3) { v = f10_zero(); v.v[0] = 2; }                 // a table entry is affine: 2 Z = 2
                    else { v = f10_unpack(cur); if (q == 2 && cur_neg) v = f10_carry(f10_neg(v)); }
                }
            } else {
                if (step == n_op && threadIdx.x == 0 && A.stamps && (pass || stamp0)) A.stamps[pass ? 5 : 2] = wall_clock64();
                const int sft = top >> (step - n_op + 1);                                // top/2, .., 1, then 0 = hand the segment's sum over
                const bool out_cached = sft == 0 && !pass && A.fuse;
                if (sft == 0 && !out_cached) break;
                F10 *const level = seg_tree + (size_t)((sft - 1) * (int)segs + (int)r * (sft - 1)) * 4;
                if (sft ? ((int)idx >= sft && (int)idx < 2 * sft) : idx == 0) {
                    const F10 cv = q10_cached(ua, q, d2);
                    if (sft) level[(idx - sft) * 4] = cv;
                    else { const Fp pk = f10_pack(cv); store_words_sc1(reinterpret_cast<Fp *>(&A.partial[row * A.nchunks + chunk_id]) + q, pk.v, 8); }
                }
                if (sft == 0) break;
                __syncthreads();
                if ((int)idx < sft) { v = level[idx * 4]; have = true; }
            }
            if (have) acc = q10_add_cached_u(ua, v, q);
        }
        if (pass == 0) {
            if (stamp0) A.stamps[3] = wall_clock64();
            if (!A.fuse) { if (qid == 0) reinterpret_cast<Fp *>(&A.partial[row * A.nchunks + chunk_id])[q] = f10_pack(acc); break; }
            last = arrive_and_check_last(A.counter, gridDim.x * gridDim.y);
            if (!last) break;
            if (A.stamps && threadIdx.x == 0) A.stamps[4] = wall_clock64();
        } else {
            if (A.stamps && threadIdx.x == 0) A.stamps[6] = wall_clock64();
            if (idx == 0) { reinterpret_cast<Fp *>(&A.host_pts[r])[q] = f10_pack(acc); __threadfence_system(); }
            __syncthreads();
            if (threadIdx.x == 0) {
                __threadfence_system();
                __hip_atomic_store(A.host_flag, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (A.stamps) A.stamps[7] = wall_clock64();
            }
        }
    }
    // ---- a bullet round persists the folded state for the next round, each element written once: a and b by the row-0 workgroups,
    // s by whichever row walks that generator slot.  State is ping-ponged, so nobody reads what this launch writes.  (After the
    // hand-over: the host is already hashing L and R while this is written.)
    if (bullet) {
        const size_t n = U.n;
        if (row == 0) {
            const size_t cx = (n + gridDim.x - 1) / gridDim.x, x0 = (size_t)chunk_id * cx;
            for (size_t x = x0 + threadIdx.x; x < min(n, x0 + cx); x += blockDim.x) { U.a_out[x] = bullet_fold_a(U, x); U.b_out[x] = bullet_fold_b(U, x); }
        }
        if (wave == 0 && (uint32_t)lane < n_here) U.s_out[bullet_slot(U, row, j0 + lane)] = keep_s;
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
// out32: device memory; host32 (may be null): the same 32 bytes straight into pinned host memory — the host reads them after the
// kernel's completion event, without a copy-engine transfer behind the kernel (~10 us of latency for 32 KB)
__global__ __launch_bounds__(64) void k_encode_points(const Pt *pts, const Pt *addend, size_t n, uint8_t *out32, uint8_t *host32) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    P10 p = p10_unpack(pts[i]);
    if (addend) p = p10_add(p, p10_unpack(addend[i]), f10_const(fp_2D()));
    uint8_t enc[32]; p10_encode(enc, p);
    uint32_t w[8];
    for (int k = 0; k < 8; k++) w[k] = (uint32_t)enc[4 * k] | ((uint32_t)enc[4 * k + 1] << 8) | ((uint32_t)enc[4 * k + 2] << 16) | ((uint32_t)enc[4 * k + 3] << 24);
    uint32_t *o = (uint32_t *)(out32 + 32 * i);
    for (int k = 0; k < 8; k++) o[k] = w[k];
    if (host32) { uint32_t *h = (uint32_t *)(host32 + 32 * i); for (int k = 0; k < 8; k++) h[k] = w[k]; }
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
    OTTI_HIP(hipMemsetAsync(c.d_counts.p, 0, sizeof(unsigned long long), c.stream));
    hipLaunchKernelGGL(k_count_small, grid_for(n), kBlock, 0, c.stream, z, n, c.d_counts.p);
    unsigned long long h = 0;
    OTTI_HIP(hipMemcpyAsync(&h, c.d_counts.p, sizeof h, hipMemcpyDeviceToHost, c.stream));
    OTTI_HIP(hipStreamSynchronize(c.stream));
    return (double)h / (double)n;
}
unsigned long long dev_bullet_round(DevCtx &c, const DeviceGens &g, size_t R, size_t n_cur, bool fold, const Fr &u, const Fr &u_inv, const Fr *a_in,
                                    const Fr *b_in, const Fr *s_in, Fr *a_out, Fr *b_out, Fr *s_out, const Fr *extra_s, const uint32_t *extra_base, bool armed) {
    BulletArgs U; U.on = armed ? 2 : 1; U.fold = fold ? 1 : 0; U.n = (uint32_t)n_cur; U.a_in = a_in; U.b_in = b_in; U.s_in = s_in;
    U.a_out = a_out; U.b_out = b_out; U.s_out = s_out; U.u = u; U.uinv = u_inv; U.u_raw = fr_to_raw(u); U.uinv_raw = fr_to_raw(u_inv);
    return msm_launch(c, g, nullptr, 0, R / 2, 2, extra_s, extra_base, 2, MSM_COMPRESSED, nullptr, &U, false);   // R/2 active terms per row
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
    const bool bulk = rows * n_dense >= ((size_t)1 << 16) && !bul;
    const bool sparse = bulk && sparse_hint && g.W <= 32;
    size_t nchunks;
    if (bulk) {
        // aim for >= 1024 workgroups (4 per CU) but keep at least one term per term lane and at most kMsmMaxChunk per workgroup;
        // the sparse variant keeps a (term, window) work list in LDS: (chunk + extras) * W <= kMsmListCap;  W <= 32 there (5-bit window field)
        nchunks = std::max<size_t>(1, (1024 + rows - 1) / rows);
        nchunks = std::min(nchunks, std::max<size_t>(1, n_dense / (size_t)A.lanes));
        const size_t max_chunk = 4096;                        // both bulk kernels walk their chunk in sub-chunks that fit the LDS
        nchunks = std::max(nchunks, (n_dense + max_chunk - 1) / max_chunk);
    } else {
        // latency-bound launches: about two (term, window) pairs per quad (one or two rows) or four (many rows), at most 2048 workgroups
        // and, when the last workgroup sums the chunk results itself (rows <= 2), at most 256 of them per row
        // (whole steps: a workgroup's pairs, extras included, should fill its 64 quads k times — the bullet rounds carry one extra
        // term per workgroup, the slice of c_L / c_R, and two in chunk 0)
        const size_t steps = rows <= 2 ? 3 : 4, per_wg = steps * (size_t)kSmallQuads / (size_t)g.W, ex_wg = bul ? 2 : n_extra;
        const size_t terms_wg = per_wg > ex_wg ? per_wg - ex_wg : 1;
        nchunks = std::max<size_t>(1, (n_dense + terms_wg - 1) / terms_wg);
        nchunks = std::min(nchunks, rows <= 2 ? (size_t)256 : std::max<size_t>(1, 2048 / rows));
        nchunks = std::min(nchunks, std::max<size_t>(1, n_dense));
        nchunks = std::max(nchunks, (n_dense + kSmallChunk - 1) / (size_t)kSmallChunk);
    }
    if (!n_dense) nchunks = 1;
    size_t chunk = n_dense ? (n_dense + nchunks - 1) / nchunks : 1;
    nchunks = n_dense ? (n_dense + chunk - 1) / chunk : 1;
    A.chunk = (uint32_t)chunk; A.nchunks = (uint32_t)nchunks;
    for (int i = 0; i < 9; i++) A.K[i] = 0;
    for (int w = 0; w < g.W; w++) { int bit = g.c - 1 + g.c * w; A.K[bit >> 5] |= 1u << (bit & 31); }
    c.ensure_points(rows, nchunks);
    A.partial = c.msm_partial.p;
    A.fuse = (!bulk && mode == MSM_COMPRESSED && !addend && rows <= 2 && rows * nchunks <= 512) ? 1 : 0;
    if (bul) A.bul = *bul; else { memset(&A.bul, 0, sizeof A.bul); }
    A.go = Armed{nullptr, nullptr, 0};
    A.rows = (uint32_t)rows; A.counter = c.d_counter2.p; A.host_pts = c.d_pts_alias; A.host_flag = c.d_flag_alias; A.seq = A.fuse ? ++c.seq : 0;
    dim3 grid((unsigned)nchunks, (unsigned)rows);
    // OTTI_MSM_STAMPS=1: phase stamps of every fused small launch on stderr (development aid; synchronises the stream)
    static const bool want_stamps = getenv("OTTI_MSM_STAMPS") != nullptr;
    static thread_local unsigned long long *h_stamps = nullptr, *d_stamps = nullptr;
    A.stamps = nullptr;
    if (want_stamps && A.fuse) {
        if (!h_stamps) { OTTI_HIP(hipHostMalloc((void **)&h_stamps, 128, hipHostMallocDefault)); OTTI_HIP(hipHostGetDevicePointer((void **)&d_stamps, h_stamps, 0)); }
        memset(h_stamps, 0, 128); A.stamps = d_stamps;
    }
    if (bul && bul->on == 2) { A.bul.on = 1; if (!A.stamps) A.go = c.arm(); else throw Error(OTTI_ERR_INTERNAL, "armed launches cannot be stamped"); }
    {
        KScope ks(c, bulk ? KC_MSM_ROWS : KC_MSM_SMALL);
        if (sparse) hipLaunchKernelGGL(k_msm_rows<MSM_BULK_SPARSE>, grid, kBlock, 0, c.stream, A);
        else if (bulk) hipLaunchKernelGGL(k_msm_rows<MSM_BULK>, grid, kBlock, 0, c.stream, A);
        else hipLaunchKernelGGL(k_msm_small, grid, kBlock, 0, c.stream, A);
    }
    if (A.stamps) {
        OTTI_HIP(hipStreamSynchronize(c.stream));
        const unsigned long long *t = h_stamps;
        auto us = [&](int a, int b) { return t[b] >= t[a] ? (double)(t[b] - t[a]) * 0.01 : -1.0; };
        fprintf(stderr, "[otti] k_msm_small rows=%zu terms=%zu chunks=%zu bullet=%d: scalars %.2f | pairs %.2f | tree %.2f | (others arrive) %.2f | chunk sums %.2f | row tree %.2f | mail %.2f | total %.2f us\n",
                rows, n_dense + n_extra, nchunks, bul ? 1 : 0, us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(0, 7));
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
        hipLaunchKernelGGL(k_encode_points, (unsigned)((rows + 63) / 64), 64, 0, c.stream, finals, addend, rows, c.d_points.p, c.d_points_host);
        if (!c.d_points_host) OTTI_HIP(hipMemcpyAsync(c.h_points, c.d_points.p, rows * 32, hipMemcpyDeviceToHost, c.stream));
        c.pending_host_encode = 0;
    } else {
        // a handful of points: the dependent inverse-square-root chain runs ~30x faster on a host core than on one GPU lane
        OTTI_HIP(hipMemcpyAsync(c.h_pts, finals, rows * sizeof(Pt), hipMemcpyDeviceToHost, c.stream));
        c.pending_host_encode = rows;
    }
    return 0;
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
__global__ __launch_bounds__(kBlock) void k_table_fill(const Pt *starts, size_t nrows, size_t nblk, size_t T, Pt *tmp, TabEntry *out) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nrows * nblk) return;
    const size_t r = t / nblk;
    const Pt B = starts[r * nblk];
    Pt acc = starts[t];
    Pt *blk = tmp + t * T; TabEntry *oblk = out + t * T;                                      // row r, block k: entries r*E + k*T ..
    Fp prod = fp_one();
    for (size_t d = 0; d < T; d++) {
        if (d) acc = pt_add(acc, B);
        Pt p = acc; p.T = prod;                                                             // T is not needed for the affine form: park the prefix product there
        blk[d] = p; prod = fp_mul(prod, acc.Z);
    }
    Fp inv = fp_inv(prod);
    for (size_t d = T; d-- > 0;) { Pt p = blk[d]; Fp zinv = fp_mul(inv, p.T); inv = fp_mul(inv, p.Z); oblk[d].n = pt_to_niels(p, zinv); }
}
std::shared_ptr<DeviceGens> build_device_gens(const Gens &g, int c) {
    DevCtx &ctx = DevCtx::get();
    auto d = std::make_shared<DeviceGens>();
    d->c = c; d->W = 253 / c + 1; d->E = (size_t)1 << (c - 1); d->nbases = g.P.size();
    const size_t per_base = (size_t)d->W * d->E;
    const int lgT = std::min(6, c - 1); const size_t T = (size_t)1 << lgT, nblk = d->E / T;
    // laps (otti_gens_build_ms): the allocation of tens of GB is the part that differs by an order of magnitude between boxes and between
    // a fresh and a used process (the driver maps and clears the pages); the kernels scale with the table and nothing else
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    d->table.alloc(d->nbases * per_base);
    size_t chunk = std::max<size_t>(1, ((size_t)4 << 30) / (per_base * sizeof(Pt)));            // <= 4 GiB of extended temporaries
    chunk = std::min(chunk, d->nbases);
    DevBuf<Pt> bases(d->nbases), starts(d->nbases * d->W * nblk), tmp(chunk * per_base);
    const double t1 = now();
    OTTI_HIP(hipMemcpy(bases.p, g.P.data(), d->nbases * sizeof(Pt), hipMemcpyHostToDevice));
    { size_t n = d->nbases * d->W; hipLaunchKernelGGL(k_table_starts, (unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, ctx.stream, (const Pt *)bases.p, d->nbases, c, d->W, nblk, lgT, starts.p); }
    for (size_t b0 = 0; b0 < d->nbases; b0 += chunk) {
        size_t nb = std::min(chunk, d->nbases - b0), nrows = nb * d->W, nthreads = nrows * nblk;
        hipLaunchKernelGGL(k_table_fill, (unsigned)((nthreads + kBlock - 1) / kBlock), kBlock, 0, ctx.stream, (const Pt *)(starts.p + b0 * d->W * nblk), nrows, nblk, T,
                           tmp.p, d->table.p + b0 * per_base);
    }
    ctx.sync();
    const double t2 = now();
    d->build_ms[0] = t1 - t0; d->build_ms[1] = t2 - t1;
    return d;
}

// ------------------------------------------------------------------------------------------------ verifier: variable-base MSM
// PolyEvalProof::verify needs  C_LZ = sum_i L[i] * C_i  over the sqrt(V) row commitments of the proof (dense_mlpoly.rs
// PolyEvalProof::verify -> vartime_multiscalar_mul [RECALL]): points nobody has a table for, uniform 253-bit scalars.  Pippenger with the
// buckets in LDS: workgroup (window w, split s) takes the signed radix-2^c digits of window w of up to kVarChunk scalars, sorts the point
// indices by bucket with an LDS counting sort (histogram by LDS atomics, prefix, scatter), then thread b walks bucket b's list with mixed
// additions (the decoded points are affine: 7 multiplications each) — about eight per bucket, by the choice of c — and the weighted
// bucket sum  sum_b b * B_b  is formed in 2 log2(#buckets) parallel steps: a suffix scan (T_b = sum_{j >= b} B_j), then a tree over the
// T_b.  The workgroup's result is one extended point per (window, split); the host adds the splits and runs the 253 doublings of the
// window recombination (a dependent chain: ~25 us on a host core against ~300 us on a GPU lane).
constexpr int kVarChunk = 2048, kVarMaxBuckets = kBlock;        // one bucket per thread at most: c <= 9
struct VarMsmArgs { const Niels *pts; const Fr *sc; uint32_t n, chunk; int c, W, splits; uint32_t K[9]; Pt *out; };
__global__ __launch_bounds__(kBlock) void k_msm_var(VarMsmArgs A) {
    __shared__ uint16_t s_key[kVarChunk], s_list[kVarChunk];        // per point: (|digit| - 1) | sign << 15, 0xffff for a zero digit; point indices sorted by bucket
    __shared__ uint32_t s_start[kVarMaxBuckets + 1], s_fill[kVarMaxBuckets];
    __shared__ P10 s_bkt[kVarMaxBuckets];
    const int w = blockIdx.x, tid = threadIdx.x;
    const uint32_t i0 = blockIdx.y * A.chunk, n_here = min(A.chunk, A.n - i0), nb = 1u << (A.c - 1);
    for (uint32_t b = tid; b <= nb; b += kBlock) { s_start[b] = 0; if (b < nb) s_fill[b] = 0; }
    __syncthreads();
    // The TOP window holds only the 253 - c (W - 1) leading bits of a scalar (often a single one) plus the recoding carry: few digit
    // values, so few buckets would take all the points (one thread adding a thousand points in a row).  A window whose digits have at most
    // vmax magnitudes therefore splits every bucket into m = #buckets / pow2(vmax) sub-buckets by point index: the lists stay ~8 long
    // whatever the window, the sub-buckets of a digit are summed by a short tree, and the weighted sum runs over pow2(vmax) buckets.
    const int top_bits = 253 - A.c * (A.W - 1);
    const uint32_t vmax = (w == A.W - 1 && top_bits < A.c - 1) ? (1u << top_bits) + 1u : nb;
    uint32_t V2 = 1; while (V2 < vmax) V2 <<= 1;
    const uint32_t m = nb / V2;                                         // sub-buckets per digit magnitude (1 for the regular windows)
    for (uint32_t i = tid; i < n_here; i += kBlock) {
        uint32_t r9[9]; recode_scalar(r9, A.sc[i0 + i], A.K);
        const int d = recoded_digit(r9, w, A.c);
        uint16_t key = 0xffffu;
        if (d) {
            const uint32_t mag = (uint32_t)(d < 0 ? -d : d), slot = (mag - 1) * m + (i & (m - 1));
            key = (uint16_t)(slot | (d < 0 ? 0x8000u : 0u)); atomicAdd(&s_start[slot + 1], 1u);      // counts land one slot up: the prefix below turns them into starts
        }
        s_key[i] = key;
    }
    __syncthreads();
    if (tid == 0) { uint32_t run = 0; for (uint32_t b = 0; b <= nb; b++) { run += s_start[b]; s_start[b] = run; } }      // s_start[b] = first list slot of bucket b (index |d| - 1)
    __syncthreads();
    for (uint32_t i = tid; i < n_here; i += kBlock) {
        const uint16_t key = s_key[i];
        if (key == 0xffffu) continue;
        const uint32_t b = key & 0x7fffu;
        s_list[s_start[b] + atomicAdd(&s_fill[b], 1u)] = (uint16_t)(i | (key & 0x8000u));
    }
    __syncthreads();
    for (uint32_t b = tid; b < nb; b += kBlock) {
        P10 acc = p10_identity();
        for (uint32_t k = s_start[b]; k < s_start[b + 1]; k++) {
            const uint16_t e16 = s_list[k];
            N10 e = n10_unpack(A.pts[i0 + (e16 & 0x7fffu)]);
            if (e16 & 0x8000u) e = n10_negate(e);
            acc = p10_madd(acc, e);
        }
        s_bkt[b] = acc;
    }
    __syncthreads();
    const F10 d2 = f10_const(fp_2D());
    // the sub-buckets of a digit: bucket v ends up at slot v * m
    for (uint32_t sft = m >> 1; sft >= 1; sft >>= 1) {
        if ((uint32_t)tid < nb && ((uint32_t)tid & (m - 1)) < sft) s_bkt[tid] = p10_add(s_bkt[tid], s_bkt[tid + sft], d2);
        __syncthreads();
    }
    // sum_v (v + 1) * B_v  (0-based v < V2, at stride m): suffix scan, then the sum of the suffix sums
    for (uint32_t off = 1; off < V2; off <<= 1) {
        const bool have = (uint32_t)tid + off < V2;
        P10 v = p10_identity();
        if (have) v = s_bkt[((uint32_t)tid + off) * m];
        __syncthreads();
        if (have) s_bkt[(uint32_t)tid * m] = p10_add(s_bkt[(uint32_t)tid * m], v, d2);
        __syncthreads();
    }
    for (uint32_t sft = V2 >> 1; sft >= 1; sft >>= 1) {
        if ((uint32_t)tid < sft) s_bkt[(uint32_t)tid * m] = p10_add(s_bkt[(uint32_t)tid * m], s_bkt[((uint32_t)tid + sft) * m], d2);
        __syncthreads();
    }
    if (tid == 0) A.out[(size_t)w * A.splits + blockIdx.y] = p10_pack(s_bkt[0]);
}
int dev_msm_var(DevCtx &c, const Niels *pts, const Fr *scalars, size_t n, Pt *out, size_t out_cap, int *n_windows, int *n_splits) {
    if (!n || n > ((size_t)1 << 24)) throw Error(OTTI_ERR_INTERNAL, "msm_var: bad size");
    VarMsmArgs A;
    const size_t splits = (n + kVarChunk - 1) / kVarChunk, chunk = (n + splits - 1) / splits;
    int lg = 0; while (((size_t)1 << lg) < chunk) lg++;
    A.c = std::max(5, std::min(9, lg - 2));                    // ~8 points per bucket: 2^(c-1) buckets for `chunk` points
    A.W = 253 / A.c + 1; A.splits = (int)splits; A.pts = pts; A.sc = scalars; A.n = (uint32_t)n; A.chunk = (uint32_t)chunk; A.out = out;
    if ((size_t)A.W * splits > out_cap) throw Error(OTTI_ERR_INTERNAL, "msm_var: result buffer too small");
    for (int i = 0; i < 9; i++) A.K[i] = 0;
    for (int w = 0; w < A.W; w++) { int bit = A.c - 1 + A.c * w; A.K[bit >> 5] |= 1u << (bit & 31); }
    KScope ks(c, KC_MSM_VAR);
    hipLaunchKernelGGL(k_msm_var, dim3((unsigned)A.W, (unsigned)splits), kBlock, 0, c.stream, A);
    if (n_windows) *n_windows = A.W;
    if (n_splits) *n_splits = (int)splits;
    return A.c;
}
// RFC 9496 4.3.1 Decode, one lane per point, straight into the affine Niels form the mixed addition reads (Z = 1 after Decode)
__global__ __launch_bounds__(64) void k_decode_niels(const uint8_t *in, size_t n, Niels *out, unsigned *bad) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t b[32];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(in + 32 * i);
    for (int k = 0; k < 8; k++) { const uint32_t x = src[k]; b[4 * k] = (uint8_t)x; b[4 * k + 1] = (uint8_t)(x >> 8); b[4 * k + 2] = (uint8_t)(x >> 16); b[4 * k + 3] = (uint8_t)(x >> 24); }
    Pt p;
    if (!pt_decode(p, b)) { atomicAdd(bad, 1u); out[i] = niels_identity(); return; }
    Niels nl; nl.yplusx = fp_add(p.Y, p.X); nl.yminusx = fp_sub(p.Y, p.X); nl.xy2d = fp_mul(p.T, fp_2D());
    out[i] = nl;
}
void dev_decode_niels(DevCtx &c, const uint8_t *compressed_dev, size_t n, Niels *out, unsigned *bad) {
    if (!n) return;
    KScope ks(c, KC_DECODE);
    hipLaunchKernelGGL(k_decode_niels, (unsigned)((n + 63) / 64), 64, 0, c.stream, compressed_dev, n, out, bad);
}

// ------------------------------------------------------------------------------------------------ the ALU roof of the MSM, measured
// Whole-chip throughput of the mixed point addition the bulk MSM is made of (p9_madd of fp9.h — the form the bulk kernel uses —, operands in registers, every CU busy): what
// bench.py prices k_msm_rows<0> against (roofline.alu.peak), measured in the run that reports it.  tools/mulbench.hip is the same loop.
__global__ __launch_bounds__(kBlock) void k_madd_peak(Fp *io, int iters) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    P9 p; p.X = f9_unpack(io[2 * i]); p.Y = f9_unpack(io[2 * i + 1]); p.Z = f9_one(); p.T = f9_mul(p.X, p.Y);
    N9 n; n.yplusx = p.X; n.yminusx = p.Y; n.xy2d = p.T;
    for (int k = 0; k < iters; k++) p = p9_madd(p, n);
    io[2 * i] = f9_pack(p.X);
}
double dev_madd_peak(DevCtx &c) {
    const int blocks = 1024, iters = 200; const size_t n = (size_t)blocks * kBlock;
    DevBuf<Fp> io(2 * n);
    std::vector<Fp> h(2 * n);
    for (size_t i = 0; i < 2 * n; i++) for (int q = 0; q < 8; q++) h[i].v[q] = (uint32_t)(0x9e3779b9u * (i * 8 + q + 1));
    OTTI_HIP(hipMemcpyAsync(io.p, h.data(), 2 * n * sizeof(Fp), hipMemcpyHostToDevice, c.stream));
    double best = 0;
    for (int rep = 0; rep < 4; rep++) {                                  // the first launch warms up; best of the rest
        OTTI_HIP(hipEventRecord(c.ev0, c.stream));
        hipLaunchKernelGGL(k_madd_peak, blocks, kBlock, 0, c.stream, io.p, iters);
        OTTI_HIP(hipEventRecord(c.ev1, c.stream)); OTTI_HIP(hipEventSynchronize(c.ev1));
        float ms = 0; OTTI_HIP(hipEventElapsedTime(&ms, c.ev0, c.ev1));
        if (rep && ms > 0) best = std::max(best, (double)n * iters / (ms * 1e-3));
    }
    return best;
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
// The reduction's last step in one launch: the fold from length 2 to 1, the folded a and b into the pinned result buffer (slots slot_a, slot_a + 1:
// the stream's next synchronisation or mailed result makes them visible) and rows[j] = d s[j], the scalars of delta's g_hat term — it used to be
// a step, two copies and a scaling launch in a row on the proof's sequential path.
__global__ __launch_bounds__(1024) void k_bullet_finish(Fr *a, Fr *b, Fr *s, size_t R, Fr u, Fr uinv, Fr d, Fr *rows, Fr *host_out) {
    if (threadIdx.x == 0) {
        const Fr af = fr_add(fr_mul(a[0], u), fr_mul(uinv, a[1])), bf = fr_add(fr_mul(b[0], uinv), fr_mul(u, b[1]));
        a[0] = af; b[0] = bf; host_out[0] = af; host_out[1] = bf;
    }
    for (size_t j = threadIdx.x; j < R; j += blockDim.x) { const Fr sj = fr_mul(s[j], ((j & 1) == 0) ? uinv : u); s[j] = sj; rows[j] = fr_mul(sj, d); }
}
void dev_bullet_finish(DevCtx &c, Fr *a, Fr *b, Fr *s, size_t R, const Fr &u, const Fr &u_inv, const Fr &d, Fr *rows, int slot_a) {
    KScope ks(c, KC_BULLET);
    hipLaunchKernelGGL(k_bullet_finish, 1, 1024, 0, c.stream, a, b, s, R, u, u_inv, d, rows, c.d_results_alias + slot_a);
}
void dev_bullet_step(DevCtx &c, Fr *a, Fr *b, Fr *s, size_t R, size_t n_cur, bool fold_first, const Fr &u, const Fr &u_inv, Fr *rows, Fr *extra_out) {
    KScope ks(c, KC_BULLET);
    hipLaunchKernelGGL(k_bullet_step, 1, 1024, 0, c.stream, a, b, s, R, n_cur, (int)fold_first, u, u_inv, rows, extra_out);
}

}  // namespace otti
