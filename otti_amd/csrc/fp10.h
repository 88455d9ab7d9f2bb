// gfx950 register-level arithmetic in GF(2^255-19) for the point kernels: ten unsaturated limbs of 26/25 bits.
//
// Why not the 8 x u32 saturated form of field.h on the device hot path: a saturated 8x8 product needs a carry-out per partial product
// (v_mad_u64_u32 + v_addc_co_u32) and a rotation of the column accumulator (moves); measured ~300 issued instructions per multiply
// with one long dependency chain.  With 26/25-bit limbs every column sum of 10 partial products (including the 19x fold of the high
// half and the 2x of odd*odd terms, both premultiplied on 32-bit operands) fits a 64-bit accumulator, so a multiply is 100 independent
// v_mad_u64_u32 into ten accumulators plus one carry sweep: ~150 instructions, ten-way instruction-level parallelism.
// Memory format stays the 32-byte packed Fp of field.h (HBM tables, LDS-free interchange); F10 lives in registers and LDS only.
//
// Bounds (unsigned limbs): "reduced" = even limbs <= 2^26 + 2^17, odd limbs <= 2^25 + 2^17 (output of mul/sqr/carry/unpack*).
// f10_mul(f, g) requires f limbs < 2^28.1 and g limbs < 2^27.7 (so that 19*g < 2^32 and ten products stay below 2^64).
// f10_sub(a, b) requires b reduced.  (*) unpack folds bit 255 of a loosely reduced Fp into limb 0.
#pragma once
#include "field.h"
#include "point.h"

namespace otti {

struct F10 { uint32_t v[10]; };
struct P10 { F10 X, Y, Z, T; };          // extended coordinates
struct N10 { F10 yplusx, yminusx, xy2d; };

#define F10_M26 0x3ffffffu
#define F10_M25 0x1ffffffu

__device__ __forceinline__ F10 f10_zero() { F10 r; for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }
__device__ __forceinline__ F10 f10_one() { F10 r = f10_zero(); r.v[0] = 1; return r; }

__device__ __forceinline__ F10 f10_unpack(const Fp &a) {
    const uint32_t *w = a.v; F10 r;
    r.v[0] = w[0] & F10_M26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & F10_M25;
    r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & F10_M26;
    r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & F10_M25;
    r.v[4] = (w[3] >> 6) & F10_M26;
    r.v[5] = w[4] & F10_M25;
    r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & F10_M26;
    r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & F10_M25;
    r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & F10_M26;
    r.v[9] = (w[7] >> 6) & F10_M25;
    r.v[0] += 19u * (w[7] >> 31);                         // bit 255 of a loosely reduced value: 2^255 = 19
    return r;
}
// one carry sweep 0 -> 9 -> 0 -> 1: limbs back to (26,25,...) widths, limb 0/1 may keep a few extra units
__device__ __forceinline__ F10 f10_carry(const F10 &a) {
    F10 r = a; uint32_t c;
#pragma unroll
    for (int i = 0; i < 9; i++) { const int bits = (i & 1) ? 25 : 26; c = r.v[i] >> bits; r.v[i] &= (1u << bits) - 1u; r.v[i + 1] += c; }
    c = r.v[9] >> 25; r.v[9] &= F10_M25; r.v[0] += 19u * c;
    c = r.v[0] >> 26; r.v[0] &= F10_M26; r.v[1] += c;
    return r;
}
// exact packing: value < 2^255 + tiny, as 8 x u32 (a valid loosely reduced Fp)
__device__ __forceinline__ Fp f10_pack(const F10 &a0) {
    F10 a = f10_carry(f10_carry(a0));                     // every limb within its width, except limb 1 by at most one unit
    const int shift[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
    uint32_t w[10];
#pragma unroll
    for (int i = 0; i < 10; i++) w[i] = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) {                        // add limb i at its bit position into a little-endian word array
        const int word = shift[i] >> 5, off = shift[i] & 31;
        uint64_t v = (uint64_t)a.v[i] << off;
        uint64_t c = (uint64_t)w[word] + (uint32_t)v; w[word] = (uint32_t)c; c >>= 32;
        c += (uint64_t)w[word + 1] + (uint32_t)(v >> 32); w[word + 1] = (uint32_t)c;
        w[word + 2] += (uint32_t)(c >> 32);
    }
    Fp r; uint64_t c = (uint64_t)w[8] * 38u;              // nothing beyond 2^256 for in-range limbs; folded anyway (2^256 = 38)
#pragma unroll
    for (int i = 0; i < 8; i++) { c += w[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    r.v[0] += (uint32_t)c * 38u;
    return r;
}

__device__ __forceinline__ F10 f10_add(const F10 &a, const F10 &b) { F10 r; for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
// a - b + 2p (b reduced)
__device__ __forceinline__ F10 f10_sub(const F10 &a, const F10 &b) {
    F10 r;
    r.v[0] = a.v[0] + 0x7ffffdau - b.v[0];                // 2 * (2^26 - 19)
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - b.v[i];   // 2 * (2^25 - 1), 2 * (2^26 - 1)
    return r;
}
__device__ __forceinline__ F10 f10_neg(const F10 &b) { return f10_sub(f10_zero(), b); }

// column sums h[0..9] (64-bit) -> reduced limbs
__device__ __forceinline__ F10 f10_reduce_columns(uint64_t h[10]) {
    F10 r; uint64_t c;
#pragma unroll
    for (int i = 0; i < 9; i++) { const int bits = (i & 1) ? 25 : 26; c = h[i] >> bits; r.v[i] = (uint32_t)h[i] & ((1u << bits) - 1u); h[i + 1] += c; }
    c = h[9] >> 25; r.v[9] = (uint32_t)h[9] & F10_M25;
    uint64_t t = (uint64_t)r.v[0] + 19u * c;              // c < 2^39: 19c < 2^44
    r.v[0] = (uint32_t)t & F10_M26; t >>= 26;
    t += r.v[1]; r.v[1] = (uint32_t)t & F10_M25; t >>= 25;
    r.v[2] += (uint32_t)t;
    return r;
}
__device__ __forceinline__ F10 f10_mul(const F10 &f, const F10 &g) {
    uint32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { g19[i] = 19u * g.v[i]; f2[i] = (i & 1) ? 2u * f.v[i] : f.v[i]; }   // (19 x as two v_lshl_add_u32 in an asm block was measured: 11 % slower than v_mul_lo_u32)
    uint64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;               // i + j == k (mod 10)
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            acc += (uint64_t)(both_odd ? f2[i] : f.v[i]) * (wrap ? g19[j] : g.v[j]);
        }
        h[k] = acc;
    }
    return f10_reduce_columns(h);
}
__device__ __forceinline__ F10 f10_sqr(const F10 &f) {
    // symmetric terms doubled on the 32-bit operand; f limbs < 2^27 here (callers pass reduced or add/sub-of-reduced values)
    uint32_t f19[10], d[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { f19[i] = 19u * f.v[i]; d[i] = 2u * f.v[i]; }
    uint64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            const int j = (k - i + 10) % 10;
            if (i > j) continue;
            const bool wrap = (i + j) >= 10;
            const bool both_odd = (i & 1) && (j & 1);
            // coefficient: (i<j ? 2 : 1) * (both_odd ? 2 : 1) * (wrap ? 19 : 1), with one factor 2 on d[] and the 19 on f19[]
            uint32_t x = (i < j) ? d[i] : f.v[i];
            uint32_t y = wrap ? f19[j] : f.v[j];
            uint64_t p = (uint64_t)x * y;
            acc += both_odd ? 2 * p : p;
        }
        h[k] = acc;
    }
    return f10_reduce_columns(h);
}
__device__ __forceinline__ F10 f10_sqr_n(F10 a, int n) { for (int i = 0; i < n; i++) a = f10_sqr(a); return a; }
__device__ __forceinline__ F10 f10_const(const Fp &c) { return f10_unpack(c); }

// ------------------------------------------------------------------------------------------------ points
__device__ __forceinline__ P10 p10_identity() { P10 p; p.X = f10_zero(); p.Y = f10_one(); p.Z = f10_one(); p.T = f10_zero(); return p; }
__device__ __forceinline__ P10 p10_unpack(const Pt &p) { P10 r; r.X = f10_unpack(p.X); r.Y = f10_unpack(p.Y); r.Z = f10_unpack(p.Z); r.T = f10_unpack(p.T); return r; }
__device__ __forceinline__ Pt p10_pack(const P10 &p) { Pt r; r.X = f10_pack(p.X); r.Y = f10_pack(p.Y); r.Z = f10_pack(p.Z); r.T = f10_pack(p.T); return r; }
__device__ __forceinline__ N10 n10_unpack(const Niels &n) { N10 r; r.yplusx = f10_unpack(n.yplusx); r.yminusx = f10_unpack(n.yminusx); r.xy2d = f10_unpack(n.xy2d); return r; }

// mixed addition, 7M.  p coordinates reduced; q (table entry) reduced.  Output reduced.
__device__ __forceinline__ P10 p10_madd(const P10 &p, const N10 &q) {
    F10 a = f10_mul(f10_sub(p.Y, p.X), q.yminusx);
    F10 b = f10_mul(f10_add(p.Y, p.X), q.yplusx);
    F10 c = f10_mul(p.T, q.xy2d);
    F10 d = f10_add(p.Z, p.Z);
    F10 e = f10_sub(b, a), f = f10_sub(d, c), g = f10_add(d, c), h = f10_add(b, a);     // e,h < 2^27.6; f < 2^28; g < 2^27.6
    // second operands (the ones f10_mul folds by 19) are e and g only: two 19-folds per addition instead of three
    P10 r; r.X = f10_mul(f, e); r.Y = f10_mul(h, g); r.T = f10_mul(h, e); r.Z = f10_mul(f, g); return r;
}
// full addition, 9M
__device__ __forceinline__ P10 p10_add(const P10 &p, const P10 &q, const F10 &d2) {
    F10 a = f10_mul(f10_sub(p.Y, p.X), f10_sub(q.Y, q.X));
    F10 b = f10_mul(f10_add(p.Y, p.X), f10_add(q.Y, q.X));
    F10 c = f10_mul(f10_mul(p.T, q.T), d2);
    F10 d = f10_mul(p.Z, q.Z); d = f10_add(d, d);
    F10 e = f10_sub(b, a), f = f10_sub(d, c), g = f10_add(d, c), h = f10_add(b, a);
    P10 r; r.X = f10_mul(f, e); r.Y = f10_mul(h, g); r.T = f10_mul(h, e); r.Z = f10_mul(f, g); return r;
}
__device__ __forceinline__ N10 n10_negate(const N10 &q) { N10 r; r.yplusx = q.yminusx; r.yminusx = q.yplusx; r.xy2d = f10_carry(f10_neg(q.xy2d)); return r; }

// ------------------------------------------------------------------------------------------------ quad-parallel point arithmetic
// The latency-bound launches (one/two-row MSMs of the evaluation proof, blinding commitments) have far fewer point additions in
// flight than the chip has lanes, and a dependent chain of full additions on one lane costs ~3 us per tree level (nine f10_mul,
// issue-bound at ~0.28 us each for a lone wave).  Here the FOUR lanes of a quad (lane & 3) hold X, Y, Z, T of ONE point and share
// every addition: each of the formula's two rounds of four independent multiplications becomes ONE f10_mul wave instruction stream
// (lane k computes product k), with the operands exchanged by DPP quad permutes.  A unified addition is then 2 multiplication
// depths (mixed: the table's Niels entry is already in "cached" form) or 3 (the cached form of the right operand first), instead
// of 7 / 9.  Same formulas as p10_madd / p10_add (add-2008-hwcd-3, a = -1), so the same group element comes out.
//   u(P)  = (Y - X, Y + X, T, Z)                on lanes (0, 1, 2, 3)
//   v(P)  = (Y - X, Y + X, 2d T, 2 Z)           the cached form; a Niels entry (yminusx, yplusx, xy2d) with 2 on lane 3 is v of an affine point
//   P + Q = stage2(u(P) * v(Q))                 lane-wise product = (A, B, C, D);  E = B - A, H = B + A, F = D - C, G = D + C;
//                                               X = E F, Y = G H, Z = F G, T = E H
#define F10_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
template <int kCtrl> __device__ __forceinline__ F10 f10_quad_perm(const F10 &a) {
    F10 r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.v[i], kCtrl, 0xf, 0xf, true);
    return r;
}
__device__ __forceinline__ F10 f10_select(bool c, const F10 &a, const F10 &b) { F10 r; for (int i = 0; i < 10; i++) r.v[i] = c ? a.v[i] : b.v[i]; return r; }
// q = lane & 3 throughout.  Identity: (0, 1, 1, 0).
__device__ __forceinline__ F10 q10_identity(int q) { F10 r = f10_zero(); r.v[0] = (q == 1 || q == 2) ? 1u : 0u; return r; }
__device__ __forceinline__ F10 q10_u(const F10 &own, int q) {
    const F10 s = f10_quad_perm<F10_QP(1, 0, 3, 2)>(own);                        // lanes 0/1 swap X and Y, lanes 2/3 swap Z and T
    const F10 sum = f10_add(own, s), diff = f10_sub(s, own);                     // lane 0: Y - X (own = X is reduced); lane 1: Y + X
    return q < 2 ? f10_select(q & 1, sum, diff) : s;
}
__device__ __forceinline__ F10 q10_cached(const F10 &u, int q, const F10 &d2) {  // u(P) -> v(P): lane 2 times 2d, lane 3 doubled
    const F10 w = f10_mul(u, d2), dbl = f10_add(u, u);
    return q == 2 ? w : (q == 3 ? dbl : u);
}
// acc + (the point whose cached form is v), given ua = q10_u(acc) (so that a caller can form it before it waits for v);
// acc coordinates reduced, v limbs < 2^27.7.  Output reduced.
__device__ __forceinline__ F10 q10_add_cached_u(const F10 &ua, const F10 &v, int q) {
    const F10 prod = f10_mul(ua, v);                                             // lanes: A, B, C, D
    const F10 s = f10_quad_perm<F10_QP(1, 0, 3, 2)>(prod);
    const F10 r = f10_select(q & 1, f10_add(prod, s), f10_sub(s, prod));         // lanes: E = B - A, H = B + A, F = D - C, G = D + C
    const F10 fa = f10_quad_perm<F10_QP(2, 3, 3, 0)>(r);                         // lanes get: F, G, G, E
    const F10 fb = f10_quad_perm<F10_QP(0, 1, 2, 1)>(r);                         // lanes get: E, H, F, H
    // X = F E, Y = G H, Z = F G, T = H E   (first operand may reach 2^28, second stays below 2^27.7)
    return f10_mul(q < 2 ? fa : fb, q < 2 ? r : fa);
}
__device__ __forceinline__ F10 q10_add_cached(const F10 &acc, const F10 &v, int q) { return q10_add_cached_u(q10_u(acc, q), v, q); }
// the lane's share of a table entry in cached form, negated when the digit is negative: lane 0 yminusx, lane 1 yplusx (swapped for
// a negative digit), lane 2 xy2d (negated), lane 3 the constant 2
__device__ __forceinline__ F10 q10_load_niels(const Niels *e, bool neg, int q) {
    if (q == 3) { F10 two = f10_zero(); two.v[0] = 2; return two; }
    const Fp *comp = reinterpret_cast<const Fp *>(e) + (q == 2 ? 2 : (((q == 0) != neg) ? 1 : 0));   // Niels = {yplusx, yminusx, xy2d}
    F10 v = f10_unpack(*comp);
    if (q == 2 && neg) v = f10_carry(f10_neg(v));
    return v;
}

// a^(2^252-3), the (p-5)/8 power of RFC 9496's SQRT_RATIO_M1
__device__ __forceinline__ F10 f10_pow22523(const F10 &a) {
    F10 z2 = f10_sqr(a), z9 = f10_mul(f10_sqr_n(z2, 2), a), z11 = f10_mul(z9, z2);
    F10 t5 = f10_mul(f10_sqr(z11), z9);
    F10 t10 = f10_mul(f10_sqr_n(t5, 5), t5), t20 = f10_mul(f10_sqr_n(t10, 10), t10), t40 = f10_mul(f10_sqr_n(t20, 20), t20);
    F10 t50 = f10_mul(f10_sqr_n(t40, 10), t10), t100 = f10_mul(f10_sqr_n(t50, 50), t50), t200 = f10_mul(f10_sqr_n(t100, 100), t100);
    F10 t250 = f10_mul(f10_sqr_n(t200, 50), t50);
    return f10_mul(f10_sqr_n(t250, 2), a);
}
// RFC 9496 4.3.2 Encode on 10-limb coordinates; the dependent power chain runs on F10, the few sign/equality tests on packed Fp
__device__ __forceinline__ void p10_encode(uint8_t out[32], const P10 &p) {
    const F10 sqrt_m1 = f10_const(fp_SQRT_M1()), invsqrt_a_minus_d = f10_const(fp_INVSQRT_A_MINUS_D());
    F10 u1 = f10_mul(f10_add(p.Z, p.Y), f10_sub(p.Z, p.Y));
    F10 u2 = f10_mul(p.X, p.Y);
    // SQRT_RATIO_M1(1, v) with v = u1 * u2^2
    F10 v = f10_mul(u1, f10_sqr(u2));
    F10 v3 = f10_mul(f10_sqr(v), v), v7 = f10_mul(f10_sqr(v3), v);
    F10 r = f10_mul(v3, f10_pow22523(v7));
    F10 check = f10_mul(v, f10_sqr(r));
    Fp chk = f10_pack(check), one = fp_one();
    Fp neg_one = fp_neg(one), neg_i = fp_mul(neg_one, fp_SQRT_M1());
    bool flipped = fp_eq(chk, neg_one), flipped_i = fp_eq(chk, neg_i);
    if (flipped || flipped_i) r = f10_mul(sqrt_m1, r);
    Fp rp = f10_pack(r);
    F10 inv = f10_unpack(fp_abs(rp));
    F10 den1 = f10_mul(inv, u1), den2 = f10_mul(inv, u2);
    F10 zinv = f10_mul(f10_mul(den1, den2), p.T);
    F10 ix = f10_mul(p.X, sqrt_m1), iy = f10_mul(p.Y, sqrt_m1);
    F10 ench = f10_mul(den1, invsqrt_a_minus_d);
    bool rotate = fp_is_negative(f10_pack(f10_mul(p.T, zinv)));
    F10 x = rotate ? iy : f10_carry(p.X), y = rotate ? ix : f10_carry(p.Y), deninv = rotate ? ench : den2;
    if (fp_is_negative(f10_pack(f10_mul(x, zinv)))) y = f10_carry(f10_neg(y));
    Fp s = fp_abs(f10_pack(f10_mul(deninv, f10_sub(p.Z, y))));
    fp_to_bytes(out, s);
}

}  // namespace otti
