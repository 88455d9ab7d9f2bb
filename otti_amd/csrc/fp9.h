// gfx950 register-level arithmetic in GF(2^255-19) for the bulk fixed-base MSM: nine unsaturated limbs of 29 bits.
//
// fp10.h's ten 26/25-bit limbs cost 100 v_mad_u64_u32 per product plus the 19-folds of one operand; with nine 29-bit limbs the 9 x 9
// product is 81 multiply-adds, the high half (columns 9..16, i.e. the multiples of 2^261 = 1216 mod p) is carried into nine 29-bit
// limbs and comes back with nine more (1216 * h_k): 90 multiply-adds, no operand pre-scaling.  The price is headroom: three spare bits
// per limb instead of six, so the point formulas normalise ONE intermediate (F = D - C) where fp10.h needs none.
// Memory format stays the packed 32-byte Fp of field.h (window tables in HBM); F9 lives in registers only.
//
// Bounds (unsigned limbs).  "reduced": limbs 0..7 < 2^29 (limbs 0 and 1 may exceed by < 2^18), limb 8 < 2^23 + 2: the output of
// f9_mul / f9_unpack / f9_carry.  f9_mul(f, g) needs 9 * max f_i * max g_j < 2^63.9 (e.g. 2^30 x 3 * 2^29, or 2^30.6 x 2^29.9) and
// limb 8 of either operand below 2^26.  f9_sub(a, b) needs b reduced.
//
// Restates, for this path, curve25519-dalek's FieldElement arithmetic under libspartan's `group.rs` [RECALL; Cargo dependency of the
// empty submodule /root/reference/Spartan, .gitmodules:4-6]; checked against the CPU oracle (fp.c) by the parity tests.
#pragma once
#include "field.h"
#include "point.h"

namespace otti {

struct F9 { uint32_t v[9]; };
struct P9 { F9 X, Y, Z, T; };
struct N9 { F9 yplusx, yminusx, xy2d; };

#define F9_M 0x1fffffffu
#define F9_TOP 0x007fffffu                    // limb 8 of a value below 2^255

HD uint32_t f9_alignbit(uint32_t hi, uint32_t lo, int s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}
HD F9 f9_zero() { F9 r; for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }
HD F9 f9_one() { F9 r = f9_zero(); r.v[0] = 1; return r; }
// any loosely reduced Fp (< 2^256): bit 255 is folded into limb 0 (2^255 = 19)
HD F9 f9_unpack(const Fp &a) {
    const uint32_t *w = a.v; F9 r;
    r.v[0] = (w[0] & F9_M) + 19u * (w[7] >> 31);
    r.v[1] = f9_alignbit(w[1], w[0], 29) & F9_M;
    r.v[2] = f9_alignbit(w[2], w[1], 26) & F9_M;
    r.v[3] = f9_alignbit(w[3], w[2], 23) & F9_M;
    r.v[4] = f9_alignbit(w[4], w[3], 20) & F9_M;
    r.v[5] = f9_alignbit(w[5], w[4], 17) & F9_M;
    r.v[6] = f9_alignbit(w[6], w[5], 14) & F9_M;
    r.v[7] = f9_alignbit(w[7], w[6], 11) & F9_M;
    r.v[8] = (w[7] >> 8) & F9_TOP;
    return r;
}
// one carry sweep 0 -> 8 -> 0 -> 1 (limbs < 2^32 in): reduced out
HD F9 f9_carry(const F9 &a) {
    F9 r = a; uint32_t c;
#pragma unroll
    for (int i = 0; i < 8; i++) { c = r.v[i] >> 29; r.v[i] &= F9_M; r.v[i + 1] += c; }
    c = r.v[8] >> 23; r.v[8] &= F9_TOP; r.v[0] += 19u * c;
    c = r.v[0] >> 29; r.v[0] &= F9_M; r.v[1] += c;
    return r;
}
// exact packing: a valid loosely reduced Fp (value < 2^255 + tiny)
HD Fp f9_pack(const F9 &a0) {
    F9 a = f9_carry(f9_carry(a0));             // every limb within its width, except limb 1 by at most one unit
    uint32_t c = a.v[1] >> 29; a.v[1] &= F9_M; a.v[2] += c;       // limb 2 <= 2^29 - 1 + 1: may reach 2^29 exactly only if limbs above absorb it
#pragma unroll
    for (int i = 2; i < 8; i++) { c = a.v[i] >> 29; a.v[i] &= F9_M; a.v[i + 1] += c; }
    const uint32_t *t = a.v; Fp r;             // limb 8 <= 2^23: the value is below 2^256
    r.v[0] = t[0] | (t[1] << 29);
    r.v[1] = (t[1] >> 3) | (t[2] << 26);
    r.v[2] = (t[2] >> 6) | (t[3] << 23);
    r.v[3] = (t[3] >> 9) | (t[4] << 20);
    r.v[4] = (t[4] >> 12) | (t[5] << 17);
    r.v[5] = (t[5] >> 15) | (t[6] << 14);
    r.v[6] = (t[6] >> 18) | (t[7] << 11);
    r.v[7] = (t[7] >> 21) | (t[8] << 8);
    return r;
}
HD F9 f9_add(const F9 &a, const F9 &b) { F9 r; for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
// a - b + 2p (b reduced): limbs of 2p are twice those of p
HD F9 f9_sub(const F9 &a, const F9 &b) {
    F9 r;
    r.v[0] = a.v[0] + 0x3fffffdau - b.v[0];    // 2 (2^29 - 19)
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = a.v[i] + 0x3ffffffeu - b.v[i];   // 2 (2^29 - 1)
    r.v[8] = a.v[8] + 0x00fffffeu - b.v[8];    // 2 (2^23 - 1)
    return r;
}
HD F9 f9_neg(const F9 &b) { return f9_sub(f9_zero(), b); }

HD F9 f9_mul(const F9 &f, const F9 &g) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OTTI_LIMB9_PLAIN_C)
    F9 r;
#include "f9_mul_gfx950.inc"
    return r;
#else
    uint32_t h[9]; uint64_t acc = 0;
#pragma unroll
    for (int k = 9; k < 17; k++) {             // the multiples of 2^261, carried into 29-bit limbs
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j < 9) acc += (uint64_t)f.v[i] * g.v[j]; }
        h[k - 9] = (uint32_t)acc & F9_M; acc >>= 29;
    }
    h[8] = (uint32_t)acc;                      // < 2^26 * 2^26 / 2^29 + carries
    F9 r; acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j < 9) acc += (uint64_t)f.v[i] * g.v[j]; }
        acc += (uint64_t)h[k] * 1216u;         // 2^261 = 19 * 2^6 (mod p)
        if (k < 8) { r.v[k] = (uint32_t)acc & F9_M; acc >>= 29; }
    }
    r.v[8] = (uint32_t)acc & F9_TOP;
    uint64_t t = (acc >> 23) * 19u + r.v[0];   // 2^255 = 19; acc >> 23 < 2^41
    r.v[0] = (uint32_t)t & F9_M; t >>= 29;
    r.v[1] += (uint32_t)t;                     // < 2^17
    return r;
#endif
}
HD F9 f9_const(const Fp &c) { return f9_unpack(c); }

HD P9 p9_identity() { P9 p; p.X = f9_zero(); p.Y = f9_one(); p.Z = f9_one(); p.T = f9_zero(); return p; }
HD P9 p9_unpack(const Pt &p) { P9 r; r.X = f9_unpack(p.X); r.Y = f9_unpack(p.Y); r.Z = f9_unpack(p.Z); r.T = f9_unpack(p.T); return r; }
HD Pt p9_pack(const P9 &p) { Pt r; r.X = f9_pack(p.X); r.Y = f9_pack(p.Y); r.Z = f9_pack(p.Z); r.T = f9_pack(p.T); return r; }
HD N9 n9_unpack(const Niels &n) { N9 r; r.yplusx = f9_unpack(n.yplusx); r.yminusx = f9_unpack(n.yminusx); r.xy2d = f9_unpack(n.xy2d); return r; }
// -q: swap the sums, negate xy2d (limbs < 2^30: fine as the second operand of T * xy2d)
HD N9 n9_negate(const N9 &q) { N9 r; r.yplusx = q.yminusx; r.yminusx = q.yplusx; r.xy2d = f9_neg(q.xy2d); return r; }

// mixed addition, 7M (add-2008-hwcd-3 with a = -1, as p10_madd / pt_madd: the same group element).  p reduced, q from n9_unpack /
// n9_negate.  Output reduced.
//   limbs: Y - X + 2p < 3 * 2^29, Y + X < 2^30, D = 2Z < 2^30, E = B - A + 2p < 3 * 2^29, H < 2^30, G = D + C < 3 * 2^29,
//   F = D - C + 2p < 2^31 is carried down (the one normalisation); products: E F, G H, F G, E H all within 9 * 3 * 2^29 * 2^30 < 2^63.8
HD P9 p9_madd(const P9 &p, const N9 &q) {
    const F9 a = f9_mul(f9_sub(p.Y, p.X), q.yminusx);
    const F9 b = f9_mul(f9_add(p.Y, p.X), q.yplusx);
    const F9 c = f9_mul(p.T, q.xy2d);
    const F9 d = f9_add(p.Z, p.Z);
    const F9 e = f9_sub(b, a), f = f9_carry(f9_sub(d, c)), g = f9_add(d, c), h = f9_add(b, a);
    P9 r; r.X = f9_mul(e, f); r.Y = f9_mul(g, h); r.T = f9_mul(e, h); r.Z = f9_mul(f, g); return r;
}

}  // namespace otti
