// gfx950 register-level arithmetic in GF(l) for the multiplier-bound streaming kernels: nine unsaturated limbs of 29 bits.
//
// Why: the 8 x u32 saturated Montgomery product of field.h pays a carry add (v_addc_co_u32) for every one of its 104 multiply-adds plus the
// rotation of a 96-bit column accumulator: ~300 issued instructions.  With 29-bit limbs every column of the 9 x 9 product AND of the
// interleaved Montgomery reduction (l has limbs 5..7 == 0 in this radix: 6 products per reduction digit) fits a 64-bit accumulator with
// room to spare, so a product is 81 + 54 v_mad_u64_u32, nine (v_mul_lo, v_and) quotient digits, one 64-bit shift per column and a mask
// per output limb: ~180 instructions, no carry flags, no asm.  Packing / unpacking costs ~16 instructions each way, so the form only pays
// inside a kernel that keeps its operands unpacked from load to store (sum-check folds and sums, eq tables, sparse products).
//
// Memory format stays the 32-byte Montgomery form of field.h (x * 2^256 mod l, canonical).  The Montgomery radix HERE is 2^261:
//     fr9_mul(a, b) = a * b * 2^-261 (mod l)
// so of the two operands of a product exactly ONE must carry an extra factor 2^5 for the result to be in the memory format again; the
// shifted unpack fr9_unpack5 (limbs of x << 5: the same instructions with other shift counts) provides it for loaded values, and
// fr9_shl5 for computed ones.
//
// Bounds.  "normalised": limbs 0..7 < 2^29, limb 8 whatever the value needs (< 2^29 for every value below 2^261).  "loose": any limbs
// for which the column sums of the product stay below 2^64: 9 * max(a_i) * max(b_j) + 6 * 2^58 < 2^64, e.g. one operand normalised and the
// other with limbs < 2^31.9, or both < 2^30.3.  Value bound of a product: a * b / 2^261 + l; every caller states what it relies on.
//
// Restates, for this path, upstream libspartan `src/scalar/ristretto255.rs::Scalar::{mul, add, sub}` (montgomery_reduce is its
// reduction; /root/reference/Spartan is an empty submodule: .gitmodules:4-6); checked against the CPU oracle (fr.c) by the parity tests.
#pragma once
#include "field.h"

namespace otti {

struct Fr9 { uint32_t v[9]; };

#define FR9_M 0x1fffffffu
#define FR9_L0 0x1cf5d3edu
#define FR9_L1 0x009318d2u
#define FR9_L2 0x1de73596u
#define FR9_L3 0x1df3bd45u
#define FR9_L4 0x0000014du
#define FR9_L8 0x00100000u
#define FR9_LINV 0x12547e1bu                  // -l^{-1} mod 2^29 (the low 29 bits of the 32-bit constant happen to be the constant itself)

// (hi:lo) >> s, low word
HD uint32_t fr9_alignbit(uint32_t hi, uint32_t lo, int s) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
#endif
}
// r = w - l if w >= l (w < 2l)
HD Fr fr9_cond_sub_l(const uint32_t w[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
    Fr r; fr_dev_cond_sub_l(r, w); return r;
#else
    return fr_cond_sub_l(w, 0);
#endif
}

HD Fr9 fr9_zero() { Fr9 r; for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }

// limbs of x << S for a 256-bit word x: limb i = bits [29 i - S, 29 i - S + 29) of x, limb 8 everything from bit 232 - S up (so the
// value must stay below 2^264: any x for S <= 5 — limb 8 < 2^29 —, a canonical x < 2^253 for S = 10 — limb 8 < 2^31).  S = 0: the plain
// unpack; S = 5, 10: the same instructions with other shift counts, for the operand that carries the radix correction(s) of a product.
template <int S> HD Fr9 fr9_unpack_s(const Fr &a) {
    static_assert(S >= 0 && S <= 10, "shifted unpack");
    const uint32_t *w = a.v; Fr9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int lo = 29 * i - S;                                   // first bit of x in this limb
        uint32_t v;
        if (lo < 0) v = w[0] << (-lo);
        else {
            const int word = lo >> 5, off = lo & 31;
            if (off == 0) v = w[word];
            else if (word == 7 || off + 29 <= 32) v = w[word] >> off;
            else v = fr9_alignbit(w[word + 1], w[word], off);
        }
        r.v[i] = i < 8 ? (v & FR9_M) : v;
    }
    return r;
}
HD Fr9 fr9_unpack(const Fr &a) { return fr9_unpack_s<0>(a); }
HD Fr9 fr9_unpack5(const Fr &a) { return fr9_unpack_s<5>(a); }
// normalised limbs, value < 2^256  ->  the 256-bit word
HD void fr9_pack_words(uint32_t w[8], const Fr9 &a) {
    const uint32_t *t = a.v;
    w[0] = t[0] | (t[1] << 29);
    w[1] = (t[1] >> 3) | (t[2] << 26);
    w[2] = (t[2] >> 6) | (t[3] << 23);
    w[3] = (t[3] >> 9) | (t[4] << 20);
    w[4] = (t[4] >> 12) | (t[5] << 17);
    w[5] = (t[5] >> 15) | (t[6] << 14);
    w[6] = (t[6] >> 18) | (t[7] << 11);
    w[7] = (t[7] >> 21) | (t[8] << 8);
}
// one carry sweep: limbs 0..7 below 2^29 again (limbs < 2^32 in; the value is unchanged)
HD Fr9 fr9_norm(const Fr9 &a) {
    Fr9 r = a;
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint32_t c = r.v[i] >> 29; r.v[i] &= FR9_M; r.v[i + 1] += c; }
    return r;
}
// normalised value < 2l  ->  canonical memory word
HD Fr fr9_pack_lt2l(const Fr9 &a) {
    uint32_t w[8]; fr9_pack_words(w, a);
    return fr9_cond_sub_l(w);
}
// normalised value < 3l  ->  canonical memory word
HD Fr fr9_pack_lt3l(const Fr9 &a) {
    uint32_t w[8]; fr9_pack_words(w, a);
    const Fr t = fr9_cond_sub_l(w);
    return fr9_cond_sub_l(t.v);
}

HD Fr9 fr9_add(const Fr9 &a, const Fr9 &b) { Fr9 r; for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
// limb i of K * l in "borrow-proof" form: K l = sum o_i 2^(29 i) with o_0 = t_0 + 2^29, o_i = t_i + 2^29 - 1 (0 < i < 8), o_8 = t_8 - 1
// (t the ordinary limbs of K l): every o_i >= 2^29 - 1, so a_i + o_i - b_i never goes negative for a normalised b below K l - 2^232.
constexpr uint32_t fr9_kl_limb(uint32_t K, int i) {
    const uint32_t L[9] = {FR9_L0, FR9_L1, FR9_L2, FR9_L3, FR9_L4, 0, 0, 0, FR9_L8};
    uint64_t c = 0; uint32_t t = 0;
    for (int j = 0; j <= i; j++) { c += (uint64_t)K * L[j]; t = (uint32_t)(j < 8 ? (c & FR9_M) : c); c >>= 29; }
    return i == 0 ? t + 0x20000000u : (i < 8 ? t + 0x1fffffffu : t - 1u);
}
// a - b + K l with every limb non-negative: b NORMALISED with value < K l - 2^232 (K a power of two <= 128).  Result limbs < a_i + 2^30.
template <uint32_t K> HD Fr9 fr9_sub_kl(const Fr9 &a, const Fr9 &b) {
    Fr9 r;
    r.v[0] = a.v[0] + fr9_kl_limb(K, 0) - b.v[0]; r.v[1] = a.v[1] + fr9_kl_limb(K, 1) - b.v[1]; r.v[2] = a.v[2] + fr9_kl_limb(K, 2) - b.v[2];
    r.v[3] = a.v[3] + fr9_kl_limb(K, 3) - b.v[3]; r.v[4] = a.v[4] + fr9_kl_limb(K, 4) - b.v[4]; r.v[5] = a.v[5] + fr9_kl_limb(K, 5) - b.v[5];
    r.v[6] = a.v[6] + fr9_kl_limb(K, 6) - b.v[6]; r.v[7] = a.v[7] + fr9_kl_limb(K, 7) - b.v[7]; r.v[8] = a.v[8] + fr9_kl_limb(K, 8) - b.v[8];
    return r;
}
HD Fr9 fr9_sub2l(const Fr9 &a, const Fr9 &b) { return fr9_sub_kl<2>(a, b); }
// any normalised value below 2^261 (limb 8 < 2^29)  ->  canonical memory word:  v = low252 + q 2^252 = low252 - q delta (mod l), q < 2^9
HD Fr fr9_canon(const Fr9 &a) {
    const uint32_t q = a.v[8] >> 20;
    const uint32_t L[5] = {FR9_L0, FR9_L1, FR9_L2, FR9_L3, FR9_L4};
    uint32_t d[6]; uint64_t acc = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) { acc += (uint64_t)q * L[j]; d[j] = (uint32_t)acc & FR9_M; acc >>= 29; }
    d[5] = (uint32_t)acc;
    Fr9 r;                                                           // low252 + l - q delta, limb by limb, in (0, 2l)
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = a.v[i] + fr9_kl_limb(1, i) - (i < 6 ? d[i] : 0u);
    r.v[8] = (a.v[8] & 0xfffffu) + fr9_kl_limb(1, 8);
    return fr9_pack_lt2l(fr9_norm(r));
}
// limbs of 32 x, normalised (x normalised, x < 2^256)
HD Fr9 fr9_shl5(const Fr9 &a) {
    Fr9 r;
    r.v[0] = (a.v[0] << 5) & FR9_M;
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = ((a.v[i] << 5) & FR9_M) | (a.v[i - 1] >> 24);
    r.v[8] = (a.v[8] << 5) | (a.v[7] >> 24);
    return r;
}

// a * b * 2^-261 mod l.  Output normalised, value < a * b / 2^261 + l.  Requires every column sum below 2^64 (header).
HD Fr9 fr9_mul(const Fr9 &a, const Fr9 &b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(OTTI_LIMB9_PLAIN_C)
    Fr9 r;
#include "fr9_mul_gfx950.inc"
    return r;
#else
    const uint32_t L[9] = {FR9_L0, FR9_L1, FR9_L2, FR9_L3, FR9_L4, 0, 0, 0, FR9_L8};
    uint32_t m[9]; Fr9 r; uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j < 9) acc += (uint64_t)a.v[i] * b.v[j]; }
#pragma unroll
        for (int i = 0; i < 9; i++) { const int j = k - i; if (i < k && j >= 1 && j < 9 && L[j] != 0) acc += (uint64_t)m[i] * L[j]; }
        if (k < 9) {
            m[k] = ((uint32_t)acc * FR9_LINV) & FR9_M;
            acc += (uint64_t)m[k] * FR9_L0;
        } else r.v[k - 9] = (uint32_t)acc & FR9_M;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
#endif
}

}  // namespace otti
