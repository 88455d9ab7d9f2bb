// Field arithmetic for the MI355X Spartan proving path: Fr = GF(l) (curve25519 scalar field, Montgomery form, R = 2^256)
// and Fp = GF(2^255-19).  One 32-byte little-endian storage format for host and device; on gfx950 the multiplies are
// 8x u32 limb chains (v_mad_u64_u32), on the host 4x u64 with 128-bit products.
//
// Replaces, for this path, upstream libspartan `src/scalar/ristretto255.rs::Scalar` and curve25519-dalek's FieldElement
// (Cargo dependencies of /root/reference/Spartan, an empty submodule: /root/reference/.gitmodules:4-6).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace otti {

struct alignas(16) Fr { uint32_t v[8]; };   // Montgomery form, always < l
struct alignas(16) Fp { uint32_t v[8]; };   // any representative < 2^256 of the class mod p (loosely reduced)

// ------------------------------------------------------------------------------------------------ constants
// l = 2^252 + 27742317777372353535851937790883648493
#define OTTI_L0 0x5cf5d3edu
#define OTTI_L1 0x5812631au
#define OTTI_L2 0xa2f79cd6u
#define OTTI_L3 0x14def9deu
#define OTTI_L7 0x10000000u
#define OTTI_LINV32 0x12547e1bu              // -l^{-1} mod 2^32

HD Fr fr_zero() { Fr r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
HD Fr fr_one() {                              // R mod l
    Fr r; r.v[0] = 0x8d98951du; r.v[1] = 0xd6ec3174u; r.v[2] = 0x737dcf70u; r.v[3] = 0xc6ef5bf4u;
    r.v[4] = 0xfffffffeu; r.v[5] = 0xffffffffu; r.v[6] = 0xffffffffu; r.v[7] = 0x0fffffffu; return r;
}
HD Fr fr_R2() {                               // R^2 mod l
    Fr r; r.v[0] = 0x449c0f01u; r.v[1] = 0xa40611e3u; r.v[2] = 0x68859347u; r.v[3] = 0xd00e1ba7u;
    r.v[4] = 0x17f5be65u; r.v[5] = 0xceec73d2u; r.v[6] = 0x7c309a3du; r.v[7] = 0x0399411bu; return r;
}
HD Fr fr_R3() {                               // R^3 mod l
    Fr r; r.v[0] = 0x7b83a2dbu; r.v[1] = 0x2a9e4968u; r.v[2] = 0xaef7f3ecu; r.v[3] = 0x278324e6u;
    r.v[4] = 0x04ec5b65u; r.v[5] = 0x8065dc6cu; r.v[6] = 0x3599cec7u; r.v[7] = 0x0e530b77u; return r;
}
HD uint32_t fr_L(int i) { return i == 0 ? OTTI_L0 : i == 1 ? OTTI_L1 : i == 2 ? OTTI_L2 : i == 3 ? OTTI_L3 : i == 7 ? OTTI_L7 : 0u; }

// ------------------------------------------------------------------------------------------------ Fr add / sub
// r = a - l if a >= l (a < 2l), branch-free
HD Fr fr_cond_sub_l(const uint32_t a[8], uint32_t top) {
    uint32_t t[8]; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a[i] - fr_L(i) - br; t[i] = (uint32_t)d; br = (d >> 32) & 1; }
    bool use_t = top || !br;
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = use_t ? t[i] : a[i];
    return r;
}
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950: explicit carry chains (v_add_co/v_addc_co, v_subrev_co/v_subbrev_co with the modulus limbs as literals).  The portable
// u64 formulation below compiles to 64-bit adds, shifts and moves on this target (~100 instructions per modular add instead of 24).
__device__ __forceinline__ void fr_dev_cond_sub_l(Fr &r, const uint32_t s[8]) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
    // (a literal and VCC in one VOP2 would exceed gfx9's one-constant-bus-read limit: the modulus limbs sit in VGPRs)
    const uint32_t l0 = OTTI_L0, l1 = OTTI_L1, l2 = OTTI_L2, l3 = OTTI_L3, l7 = OTTI_L7;
    asm("v_sub_co_u32_e32 %8, vcc, %16, %24\n\t"
        "v_subb_co_u32_e32 %9, vcc, %17, %25, vcc\n\t"
        "v_subb_co_u32_e32 %10, vcc, %18, %26, vcc\n\t"
        "v_subb_co_u32_e32 %11, vcc, %19, %27, vcc\n\t"
        "v_subbrev_co_u32_e32 %12, vcc, 0, %20, vcc\n\t"
        "v_subbrev_co_u32_e32 %13, vcc, 0, %21, vcc\n\t"
        "v_subbrev_co_u32_e32 %14, vcc, 0, %22, vcc\n\t"
        "v_subb_co_u32_e32 %15, vcc, %23, %28, vcc\n\t"
        "v_cndmask_b32_e32 %0, %8, %16, vcc\n\t"
        "v_cndmask_b32_e32 %1, %9, %17, vcc\n\t"
        "v_cndmask_b32_e32 %2, %10, %18, vcc\n\t"
        "v_cndmask_b32_e32 %3, %11, %19, vcc\n\t"
        "v_cndmask_b32_e32 %4, %12, %20, vcc\n\t"
        "v_cndmask_b32_e32 %5, %13, %21, vcc\n\t"
        "v_cndmask_b32_e32 %6, %14, %22, vcc\n\t"
        "v_cndmask_b32_e32 %7, %15, %23, vcc"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]),
          "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]),
          "v"(l0), "v"(l1), "v"(l2), "v"(l3), "v"(l7)
        : "vcc");
}
#endif
HD Fr fr_add(const Fr &a, const Fr &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t s[8]; Fr r;
    asm("v_add_co_u32_e32 %0, vcc, %8, %16\n\t"
        "v_addc_co_u32_e32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, %12, %20, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, %13, %21, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, %14, %22, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %15, %23, vcc"
        : "=&v"(s[0]), "=&v"(s[1]), "=&v"(s[2]), "=&v"(s[3]), "=&v"(s[4]), "=&v"(s[5]), "=&v"(s[6]), "=&v"(s[7])
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
        : "vcc");
    fr_dev_cond_sub_l(r, s);                  // a,b < l < 2^253: no carry out of 256 bits; subtract l iff s >= l
    return r;
#else
    uint32_t s[8]; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; s[i] = (uint32_t)c; c >>= 32; }
    return fr_cond_sub_l(s, 0);               // a,b < l < 2^253: no carry out of 256 bits
#endif
}
HD Fr fr_sub(const Fr &a, const Fr &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d[8], m; Fr r;
    asm("v_sub_co_u32_e32 %0, vcc, %9, %17\n\t"
        "v_subb_co_u32_e32 %1, vcc, %10, %18, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %11, %19, vcc\n\t"
        "v_subb_co_u32_e32 %3, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32_e32 %4, vcc, %13, %21, vcc\n\t"
        "v_subb_co_u32_e32 %5, vcc, %14, %22, vcc\n\t"
        "v_subb_co_u32_e32 %6, vcc, %15, %23, vcc\n\t"
        "v_subb_co_u32_e32 %7, vcc, %16, %24, vcc\n\t"
        "v_cndmask_b32_e64 %8, 0, -1, vcc"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(m)
        : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
          "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
        : "vcc");
    uint32_t l0 = OTTI_L0 & m, l1 = OTTI_L1 & m, l2 = OTTI_L2 & m, l3 = OTTI_L3 & m, l7 = OTTI_L7 & m;
    asm("v_add_co_u32_e32 %0, vcc, %8, %16\n\t"
        "v_addc_co_u32_e32 %1, vcc, %9, %17, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, %10, %18, vcc\n\t"
        "v_addc_co_u32_e32 %3, vcc, %11, %19, vcc\n\t"
        "v_addc_co_u32_e32 %4, vcc, 0, %12, vcc\n\t"
        "v_addc_co_u32_e32 %5, vcc, 0, %13, vcc\n\t"
        "v_addc_co_u32_e32 %6, vcc, 0, %14, vcc\n\t"
        "v_addc_co_u32_e32 %7, vcc, %20, %15, vcc"
        : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
        : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]),
          "v"(l0), "v"(l1), "v"(l2), "v"(l3), "v"(l7)
        : "vcc");
    return r;
#else
    uint32_t s[8]; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - br; s[i] = (uint32_t)d; br = (d >> 32) & 1; }
    uint32_t m = (uint32_t)0 - (uint32_t)br;  // all ones if borrow
    Fr r; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)s[i] + (fr_L(i) & m); r.v[i] = (uint32_t)c; c >>= 32; }
    return r;
#endif
}
HD Fr fr_neg(const Fr &a) { return fr_sub(fr_zero(), a); }
HD Fr fr_dbl(const Fr &a) { return fr_add(a, a); }
HD bool fr_eq(const Fr &a, const Fr &b) { uint32_t d = 0; for (int i = 0; i < 8; i++) d |= a.v[i] ^ b.v[i]; return d == 0; }
HD bool fr_is_zero(const Fr &a) { uint32_t d = 0; for (int i = 0; i < 8; i++) d |= a.v[i]; return d == 0; }

// ------------------------------------------------------------------------------------------------ gfx950 multiply-accumulate
#if defined(__HIP_DEVICE_COMPILE__)
// 96-bit column accumulator (lohi: low 64 bits, ex: carries) += x * y.  v_mad_u64_u32 leaves the carry-out of its 64-bit add in
// VCC; v_addc_co_u32 folds it into the third word: two VALU instructions per 32x32 partial product, no 64-bit adds, no moves.
__device__ __forceinline__ void mac96(uint64_t &lohi, uint32_t &ex, uint32_t x, uint32_t y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(lohi), "+v"(ex) : "v"(x), "v"(y) : "vcc");
}
// same with a wave-uniform multiplier kept in an SGPR (modulus limbs, 38)
__device__ __forceinline__ void mac96s(uint64_t &lohi, uint32_t &ex, uint32_t x, uint32_t y_uniform) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" : "+v"(lohi), "+v"(ex) : "v"(x), "s"(y_uniform) : "vcc");
}
// next column: drop the finished low word
__device__ __forceinline__ uint32_t col_shift(uint64_t &lohi, uint32_t &ex) {
    uint32_t out = (uint32_t)lohi;
    lohi = (lohi >> 32) | ((uint64_t)ex << 32); ex = 0;
    return out;
}
#endif

// ------------------------------------------------------------------------------------------------ Fr Montgomery multiply
// Device: CIOS over 8 u32 limbs (l has limbs 4..6 == 0 and limb 7 == 2^28, so a reduction row needs 4 real multiplies).
// Host: the same recurrence over 4 u64 limbs.
HD Fr fr_mul(const Fr &a, const Fr &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // product scanning with the Montgomery reduction interleaved column by column (FIPS): column k collects a_i*b_(k-i) and
    // m_i*l_(k-i); m_k makes the column's low word vanish.  l has limbs 4..6 == 0.  Body generated by gen_fr_mul.py.
#include "fr_mul_gfx950.inc"
    // a, b < l: the Montgomery product is < 2l < 2^254, so t[8] == 0 and one conditional subtraction finishes
    Fr r; fr_dev_cond_sub_l(r, t); return r;
#else
    typedef unsigned __int128 u128;
    static const uint64_t L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
    const uint64_t INV = 0xd2b51da312547e1bULL;
    uint64_t x[4], y[4], t[6] = {0, 0, 0, 0, 0, 0};
    memcpy(x, a.v, 32); memcpy(y, b.v, 32);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)x[j] * y[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * INV;
        c = (u128)m * L[0] + t[0]; c >>= 64;
        c += (u128)m * L[1] + t[1]; t[0] = (uint64_t)c; c >>= 64;
        c += t[2]; t[1] = (uint64_t)c; c >>= 64;
        c += (u128)m * L[3] + t[3]; t[2] = (uint64_t)c; c >>= 64;
        c += t[4]; t[3] = (uint64_t)c; c >>= 64;
        t[4] = t[5] + (uint64_t)c;
    }
    uint32_t w[8]; memcpy(w, t, 32);
    return fr_cond_sub_l(w, (uint32_t)t[4]);
#endif
}
HD Fr fr_sqr(const Fr &a) { return fr_mul(a, a); }

// x * c mod l for a small signed integer c (|c| < 2^31); x in either form (the form is kept: (xR) c = (xc) R).  The coefficients of a
// compiled circuit are mostly such integers: ~15 multiply-accumulates instead of a 128-product Montgomery multiplication.
//   t = x |c| < 2^283;  q = t >> 252 is floor(t / l) or one more (l = 2^252 + delta, delta < 2^125);  t - q l = (t mod 2^252) - q delta
HD Fr fr_mul_small(const Fr &x, int32_t c) {
    const uint32_t mag = c < 0 ? (uint32_t)(-(int64_t)c) : (uint32_t)c;
    uint32_t t[9]; uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { acc += (uint64_t)x.v[i] * mag; t[i] = (uint32_t)acc; acc >>= 32; }
    t[8] = (uint32_t)acc;
    const uint32_t q = (t[8] << 4) | (t[7] >> 28);
    Fr low; for (int i = 0; i < 7; i++) low.v[i] = t[i];
    low.v[7] = t[7] & 0x0fffffffu;
    Fr qd = fr_zero(); acc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { acc += (uint64_t)fr_L(i) * q; qd.v[i] = (uint32_t)acc; acc >>= 32; }
    qd.v[4] = (uint32_t)acc;
    const Fr r = fr_sub(low, qd);                              // both below l: the difference mod l
    return c < 0 ? fr_neg(r) : r;
}
HD Fr fr_from_u64(uint64_t x) { Fr t = fr_zero(); t.v[0] = (uint32_t)x; t.v[1] = (uint32_t)(x >> 32); return fr_mul(t, fr_R2()); }
// canonical integer (out of Montgomery form)
HD Fr fr_to_raw(const Fr &a) { Fr one = fr_zero(); one.v[0] = 1; return fr_mul(a, one); }
constexpr int32_t kNotSmall = INT32_MIN;                         // fr_small_code of a value that is no such integer
// the small signed integer a Montgomery-form value stands for, or kNotSmall
HD int32_t fr_small_code(const Fr &mont) {
    const Fr raw = fr_to_raw(mont);
    if ((raw.v[1] | raw.v[2] | raw.v[3] | raw.v[4] | raw.v[5] | raw.v[6] | raw.v[7]) == 0 && raw.v[0] < 0x7fffffffu) return (int32_t)raw.v[0];
    const Fr neg = fr_to_raw(fr_neg(mont));
    if ((neg.v[1] | neg.v[2] | neg.v[3] | neg.v[4] | neg.v[5] | neg.v[6] | neg.v[7]) == 0 && neg.v[0] < 0x7fffffffu) return -(int32_t)neg.v[0];
    return kNotSmall;
}
HD void fr_to_bytes(uint8_t b[32], const Fr &a) {
    Fr r = fr_to_raw(a);
    for (int i = 0; i < 8; i++) { b[4 * i] = (uint8_t)r.v[i]; b[4 * i + 1] = (uint8_t)(r.v[i] >> 8); b[4 * i + 2] = (uint8_t)(r.v[i] >> 16); b[4 * i + 3] = (uint8_t)(r.v[i] >> 24); }
}
HD bool fr_raw_is_canonical(const uint32_t w[8]) {
    for (int i = 7; i >= 0; i--) { uint32_t li = fr_L(i); if (w[i] < li) return true; if (w[i] > li) return false; }
    return false;
}
HD bool fr_from_bytes(Fr &o, const uint8_t b[32]) {       // false if >= l (upstream: InvalidScalar)
    Fr t;
    for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    if (!fr_raw_is_canonical(t.v)) { o = fr_zero(); return false; }
    o = fr_mul(t, fr_R2());
    return true;
}
HD Fr fr_from_bytes_wide(const uint8_t b[64]) {           // 512-bit LE mod l: lo*R2/R + hi*R3/R
    Fr lo, hi;
    for (int i = 0; i < 8; i++) {
        lo.v[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
        hi.v[i] = (uint32_t)b[32 + 4 * i] | ((uint32_t)b[32 + 4 * i + 1] << 8) | ((uint32_t)b[32 + 4 * i + 2] << 16) | ((uint32_t)b[32 + 4 * i + 3] << 24);
    }
    return fr_add(fr_mul(lo, fr_R2()), fr_mul(hi, fr_R3()));
}
HD Fr fr_inv(const Fr &a) {                               // a^(l-2); 0 -> 0
    Fr e; e.v[0] = OTTI_L0 - 2; e.v[1] = OTTI_L1; e.v[2] = OTTI_L2; e.v[3] = OTTI_L3; e.v[4] = e.v[5] = e.v[6] = 0; e.v[7] = OTTI_L7;
    Fr acc = fr_one();
    for (int i = 252; i >= 0; i--) { acc = fr_sqr(acc); if ((e.v[i >> 5] >> (i & 31)) & 1) acc = fr_mul(acc, a); }
    return acc;
}

// ================================================================================================ Fp = GF(2^255 - 19)
HD Fp fp_zero() { Fp r; for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
HD Fp fp_one() { Fp r = fp_zero(); r.v[0] = 1; return r; }
HD Fp fp_from_words(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5, uint32_t a6, uint32_t a7) {
    Fp r; r.v[0] = a0; r.v[1] = a1; r.v[2] = a2; r.v[3] = a3; r.v[4] = a4; r.v[5] = a5; r.v[6] = a6; r.v[7] = a7; return r;
}
HD Fp fp_D() { return fp_from_words(0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu); }
HD Fp fp_2D() { return fp_from_words(0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu); }
HD Fp fp_SQRT_M1() { return fp_from_words(0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u); }
HD Fp fp_SQRT_AD_MINUS_ONE() { return fp_from_words(0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu); }
HD Fp fp_INVSQRT_A_MINUS_D() { return fp_from_words(0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u); }
HD Fp fp_ONE_MINUS_D_SQ() { return fp_from_words(0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u); }
HD Fp fp_D_MINUS_ONE_SQ() { return fp_from_words(0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au); }

// 2^256 = 38 (mod p): a carry out of 256 bits folds back as +38, a borrow as -38
HD Fp fp_add(const Fp &a, const Fp &b) {
    Fp r; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    c *= 38;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += r.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    r.v[0] += (uint32_t)c * 38;               // second wrap leaves the low limb tiny: cannot carry again
    return r;
}
HD Fp fp_sub(const Fp &a, const Fp &b) {
    Fp r; uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)a.v[i] - b.v[i] - br; r.v[i] = (uint32_t)d; br = (d >> 32) & 1; }
    uint64_t s = br * 38; br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)r.v[i] - s - br; r.v[i] = (uint32_t)d; br = (d >> 32) & 1; s = 0; }
    r.v[0] -= (uint32_t)br * 38;              // second wrap: value is now close to 2^256, cannot borrow again
    return r;
}
HD Fp fp_neg(const Fp &a) { return fp_sub(fp_zero(), a); }

HD Fp fp_reduce512(const uint32_t t[16]) {
    // r = lo + 38*hi (two multiply-adds per limb, sums stay below 2^64), then fold the (<= 38) top word twice
    Fp r; uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (uint64_t)t[8 + i] * 38u; c += t[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    c *= 38;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += r.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    r.v[0] += (uint32_t)c * 38;
    return r;
}
HD Fp fp_mul(const Fp &a, const Fp &b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t t[16];
    uint64_t acc = 0; uint32_t ex = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = (k < 8 ? 0 : k - 7); i <= (k < 8 ? k : 7); i++) mac96(acc, ex, a.v[i], b.v[k - i]);
        t[k] = col_shift(acc, ex);
    }
    t[15] = (uint32_t)acc;
    return fp_reduce512(t);
#else
    typedef unsigned __int128 u128;
    uint64_t x[4], y[4], t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(x, a.v, 32); memcpy(y, b.v, 32);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)x[j] * y[i] + t[i + j]; t[i + j] = (uint64_t)c; c >>= 64; }
        t[i + 4] = (uint64_t)c;
    }
    uint64_t r[4]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)t[4 + i] * 38u + t[i]; r[i] = (uint64_t)c; c >>= 64; }
    c *= 38;
    for (int i = 0; i < 4; i++) { c += r[i]; r[i] = (uint64_t)c; c >>= 64; }
    r[0] += (uint64_t)c * 38;
    Fp o; memcpy(o.v, r, 32); return o;
#endif
}
HD Fp fp_sqr(const Fp &a) { return fp_mul(a, a); }

HD Fp fp_sqr_n(Fp a, int n) { for (int i = 0; i < n; i++) a = fp_sqr(a); return a; }
// a^(2^250-1) and a^11 (shared prefix of the inversion and (p-5)/8 chains)
HD void fp_pow_2_250_1(const Fp &a, Fp &t250, Fp &a11) {
    Fp z2 = fp_sqr(a), z9 = fp_mul(fp_sqr_n(z2, 2), a), z11 = fp_mul(z9, z2);
    Fp t5 = fp_mul(fp_sqr(z11), z9);
    Fp t10 = fp_mul(fp_sqr_n(t5, 5), t5), t20 = fp_mul(fp_sqr_n(t10, 10), t10), t40 = fp_mul(fp_sqr_n(t20, 20), t20);
    Fp t50 = fp_mul(fp_sqr_n(t40, 10), t10), t100 = fp_mul(fp_sqr_n(t50, 50), t50), t200 = fp_mul(fp_sqr_n(t100, 100), t100);
    t250 = fp_mul(fp_sqr_n(t200, 50), t50); a11 = z11;
}
HD Fp fp_inv(const Fp &a) { Fp t, a11; fp_pow_2_250_1(a, t, a11); return fp_mul(fp_sqr_n(t, 5), a11); }
HD Fp fp_pow22523(const Fp &a) { Fp t, a11; fp_pow_2_250_1(a, t, a11); return fp_mul(fp_sqr_n(t, 2), a); }

// canonical representative in [0, p)
HD Fp fp_canon(const Fp &a) {
    Fp r = a;
#pragma unroll
    for (int k = 0; k < 2; k++) {             // fold bit 255 (2^255 = 19) twice
        uint64_t c = (uint64_t)(r.v[7] >> 31) * 19; r.v[7] &= 0x7fffffffu;
        for (int i = 0; i < 8; i++) { c += r.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
    }
    // now r < 2^255 + small; subtract p if r >= p:  r + 19 >= 2^255 ?
    uint32_t t[8]; uint64_t c = 19;
    for (int i = 0; i < 8; i++) { c += r.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    bool ge = (t[7] >> 31) != 0;
    t[7] &= 0x7fffffffu;
    for (int i = 0; i < 8; i++) r.v[i] = ge ? t[i] : r.v[i];
    return r;
}
HD void fp_to_bytes(uint8_t b[32], const Fp &a) {
    Fp r = fp_canon(a);
    for (int i = 0; i < 8; i++) { b[4 * i] = (uint8_t)r.v[i]; b[4 * i + 1] = (uint8_t)(r.v[i] >> 8); b[4 * i + 2] = (uint8_t)(r.v[i] >> 16); b[4 * i + 3] = (uint8_t)(r.v[i] >> 24); }
}
HD Fp fp_from_bytes(const uint8_t b[32]) {                 // top bit ignored
    Fp t;
    for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    t.v[7] &= 0x7fffffffu;
    return t;
}
HD bool fp_bytes_canonical(const uint8_t b[32]) {
    if (b[31] & 0x80) return false;
    Fp t = fp_from_bytes(b), c = fp_canon(t);
    uint32_t d = 0; for (int i = 0; i < 8; i++) d |= t.v[i] ^ c.v[i];
    return d == 0;
}
HD bool fp_is_negative(const Fp &a) { return fp_canon(a).v[0] & 1; }
HD bool fp_is_zero(const Fp &a) { Fp c = fp_canon(a); uint32_t d = 0; for (int i = 0; i < 8; i++) d |= c.v[i]; return d == 0; }
HD bool fp_eq(const Fp &a, const Fp &b) { return fp_is_zero(fp_sub(a, b)); }
HD Fp fp_abs(const Fp &a) { return fp_is_negative(a) ? fp_neg(a) : a; }

// RFC 9496 4.2 SQRT_RATIO_M1
HD bool fp_sqrt_ratio_m1(Fp &r_out, const Fp &u, const Fp &v) {
    Fp v3 = fp_mul(fp_sqr(v), v), v7 = fp_mul(fp_sqr(v3), v);
    Fp r = fp_mul(fp_mul(u, v3), fp_pow22523(fp_mul(u, v7)));
    Fp check = fp_mul(v, fp_sqr(r));
    Fp neg_u = fp_neg(u), neg_u_i = fp_mul(neg_u, fp_SQRT_M1());
    bool correct = fp_eq(check, u), flipped = fp_eq(check, neg_u), flipped_i = fp_eq(check, neg_u_i);
    if (flipped || flipped_i) r = fp_mul(fp_SQRT_M1(), r);
    r_out = fp_abs(r);
    return correct || flipped;
}

}  // namespace otti
