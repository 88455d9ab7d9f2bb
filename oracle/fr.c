/*
 * ORACLE — test infrastructure only (see fr.h).  Scalar field GF(l) of curve25519 / ristretto255.
 * Restates upstream libspartan src/scalar/ristretto255.rs [RECALL: Scalar::{add,sub,mul,square,invert,
 * from_bytes,from_bytes_wide,to_bytes,batch_invert}]; the reference mount holds no copy (empty Spartan/).
 */
#include "fr.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;

static const uint64_t L[4]  = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};
static const fr_t R2 = {{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL, 0x0399411b7c309a3dULL}};
static const fr_t R3 = {{0x2a9e49687b83a2dbULL, 0x278324e6aef7f3ecULL, 0x8065dc6c04ec5b65ULL, 0x0e530b773599cec7ULL}};
static const uint64_t INV = 0xd2b51da312547e1bULL;   /* -l^{-1} mod 2^64 */

const fr_t FR_ZERO = {{0, 0, 0, 0}};
const fr_t FR_ONE  = {{0xd6ec31748d98951dULL, 0xc6ef5bf4737dcf70ULL, 0xfffffffffffffffeULL, 0x0fffffffffffffffULL}}; /* R mod l */

/* o = a - l if a >= l (a < 2l assumed) */
static inline void cond_sub_l(uint64_t o[4], const uint64_t a[4], uint64_t carry_in) {
    uint64_t t[4]; u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - L[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    /* if carry_in set, a really is >= 2^256 > l, subtraction is right; else use t only when no borrow */
    int use_t = carry_in || !br;
    for (int i = 0; i < 4; i++) o[i] = use_t ? t[i] : a[i];
}

void fr_add(fr_t *o, const fr_t *a, const fr_t *b) {
    uint64_t s[4]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; s[i] = (uint64_t)c; c >>= 64; }
    cond_sub_l(o->v, s, (uint64_t)c);
}

void fr_sub(fr_t *o, const fr_t *a, const fr_t *b) {
    uint64_t s[4]; u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - (uint64_t)br;
        s[i] = (uint64_t)d; br = (d >> 64) & 1;
    }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)s[i] + L[i]; s[i] = (uint64_t)c; c >>= 64; } }
    memcpy(o->v, s, 32);
}

void fr_neg(fr_t *o, const fr_t *a) { fr_sub(o, &FR_ZERO, a); }

/* Montgomery product, CIOS */
void fr_mul(fr_t *o, const fr_t *a, const fr_t *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a->v[j] * b->v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * INV;
        c = (u128)m * L[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * L[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; c >>= 64;
        t[4] = t[5] + (uint64_t)c;
    }
    cond_sub_l(o->v, t, t[4]);
}

void fr_sqr(fr_t *o, const fr_t *a) { fr_mul(o, a, a); }

int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a->v, b->v, 32) == 0; }
int fr_is_zero(const fr_t *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }

void fr_from_u64(fr_t *o, uint64_t x) { fr_t t = {{x, 0, 0, 0}}; fr_mul(o, &t, &R2); }

static uint64_t load64(const uint8_t *p) { uint64_t x = 0; for (int i = 7; i >= 0; i--) x = (x << 8) | p[i]; return x; }
static void store64(uint8_t *p, uint64_t x) { for (int i = 0; i < 8; i++) { p[i] = (uint8_t)x; x >>= 8; } }

int fr_from_bytes(fr_t *o, const uint8_t b[32]) {
    fr_t t; for (int i = 0; i < 4; i++) t.v[i] = load64(b + 8 * i);
    /* canonical iff t < l */
    int lt = 0;
    for (int i = 3; i >= 0; i--) { if (t.v[i] < L[i]) { lt = 1; break; } if (t.v[i] > L[i]) { lt = 0; break; } }
    if (!lt) { *o = FR_ZERO; return 0; }
    fr_mul(o, &t, &R2);
    return 1;
}

void fr_from_bytes_wide(fr_t *o, const uint8_t b[64]) {
    /* value = d0 + d1*2^256 ; mont(d0,R2) = d0*R ; mont(d1,R3) = d1*R^2 = (d1*2^256)*R   [RECALL Scalar::from_u512] */
    fr_t d0, d1, x, y;
    for (int i = 0; i < 4; i++) { d0.v[i] = load64(b + 8 * i); d1.v[i] = load64(b + 32 + 8 * i); }
    /* fr_mul tolerates unreduced 256-bit inputs on one side: result < 2l is handled by cond_sub since
       a*b/R + l < 2^256*l/R... keep it simple: reduce operands first by one Montgomery round trip. */
    fr_mul(&x, &d0, &R2);
    fr_mul(&y, &d1, &R3);
    fr_add(o, &x, &y);
}

void fr_to_raw(uint64_t r[4], const fr_t *a) { fr_t one = {{1, 0, 0, 0}}, t; fr_mul(&t, a, &one); memcpy(r, t.v, 32); }
void fr_to_bytes(uint8_t b[32], const fr_t *a) { uint64_t r[4]; fr_to_raw(r, a); for (int i = 0; i < 4; i++) store64(b + 8 * i, r[i]); }
void fr_mont_bytes(uint8_t b[32], const fr_t *a) { for (int i = 0; i < 4; i++) store64(b + 8 * i, a->v[i]); }
int fr_from_mont_bytes(fr_t *o, const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) o->v[i] = load64(b + 8 * i);
    for (int i = 3; i >= 0; i--) { if (o->v[i] < L[i]) return 1; if (o->v[i] > L[i]) return 0; }
    return 0;
}

void fr_inv(fr_t *o, const fr_t *a) {
    /* a^(l-2), plain square-and-multiply, MSB first */
    static const uint64_t E[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};
    fr_t acc = FR_ONE, base = *a;
    for (int i = 252; i >= 0; i--) {
        fr_sqr(&acc, &acc);
        if ((E[i >> 6] >> (i & 63)) & 1) fr_mul(&acc, &acc, &base);
    }
    *o = acc;
}

void fr_batch_inv(fr_t *x, size_t n) {
    if (!n) return;
    fr_t *pre = (fr_t *)malloc(n * sizeof(fr_t));
    fr_t acc = FR_ONE;
    for (size_t i = 0; i < n; i++) { pre[i] = acc; if (!fr_is_zero(&x[i])) fr_mul(&acc, &acc, &x[i]); }
    fr_inv(&acc, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fr_is_zero(&x[i])) continue;
        fr_t t; fr_mul(&t, &acc, &pre[i]);
        fr_mul(&acc, &acc, &x[i]);
        x[i] = t;
    }
    free(pre);
}
