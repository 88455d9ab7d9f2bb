/*
 * ORACLE — test infrastructure only (see fp.h).  GF(2^255-19) in five 51-bit limbs.
 */
#include "fp.h"
#include <string.h>

typedef unsigned __int128 u128;
#define M51 0x7ffffffffffffULL

const fp_t FP_ZERO = {{0, 0, 0, 0, 0}};
const fp_t FP_ONE  = {{1, 0, 0, 0, 0}};
const fp_t FP_D = {{0x34dca135978a3ULL, 0x1a8283b156ebdULL, 0x5e7a26001c029ULL, 0x739c663a03cbbULL, 0x52036cee2b6ffULL}};
const fp_t FP_2D = {{0x69b9426b2f159ULL, 0x35050762add7aULL, 0x3cf44c0038052ULL, 0x6738cc7407977ULL, 0x2406d9dc56dffULL}};
const fp_t FP_SQRT_M1 = {{0x61b274a0ea0b0ULL, 0x0d5a5fc8f189dULL, 0x7ef5e9cbd0c60ULL, 0x78595a6804c9eULL, 0x2b8324804fc1dULL}};
const fp_t FP_SQRT_AD_MINUS_ONE = {{0x7f6a0497b2e1bULL, 0x1836f0a97afd2ULL, 0x7d747f6be7638ULL, 0x456079e7e6498ULL, 0x376931bf2b834ULL}};
const fp_t FP_INVSQRT_A_MINUS_D = {{0x0fdaa805d40eaULL, 0x2eb482e57d339ULL, 0x007610274bc58ULL, 0x6510b613dc8ffULL, 0x786c8905cfaffULL}};
const fp_t FP_ONE_MINUS_D_SQ = {{0x409c1945fc176ULL, 0x719abc6a1fc4fULL, 0x1c37f90b20684ULL, 0x06bccca55eedfULL, 0x029072a8b2b3eULL}};
const fp_t FP_D_MINUS_ONE_SQ = {{0x55aaa44ed4d20ULL, 0x59603c3332635ULL, 0x26d3baf4a7928ULL, 0x120a66e6997a9ULL, 0x5968b37af66c2ULL}};

static inline void carry(fp_t *o) {
    uint64_t c;
    c = o->v[0] >> 51; o->v[0] &= M51; o->v[1] += c;
    c = o->v[1] >> 51; o->v[1] &= M51; o->v[2] += c;
    c = o->v[2] >> 51; o->v[2] &= M51; o->v[3] += c;
    c = o->v[3] >> 51; o->v[3] &= M51; o->v[4] += c;
    c = o->v[4] >> 51; o->v[4] &= M51; o->v[0] += 19 * c;
    c = o->v[0] >> 51; o->v[0] &= M51; o->v[1] += c;
}

void fp_add(fp_t *o, const fp_t *a, const fp_t *b) {
    fp_t t; for (int i = 0; i < 5; i++) t.v[i] = a->v[i] + b->v[i];
    carry(&t); *o = t;
}

void fp_sub(fp_t *o, const fp_t *a, const fp_t *b) {
    /* add 4p so limbs stay non-negative (inputs have limbs < 2^52) */
    fp_t t;
    t.v[0] = a->v[0] + 0x1fffffffffffb4ULL - b->v[0];
    t.v[1] = a->v[1] + 0x1ffffffffffffcULL - b->v[1];
    t.v[2] = a->v[2] + 0x1ffffffffffffcULL - b->v[2];
    t.v[3] = a->v[3] + 0x1ffffffffffffcULL - b->v[3];
    t.v[4] = a->v[4] + 0x1ffffffffffffcULL - b->v[4];
    carry(&t); *o = t;
}

void fp_neg(fp_t *o, const fp_t *a) { fp_sub(o, &FP_ZERO, a); }

void fp_mul(fp_t *o, const fp_t *a, const fp_t *b) {
    const uint64_t *x = a->v, *y = b->v;
    uint64_t y1 = 19 * y[1], y2 = 19 * y[2], y3 = 19 * y[3], y4 = 19 * y[4];
    u128 r0 = (u128)x[0] * y[0] + (u128)x[1] * y4 + (u128)x[2] * y3 + (u128)x[3] * y2 + (u128)x[4] * y1;
    u128 r1 = (u128)x[0] * y[1] + (u128)x[1] * y[0] + (u128)x[2] * y4 + (u128)x[3] * y3 + (u128)x[4] * y2;
    u128 r2 = (u128)x[0] * y[2] + (u128)x[1] * y[1] + (u128)x[2] * y[0] + (u128)x[3] * y4 + (u128)x[4] * y3;
    u128 r3 = (u128)x[0] * y[3] + (u128)x[1] * y[2] + (u128)x[2] * y[1] + (u128)x[3] * y[0] + (u128)x[4] * y4;
    u128 r4 = (u128)x[0] * y[4] + (u128)x[1] * y[3] + (u128)x[2] * y[2] + (u128)x[3] * y[1] + (u128)x[4] * y[0];
    fp_t t; uint64_t c;
    r1 += (uint64_t)(r0 >> 51); t.v[0] = (uint64_t)r0 & M51;
    r2 += (uint64_t)(r1 >> 51); t.v[1] = (uint64_t)r1 & M51;
    r3 += (uint64_t)(r2 >> 51); t.v[2] = (uint64_t)r2 & M51;
    r4 += (uint64_t)(r3 >> 51); t.v[3] = (uint64_t)r3 & M51;
    c = (uint64_t)(r4 >> 51);   t.v[4] = (uint64_t)r4 & M51;
    t.v[0] += 19 * c;
    c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
    *o = t;
}

void fp_sqr(fp_t *o, const fp_t *a) { fp_mul(o, a, a); }

static void sqr_n(fp_t *o, const fp_t *a, int n) { fp_t t = *a; for (int i = 0; i < n; i++) fp_sqr(&t, &t); *o = t; }

/* t250 = a^(2^250-1), also returns a^11 */
static void pow_2_250_1(fp_t *t250, fp_t *a11, const fp_t *a) {
    fp_t z2, z9, z11, t, t5, t10, t20, t40, t50, t100, t200;
    fp_sqr(&z2, a);
    sqr_n(&t, &z2, 2);          /* a^8 */
    fp_mul(&z9, &t, a);
    fp_mul(&z11, &z9, &z2);
    fp_sqr(&t, &z11);           /* a^22 */
    fp_mul(&t5, &t, &z9);       /* 2^5-1 */
    sqr_n(&t, &t5, 5);   fp_mul(&t10, &t, &t5);
    sqr_n(&t, &t10, 10); fp_mul(&t20, &t, &t10);
    sqr_n(&t, &t20, 20); fp_mul(&t40, &t, &t20);
    sqr_n(&t, &t40, 10); fp_mul(&t50, &t, &t10);
    sqr_n(&t, &t50, 50); fp_mul(&t100, &t, &t50);
    sqr_n(&t, &t100, 100); fp_mul(&t200, &t, &t100);
    sqr_n(&t, &t200, 50); fp_mul(t250, &t, &t50);
    *a11 = z11;
}

void fp_inv(fp_t *o, const fp_t *a) {
    fp_t t250, a11, t;
    pow_2_250_1(&t250, &a11, a);
    sqr_n(&t, &t250, 5);        /* 2^255 - 32 */
    fp_mul(o, &t, &a11);        /* 2^255 - 21 */
}

void fp_pow22523(fp_t *o, const fp_t *a) {
    fp_t t250, a11, t;
    pow_2_250_1(&t250, &a11, a);
    sqr_n(&t, &t250, 2);        /* 2^252 - 4 */
    fp_mul(o, &t, a);           /* 2^252 - 3 */
}

static uint64_t load64(const uint8_t *p) { uint64_t x = 0; for (int i = 7; i >= 0; i--) x = (x << 8) | p[i]; return x; }

void fp_from_bytes(fp_t *o, const uint8_t b[32]) {
    o->v[0] = load64(b) & M51;
    o->v[1] = (load64(b + 6) >> 3) & M51;
    o->v[2] = (load64(b + 12) >> 6) & M51;
    o->v[3] = (load64(b + 19) >> 1) & M51;
    o->v[4] = (load64(b + 24) >> 12) & M51;
}

void fp_to_bytes(uint8_t b[32], const fp_t *a) {
    fp_t t = *a; carry(&t); carry(&t);
    /* now 0 <= t < 2^255 + small; compute t mod p: q = (t + 19) >> 255 */
    uint64_t q = (t.v[0] + 19) >> 51;
    q = (t.v[1] + q) >> 51; q = (t.v[2] + q) >> 51; q = (t.v[3] + q) >> 51; q = (t.v[4] + q) >> 51;
    t.v[0] += 19 * q;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
    t.v[4] &= M51;
    uint64_t w0 = t.v[0] | (t.v[1] << 51);
    uint64_t w1 = (t.v[1] >> 13) | (t.v[2] << 38);
    uint64_t w2 = (t.v[2] >> 26) | (t.v[3] << 25);
    uint64_t w3 = (t.v[3] >> 39) | (t.v[4] << 12);
    uint64_t w[4] = {w0, w1, w2, w3};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 8; j++) b[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

int fp_is_canonical_bytes(const uint8_t b[32]) {
    fp_t t; uint8_t c[32];
    if (b[31] & 0x80) return 0;
    fp_from_bytes(&t, b); fp_to_bytes(c, &t);
    return memcmp(b, c, 32) == 0;
}

int fp_is_negative(const fp_t *a) { uint8_t b[32]; fp_to_bytes(b, a); return b[0] & 1; }
int fp_is_zero(const fp_t *a) { uint8_t b[32]; fp_to_bytes(b, a); uint8_t x = 0; for (int i = 0; i < 32; i++) x |= b[i]; return x == 0; }
int fp_eq(const fp_t *a, const fp_t *b) { uint8_t x[32], y[32]; fp_to_bytes(x, a); fp_to_bytes(y, b); return memcmp(x, y, 32) == 0; }
void fp_cmov(fp_t *o, const fp_t *a, int c) { if (c) *o = *a; }
void fp_abs(fp_t *o, const fp_t *a) { if (fp_is_negative(a)) fp_neg(o, a); else *o = *a; }

int fp_sqrt_ratio_m1(fp_t *r_out, const fp_t *u, const fp_t *v) {
    fp_t v3, v7, r, check, t, neg_u, neg_u_i, r_prime;
    fp_sqr(&t, v); fp_mul(&v3, &t, v);            /* v^3 */
    fp_sqr(&t, &v3); fp_mul(&v7, &t, v);          /* v^7 */
    fp_mul(&t, u, &v7); fp_pow22523(&t, &t);
    fp_mul(&r, u, &v3); fp_mul(&r, &r, &t);
    fp_sqr(&t, &r); fp_mul(&check, v, &t);
    fp_neg(&neg_u, u); fp_mul(&neg_u_i, &neg_u, &FP_SQRT_M1);
    int correct = fp_eq(&check, u), flipped = fp_eq(&check, &neg_u), flipped_i = fp_eq(&check, &neg_u_i);
    fp_mul(&r_prime, &FP_SQRT_M1, &r);
    if (flipped | flipped_i) r = r_prime;
    fp_abs(&r, &r);
    *r_out = r;
    return correct | flipped;
}
