/*
 * ORACLE — test infrastructure only (see ristretto.h).
 * Formulas: RFC 9496 section 4.3 (decode/encode/one-way map), HWCD'08 extended-coordinate add/double (a = -1).
 */
#include "ristretto.h"
#include <string.h>
#include <stdlib.h>

const uint8_t RISTRETTO_BASEPOINT_COMPRESSED[32] = {
    0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
    0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};

void ge_identity(ge_t *o) { o->X = FP_ZERO; o->Y = FP_ONE; o->Z = FP_ONE; o->T = FP_ZERO; }

void ge_add(ge_t *o, const ge_t *p, const ge_t *q) {
    fp_t a, b, c, d, e, f, g, h, t0, t1;
    fp_sub(&t0, &p->Y, &p->X); fp_sub(&t1, &q->Y, &q->X); fp_mul(&a, &t0, &t1);
    fp_add(&t0, &p->Y, &p->X); fp_add(&t1, &q->Y, &q->X); fp_mul(&b, &t0, &t1);
    fp_mul(&c, &p->T, &q->T); fp_mul(&c, &c, &FP_2D);
    fp_mul(&d, &p->Z, &q->Z); fp_add(&d, &d, &d);
    fp_sub(&e, &b, &a); fp_sub(&f, &d, &c); fp_add(&g, &d, &c); fp_add(&h, &b, &a);
    fp_mul(&o->X, &e, &f); fp_mul(&o->Y, &g, &h); fp_mul(&o->T, &e, &h); fp_mul(&o->Z, &f, &g);
}

void ge_neg(ge_t *o, const ge_t *a) { fp_neg(&o->X, &a->X); o->Y = a->Y; o->Z = a->Z; fp_neg(&o->T, &a->T); }
void ge_sub(ge_t *o, const ge_t *a, const ge_t *b) { ge_t n; ge_neg(&n, b); ge_add(o, a, &n); }

void ge_dbl(ge_t *o, const ge_t *p) {
    fp_t a, b, c, e, f, g, h, t;
    fp_sqr(&a, &p->X); fp_sqr(&b, &p->Y);
    fp_sqr(&c, &p->Z); fp_add(&c, &c, &c);
    fp_add(&t, &p->X, &p->Y); fp_sqr(&t, &t);
    fp_sub(&e, &t, &a); fp_sub(&e, &e, &b);     /* E = (X+Y)^2 - A - B */
    fp_sub(&g, &b, &a);                         /* G = D + B with D = -A */
    fp_sub(&f, &g, &c);                         /* F = G - C */
    fp_add(&h, &a, &b); fp_neg(&h, &h);         /* H = D - B = -A - B */
    fp_mul(&o->X, &e, &f); fp_mul(&o->Y, &g, &h); fp_mul(&o->T, &e, &h); fp_mul(&o->Z, &f, &g);
}

int ge_decode(ge_t *o, const uint8_t bytes[32]) {
    fp_t s, ss, u1, u2, u2s, v, t, inv, dx, dy, x, y;
    if (!fp_is_canonical_bytes(bytes) || (bytes[0] & 1)) return 0;
    fp_from_bytes(&s, bytes);
    fp_sqr(&ss, &s);
    fp_sub(&u1, &FP_ONE, &ss); fp_add(&u2, &FP_ONE, &ss);
    fp_sqr(&u2s, &u2);
    fp_sqr(&t, &u1); fp_mul(&t, &t, &FP_D); fp_neg(&t, &t); fp_sub(&v, &t, &u2s);     /* v = -(D*u1^2) - u2^2 */
    fp_mul(&t, &v, &u2s);
    int was_square = fp_sqrt_ratio_m1(&inv, &FP_ONE, &t);
    fp_mul(&dx, &inv, &u2);
    fp_mul(&dy, &inv, &dx); fp_mul(&dy, &dy, &v);
    fp_mul(&x, &s, &dx); fp_add(&x, &x, &x); fp_abs(&x, &x);
    fp_mul(&y, &u1, &dy);
    fp_mul(&t, &x, &y);
    if (!was_square || fp_is_negative(&t) || fp_is_zero(&y)) return 0;
    o->X = x; o->Y = y; o->Z = FP_ONE; o->T = t;
    return 1;
}

void ge_encode(uint8_t out[32], const ge_t *p) {
    fp_t u1, u2, t, inv, den1, den2, zinv, ix, iy, ench, x, y, deninv, s;
    fp_add(&u1, &p->Z, &p->Y); fp_sub(&t, &p->Z, &p->Y); fp_mul(&u1, &u1, &t);
    fp_mul(&u2, &p->X, &p->Y);
    fp_sqr(&t, &u2); fp_mul(&t, &t, &u1);
    fp_sqrt_ratio_m1(&inv, &FP_ONE, &t);
    fp_mul(&den1, &inv, &u1); fp_mul(&den2, &inv, &u2);
    fp_mul(&zinv, &den1, &den2); fp_mul(&zinv, &zinv, &p->T);
    fp_mul(&ix, &p->X, &FP_SQRT_M1); fp_mul(&iy, &p->Y, &FP_SQRT_M1);
    fp_mul(&ench, &den1, &FP_INVSQRT_A_MINUS_D);
    fp_mul(&t, &p->T, &zinv);
    int rotate = fp_is_negative(&t);
    x = p->X; y = p->Y; deninv = den2;
    if (rotate) { x = iy; y = ix; deninv = ench; }
    fp_mul(&t, &x, &zinv);
    if (fp_is_negative(&t)) fp_neg(&y, &y);
    fp_sub(&t, &p->Z, &y); fp_mul(&s, &deninv, &t); fp_abs(&s, &s);
    fp_to_bytes(out, &s);
}

static void elligator_map(ge_t *o, const fp_t *t0) {
    fp_t r, u, v, s, sp, c, n, w0, w1, w2, w3, t;
    fp_sqr(&r, t0); fp_mul(&r, &r, &FP_SQRT_M1);
    fp_add(&u, &r, &FP_ONE); fp_mul(&u, &u, &FP_ONE_MINUS_D_SQ);
    fp_mul(&t, &r, &FP_D); fp_add(&t, &t, &FP_ONE); fp_neg(&t, &t);      /* -1 - r*D */
    fp_add(&v, &r, &FP_D); fp_mul(&v, &v, &t);
    int was_square = fp_sqrt_ratio_m1(&s, &u, &v);
    fp_mul(&sp, &s, t0); fp_abs(&sp, &sp); fp_neg(&sp, &sp);
    fp_neg(&c, &FP_ONE);
    if (!was_square) { s = sp; c = r; }
    fp_sub(&t, &r, &FP_ONE); fp_mul(&n, &c, &t); fp_mul(&n, &n, &FP_D_MINUS_ONE_SQ); fp_sub(&n, &n, &v);
    fp_mul(&w0, &s, &v); fp_add(&w0, &w0, &w0);
    fp_mul(&w1, &n, &FP_SQRT_AD_MINUS_ONE);
    fp_sqr(&t, &s); fp_sub(&w2, &FP_ONE, &t); fp_add(&w3, &FP_ONE, &t);
    fp_mul(&o->X, &w0, &w3); fp_mul(&o->Y, &w2, &w1); fp_mul(&o->Z, &w1, &w3); fp_mul(&o->T, &w0, &w2);
}

void ge_from_uniform_bytes(ge_t *o, const uint8_t b[64]) {
    fp_t t0, t1; ge_t p0, p1;
    fp_from_bytes(&t0, b); fp_from_bytes(&t1, b + 32);     /* top bit masked by fp_from_bytes */
    elligator_map(&p0, &t0); elligator_map(&p1, &t1);
    ge_add(o, &p0, &p1);
}

int ge_eq(const ge_t *a, const ge_t *b) {
    /* X1*Y2 == Y1*X2  or  Y1*Y2 == X1*X2  (RFC 9496 4.3.3) */
    fp_t l, r;
    fp_mul(&l, &a->X, &b->Y); fp_mul(&r, &a->Y, &b->X);
    if (fp_eq(&l, &r)) return 1;
    fp_mul(&l, &a->Y, &b->Y); fp_mul(&r, &a->X, &b->X);
    return fp_eq(&l, &r);
}

static inline unsigned raw_bits(const uint64_t r[4], unsigned pos, unsigned w) {
    if (pos >= 256) return 0;
    unsigned limb = pos >> 6, off = pos & 63;
    uint64_t x = r[limb] >> off;
    if (off + w > 64 && limb < 3) x |= r[limb + 1] << (64 - off);
    return (unsigned)(x & ((1u << w) - 1));
}

void ge_scalarmul(ge_t *o, const ge_t *p, const fr_t *s) {
    uint64_t r[4]; fr_to_raw(r, s);
    ge_t tab[16]; ge_identity(&tab[0]); tab[1] = *p;
    for (int i = 2; i < 16; i++) ge_add(&tab[i], &tab[i - 1], p);
    ge_t acc; ge_identity(&acc);
    for (int w = 63; w >= 0; w--) {
        for (int k = 0; k < 4; k++) ge_dbl(&acc, &acc);
        unsigned d = raw_bits(r, 4 * w, 4);
        if (d) ge_add(&acc, &acc, &tab[d]);
    }
    *o = acc;
}

static void msm_straus(ge_t *o, const fr_t *s, const ge_t *P, size_t n) {
    ge_t *tab = (ge_t *)malloc(n * 16 * sizeof(ge_t));
    uint64_t (*raw)[4] = (uint64_t (*)[4])malloc(n * 32);
    for (size_t i = 0; i < n; i++) {
        fr_to_raw(raw[i], &s[i]);
        ge_identity(&tab[16 * i]); tab[16 * i + 1] = P[i];
        for (int k = 2; k < 16; k++) ge_add(&tab[16 * i + k], &tab[16 * i + k - 1], &P[i]);
    }
    ge_t acc; ge_identity(&acc);
    for (int w = 63; w >= 0; w--) {
        for (int k = 0; k < 4; k++) ge_dbl(&acc, &acc);
        for (size_t i = 0; i < n; i++) { unsigned d = raw_bits(raw[i], 4 * w, 4); if (d) ge_add(&acc, &acc, &tab[16 * i + d]); }
    }
    *o = acc; free(tab); free(raw);
}

static void msm_pippenger(ge_t *o, const fr_t *s, const ge_t *P, size_t n) {
    unsigned c = n < 256 ? 5 : n < 2048 ? 7 : n < 16384 ? 9 : 11;
    unsigned nwin = (253 + c) / c + 1;            /* one spare window for the signed-digit carry */
    size_t nb = (size_t)1 << (c - 1);
    int16_t *dig = (int16_t *)malloc(n * nwin * sizeof(int16_t));
    for (size_t i = 0; i < n; i++) {
        uint64_t r[4]; fr_to_raw(r, &s[i]);
        int carry = 0;
        for (unsigned w = 0; w < nwin; w++) {
            int d = (int)raw_bits(r, w * c, c) + carry;
            carry = 0;
            if (d > (int)nb) { d -= (1 << c); carry = 1; }
            dig[i * nwin + w] = (int16_t)d;
        }
    }
    ge_t *bucket = (ge_t *)malloc(nb * sizeof(ge_t));
    ge_t acc; ge_identity(&acc);
    for (int w = (int)nwin - 1; w >= 0; w--) {
        for (unsigned k = 0; k < c; k++) ge_dbl(&acc, &acc);
        for (size_t b = 0; b < nb; b++) ge_identity(&bucket[b]);
        int any = 0;
        for (size_t i = 0; i < n; i++) {
            int d = dig[i * nwin + w];
            if (d > 0) { ge_add(&bucket[d - 1], &bucket[d - 1], &P[i]); any = 1; }
            else if (d < 0) { ge_sub(&bucket[-d - 1], &bucket[-d - 1], &P[i]); any = 1; }
        }
        if (!any) continue;
        ge_t run, sum; ge_identity(&run); ge_identity(&sum);
        for (size_t b = nb; b-- > 0;) { ge_add(&run, &run, &bucket[b]); ge_add(&sum, &sum, &run); }
        ge_add(&acc, &acc, &sum);
    }
    *o = acc; free(bucket); free(dig);
}

void ge_msm(ge_t *o, const fr_t *s, const ge_t *P, size_t n) {
    if (n == 0) { ge_identity(o); return; }
    if (n < 48) msm_straus(o, s, P, n); else msm_pippenger(o, s, P, n);
}
