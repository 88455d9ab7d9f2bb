/*
 * ORACLE — test infrastructure only (see fr.h).  Base field GF(2^255-19), radix-2^51.
 * Restates what the reference reaches through curve25519-dalek 3.x `FieldElement` [RECALL; the crate is a
 * Cargo dependency of the absent Spartan/ submodule, not vendored under /root/reference]; semantics follow
 * RFC 9496 section 4.2 (SQRT_RATIO_M1, IS_NEGATIVE, CT_ABS).
 */
#ifndef OTTI_ORACLE_FP_H
#define OTTI_ORACLE_FP_H
#include <stdint.h>

typedef struct { uint64_t v[5]; } fp_t;   /* limbs < 2^52 between operations */

extern const fp_t FP_ZERO, FP_ONE, FP_D, FP_2D, FP_SQRT_M1, FP_SQRT_AD_MINUS_ONE, FP_INVSQRT_A_MINUS_D,
                  FP_ONE_MINUS_D_SQ, FP_D_MINUS_ONE_SQ;

void fp_add(fp_t *o, const fp_t *a, const fp_t *b);
void fp_sub(fp_t *o, const fp_t *a, const fp_t *b);
void fp_neg(fp_t *o, const fp_t *a);
void fp_mul(fp_t *o, const fp_t *a, const fp_t *b);
void fp_sqr(fp_t *o, const fp_t *a);
void fp_inv(fp_t *o, const fp_t *a);
void fp_pow22523(fp_t *o, const fp_t *a);           /* a^((p-5)/8) */
void fp_from_bytes(fp_t *o, const uint8_t b[32]);   /* top bit ignored, value reduced lazily */
void fp_to_bytes(uint8_t b[32], const fp_t *a);     /* canonical */
int  fp_is_canonical_bytes(const uint8_t b[32]);
int  fp_is_negative(const fp_t *a);                 /* LSB of canonical encoding */
int  fp_is_zero(const fp_t *a);
int  fp_eq(const fp_t *a, const fp_t *b);
void fp_abs(fp_t *o, const fp_t *a);
void fp_cmov(fp_t *o, const fp_t *a, int c);        /* o = c ? a : o */
int  fp_sqrt_ratio_m1(fp_t *r, const fp_t *u, const fp_t *v);  /* RFC 9496 4.2; returns was_square */

#endif
