/*
 * ORACLE — test infrastructure only.  CPU restatement of the Spartan NIZK hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
 * The product (otti_amd/) never includes, links or calls this code.
 *
 * PARITY UNPINNED against the reference: the reference prover's source is an empty submodule
 * (/root/reference/.gitmodules:4-6, Spartan/ is an empty directory) and it ships no golden vectors.
 * This file restates the *published* arithmetic of upstream libspartan `src/scalar/ristretto255.rs`
 * (Scalar = GF(l), l = 2^252 + 27742317777372353535851937790883648493, Montgomery 4x u64, R = 2^256);
 * it is pinned by the known answers in SURVEY.md App. B and by libsodium cross-checks (tests/golden/).
 */
#ifndef OTTI_ORACLE_FR_H
#define OTTI_ORACLE_FR_H
#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t v[4]; } fr_t;   /* Montgomery form: value * 2^256 mod l, little-endian limbs */

extern const fr_t FR_ZERO, FR_ONE;

void fr_add(fr_t *o, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *o, const fr_t *a, const fr_t *b);
void fr_neg(fr_t *o, const fr_t *a);
void fr_mul(fr_t *o, const fr_t *a, const fr_t *b);
void fr_sqr(fr_t *o, const fr_t *a);
void fr_inv(fr_t *o, const fr_t *a);                 /* 0 -> 0 */
int  fr_eq(const fr_t *a, const fr_t *b);
int  fr_is_zero(const fr_t *a);
void fr_from_u64(fr_t *o, uint64_t x);
int  fr_from_bytes(fr_t *o, const uint8_t b[32]);    /* canonical LE; returns 0 if >= l (InvalidScalar) */
void fr_from_bytes_wide(fr_t *o, const uint8_t b[64]);/* 512-bit LE reduced mod l */
void fr_to_bytes(uint8_t b[32], const fr_t *a);      /* canonical LE */
void fr_to_raw(uint64_t o[4], const fr_t *a);        /* canonical integer limbs */
void fr_mont_bytes(uint8_t b[32], const fr_t *a);    /* Montgomery-form limbs LE (bincode of upstream Scalar([u64;4])) */
int  fr_from_mont_bytes(fr_t *o, const uint8_t b[32]);
void fr_batch_inv(fr_t *x, size_t n);                /* in place; zeros stay zero */

#endif
