/*
 * ORACLE — test infrastructure only (see fr.h).  ristretto255 group (RFC 9496) on edwards25519 extended coordinates.
 * Restates what upstream libspartan `src/group.rs` [RECALL] takes from curve25519-dalek 3.x:
 * RistrettoPoint::{from_uniform_bytes, compress, decompress, +, -, *}, VartimeMultiscalarMul.
 */
#ifndef OTTI_ORACLE_RISTRETTO_H
#define OTTI_ORACLE_RISTRETTO_H
#include "fp.h"
#include "fr.h"
#include <stddef.h>

typedef struct { fp_t X, Y, Z, T; } ge_t;                 /* extended twisted Edwards, a = -1 */

extern const uint8_t RISTRETTO_BASEPOINT_COMPRESSED[32];

void ge_identity(ge_t *o);
void ge_add(ge_t *o, const ge_t *a, const ge_t *b);
void ge_sub(ge_t *o, const ge_t *a, const ge_t *b);
void ge_neg(ge_t *o, const ge_t *a);
void ge_dbl(ge_t *o, const ge_t *a);
void ge_scalarmul(ge_t *o, const ge_t *p, const fr_t *s);
int  ge_decode(ge_t *o, const uint8_t b[32]);             /* 1 ok, 0 fail (DecompressionError) */
void ge_encode(uint8_t b[32], const ge_t *p);
void ge_from_uniform_bytes(ge_t *o, const uint8_t b[64]); /* one-way map, RFC 9496 4.3.4 */
int  ge_eq(const ge_t *a, const ge_t *b);                 /* ristretto equality */
/* sum_i s[i]*P[i]; Pippenger for n >= 64, double-and-add below (result is method independent) */
void ge_msm(ge_t *o, const fr_t *s, const ge_t *P, size_t n);

#endif
