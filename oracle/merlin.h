/*
 * ORACLE — test infrastructure only (see fr.h).  Keccak-f[1600], SHAKE256, STROBE-128 (Merlin subset), Merlin v1.0.
 * Restates the reference's dependencies merlin 3.x / sha3 0.8 [RECALL; Cargo deps of the absent Spartan/ submodule]
 * per FIPS 202 and the STROBE / Merlin specifications; upstream glue: libspartan src/transcript.rs, src/random.rs.
 * Pinned by SURVEY.md App. B (hashlib SHA3/SHAKE equality, STROBE conformance PRFs, Merlin "test protocol").
 */
#ifndef OTTI_ORACLE_MERLIN_H
#define OTTI_ORACLE_MERLIN_H
#include <stdint.h>
#include <stddef.h>
#include "fr.h"

void keccak_f1600(uint64_t st[25]);

typedef struct { uint64_t st[25]; size_t pos; int squeezing; } shake256_t;
void shake256_init(shake256_t *s);
void shake256_absorb(shake256_t *s, const uint8_t *in, size_t n);
void shake256_squeeze(shake256_t *s, uint8_t *out, size_t n);
void sha3_256(uint8_t out[32], const uint8_t *in, size_t n);   /* for the Keccak known-answer test */

typedef struct { uint8_t st[200]; uint8_t pos, pos_begin, cur_flags; } strobe_t;
void strobe_init(strobe_t *s, const uint8_t *label, size_t n);
void strobe_meta_ad(strobe_t *s, const uint8_t *d, size_t n, int more);
void strobe_ad(strobe_t *s, const uint8_t *d, size_t n, int more);
void strobe_prf(strobe_t *s, uint8_t *out, size_t n, int more);
void strobe_key(strobe_t *s, const uint8_t *d, size_t n, int more);

typedef struct { strobe_t s; } transcript_t;
void tr_init(transcript_t *t, const char *label, size_t n);
void tr_append(transcript_t *t, const char *label, const uint8_t *msg, size_t n);
void tr_challenge_bytes(transcript_t *t, const char *label, uint8_t *out, size_t n);
/* libspartan ProofTranscript trait [RECALL src/transcript.rs] */
void tr_protocol_name(transcript_t *t, const char *name);
void tr_append_scalar(transcript_t *t, const char *label, const fr_t *s);
void tr_append_point(transcript_t *t, const char *label, const uint8_t p[32]);
void tr_append_scalars(transcript_t *t, const char *label, const fr_t *s, size_t n);
void tr_challenge_scalar(transcript_t *t, const char *label, fr_t *o);
void tr_challenge_vector(transcript_t *t, const char *label, fr_t *o, size_t n);

#endif
