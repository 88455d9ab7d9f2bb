/*
 * ORACLE — test infrastructure only (see merlin.h).
 */
#include "merlin.h"
#include <string.h>
#include <assert.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void keccak_f1600(uint64_t st[25]) {
    uint64_t bc[5], t;
    for (int r = 0; r < 24; r++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) { t = bc[(i + 4) % 5] ^ ROL(bc[(i + 1) % 5], 1); for (int j = 0; j < 25; j += 5) st[j + i] ^= t; }
        t = st[1];
        for (int i = 0; i < 24; i++) { int j = PILN[i]; bc[0] = st[j]; st[j] = ROL(t, ROTC[i]); t = bc[0]; }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[r];
    }
}

/* byte access into the little-endian lane state (host is little-endian x86-64) */
static void sponge_absorb(uint64_t st[25], size_t *pos, size_t rate, const uint8_t *in, size_t n) {
    uint8_t *b = (uint8_t *)st;
    for (size_t i = 0; i < n; i++) { b[(*pos)++] ^= in[i]; if (*pos == rate) { keccak_f1600(st); *pos = 0; } }
}
static void sponge_finish(uint64_t st[25], size_t pos, size_t rate, uint8_t dom) {
    uint8_t *b = (uint8_t *)st; b[pos] ^= dom; b[rate - 1] ^= 0x80; keccak_f1600(st);
}

void shake256_init(shake256_t *s) { memset(s, 0, sizeof *s); }
void shake256_absorb(shake256_t *s, const uint8_t *in, size_t n) { assert(!s->squeezing); sponge_absorb(s->st, &s->pos, 136, in, n); }
void shake256_squeeze(shake256_t *s, uint8_t *out, size_t n) {
    if (!s->squeezing) { sponge_finish(s->st, s->pos, 136, 0x1f); s->pos = 0; s->squeezing = 1; }
    uint8_t *b = (uint8_t *)s->st;
    for (size_t i = 0; i < n; i++) { if (s->pos == 136) { keccak_f1600(s->st); s->pos = 0; } out[i] = b[s->pos++]; }
}
void sha3_256(uint8_t out[32], const uint8_t *in, size_t n) {
    uint64_t st[25]; size_t pos = 0; memset(st, 0, sizeof st);
    sponge_absorb(st, &pos, 136, in, n); sponge_finish(st, pos, 136, 0x06);
    memcpy(out, st, 32);
}

/* ---- STROBE-128, the subset Merlin uses [RECALL merlin/src/strobe.rs] ---- */
#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void strobe_run_f(strobe_t *s) {
    s->st[s->pos] ^= s->pos_begin;
    s->st[s->pos + 1] ^= 0x04;
    s->st[STROBE_R + 1] ^= 0x80;
    uint64_t lanes[25]; memcpy(lanes, s->st, 200); keccak_f1600(lanes); memcpy(s->st, lanes, 200);
    s->pos = 0; s->pos_begin = 0;
}
static void strobe_absorb(strobe_t *s, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { s->st[s->pos++] ^= d[i]; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void strobe_overwrite(strobe_t *s, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { s->st[s->pos++] = d[i]; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void strobe_squeeze(strobe_t *s, uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) { d[i] = s->st[s->pos]; s->st[s->pos++] = 0; if (s->pos == STROBE_R) strobe_run_f(s); }
}
static void strobe_begin_op(strobe_t *s, uint8_t flags, int more) {
    if (more) { assert(flags == s->cur_flags); return; }
    assert(!(flags & FLAG_T));
    uint8_t old_begin = s->pos_begin;
    s->pos_begin = s->pos + 1;
    s->cur_flags = flags;
    uint8_t hdr[2] = {old_begin, flags};
    strobe_absorb(s, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && s->pos != 0) strobe_run_f(s);
}
void strobe_init(strobe_t *s, const uint8_t *label, size_t n) {
    memset(s, 0, sizeof *s);
    static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(s->st, hdr, 6); memcpy(s->st + 6, "STROBEv1.0.2", 12);
    uint64_t lanes[25]; memcpy(lanes, s->st, 200); keccak_f1600(lanes); memcpy(s->st, lanes, 200);
    strobe_meta_ad(s, label, n, 0);
}
void strobe_meta_ad(strobe_t *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_M | FLAG_A, more); strobe_absorb(s, d, n); }
void strobe_ad(strobe_t *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_A, more); strobe_absorb(s, d, n); }
void strobe_prf(strobe_t *s, uint8_t *out, size_t n, int more) { strobe_begin_op(s, FLAG_I | FLAG_A | FLAG_C, more); strobe_squeeze(s, out, n); }
void strobe_key(strobe_t *s, const uint8_t *d, size_t n, int more) { strobe_begin_op(s, FLAG_A | FLAG_C, more); strobe_overwrite(s, d, n); }

/* ---- Merlin v1.0 [RECALL merlin/src/transcript.rs] ---- */
static void le32(uint8_t b[4], size_t n) { b[0] = (uint8_t)n; b[1] = (uint8_t)(n >> 8); b[2] = (uint8_t)(n >> 16); b[3] = (uint8_t)(n >> 24); }
void tr_append(transcript_t *t, const char *label, const uint8_t *msg, size_t n) {
    uint8_t len[4]; le32(len, n);
    strobe_meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(&t->s, len, 4, 1);
    strobe_ad(&t->s, msg, n, 0);
}
void tr_init(transcript_t *t, const char *label, size_t n) {
    strobe_init(&t->s, (const uint8_t *)"Merlin v1.0", 11);
    uint8_t len[4]; le32(len, n);
    strobe_meta_ad(&t->s, (const uint8_t *)"dom-sep", 7, 0);
    strobe_meta_ad(&t->s, len, 4, 1);
    strobe_ad(&t->s, (const uint8_t *)label, n, 0);
}
void tr_challenge_bytes(transcript_t *t, const char *label, uint8_t *out, size_t n) {
    uint8_t len[4]; le32(len, n);
    strobe_meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(&t->s, len, 4, 1);
    strobe_prf(&t->s, out, n, 0);
}

void tr_protocol_name(transcript_t *t, const char *name) { tr_append(t, "protocol-name", (const uint8_t *)name, strlen(name)); }
void tr_append_scalar(transcript_t *t, const char *label, const fr_t *s) { uint8_t b[32]; fr_to_bytes(b, s); tr_append(t, label, b, 32); }
void tr_append_point(transcript_t *t, const char *label, const uint8_t p[32]) { tr_append(t, label, p, 32); }
void tr_append_scalars(transcript_t *t, const char *label, const fr_t *s, size_t n) {
    tr_append(t, label, (const uint8_t *)"begin_append_vector", 19);
    for (size_t i = 0; i < n; i++) tr_append_scalar(t, label, &s[i]);
    tr_append(t, label, (const uint8_t *)"end_append_vector", 17);
}
void tr_challenge_scalar(transcript_t *t, const char *label, fr_t *o) { uint8_t b[64]; tr_challenge_bytes(t, label, b, 64); fr_from_bytes_wide(o, b); }
void tr_challenge_vector(transcript_t *t, const char *label, fr_t *o, size_t n) { for (size_t i = 0; i < n; i++) tr_challenge_scalar(t, label, &o[i]); }
