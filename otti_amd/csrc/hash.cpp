// See hash.h.  FIPS 202 Keccak-f[1600]; STROBE-128 per merlin's strobe.rs subset; Merlin v1.0 framing.
#include "hash.h"
#include <algorithm>
#include <stdexcept>
#include <string>
#include <stdio.h>

namespace otti {

static inline uint64_t rotl(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

void keccak_f1600(uint64_t A[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    // rotation offsets r[x][y] indexed as [x + 5*y]
    static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; round++) {
        uint64_t C[5], D[5], B[25];
        for (int x = 0; x < 5; x++) C[x] = A[x] ^ A[x + 5] ^ A[x + 10] ^ A[x + 15] ^ A[x + 20];
        for (int x = 0; x < 5; x++) D[x] = C[(x + 4) % 5] ^ rotl(C[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) A[i] ^= D[i % 5];
        // rho + pi: B[y, 2x+3y] = rot(A[x,y], r[x,y])
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) B[y + 5 * ((2 * x + 3 * y) % 5)] = rotl(A[x + 5 * y], RHO[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) A[x + 5 * y] = B[x + 5 * y] ^ (~B[(x + 1) % 5 + 5 * y] & B[(x + 2) % 5 + 5 * y]);
        A[0] ^= RC[round];
    }
}

void Shake256::absorb(const void *data, size_t n) {
    if (squeezing_) throw std::logic_error("Shake256: absorb after squeeze");
    const uint8_t *in = (const uint8_t *)data; uint8_t *b = (uint8_t *)st_;
    for (size_t i = 0; i < n; i++) { b[pos_++] ^= in[i]; if (pos_ == 136) { keccak_f1600(st_); pos_ = 0; } }
}
void Shake256::squeeze(void *out, size_t n) {
    uint8_t *b = (uint8_t *)st_, *o = (uint8_t *)out;
    if (!squeezing_) { b[pos_] ^= 0x1f; b[135] ^= 0x80; keccak_f1600(st_); pos_ = 0; squeezing_ = true; }
    while (n) {                                                // generator derivation squeezes 64 bytes per point: whole blocks at a time
        if (pos_ == 136) { keccak_f1600(st_); pos_ = 0; }
        const size_t take = std::min<size_t>(n, 136 - pos_);
        memcpy(o, b + pos_, take); pos_ += take; o += take; n -= take;
    }
}

namespace {
constexpr int kRate = 166;
constexpr uint8_t FI = 1, FA = 2, FC = 4, FT = 8, FM = 16, FK = 32;
}

Strobe128::Strobe128(const char *protocol_label) {
    memset(st_, 0, sizeof st_);
    const uint8_t hdr[6] = {1, kRate + 2, 1, 0, 1, 96};
    memcpy(st_, hdr, 6); memcpy(st_ + 6, "STROBEv1.0.2", 12);
    keccak_f1600((uint64_t *)st_);
    meta_ad(protocol_label, strlen(protocol_label), false);
}
void Strobe128::run_f() {
    st_[pos_] ^= pos_begin_; st_[pos_ + 1] ^= 0x04; st_[kRate + 1] ^= 0x80;
    keccak_f1600((uint64_t *)st_);
    pos_ = 0; pos_begin_ = 0;
}
// the duplex operations in chunks up to the rate boundary (a 2^20 proof absorbs ~100 KB: a thousand commitments, then ~10 messages per round)
static inline void xor_bytes(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t a, b; memcpy(&a, dst + i, 8); memcpy(&b, src + i, 8); a ^= b; memcpy(dst + i, &a, 8); }
    for (; i < n; i++) dst[i] ^= src[i];
}
void Strobe128::absorb(const uint8_t *d, size_t n) {
    while (n) {
        const size_t take = std::min<size_t>(n, (size_t)kRate - pos_);
        xor_bytes(st_ + pos_, d, take);
        pos_ = (uint8_t)(pos_ + take); d += take; n -= take;
        if (pos_ == kRate) run_f();
    }
}
void Strobe128::overwrite(const uint8_t *d, size_t n) {
    while (n) {
        const size_t take = std::min<size_t>(n, (size_t)kRate - pos_);
        memcpy(st_ + pos_, d, take);
        pos_ = (uint8_t)(pos_ + take); d += take; n -= take;
        if (pos_ == kRate) run_f();
    }
}
void Strobe128::squeeze(uint8_t *d, size_t n) {
    while (n) {
        const size_t take = std::min<size_t>(n, (size_t)kRate - pos_);
        memcpy(d, st_ + pos_, take); memset(st_ + pos_, 0, take);
        pos_ = (uint8_t)(pos_ + take); d += take; n -= take;
        if (pos_ == kRate) run_f();
    }
}
void Strobe128::begin_op(uint8_t flags, bool more) {
    if (more) { if (flags != cur_flags_) throw std::logic_error("strobe: continued op with different flags"); return; }
    if (flags & FT) throw std::logic_error("strobe: transport ops unsupported");
    uint8_t old_begin = pos_begin_;
    pos_begin_ = pos_ + 1; cur_flags_ = flags;
    uint8_t hdr[2] = {old_begin, flags};
    absorb(hdr, 2);
    if ((flags & (FC | FK)) && pos_ != 0) run_f();
}
void Strobe128::meta_ad(const void *d, size_t n, bool more) { begin_op(FM | FA, more); absorb((const uint8_t *)d, n); }
void Strobe128::ad(const void *d, size_t n, bool more) { begin_op(FA, more); absorb((const uint8_t *)d, n); }
void Strobe128::prf(void *out, size_t n, bool more) { begin_op(FI | FA | FC, more); squeeze((uint8_t *)out, n); }
void Strobe128::key(const void *d, size_t n, bool more) { begin_op(FA | FC, more); overwrite((const uint8_t *)d, n); }

// framing of one message: [old_begin, M|A] label len4 [begin of the first op, flags2]; returns its length.  p0 = position before the message.
static inline size_t merlin_frame(uint8_t *b, uint8_t old_begin, uint8_t p0, const char *label, size_t L, size_t n, uint8_t flags2) {
    b[0] = old_begin; b[1] = FM | FA;
    memcpy(b + 2, label, L);
    b[2 + L] = (uint8_t)n; b[3 + L] = (uint8_t)(n >> 8); b[4 + L] = (uint8_t)(n >> 16); b[5 + L] = (uint8_t)(n >> 24);
    b[6 + L] = (uint8_t)(p0 + 1); b[7 + L] = flags2;
    return 8 + L;
}
void Strobe128::merlin_append(const char *label, size_t L, const void *msg, size_t n) {
    const size_t total = 8 + L + n;
    if (L > 64 || n > 64 || (size_t)pos_ + total >= (size_t)kRate) {       // near the end of the block (or an unusually long message): the separate operations
        uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
        meta_ad(label, L, false); meta_ad(len, 4, true); ad(msg, n, false);
        return;
    }
    uint8_t b[8 + 64 + 64];
    const uint8_t p0 = pos_;
    const size_t f = merlin_frame(b, pos_begin_, p0, label, L, n, FA);
    memcpy(b + f, msg, n);
    xor_bytes(st_ + p0, b, total);
    pos_ = (uint8_t)(p0 + total); pos_begin_ = (uint8_t)(p0 + 6 + L + 1); cur_flags_ = FA;     // the second operation began after header, label and length
}
void Strobe128::merlin_challenge(const char *label, size_t L, void *out, size_t n) {
    if (L > 64 || (size_t)pos_ + 8 + L >= (size_t)kRate) {
        uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
        meta_ad(label, L, false); meta_ad(len, 4, true); prf(out, n, false);
        return;
    }
    uint8_t b[8 + 64];
    const uint8_t p0 = pos_;
    const size_t f = merlin_frame(b, pos_begin_, p0, label, L, n, FI | FA | FC);
    xor_bytes(st_ + p0, b, f);
    pos_ = (uint8_t)(p0 + f); pos_begin_ = (uint8_t)(p0 + 6 + L + 1); cur_flags_ = FI | FA | FC;
    run_f();                                                                // a C operation starts on a fresh block (pos_ != 0 here)
    squeeze((uint8_t *)out, n);
}

Transcript::Transcript(const void *label, size_t n) : s_("Merlin v1.0") { append_message("dom-sep", label, n); }
void Transcript::append_message(const char *label, const void *msg, size_t n) { s_.merlin_append(label, strlen(label), msg, n); }
void Transcript::challenge_bytes(const char *label, void *out, size_t n) { s_.merlin_challenge(label, strlen(label), out, n); }
void Transcript::append_scalars(const char *label, const Fr *s, size_t n) {
    append_message(label, "begin_append_vector", 19);
    for (size_t i = 0; i < n; i++) append_scalar(label, s[i]);
    append_message(label, "end_append_vector", 17);
}

RandomTape::RandomTape(const uint8_t seed32[32]) : t_("proof", 5) {
    uint8_t w[64]; memset(w, 0, sizeof w);
    if (seed32) memcpy(w, seed32, 32);
    else {
        FILE *f = fopen("/dev/urandom", "rb");
        if (!f || fread(w, 1, 64, f) != 64) { if (f) fclose(f); throw std::runtime_error("RandomTape: no OS entropy"); }
        fclose(f);
    }
    t_.append_scalar("init_randomness", fr_from_bytes_wide(w));
}

Fr RandomTape::random_scalar(const char *label) {
    if (head_ < queue_.size()) {
        if (strcmp(queue_[head_].first, label) != 0) throw std::logic_error(std::string("RandomTape: prefetched label '") + queue_[head_].first + "' but '" + label + "' was asked for");
        return queue_[head_++].second;
    }
    return t_.challenge_scalar(label);
}
Fr RandomTape::Cursor::random_scalar(const char *label) {
    if (pos_ < t_.head_ || pos_ >= t_.queue_.size()) throw std::logic_error("RandomTape::Cursor: read outside the prefetched, unconsumed part of the tape");
    if (strcmp(t_.queue_[pos_].first, label) != 0) throw std::logic_error(std::string("RandomTape::Cursor: prefetched label '") + t_.queue_[pos_].first + "' but '" + label + "' was asked for");
    return t_.queue_[pos_++].second;
}
void RandomTape::prefetch(const std::vector<std::pair<const char *, size_t>> &schedule) {
    if (head_ != queue_.size()) throw std::logic_error("RandomTape: prefetch with unread values pending");
    queue_.clear(); head_ = 0;
    for (auto &e : schedule) for (size_t i = 0; i < e.second; i++) queue_.emplace_back(e.first, t_.challenge_scalar(e.first));
}

}  // namespace otti
