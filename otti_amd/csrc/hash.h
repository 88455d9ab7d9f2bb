// Host-side Fiat-Shamir machinery: Keccak-f[1600], SHAKE256, STROBE-128 (Merlin subset), Merlin transcripts, and the
// libspartan ProofTranscript / RandomTape conventions on top of them.
// Replaces upstream libspartan `src/transcript.rs`, `src/random.rs` and the merlin / sha3 crates [RECALL].
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <string>
#include <vector>
#include <utility>
#include "field.h"

namespace otti {

void keccak_f1600(uint64_t s[25]);

class Shake256 {
public:
    Shake256() { memset(st_, 0, sizeof st_); }
    void absorb(const void *data, size_t n);
    void squeeze(void *out, size_t n);
private:
    uint64_t st_[25]; size_t pos_ = 0; bool squeezing_ = false;
};

class Strobe128 {
public:
    explicit Strobe128(const char *protocol_label);
    void meta_ad(const void *d, size_t n, bool more);
    void ad(const void *d, size_t n, bool more);
    void prf(void *out, size_t n, bool more);
    void key(const void *d, size_t n, bool more);
    // One whole Merlin message in a single pass over the state: meta_ad(label) ‖ meta_ad(len, more) ‖ ad(msg)  (append_message), or the
    // same framing followed by prf(out)  (challenge_bytes).  A proof makes ~25 of these per sum-check round, each 40-60 bytes: as
    // separate operations that is five small absorbs and two operation headers apiece, most of the transcript's time on the
    // sequential path.  Same bytes into the same positions as the separate operations (falls back to them at a rate boundary).
    void merlin_append(const char *label, size_t label_len, const void *msg, size_t n);
    void merlin_challenge(const char *label, size_t label_len, void *out, size_t n);
private:
    void run_f(); void absorb(const uint8_t *d, size_t n); void overwrite(const uint8_t *d, size_t n);
    void squeeze(uint8_t *d, size_t n); void begin_op(uint8_t flags, bool more);
    alignas(8) uint8_t st_[200]; uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
};

// merlin::Transcript + libspartan's ProofTranscript / AppendToTranscript traits
class Transcript {
public:
    Transcript(const void *label, size_t n);
    void append_message(const char *label, const void *msg, size_t n);
    void challenge_bytes(const char *label, void *out, size_t n);
    void append_protocol_name(const char *name) { append_message("protocol-name", name, strlen(name)); }
    void append_scalar(const char *label, const Fr &s) { uint8_t b[32]; fr_to_bytes(b, s); append_message(label, b, 32); }
    void append_point(const char *label, const uint8_t p[32]) { append_message(label, p, 32); }
    void append_scalars(const char *label, const Fr *s, size_t n);
    Fr challenge_scalar(const char *label) { uint8_t b[64]; challenge_bytes(label, b, 64); return fr_from_bytes_wide(b); }
    std::vector<Fr> challenge_vector(const char *label, size_t n) { std::vector<Fr> v(n); for (auto &x : v) x = challenge_scalar(label); return v; }
private:
    Strobe128 s_;
};

// libspartan RandomTape: a second transcript whose challenges are the prover's blinds.  Upstream seeds it from OsRng;
// here the caller supplies 32 bytes (NULL => OS entropy) so that a proof is a deterministic function of its inputs.
// anything that hands out the tape's scalars in order (the tape itself, or a read-ahead cursor over its prefetched queue)
class ScalarSource {
public:
    virtual ~ScalarSource() {}
    virtual Fr random_scalar(const char *label) = 0;
    std::vector<Fr> random_vector(const char *label, size_t n) { std::vector<Fr> v(n); for (auto &x : v) x = random_scalar(label); return v; }
};
class RandomTape : public ScalarSource {
public:
    explicit RandomTape(const uint8_t seed32[32]);
    Fr random_scalar(const char *label) override;
    // The tape is a transcript of its own: its outputs depend only on the seed and on the sequence of labels asked for.  The prover
    // knows that sequence in advance, so it draws everything in one go (while the device is busy with the witness commitment);
    // later random_scalar calls pop the queue and check that the label is the scheduled one.
    void prefetch(const std::vector<std::pair<const char *, size_t>> &schedule);
    // Reads prefetched values ahead of the protocol WITHOUT consuming them (label-checked like the real draws): lets the prover start
    // work that depends on the tape alone — e.g. the second sum-check's blinding commitments — long before the protocol gets there.
    class Cursor : public ScalarSource {
    public:
        Cursor(const RandomTape &t, size_t queue_index) : t_(t), pos_(queue_index) {}
        Fr random_scalar(const char *label) override;
    private:
        const RandomTape &t_; size_t pos_;
    };
private:
    Transcript t_;
    std::vector<std::pair<const char *, Fr>> queue_; size_t head_ = 0;
};

}  // namespace otti
