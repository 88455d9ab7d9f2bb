// Host-side group helpers: variable-base MSM (verifier), fixed-base window tables (the prover's per-round O(1)
// Pedersen commitments, which sit on the sequential Fiat-Shamir critical path and are latency- not throughput-bound).
// Replaces upstream libspartan `src/commitments.rs` (MultiCommitGens, Commitments) on the host side [RECALL].
#pragma once
#include <vector>
#include <utility>
#include "point.h"
#include "hostfast.h"
#include "hostifma.h"

namespace otti {

Pt host_scalarmul(const Pt &p, const Fr &s);
Pt host_msm(const Fr *s, const Pt *P, size_t n);            // sum s[i] * P[i]
// A point that will be multiplied by scalars not known yet (a verifier's sum-check commitments: in the proof from the start, their
// scalars are transcript challenges): the multiples 1..8 of P, 2^64 P, 2^128 P, 2^192 P prepared ahead (192 doublings, off the
// sequential path), so that the multiplication itself is a four-way Straus walk of 64 doublings and at most 64 additions — a third
// of host_scalarmul's chain.
struct SplitTable { CachedFe tab[4][8]; Niels4 tab4[4][8]; bool ifma = false; };   // tab4: the same multiples in four-lane layout when the CPU has AVX-512 IFMA (then tab is not filled)
void split_table_build(SplitTable &T, const Pt &p);
Pt split_table_mul(const SplitTable &T, const Fr &s);
void host_batch_invert(Fp *x, size_t n);                    // in place, no zeros allowed

// signed radix-2^c digits (c <= 16) of a Montgomery-form scalar; nwin = floor(253/c) + 1
void scalar_digits(const Fr &s, int c, int nwin, int *digits);

constexpr int kHostWinBits = 8, kHostWindows = 32, kHostWinEntries = 128;   // 2^(c-1) entries per window
struct FixedBaseTable {
    std::vector<NielsFe> t;                                 // [w * 128 + (|d| - 1)] = |d| * 2^(8w) * B, five 51-bit limbs per coordinate (hostfast.h)
    std::vector<Niels4> t4;                                 // the same entries in four-lane layout when the CPU has AVX-512 IFMA (hostifma.h); empty otherwise
    void build(const Pt &base);
    void accumulate(Pt &acc, const Fr &s) const;            // acc += s * B  (32 mixed additions)
    void accumulate(PtFe &acc, const Fr &s) const;          // the same without converting the accumulator in and out
};

}  // namespace otti
