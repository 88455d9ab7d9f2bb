// Four GF(2^255-19) multiplications at a time with AVX-512 IFMA (vpmadd52luq / vpmadd52huq on 256-bit vectors), for the host's
// fixed-base scalar multiplications on the sequential Fiat-Shamir path (hostgroup.h FixedBaseTable): a mixed point addition is two
// such 4-way products — (Y-X, Y+X, T, Z) x (y-x, y+x, 2dxy, 2), then (E, G, E, F) x (F, H, H, G) — instead of seven scalar ones,
// ~3x shorter (tools/roundbench).  Same group elements as hostfast.h / point.h: checked against them by otti_host_selftest.
// Used only when the CPU has the instructions (EPYC Zen 4/5, Xeon Ice Lake and later) and OTTI_HOST_IFMA is not 0; everything
// else in the library is built for x86-64-v3.
#pragma once
#include "hostfast.h"

namespace otti {

bool host_ifma_available();                                  // CPU feature test (cached) and the OTTI_HOST_IFMA switch

// a table entry in vector layout: limb k of (y-x, y+x, 2dxy, 2) side by side — 5 x 32 bytes
struct alignas(32) Niels4 { uint64_t v[5][4]; };
Niels4 niels4_from(const NielsFe &n);
// acc += sum over the 32 windows of +-table[w * 128 + |digit_w| - 1]  (digits from scalar_digits(s, 8, 32, .))
void ifma_accumulate(PtFe &acc, const Niels4 *table, const int *digits, int w0, int w1);
// one mixed addition (selftest)
void ifma_madd(PtFe &p, const Niels4 &q, bool negate);

// ---- the verifiers' variable-base work in the same form: a doubling is a 4-way squaring of (X, Y, Z, X+Y) and the same second
// product; an addition of a point kept in cached form (Y-X, Y+X, 2dT, 2Z — the layout of Niels4, fourth lane 2Z) is the mixed addition
// above with that lane in place of the constant.
Niels4 cached4_from(const CachedFe &c);
void ifma_dbl_n(PtFe &p, int n);                             // p = 2^n p
// Straus: acc = sum_i s_i P_i, where tabs[8 i + m - 1] = m P_i (m = 1 .. 8, cached form) and digs[i * nwin + w] is signed radix-16
// digit w of s_i (|digit| <= 8); four doublings per window, windows from nwin - 1 down to 0
void ifma_straus(PtFe &acc, const Niels4 *tabs, int ntabs, const int *digs, int nwin);

}  // namespace otti
