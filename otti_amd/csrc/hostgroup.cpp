#include "hostgroup.h"
#include <stdlib.h>
#include <algorithm>

namespace otti {

void host_batch_invert(Fp *x, size_t n) {
    if (!n) return;
    std::vector<Fp> pre(n);
    Fp acc = fp_one();
    for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = fp_mul(acc, x[i]); }
    acc = fp_inv(acc);
    for (size_t i = n; i-- > 0;) { Fp t = fp_mul(acc, pre[i]); acc = fp_mul(acc, x[i]); x[i] = t; }
}

void scalar_digits(const Fr &s, int c, int nwin, int *digits) {
    Fr raw = fr_to_raw(s);
    int carry = 0, half = 1 << (c - 1);
    for (int w = 0; w < nwin; w++) {
        int d = scalar_window(raw.v, w * c, c) + carry;
        carry = 0;
        if (d > half) { d -= (1 << c); carry = 1; }
        digits[w] = d;
    }
}

// Variable-base work of the verifiers, in the five-limb field of hostfast.h (the generic 4 x u64 code is ~3x slower per point operation;
// a NIZK verification makes 82 scalar multiplications of proof points on its sequential path, a SNARK verification six MSMs of 2-4 k points)
// the multiples 1 .. 8 of a point in cached form, four-lane layout (hostifma.h)
static void multiples8_ifma(Niels4 *out, PtFe m) {
    const CachedFe c1 = ptfe_cache(m);
    out[0] = cached4_from(c1);
    for (int i = 1; i < 8; i++) { ptfe_add_cached(m, c1, false); out[i] = cached4_from(ptfe_cache(m)); }
}
Pt host_scalarmul(const Pt &p, const Fr &s) {
    // signed 4-bit windows, table of 1..8 multiples in cached form
    int dig[64]; scalar_digits(s, 4, 64, dig);
    if (host_ifma_available()) {                                // doublings and additions as two 4-way products each (hostifma.h)
        Niels4 tab4[8]; multiples8_ifma(tab4, ptfe_from(p));
        PtFe acc; ifma_straus(acc, tab4, 1, dig, 64);
        return ptfe_to(acc);
    }
    CachedFe tab[8]; PtFe m = ptfe_from(p); const CachedFe c1 = ptfe_cache(m);
    tab[0] = c1;
    for (int i = 1; i < 8; i++) { ptfe_add_cached(m, c1, false); tab[i] = ptfe_cache(m); }
    PtFe acc = ptfe_identity();
    for (int w = 63; w >= 0; w--) {
        for (int k = 0; k < 4; k++) ptfe_dbl(acc);
        if (dig[w]) ptfe_add_cached(acc, tab[(dig[w] > 0 ? dig[w] : -dig[w]) - 1], dig[w] < 0);
    }
    return ptfe_to(acc);
}

void split_table_build(SplitTable &T, const Pt &p) {
    PtFe base = ptfe_from(p);
    T.ifma = host_ifma_available();
    if (T.ifma) {
        for (int q = 0; q < 4; q++) { if (q) ifma_dbl_n(base, 64); multiples8_ifma(T.tab4[q], base); }
        return;
    }
    for (int q = 0; q < 4; q++) {
        if (q) for (int k = 0; k < 64; k++) ptfe_dbl(base);
        PtFe m = base; const CachedFe c1 = ptfe_cache(m);
        T.tab[q][0] = c1;
        for (int i = 1; i < 8; i++) { ptfe_add_cached(m, c1, false); T.tab[q][i] = ptfe_cache(m); }
    }
}
Pt split_table_mul(const SplitTable &T, const Fr &s) {
    int dig[64]; scalar_digits(s, 4, 64, dig);               // s = sum_q 2^(64 q) sum_{j < 16} dig[16 q + j] 16^j
    if (T.ifma) { PtFe acc; ifma_straus(acc, &T.tab4[0][0], 4, dig, 16); return ptfe_to(acc); }   // table q, window j: digit dig[16 q + j]
    PtFe acc = ptfe_identity();
    for (int j = 15; j >= 0; j--) {
        if (j != 15) for (int k = 0; k < 4; k++) ptfe_dbl(acc);
        for (int q = 0; q < 4; q++) {
            const int d = dig[16 * q + j];
            if (d) ptfe_add_cached(acc, T.tab[q][(d > 0 ? d : -d) - 1], d < 0);
        }
    }
    return ptfe_to(acc);
}

Pt host_msm(const Fr *s, const Pt *P, size_t n) {
    if (n == 0) return pt_identity();
    if (n < 24 && host_ifma_available()) {
        std::vector<Niels4> tab4(n * 8); std::vector<int> dig(n * 64);
        for (size_t i = 0; i < n; i++) { multiples8_ifma(&tab4[8 * i], ptfe_from(P[i])); scalar_digits(s[i], 4, 64, &dig[64 * i]); }
        PtFe acc; ifma_straus(acc, tab4.data(), (int)n, dig.data(), 64);
        return ptfe_to(acc);
    }
    if (n < 24) {
        // Straus: shared doublings, per-point 4-bit signed tables
        std::vector<CachedFe> tab(n * 8); std::vector<int> dig(n * 64);
        for (size_t i = 0; i < n; i++) {
            PtFe m = ptfe_from(P[i]); const CachedFe c1 = ptfe_cache(m);
            tab[8 * i] = c1;
            for (int k = 1; k < 8; k++) { ptfe_add_cached(m, c1, false); tab[8 * i + k] = ptfe_cache(m); }
            scalar_digits(s[i], 4, 64, &dig[64 * i]);
        }
        PtFe acc = ptfe_identity();
        for (int w = 63; w >= 0; w--) {
            for (int k = 0; k < 4; k++) ptfe_dbl(acc);
            for (size_t i = 0; i < n; i++) {
                const int d = dig[64 * i + w];
                if (d) ptfe_add_cached(acc, tab[8 * i + (d > 0 ? d : -d) - 1], d < 0);
            }
        }
        return ptfe_to(acc);
    }
    // bucket method, signed digits
    int c = n < 128 ? 5 : n < 1024 ? 7 : n < 8192 ? 9 : 12;
    int nwin = 253 / c + 1; size_t nb = (size_t)1 << (c - 1);
    std::vector<int> dig(n * nwin);
    std::vector<CachedFe> pc(n);
    for (size_t i = 0; i < n; i++) { scalar_digits(s[i], c, nwin, &dig[i * nwin]); pc[i] = ptfe_cache(ptfe_from(P[i])); }
    std::vector<PtFe> bucket(nb); std::vector<uint8_t> used(nb);
    PtFe acc = ptfe_identity();
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) ptfe_dbl(acc);
        std::fill(used.begin(), used.end(), 0);
        for (size_t i = 0; i < n; i++) {
            const int d = dig[i * nwin + w];
            if (!d) continue;
            const size_t bi = (size_t)(d > 0 ? d : -d) - 1;
            if (!used[bi]) { bucket[bi] = ptfe_identity(); used[bi] = 1; }
            ptfe_add_cached(bucket[bi], pc[i], d < 0);
        }
        PtFe run = ptfe_identity(), sum = ptfe_identity(); bool any = false;
        for (size_t bkt = nb; bkt-- > 0;) {
            if (used[bkt]) { ptfe_add(run, bucket[bkt]); any = true; }
            if (any) ptfe_add(sum, run);
        }
        ptfe_add(acc, sum);
    }
    return ptfe_to(acc);
}

void FixedBaseTable::build(const Pt &base) {
    std::vector<Pt> ext((size_t)kHostWindows * kHostWinEntries);
    Pt b = base;
    for (int w = 0; w < kHostWindows; w++) {
        Pt *row = &ext[(size_t)w * kHostWinEntries];
        row[0] = b;
        for (int d = 1; d < kHostWinEntries; d++) row[d] = pt_add(row[d - 1], b);
        b = pt_dbl(row[kHostWinEntries - 1]);                 // 2 * 128 * b = 2^8 * b
    }
    std::vector<Fp> z(ext.size());
    for (size_t i = 0; i < ext.size(); i++) z[i] = ext[i].Z;
    host_batch_invert(z.data(), z.size());
    t.resize(ext.size());
    for (size_t i = 0; i < ext.size(); i++) t[i] = nielsfe_from(pt_to_niels(ext[i], z[i]));
    if (host_ifma_available()) { t4.resize(t.size()); for (size_t i = 0; i < t.size(); i++) t4[i] = niels4_from(t[i]); }
}

void FixedBaseTable::accumulate(PtFe &acc, const Fr &s) const {
    int dig[kHostWindows]; scalar_digits(s, kHostWinBits, kHostWindows, dig);
    if (!t4.empty()) { ifma_accumulate(acc, t4.data(), dig, 0, kHostWindows); return; }     // two 4-way products per addition instead of seven scalar ones
    for (int w = 0; w < kHostWindows; w++) {
        const int d = dig[w];
        if (d) ptfe_madd(acc, t[(size_t)w * kHostWinEntries + (d > 0 ? d : -d) - 1], d < 0);
    }
}
void FixedBaseTable::accumulate(Pt &acc, const Fr &s) const { PtFe a = ptfe_from(acc); accumulate(a, s); acc = ptfe_to(a); }

}  // namespace otti
