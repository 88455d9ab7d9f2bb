#include "hostgroup.h"
#include <stdlib.h>

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

Pt host_scalarmul(const Pt &p, const Fr &s) {
    // signed 4-bit windows, table of 1..8 multiples
    Pt tab[8]; tab[0] = p;
    for (int i = 1; i < 8; i++) tab[i] = pt_add(tab[i - 1], p);
    int dig[64]; scalar_digits(s, 4, 64, dig);
    Pt acc = pt_identity();
    for (int w = 63; w >= 0; w--) {
        for (int k = 0; k < 4; k++) acc = pt_dbl(acc);
        if (dig[w] > 0) acc = pt_add(acc, tab[dig[w] - 1]);
        else if (dig[w] < 0) acc = pt_sub(acc, tab[-dig[w] - 1]);
    }
    return acc;
}

Pt host_msm(const Fr *s, const Pt *P, size_t n) {
    if (n == 0) return pt_identity();
    if (n < 24) {
        // Straus: shared doublings, per-point 4-bit signed tables
        std::vector<Pt> tab(n * 8); std::vector<int> dig(n * 64);
        for (size_t i = 0; i < n; i++) {
            tab[8 * i] = P[i];
            for (int k = 1; k < 8; k++) tab[8 * i + k] = pt_add(tab[8 * i + k - 1], P[i]);
            scalar_digits(s[i], 4, 64, &dig[64 * i]);
        }
        Pt acc = pt_identity();
        for (int w = 63; w >= 0; w--) {
            for (int k = 0; k < 4; k++) acc = pt_dbl(acc);
            for (size_t i = 0; i < n; i++) {
                int d = dig[64 * i + w];
                if (d > 0) acc = pt_add(acc, tab[8 * i + d - 1]); else if (d < 0) acc = pt_sub(acc, tab[8 * i - d - 1]);
            }
        }
        return acc;
    }
    // bucket method, signed digits
    int c = n < 128 ? 5 : n < 1024 ? 7 : n < 8192 ? 9 : 12;
    int nwin = 253 / c + 1; size_t nb = (size_t)1 << (c - 1);
    std::vector<int> dig(n * nwin);
    for (size_t i = 0; i < n; i++) scalar_digits(s[i], c, nwin, &dig[i * nwin]);
    std::vector<Pt> bucket(nb);
    Pt acc = pt_identity();
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) acc = pt_dbl(acc);
        for (auto &b : bucket) b = pt_identity();
        for (size_t i = 0; i < n; i++) {
            int d = dig[i * nwin + w];
            if (d > 0) bucket[d - 1] = pt_add(bucket[d - 1], P[i]); else if (d < 0) bucket[-d - 1] = pt_sub(bucket[-d - 1], P[i]);
        }
        Pt run = pt_identity(), sum = pt_identity();
        for (size_t b = nb; b-- > 0;) { run = pt_add(run, bucket[b]); sum = pt_add(sum, run); }
        acc = pt_add(acc, sum);
    }
    return acc;
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
}

void FixedBaseTable::accumulate(PtFe &acc, const Fr &s) const {
    int dig[kHostWindows]; scalar_digits(s, kHostWinBits, kHostWindows, dig);
    for (int w = 0; w < kHostWindows; w++) {
        const int d = dig[w];
        if (d) ptfe_madd(acc, t[(size_t)w * kHostWinEntries + (d > 0 ? d : -d) - 1], d < 0);
    }
}
void FixedBaseTable::accumulate(Pt &acc, const Fr &s) const { PtFe a = ptfe_from(acc); accumulate(a, s); acc = ptfe_to(a); }

}  // namespace otti
