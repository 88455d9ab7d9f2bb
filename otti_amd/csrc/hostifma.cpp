// See hostifma.h.  Radix 2^51, five limbs per element, four elements per vector (one per 64-bit lane).  vpmadd52 multiplies the low
// 52 bits of its operands, so every operand of a product is kept below 2^52: products come out below 2^51 + 2^13 per limb (one
// parallel carry pass), sums and differences are brought back under 2^51 + 2^7 by the same pass before they are multiplied.
#include "hostifma.h"
#include <stdlib.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)       // host pass only (hipcc also walks this file for gfx950)
#include <immintrin.h>

namespace otti {

#define OTTI_IFMA __attribute__((target("avx2,avx512f,avx512vl,avx512ifma"), always_inline)) static inline
#define OTTI_IFMA_FN __attribute__((target("avx2,avx512f,avx512vl,avx512ifma")))

bool host_ifma_available() {
    static const bool on = [] {
        const char *e = getenv("OTTI_HOST_IFMA");
        if (e && e[0] == '0') return false;
        __builtin_cpu_init();
        return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512ifma");
    }();
    return on;
}

static inline void canon(Fe &a) {                            // every limb below 2^51 (limb 1 by at most one unit more)
    const uint64_t M = ((uint64_t)1 << 51) - 1;
    for (int pass = 0; pass < 2; pass++) {
        uint64_t c;
        c = a.v[0] >> 51; a.v[0] &= M; a.v[1] += c;
        c = a.v[1] >> 51; a.v[1] &= M; a.v[2] += c;
        c = a.v[2] >> 51; a.v[2] &= M; a.v[3] += c;
        c = a.v[3] >> 51; a.v[3] &= M; a.v[4] += c;
        c = a.v[4] >> 51; a.v[4] &= M; a.v[0] += 19 * c;
        c = a.v[0] >> 51; a.v[0] &= M; a.v[1] += c;
    }
}
Niels4 cached4_from(const CachedFe &c0) {
    CachedFe c = c0; canon(c.yplusx); canon(c.yminusx); canon(c.z2); canon(c.t2d);
    Niels4 r;
    for (int k = 0; k < 5; k++) { r.v[k][0] = c.yminusx.v[k]; r.v[k][1] = c.yplusx.v[k]; r.v[k][2] = c.t2d.v[k]; r.v[k][3] = c.z2.v[k]; }
    return r;
}
Niels4 niels4_from(const NielsFe &n0) {
    NielsFe n = n0; canon(n.yplusx); canon(n.yminusx); canon(n.xy2d);
    Niels4 r;
    for (int k = 0; k < 5; k++) { r.v[k][0] = n.yminusx.v[k]; r.v[k][1] = n.yplusx.v[k]; r.v[k][2] = n.xy2d.v[k]; r.v[k][3] = k == 0 ? 2 : 0; }
    return r;
}

namespace {
struct V5 { __m256i l[5]; };                                 // lanes: X, Y, T, Z of a point — or any four field elements

OTTI_IFMA __m256i mask51() { return _mm256_set1_epi64x(((long long)1 << 51) - 1); }
OTTI_IFMA __m256i times19(__m256i c) { return _mm256_add_epi64(_mm256_add_epi64(_mm256_slli_epi64(c, 4), _mm256_slli_epi64(c, 1)), c); }
// one parallel carry pass: limb k keeps its low 51 bits and takes the carry of limb k-1 (limb 0: 19 x the carry of limb 4)
OTTI_IFMA V5 vreduce(const V5 &a) {
    const __m256i m = mask51();
    __m256i c[5];
    for (int k = 0; k < 5; k++) c[k] = _mm256_srli_epi64(a.l[k], 51);
    V5 r;
    r.l[0] = _mm256_add_epi64(_mm256_and_si256(a.l[0], m), times19(c[4]));
    for (int k = 1; k < 5; k++) r.l[k] = _mm256_add_epi64(_mm256_and_si256(a.l[k], m), c[k - 1]);
    return r;
}
OTTI_IFMA V5 vadd(const V5 &a, const V5 &b) { V5 r; for (int k = 0; k < 5; k++) r.l[k] = _mm256_add_epi64(a.l[k], b.l[k]); return r; }
// 2p - b per limb (b below 2^52 - 38): never negative
OTTI_IFMA V5 vneg2p(const V5 &b) {
    V5 r;
    r.l[0] = _mm256_sub_epi64(_mm256_set1_epi64x(0xFFFFFFFFFFFDALL), b.l[0]);
    const __m256i p = _mm256_set1_epi64x(0xFFFFFFFFFFFFELL);
    for (int k = 1; k < 5; k++) r.l[k] = _mm256_sub_epi64(p, b.l[k]);
    return r;
}
// four products; operand limbs below 2^52, result limbs below 2^51 + 2^13
OTTI_IFMA V5 vmul(const V5 &a, const V5 &b) {
    const __m256i z = _mm256_setzero_si256();
    __m256i L[9], H[9];
    for (int k = 0; k < 9; k++) { L[k] = z; H[k] = z; }
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) {
            L[i + j] = _mm256_madd52lo_epu64(L[i + j], a.l[i], b.l[j]);
            H[i + j] = _mm256_madd52hi_epu64(H[i + j], a.l[i], b.l[j]);
        }
    // column k of weight 2^(51 k): L_k + 2 H_(k-1)  (the high halves start at bit 52 = 2 x 2^51); columns 5 .. 9 wrap with a factor 19
    __m256i C[10];
    C[0] = L[0];
    for (int k = 1; k < 9; k++) C[k] = _mm256_add_epi64(L[k], _mm256_slli_epi64(H[k - 1], 1));
    C[9] = _mm256_slli_epi64(H[8], 1);
    V5 r;
    for (int k = 0; k < 5; k++) r.l[k] = _mm256_add_epi64(C[k], times19(C[k + 5]));
    return vreduce(r);
}
OTTI_IFMA V5 vload_point(const PtFe &p) {
    V5 r;
    for (int k = 0; k < 5; k++) r.l[k] = _mm256_set_epi64x((long long)p.Z.v[k], (long long)p.T.v[k], (long long)p.Y.v[k], (long long)p.X.v[k]);
    return vreduce(vreduce(r));                              // host points may carry lazily reduced limbs (up to 2^54): two passes bring any of them under 2^51 + 2^7
}
OTTI_IFMA void vstore_point(PtFe &p, const V5 &a) {
    alignas(32) uint64_t t[5][4];
    for (int k = 0; k < 5; k++) _mm256_store_si256((__m256i *)t[k], a.l[k]);
    for (int k = 0; k < 5; k++) { p.X.v[k] = t[k][0]; p.Y.v[k] = t[k][1]; p.T.v[k] = t[k][2]; p.Z.v[k] = t[k][3]; }
}
// add-2008-hwcd-3 mixed addition as two 4-way products (lanes X, Y, T, Z in and out)
OTTI_IFMA V5 vmadd(const V5 &P, const Niels4 &q, bool negate) {
    const __m256i z = _mm256_setzero_si256();
    V5 A, B, Q;
    for (int k = 0; k < 5; k++) {
        A.l[k] = _mm256_permute4x64_epi64(P.l[k], 0xE5);                  // (Y, Y, T, Z)
        B.l[k] = _mm256_permute4x64_epi64(P.l[k], 0x00);                  // (X, X, X, X)
        Q.l[k] = _mm256_load_si256((const __m256i *)q.v[k]);              // (y-x, y+x, 2dxy, 2)
    }
    const V5 Bn = vneg2p(B);
    V5 U;
    for (int k = 0; k < 5; k++) {
        const __m256i t = _mm256_blend_epi32(_mm256_blend_epi32(z, Bn.l[k], 0x03), B.l[k], 0x0C);   // (2p - X, X, 0, 0)
        U.l[k] = _mm256_add_epi64(A.l[k], t);                             // (Y - X, Y + X, T, Z)
    }
    U = vreduce(U);
    if (negate) {                                                         // -Q = (y+x, y-x, -2dxy, 2)
        V5 S; for (int k = 0; k < 5; k++) S.l[k] = _mm256_permute4x64_epi64(Q.l[k], 0xE1);
        const V5 N = vneg2p(S);
        for (int k = 0; k < 5; k++) Q.l[k] = _mm256_blend_epi32(S.l[k], N.l[k], 0x30);
    }
    const V5 V = vmul(U, Q);                                              // (A, B, C, D)
    V5 S1, S2;
    for (int k = 0; k < 5; k++) { S1.l[k] = _mm256_permute4x64_epi64(V.l[k], 0xDD); S2.l[k] = _mm256_permute4x64_epi64(V.l[k], 0x88); }   // (B, D, B, D), (A, C, A, C)
    const V5 diff = vadd(S1, vneg2p(S2)), sum = vadd(S1, S2);             // (E, F, E, F), (H, G, H, G)
    V5 M1, M2;
    for (int k = 0; k < 5; k++) {
        M1.l[k] = _mm256_blend_epi32(diff.l[k], sum.l[k], 0x0C);          // (E, G, E, F)
        M2.l[k] = _mm256_blend_epi32(_mm256_permute4x64_epi64(sum.l[k], 0xE0), _mm256_permute4x64_epi64(diff.l[k], 0x01), 0x03);   // (F, H, H, G)
    }
    return vmul(vreduce(M1), vreduce(M2));                                // (E F, G H, E H, F G) = (X3, Y3, T3, Z3)
}
// 4p - b per limb (b below 2^53 - 76)
OTTI_IFMA V5 vneg4p(const V5 &b) {
    V5 r;
    r.l[0] = _mm256_sub_epi64(_mm256_set1_epi64x(0x1FFFFFFFFFFFB4LL), b.l[0]);
    const __m256i p = _mm256_set1_epi64x(0x1FFFFFFFFFFFFCLL);
    for (int k = 1; k < 5; k++) r.l[k] = _mm256_sub_epi64(p, b.l[k]);
    return r;
}
// dbl-2008-hwcd (a = -1) as a 4-way squaring and a 4-way product (lanes X, Y, T, Z in and out)
OTTI_IFMA V5 vdbl(const V5 &P) {
    const __m256i z = _mm256_setzero_si256();
    V5 S;
    for (int k = 0; k < 5; k++) {
        const __m256i a = _mm256_permute4x64_epi64(P.l[k], 0x34);                     // (X, Y, Z, X): 00 01 11 00 -> imm 0b00_11_01_00
        const __m256i b = _mm256_blend_epi32(z, _mm256_permute4x64_epi64(P.l[k], 0x55), 0xC0);   // (0, 0, 0, Y)
        S.l[k] = _mm256_add_epi64(a, b);                                              // (X, Y, Z, X + Y)
    }
    S = vreduce(S);
    const V5 Q = vmul(S, S);                                                          // (A, B, ZZ, W) = (X^2, Y^2, Z^2, (X + Y)^2)
    V5 a4, b4, zz4, w4;
    for (int k = 0; k < 5; k++) {
        a4.l[k] = _mm256_permute4x64_epi64(Q.l[k], 0x00); b4.l[k] = _mm256_permute4x64_epi64(Q.l[k], 0x55);
        zz4.l[k] = _mm256_permute4x64_epi64(Q.l[k], 0xAA); w4.l[k] = _mm256_permute4x64_epi64(Q.l[k], 0xFF);
    }
    const V5 na = vneg2p(a4), nb = vneg2p(b4);
    const V5 G = vadd(b4, na), H = vadd(na, nb);                                      // G = B - A, H = -A - B  (D = -A)
    const V5 E = vadd(w4, H);                                                         // (X + Y)^2 - A - B
    const V5 F = vadd(G, vneg4p(vadd(zz4, zz4)));                                     // G - 2 ZZ
    V5 M1, M2;
    for (int k = 0; k < 5; k++) {
        M1.l[k] = _mm256_blend_epi32(_mm256_blend_epi32(E.l[k], G.l[k], 0x0C), F.l[k], 0xC0);        // (E, G, E, F)
        M2.l[k] = _mm256_blend_epi32(_mm256_blend_epi32(H.l[k], F.l[k], 0x03), G.l[k], 0xC0);        // (F, H, H, G)
    }
    return vmul(vreduce(M1), vreduce(M2));                                            // (E F, G H, E H, F G) = (X3, Y3, T3, Z3)
}
OTTI_IFMA V5 videntity() {
    V5 r;
    r.l[0] = _mm256_set_epi64x(1, 0, 1, 0);                                           // lanes X, Y, T, Z = 0, 1, 0, 1
    for (int k = 1; k < 5; k++) r.l[k] = _mm256_setzero_si256();
    return r;
}
}  // namespace

OTTI_IFMA_FN void ifma_dbl_n(PtFe &p, int n) { V5 P = vload_point(p); for (int i = 0; i < n; i++) P = vdbl(P); vstore_point(p, P); }

OTTI_IFMA_FN void ifma_straus(PtFe &acc, const Niels4 *tabs, int ntabs, const int *digs, int nwin) {
    V5 P = videntity();
    for (int w = nwin - 1; w >= 0; w--) {
        if (w != nwin - 1) { P = vdbl(P); P = vdbl(P); P = vdbl(P); P = vdbl(P); }
        for (int i = 0; i < ntabs; i++) {
            const int d = digs[(size_t)i * nwin + w];
            if (d) P = vmadd(P, tabs[(size_t)8 * i + (size_t)((d > 0 ? d : -d) - 1)], d < 0);
        }
    }
    vstore_point(acc, P);
}

OTTI_IFMA_FN void ifma_madd(PtFe &p, const Niels4 &q, bool negate) { vstore_point(p, vmadd(vload_point(p), q, negate)); }

OTTI_IFMA_FN void ifma_accumulate(PtFe &acc, const Niels4 *table, const int *dig, int w0, int w1) {
    V5 P = vload_point(acc);
    for (int w = w0; w < w1; w++) {
        const int d = dig[w];
        if (d) P = vmadd(P, table[(size_t)w * 128 + (size_t)((d > 0 ? d : -d) - 1)], d < 0);
    }
    vstore_point(acc, P);
}

}  // namespace otti

#endif
