// See hosttail.h.
#include "hosttail.h"
#include <chrono>
#include <functional>
#include <stdlib.h>
#include <string.h>
#include "hash.h"
#include "hostifma.h"
#include "pool.h"
#include "spartan.h"

namespace otti {

// ================================================================================================ scalar (field.h's 4 x u64 Montgomery code)
namespace {
class HostTailScalar final : public HostTail {
    int np_, ni_, threads_; size_t len_;
    std::vector<std::vector<Fr>> A_, B_, C_; std::vector<Fr> E_, coeff_;
public:
    HostTailScalar(int np, int nd, size_t T, const Fr *const *A, const Fr *const *B, const Fr *const *C, const Fr *E, const Fr *coeff, int threads)
        : np_(np), ni_(np + nd), threads_(std::max(1, std::min(8, threads))), len_(T), A_(ni_), B_(ni_), C_(ni_) {
        for (int k = 0; k < ni_; k++) { A_[k].assign(A[k], A[k] + T); B_[k].assign(B[k], B[k] + T); if (k >= np) C_[k].assign(C[k], C[k] + T); }
        if (np) E_.assign(E, E + T);
        coeff_.assign(coeff, coeff + ni_);
    }
    const char *kind() const override { return "scalar"; }
    size_t len() const override { return len_; }
    void sums(Fr out[3]) override {
        const size_t half = len_ / 2;
        // the instances are independent: spread over the prover's helper threads while a round is more than a few microseconds of work
        const int nt = (half * (size_t)ni_ >= 32) ? std::min(threads_, ni_) : 1;
        Fr part[8][3];
        auto share = [&](int t) {
            Fr q0 = fr_zero(), q2 = fr_zero(), q3 = fr_zero();
            for (int k = t; k < ni_; k += nt) {
                const std::vector<Fr> &A = A_[k], &B = B_[k], &Cc = k < np_ ? E_ : C_[k];
                Fr s0 = fr_zero(), s2 = fr_zero(), s3 = fr_zero();
                for (size_t i = 0; i < half; i++) {
                    const Fr da = fr_sub(A[i + half], A[i]), db = fr_sub(B[i + half], B[i]), dc = fr_sub(Cc[i + half], Cc[i]);
                    s0 = fr_add(s0, fr_mul(fr_mul(A[i], B[i]), Cc[i]));
                    Fr x = fr_add(A[i + half], da), y = fr_add(B[i + half], db), z = fr_add(Cc[i + half], dc);
                    s2 = fr_add(s2, fr_mul(fr_mul(x, y), z));
                    x = fr_add(x, da); y = fr_add(y, db); z = fr_add(z, dc);
                    s3 = fr_add(s3, fr_mul(fr_mul(x, y), z));
                }
                q0 = fr_add(q0, fr_mul(s0, coeff_[k])); q2 = fr_add(q2, fr_mul(s2, coeff_[k])); q3 = fr_add(q3, fr_mul(s3, coeff_[k]));
            }
            part[t][0] = q0; part[t][1] = q2; part[t][2] = q3;
        };
        if (nt > 1) { std::function<void()> tasks[8]; for (int t = 0; t < nt; t++) tasks[t] = [&share, t] { share(t); }; SpinPool::get().parallel(tasks, nt); }
        else share(0);
        out[0] = out[1] = out[2] = fr_zero();
        for (int t = 0; t < nt; t++) for (int p = 0; p < 3; p++) out[p] = fr_add(out[p], part[t][p]);
    }
    void fold(const Fr &r) override {
        const size_t half = len_ / 2;
        auto fold1 = [&](std::vector<Fr> &t) { for (size_t i = 0; i < half; i++) t[i] = fr_add(t[i], fr_mul(r, fr_sub(t[i + half], t[i]))); t.resize(half); };
        const int nt = (half * (size_t)ni_ >= 64) ? std::min(threads_, ni_) : 1;
        auto share = [&](int t) { for (int k = t; k < ni_; k += nt) { fold1(A_[k]); fold1(B_[k]); if (k >= np_) fold1(C_[k]); } if (t == nt - 1 && np_) fold1(E_); };
        if (nt > 1) { std::function<void()> tasks[8]; for (int t = 0; t < nt; t++) tasks[t] = [&share, t] { share(t); }; SpinPool::get().parallel(tasks, nt); }
        else share(0);
        len_ = half;
    }
    void last(int k, Fr out[3]) const override { out[0] = A_[k][0]; out[1] = B_[k][0]; out[2] = k < np_ ? E_[0] : C_[k][0]; }
};
}  // namespace

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)       // host pass only (hipcc also walks this file for gfx950)
}  // namespace otti
#include <immintrin.h>
namespace otti {

bool host_fr8_available() {
    static const bool on = [] {
        const char *e = getenv("OTTI_HOST_FR8");
        if (e && e[0] == '0') return false;
        __builtin_cpu_init();
        return host_ifma_available() && __builtin_cpu_supports("avx512dq");
    }();
    return on;
}

namespace {
#define OTTI_FR8 __attribute__((target("avx2,avx512f,avx512vl,avx512dq,avx512ifma"), always_inline)) static inline
#define OTTI_FR8_FN __attribute__((target("avx2,avx512f,avx512vl,avx512dq,avx512ifma")))

constexpr uint64_t M52 = ((uint64_t)1 << 52) - 1;
constexpr int kKl = 16;                                      // differences are kept non-negative by adding kKl * l: table entries stay below kKl * l (see bounds below)
constexpr size_t kMaxT = 256;                                // eight rounds (bounds below)
struct Consts { uint64_t L[5], linv, KL[5]; };
const Consts &consts() {
    static const Consts c = [] {
        Consts k{};
        const uint64_t w[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
        k.L[0] = w[0] & M52; k.L[1] = ((w[0] >> 52) | (w[1] << 12)) & M52; k.L[2] = ((w[1] >> 40) | (w[2] << 24)) & M52; k.L[3] = ((w[2] >> 28) | (w[3] << 36)) & M52; k.L[4] = w[3] >> 16;
        uint64_t x = 1;                                      // Newton: x <- x (2 - l0 x) doubles the number of correct low bits
        for (int i = 0; i < 6; i++) x *= 2 - k.L[0] * x;
        k.linv = (0 - x) & M52;                              // -l^-1 mod 2^52
        uint64_t cy = 0;
        for (int i = 0; i < 5; i++) { const uint64_t v = k.L[i] * kKl + cy; k.KL[i] = i < 4 ? (v & M52) : v; cy = i < 4 ? (v >> 52) : 0; }
        return k;
    }();
    return c;
}
inline void fr_to_limbs52(const Fr &a, uint64_t o[5]) {
    uint64_t w[4]; memcpy(w, a.v, 32);
    o[0] = w[0] & M52; o[1] = ((w[0] >> 52) | (w[1] << 12)) & M52; o[2] = ((w[1] >> 40) | (w[2] << 24)) & M52; o[3] = ((w[2] >> 28) | (w[3] << 36)) & M52; o[4] = w[3] >> 16;
}
// any value below 2^260 in normalised limbs -> the canonical element.  q = floor(v / 2^252) is floor(v / l) or one more (l = 2^252 + c, c < 2^125,
// q < 2^8): v - q l lies in (-2^133, 2^252), one conditional addition of l finishes.
inline Fr fr_from_limbs52(const uint64_t l[5]) {
    const Consts &K = consts();
    const int64_t q = (int64_t)(l[4] >> 44);
    int64_t r[5];
    for (int i = 0; i < 5; i++) r[i] = (int64_t)l[i] - q * (int64_t)K.L[i];
    for (int i = 0; i < 4; i++) { const int64_t c = r[i] >> 52; r[i] &= (int64_t)M52; r[i + 1] += c; }
    if (r[4] < 0) {
        for (int i = 0; i < 5; i++) r[i] += (int64_t)K.L[i];
        for (int i = 0; i < 4; i++) { const int64_t c = r[i] >> 52; r[i] &= (int64_t)M52; r[i + 1] += c; }
    }
    const uint64_t u[5] = {(uint64_t)r[0], (uint64_t)r[1], (uint64_t)r[2], (uint64_t)r[3], (uint64_t)r[4]};
    const uint64_t w[4] = {u[0] | (u[1] << 52), (u[1] >> 12) | (u[2] << 40), (u[2] >> 24) | (u[3] << 28), (u[3] >> 36) | (u[4] << 16)};
    Fr out; memcpy(out.v, w, 32);
    return out;
}

struct alignas(64) Vec { uint64_t l[5][8]; };               // limb k of eight elements side by side
struct V5 { __m512i l[5]; };
struct VC { __m512i L0, L1, L2, L4, linv, mask; V5 KL; };
OTTI_FR8 VC vconsts() {
    const Consts &K = consts(); VC c;
    c.L0 = _mm512_set1_epi64((long long)K.L[0]); c.L1 = _mm512_set1_epi64((long long)K.L[1]); c.L2 = _mm512_set1_epi64((long long)K.L[2]); c.L4 = _mm512_set1_epi64((long long)K.L[4]);
    c.linv = _mm512_set1_epi64((long long)K.linv); c.mask = _mm512_set1_epi64((long long)M52);
    for (int i = 0; i < 5; i++) c.KL.l[i] = _mm512_set1_epi64((long long)K.KL[i]);
    return c;
}
OTTI_FR8 V5 vload(const Vec *p) { V5 r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_load_si512((const void *)p->l[i]); return r; }
OTTI_FR8 void vstore(Vec *p, const V5 &a) { for (int i = 0; i < 5; i++) _mm512_store_si512((void *)p->l[i], a.l[i]); }
OTTI_FR8 V5 vzero() { V5 r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_setzero_si512(); return r; }
OTTI_FR8 V5 vbroadcast(const uint64_t l[5]) { V5 r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_set1_epi64((long long)l[i]); return r; }
OTTI_FR8 V5 vadd(const V5 &a, const V5 &b) { V5 r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_add_epi64(a.l[i], b.l[i]); return r; }
// a - b + kKl l, limb by limb (signed lanes; normalise before multiplying)
OTTI_FR8 V5 vsub_kl(const V5 &a, const V5 &b, const VC &c) { V5 r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_add_epi64(_mm512_sub_epi64(a.l[i], b.l[i]), c.KL.l[i]); return r; }
// the 256-bit halves exchanged: lanes 0-3 <-> 4-7
OTTI_FR8 V5 vswap(const V5 &a) { V5 r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_shuffle_i64x2(a.l[i], a.l[i], 0x4E); return r; }
// carries propagated (arithmetic shifts: limbs may be negative on the way in); the value must lie in [0, 2^260)
OTTI_FR8 V5 vnorm(V5 x, const VC &c) {
#pragma unroll
    for (int i = 0; i < 4; i++) { const __m512i cy = _mm512_srai_epi64(x.l[i], 52); x.l[i] = _mm512_and_si512(x.l[i], c.mask); x.l[i + 1] = _mm512_add_epi64(x.l[i + 1], cy); }
    return x;
}
// eight Montgomery products a b / 2^260 mod l: operand limbs in [0, 2^52), values below 2^260 with a b < 2^260 * 255 l; result normalised, below a b / 2^260 + l.
// Row i: t += a * b_i; m = t_0 * (-1/l) mod 2^52; t += m * l (l_3 = 0); t >>= 52.  High halves of the 104-bit products go to the next limb as they come.
OTTI_FR8 V5 vmul(const V5 &a, const V5 &b, const VC &c) {
    const __m512i z = _mm512_setzero_si512();
    __m512i t[6] = {z, z, z, z, z, z};
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 5; j++) { t[j] = _mm512_madd52lo_epu64(t[j], a.l[j], b.l[i]); t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], a.l[j], b.l[i]); }
        const __m512i m = _mm512_madd52lo_epu64(z, t[0], c.linv);
        t[0] = _mm512_madd52lo_epu64(t[0], m, c.L0); t[1] = _mm512_madd52hi_epu64(t[1], m, c.L0);
        t[1] = _mm512_madd52lo_epu64(t[1], m, c.L1); t[2] = _mm512_madd52hi_epu64(t[2], m, c.L1);
        t[2] = _mm512_madd52lo_epu64(t[2], m, c.L2); t[3] = _mm512_madd52hi_epu64(t[3], m, c.L2);
        t[4] = _mm512_madd52lo_epu64(t[4], m, c.L4); t[5] = _mm512_madd52hi_epu64(t[5], m, c.L4);
        t[0] = _mm512_add_epi64(t[1], _mm512_srli_epi64(t[0], 52));
        t[1] = t[2]; t[2] = t[3]; t[3] = t[4]; t[4] = t[5]; t[5] = z;
    }
    V5 r; for (int i = 0; i < 5; i++) r.l[i] = t[i];
    return vnorm(r, c);
}

// Bounds.  Table entries enter canonical (< l).  A fold gives lo + r' d / 2^260 + (< l) with d = hi - lo + 16 l and r' < l: after j folds the entries
// are below (1 + 1.13 j) l — below 9 l through the seven folds that precede the last round of a 256-element table, far below the 16 l that keeps
// differences non-negative.  Evaluation points in round j (entries < e l): d < (e + 16) l, x_2 = hi + d, x_3 = x_2 + d < (3 e + 32) l <= 59 l, all far
// below 2^260 = 255.9 l; products x y / 2^260 + l: round 0 (e = 1): 35 l * 35 l -> < 5.9 l, then 5.9 l * 35 l -> < 1.9 l, 64 terms per lane
// at most (< 120 l); round j has 64 / 2^j terms of < 4.3 l.  Everything a product takes in or a lane adds up stays below 2^260.
class HostTailFr8 final : public HostTail {
    int np_, nup_, nud_, nu_, threads_; size_t len_, nv0_;
    std::vector<Vec> tabs_, ev_, coeff_, coeff_lo_;       // tabs_[(u * 3 + t) * nv0_ + j]; ev_[j]: E[2 j] in lanes 0-3, E[2 j + 1] in lanes 4-7
    std::vector<Vec> part_;                               // per-thread partial sums: [thread][unit][point]
    std::vector<Vec *> tables_; int ntab_ = 0;            // every table there is to fold (two per product unit, three per triple unit, the eq table)
    Vec *tab(int u, int t) { return tabs_.data() + ((size_t)u * 3 + t) * nv0_; }
    const Vec *tab(int u, int t) const { return tabs_.data() + ((size_t)u * 3 + t) * nv0_; }
    int unit_of(int k) const { return k < np_ ? k / 4 : nup_ + (k - np_) / 4; }
    int lane_of(int k) const { return k < np_ ? k % 4 : (k - np_) % 4; }
    static void put(Vec &v, int lane, const Fr &x) { uint64_t l[5]; fr_to_limbs52(x, l); for (int i = 0; i < 5; i++) v.l[i][lane] = l[i]; }
    static Fr get(const Vec &v, int lane) { uint64_t l[5]; for (int i = 0; i < 5; i++) l[i] = v.l[i][lane]; return fr_from_limbs52(l); }

    OTTI_FR8_FN void load_tables(int np, int nd, size_t T, const Fr *const *A, const Fr *const *B, const Fr *const *C, const Fr *E);
    OTTI_FR8_FN void sums_range(int t, int nt, size_t nv, bool single);
    OTTI_FR8_FN void combine(int nt, bool single, Fr out[3]);
    OTTI_FR8_FN void fold_range(int t, int nt, size_t nv, bool single, const uint64_t r16[5]);
public:
    HostTailFr8(int np, int nd, size_t T, const Fr *const *A, const Fr *const *B, const Fr *const *C, const Fr *E, const Fr *coeff, int threads)
        : np_(np), nup_((np + 3) / 4), nud_((nd + 3) / 4), nu_(nup_ + nud_), threads_(std::max(1, std::min(8, threads))), len_(T), nv0_(T / 2) {
        Vec zero; memset(&zero, 0, sizeof zero);
        tabs_.resize((size_t)nu_ * 3 * nv0_); ev_.resize(np ? nv0_ : 0); coeff_.assign(nu_, zero); coeff_lo_.assign(nu_, zero);
        part_.assign((size_t)8 * nu_ * 3, zero);
        const Fr two12 = fr_from_u64(4096);                  // (a b) c and the coefficient: three reductions by 2^260 where 2^256 is meant
        for (int k = 0; k < np + nd; k++) {
            const int u = unit_of(k), q = lane_of(k);
            const Fr cs = fr_mul(coeff[k], two12);
            put(coeff_[u], q, cs); put(coeff_[u], 4 + q, cs); put(coeff_lo_[u], q, cs);
        }
        load_tables(np, nd, T, A, B, C, E);
        for (int u = 0; u < nu_; u++) { tables_.push_back(tab(u, 0)); tables_.push_back(tab(u, 1)); if (u >= nup_) tables_.push_back(tab(u, 2)); }
        if (np) tables_.push_back(ev_.data());
        ntab_ = (int)tables_.size();
    }
    const char *kind() const override { return "avx512ifma"; }
    size_t len() const override { return len_; }
    // work is dealt out in items — (unit, vector pair) for the sums, (table, vector pair) for the fold — in contiguous ranges; a thread is only
    // worth waking for `grain` products or more (a hand-over costs about what 20 of them do)
    int threads_for(size_t items, size_t products_per_item) const {
        static const size_t grain = [] { const char *e = getenv("OTTI_HOST_TAIL_GRAIN"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 ? v : 48); }();
        return (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>((size_t)threads_, items), items * products_per_item / grain));
    }
    void sums(Fr out[3]) override {
        const size_t half = len_ / 2; const bool single = half < 2; const size_t nv = single ? 1 : half / 2;
        const int nt = threads_for(nv * (size_t)nu_, 6);
        if (nt > 1) { std::function<void()> tasks[8]; for (int t = 0; t < nt; t++) tasks[t] = [this, t, nt, nv, single] { sums_range(t, nt, nv, single); }; SpinPool::get().parallel(tasks, nt); }
        else sums_range(0, 1, nv, single);
        combine(nt, single, out);
    }
    void fold(const Fr &r) override {
        const size_t half = len_ / 2; const bool single = half < 2; const size_t nv = single ? 1 : half / 2;
        uint64_t r16[5]; fr_to_limbs52(fr_mul(r, fr_from_u64(16)), r16);     // one reduction by 2^260 where 2^256 is meant
        const int nt = threads_for(nv * (size_t)ntab_, 1);
        if (nt > 1) { std::function<void()> tasks[8]; for (int t = 0; t < nt; t++) tasks[t] = [this, t, nt, nv, single, &r16] { fold_range(t, nt, nv, single, r16); }; SpinPool::get().parallel(tasks, nt); }
        else fold_range(0, 1, nv, single, r16);
        len_ = half;
    }
    void last(int k, Fr out[3]) const override {
        const int u = unit_of(k), q = lane_of(k);
        out[0] = get(tab(u, 0)[0], q); out[1] = get(tab(u, 1)[0], q); out[2] = k < np_ ? get(ev_[0], 0) : get(tab(u, 2)[0], q);
    }
};
// The tables into the packed form, a vector (eight elements: two consecutive ones of four instances) at a time: the four 64-bit words of the eight
// elements are gathered, cut into 52-bit limbs side by side and stored as five rows — element by element this cost more than a layer's arithmetic.
OTTI_FR8 void pack8(Vec *dst, const Fr *const src[8]) {
    long long addr[8]; for (int l = 0; l < 8; l++) addr[l] = (long long)(uintptr_t)src[l];
    const __m512i a = _mm512_loadu_si512((const void *)addr), m = _mm512_set1_epi64((long long)M52);
    const __m512i w0 = _mm512_i64gather_epi64(a, (const void *)0, 1), w1 = _mm512_i64gather_epi64(a, (const void *)8, 1),
                  w2 = _mm512_i64gather_epi64(a, (const void *)16, 1), w3 = _mm512_i64gather_epi64(a, (const void *)24, 1);
    _mm512_store_si512((void *)dst->l[0], _mm512_and_si512(w0, m));
    _mm512_store_si512((void *)dst->l[1], _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w0, 52), _mm512_slli_epi64(w1, 12)), m));
    _mm512_store_si512((void *)dst->l[2], _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w1, 40), _mm512_slli_epi64(w2, 24)), m));
    _mm512_store_si512((void *)dst->l[3], _mm512_and_si512(_mm512_or_si512(_mm512_srli_epi64(w2, 28), _mm512_slli_epi64(w3, 36)), m));
    _mm512_store_si512((void *)dst->l[4], _mm512_srli_epi64(w3, 16));
}
OTTI_FR8_FN void HostTailFr8::load_tables(int np, int nd, size_t T, const Fr *const *A, const Fr *const *B, const Fr *const *C, const Fr *E) {
    static const Fr zero_fr = fr_zero();
    for (int u = 0; u < nu_; u++) {
        const bool triple = u >= nup_; const int k0 = triple ? np + 4 * (u - nup_) : 4 * u, kend = triple ? np + nd : np;
        for (int t = 0; t < (triple ? 3 : 2); t++) {
            const Fr *const *tab_src = t == 0 ? A : t == 1 ? B : C;
            for (size_t j = 0; j < nv0_; j++) {
                const Fr *src[8];
                for (int l = 0; l < 8; l++) { const int k = k0 + (l & 3); src[l] = k < kend ? tab_src[k] + 2 * j + (size_t)(l >> 2) : &zero_fr; }
                pack8(&tab(u, t)[j], src);
            }
        }
        if (!triple) for (size_t j = 0; j < nv0_; j++) memset(&tab(u, 2)[j], 0, sizeof(Vec));
    }
    for (size_t j = 0; np && j < nv0_; j++) {
        const Fr *src[8]; for (int l = 0; l < 8; l++) src[l] = E + 2 * j + (size_t)(l >> 2);
        pack8(&ev_[j], src);
    }
    (void)T;
}
// items [t n / nt, (t + 1) n / nt) of the n = units x nv (unit, vector pair) items, unit-major; a pair is (vector j, vector j + nv), or the two halves
// of vector 0 in the last round
OTTI_FR8_FN void HostTailFr8::sums_range(int t, int nt, size_t nv, bool single) {
    const VC c = vconsts();
    const size_t n = (size_t)nu_ * nv, i0 = n * (size_t)t / (size_t)nt, i1 = n * (size_t)(t + 1) / (size_t)nt;
    Vec *mine = &part_[(size_t)t * nu_ * 3];
    for (int u = 0; u < nu_; u++) {
        const size_t a = std::max(i0, (size_t)u * nv), b = std::min(i1, (size_t)(u + 1) * nv);
        V5 acc0 = vzero(), acc2 = vzero(), acc3 = vzero();
        for (size_t i = a; i < b; i++) {
            const size_t j = i - (size_t)u * nv;
            const V5 alo = vload(&tab(u, 0)[j]), ahi = single ? vswap(alo) : vload(&tab(u, 0)[j + nv]);
            const V5 blo = vload(&tab(u, 1)[j]), bhi = single ? vswap(blo) : vload(&tab(u, 1)[j + nv]);
            const V5 da = vnorm(vsub_kl(ahi, alo, c), c), db = vnorm(vsub_kl(bhi, blo, c), c);
            const V5 a2 = vnorm(vadd(ahi, da), c), b2 = vnorm(vadd(bhi, db), c), a3 = vnorm(vadd(a2, da), c), b3 = vnorm(vadd(b2, db), c);
            const Vec *third = u >= nup_ ? tab(u, 2) : ev_.data();     // a triple's own third table, or the eq table the product instances share
            const V5 clo = vload(&third[j]), chi = single ? vswap(clo) : vload(&third[j + nv]);
            const V5 dc = vnorm(vsub_kl(chi, clo, c), c), c2 = vnorm(vadd(chi, dc), c), c3 = vnorm(vadd(c2, dc), c);
            acc0 = vadd(acc0, vmul(vmul(alo, blo, c), clo, c));
            acc2 = vadd(acc2, vmul(vmul(a2, b2, c), c2, c));
            acc3 = vadd(acc3, vmul(vmul(a3, b3, c), c3, c));
        }
        vstore(&mine[3 * u], acc0); vstore(&mine[3 * u + 1], acc2); vstore(&mine[3 * u + 2], acc3);
    }
}
OTTI_FR8_FN void HostTailFr8::combine(int nt, bool single, Fr out[3]) {
    const VC c = vconsts();
    V5 tot[3] = {vzero(), vzero(), vzero()};
    for (int u = 0; u < nu_; u++) {
        const V5 cf = vload(single ? &coeff_lo_[u] : &coeff_[u]);
        for (int p = 0; p < 3; p++) {
            V5 s = vload(&part_[((size_t)0 * nu_ + u) * 3 + p]);
            for (int t = 1; t < nt; t++) s = vadd(s, vload(&part_[((size_t)t * nu_ + u) * 3 + p]));
            tot[p] = vadd(tot[p], vmul(vnorm(s, c), cf, c));   // each < 34 l * l / 2^260 + l < 1.2 l; at most 8 units
        }
    }
    for (int p = 0; p < 3; p++) {
        Vec v; vstore(&v, vnorm(tot[p], c));
        Fr s = get(v, 0);
        for (int lane = 1; lane < 8; lane++) s = fr_add(s, get(v, lane));
        out[p] = s;
    }
}
OTTI_FR8 void fold_vec(Vec *T, size_t j, size_t nv, bool single, const V5 &r, const VC &c) {
    const V5 lo = vload(&T[j]), hi = single ? vswap(lo) : vload(&T[j + nv]);
    vstore(&T[j], vnorm(vadd(lo, vmul(r, vnorm(vsub_kl(hi, lo, c), c), c)), c));
}
OTTI_FR8_FN void HostTailFr8::fold_range(int t, int nt, size_t nv, bool single, const uint64_t r16[5]) {
    const VC c = vconsts();
    const V5 r = vbroadcast(r16);
    const size_t n = (size_t)ntab_ * nv, i0 = n * (size_t)t / (size_t)nt, i1 = n * (size_t)(t + 1) / (size_t)nt;
    for (size_t i = i0; i < i1; i++) fold_vec(tables_[i / nv], i % nv, nv, single, r, c);
}
OTTI_FR8_FN void weighted_sums3_fr8(const Fr *s, const Fr *w, int n0, int n, Fr out0[3], Fr out1[3]) {
    static const Fr zero_fr = fr_zero(), sixteen = fr_from_u64(16);      // one reduction by 2^260 where 2^256 is meant
    const VC c = vconsts();
    V5 acc[2][3];
    for (int g = 0; g < 2; g++) for (int t = 0; t < 3; t++) acc[g][t] = vzero();
    Fr w16[24];
    for (int k = 0; k < n; k++) w16[k] = fr_mul(w[k], sixteen);
    for (int g = 0; g < 2; g++) {
        const int k0 = g ? n0 : 0, k1 = g ? n : n0;
        for (int base = k0; base < k1; base += 8) {
            const Fr *wp[8]; for (int l = 0; l < 8; l++) wp[l] = base + l < k1 ? &w16[base + l] : &zero_fr;
            Vec wv; pack8(&wv, wp);
            const V5 wvv = vload(&wv);
            for (int t = 0; t < 3; t++) {
                const Fr *sp[8]; for (int l = 0; l < 8; l++) sp[l] = base + l < k1 ? &s[3 * (base + l) + t] : &zero_fr;
                Vec sv; pack8(&sv, sp);
                acc[g][t] = vadd(acc[g][t], vmul(vload(&sv), wvv, c));         // each below 2 l; at most three vectors per group
            }
        }
    }
    for (int g = 0; g < 2; g++) for (int t = 0; t < 3; t++) {
        Vec v; vstore(&v, vnorm(acc[g][t], c));
        Fr x = fr_zero();
        for (int lane = 0; lane < 8; lane++) { uint64_t l[5]; for (int i = 0; i < 5; i++) l[i] = v.l[i][lane]; x = fr_add(x, fr_from_limbs52(l)); }
        (g ? out1 : out0)[t] = x;
    }
}
}  // namespace
#else
bool host_fr8_available() { return false; }
#endif
void weighted_sums3(const Fr *s, const Fr *w, int n0, int n, Fr out0[3], Fr out1[3]) {
    if (n < 0 || n > 24 || n0 < 0 || n0 > n) throw Error(OTTI_ERR_INTERNAL, "weighted sums: bad sizes");
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (host_fr8_available() && n >= 6) { weighted_sums3_fr8(s, w, n0, n, out0, out1); return; }
#endif
    for (int t = 0; t < 3; t++) { out0[t] = fr_zero(); out1[t] = fr_zero(); }
    for (int k = 0; k < n; k++) for (int t = 0; t < 3; t++) { Fr &a = (k < n0 ? out0 : out1)[t]; a = fr_add(a, fr_mul(s[3 * k + t], w[k])); }
}

std::unique_ptr<HostTail> HostTail::make(int np, int nd, size_t T, const Fr *const *A, const Fr *const *B, const Fr *const *C, const Fr *E, const Fr *coeff, int threads, bool force_scalar) {
    if (np < 0 || nd < 0 || np + nd < 1 || T < 2 || (T & (T - 1))) throw Error(OTTI_ERR_INTERNAL, "host sum-check tail: bad geometry");
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    if (!force_scalar && host_fr8_available() && T <= kMaxT && (np + 3) / 4 + (nd + 3) / 4 <= 8) return std::unique_ptr<HostTail>(new HostTailFr8(np, nd, T, A, B, C, E, coeff, threads));
#endif
    return std::unique_ptr<HostTail>(new HostTailScalar(np, nd, T, A, B, C, E, coeff, threads));
}

void hosttail_selftest(uint32_t seed) {
    Shake256 xof; xof.absorb("otti-hosttail-selftest", 22); xof.absorb(&seed, 4);
    auto rnd_fr = [&](int special) {
        if (special == 0) return fr_zero();
        if (special == 1) return fr_neg(fr_one());
        uint8_t w[64]; xof.squeeze(w, 64); return fr_from_bytes_wide(w);
    };
    uint8_t pick[4]; xof.squeeze(pick, 4);
    const size_t T = (size_t)2 << (pick[0] % 8);             // 2 .. 256
    int np = pick[1] % 13, nd = pick[2] % 7;
    if (np + nd == 0) np = 1;
    const int ni = np + nd;
    std::vector<std::vector<Fr>> A(ni), B(ni), C(ni); std::vector<Fr> E(T), coeff(ni);
    for (int k = 0; k < ni; k++) {
        A[k].resize(T); B[k].resize(T); C[k].resize(T);
        // every eighth run: the extremes (upper halves l - 1, lower halves 0, or the reverse), which make the evaluation points as large as they get
        const int extreme = (seed % 8 == 7) ? 1 + (k & 1) : 0;
        for (size_t e = 0; e < T; e++) {
            uint8_t s[3]; xof.squeeze(s, 3);
            if (extreme) { const int hi_half = e >= T / 2; s[0] = s[1] = s[2] = (uint8_t)((hi_half == (extreme == 1)) ? 1 : 0); }
            A[k][e] = rnd_fr(s[0] % 16); B[k][e] = rnd_fr(s[1] % 16); C[k][e] = rnd_fr(s[2] % 16);
        }
        coeff[k] = rnd_fr(7);
    }
    for (size_t e = 0; e < T; e++) E[e] = rnd_fr(9);
    std::vector<const Fr *> pa(ni), pb(ni), pc(ni);
    for (int k = 0; k < ni; k++) { pa[k] = A[k].data(); pb[k] = B[k].data(); pc[k] = C[k].data(); }
    const int threads = 1 + pick[3] % 4;
    std::unique_ptr<HostTail> ref = HostTail::make(np, nd, T, pa.data(), pb.data(), pc.data(), E.data(), coeff.data(), threads, true);
    std::unique_ptr<HostTail> got = HostTail::make(np, nd, T, pa.data(), pb.data(), pc.data(), E.data(), coeff.data(), threads);
    while (ref->len() > 1) {
        Fr s1[3], s2[3]; ref->sums(s1); got->sums(s2);
        for (int p = 0; p < 3; p++) if (!fr_eq(s1[p], s2[p])) throw Error(OTTI_ERR_INTERNAL, "host sum-check tail: the vector form's round sums differ from the scalar form's");
        const Fr r = rnd_fr(ref->len() == 4 ? 1 : 5);
        ref->fold(r); got->fold(r);
        if (ref->len() != got->len()) throw Error(OTTI_ERR_INTERNAL, "host sum-check tail: lengths differ");
    }
    {   // the weighted sums of a device round, vector form against the plain loop
        std::vector<Fr> s3((size_t)3 * ni); for (auto &x : s3) x = rnd_fr(3);
        Fr a0[3], a1[3], b0[3] = {fr_zero(), fr_zero(), fr_zero()}, b1[3] = {fr_zero(), fr_zero(), fr_zero()};
        weighted_sums3(s3.data(), coeff.data(), np, ni, a0, a1);
        for (int k = 0; k < ni; k++) for (int t = 0; t < 3; t++) { Fr &x = (k < np ? b0 : b1)[t]; x = fr_add(x, fr_mul(s3[3 * k + t], coeff[k])); }
        for (int t = 0; t < 3; t++) if (!fr_eq(a0[t], b0[t]) || !fr_eq(a1[t], b1[t])) throw Error(OTTI_ERR_INTERNAL, "weighted sums: the vector form differs from the plain loop");
    }
    for (int k = 0; k < ni; k++) {
        Fr l1[3], l2[3]; ref->last(k, l1); got->last(k, l2);
        for (int p = 0; p < 3; p++) if (!fr_eq(l1[p], l2[p])) throw Error(OTTI_ERR_INTERNAL, "host sum-check tail: the vector form's final elements differ from the scalar form's");
    }
}

void hosttail_bench(int np, int nd, size_t T, int threads, int reps, double out[2]) {
    Shake256 xof; xof.absorb("otti-hosttail-bench", 19);
    auto rnd_fr = [&] { uint8_t w[64]; xof.squeeze(w, 64); return fr_from_bytes_wide(w); };
    const int ni = np + nd;
    if (ni < 1 || reps < 1) throw Error(OTTI_ERR_BAD_ARG, "host tail bench: nothing to do");
    std::vector<std::vector<Fr>> A(ni), B(ni), C(ni); std::vector<Fr> E(T), coeff(ni);
    for (int k = 0; k < ni; k++) { A[k].resize(T); B[k].resize(T); C[k].resize(T); for (size_t e = 0; e < T; e++) { A[k][e] = rnd_fr(); B[k][e] = rnd_fr(); C[k][e] = rnd_fr(); } coeff[k] = rnd_fr(); }
    for (size_t e = 0; e < T; e++) E[e] = rnd_fr();
    std::vector<const Fr *> pa(ni), pb(ni), pc(ni);
    for (int k = 0; k < ni; k++) { pa[k] = A[k].data(); pb[k] = B[k].data(); pc[k] = C[k].data(); }
    const Fr r = rnd_fr();
    SpinPool::Session session;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int form = 0; form < 2; form++) {
        out[form] = 0;
        if (form == 0 && !host_fr8_available()) continue;
        Fr sink = fr_zero();
        for (int rep = -2; rep < reps; rep++) {
            const double t0 = now();
            std::unique_ptr<HostTail> h = HostTail::make(np, nd, T, pa.data(), pb.data(), pc.data(), E.data(), coeff.data(), threads, form == 1);
            while (h->len() > 1) { Fr s[3]; h->sums(s); sink = fr_add(sink, s[0]); h->fold(fr_add(r, sink)); }
            Fr l3[3]; h->last(0, l3); sink = fr_add(sink, l3[0]);
            if (rep >= 0) out[form] += now() - t0;
        }
        out[form] /= reps;
        if (fr_eq(sink, fr_one())) out[form] += 1e-9;          // (keeps the loop's results alive)
    }
}

}  // namespace otti
