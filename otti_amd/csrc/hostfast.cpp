#include "hostfast.h"

namespace otti {

typedef unsigned __int128 u128;
static const uint64_t M51 = ((uint64_t)1 << 51) - 1;

Fe fe_from_fp(const Fp &a) {
    uint64_t t[4]; memcpy(t, a.v, 32);
    Fe r;
    r.v[0] = t[0] & M51;
    r.v[1] = ((t[0] >> 51) | (t[1] << 13)) & M51;
    r.v[2] = ((t[1] >> 38) | (t[2] << 26)) & M51;
    r.v[3] = ((t[2] >> 25) | (t[3] << 39)) & M51;
    r.v[4] = t[3] >> 12;                                   // a loosely reduced Fp is < 2^256: this limb may use 52 bits
    return r;
}
static inline void fe_carry(Fe &a) {                         // limbs below 2^51 (limb 0 may keep a few units above after the wrap)
    uint64_t c;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
    c = a.v[1] >> 51; a.v[1] &= M51; a.v[2] += c;
    c = a.v[2] >> 51; a.v[2] &= M51; a.v[3] += c;
    c = a.v[3] >> 51; a.v[3] &= M51; a.v[4] += c;
    c = a.v[4] >> 51; a.v[4] &= M51; a.v[0] += 19 * c;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
}
Fp fe_to_fp(const Fe &a0) {
    Fe a = a0; fe_carry(a); fe_carry(a);                     // every limb < 2^51 (limb 1 by at most one unit more): value < 2^255 + 2^52
    u128 acc = a.v[0];
    uint64_t t[4];
    acc += (u128)a.v[1] << 51; t[0] = (uint64_t)acc; acc >>= 64;
    acc += (u128)a.v[2] << 38; t[1] = (uint64_t)acc; acc >>= 64;
    acc += (u128)a.v[3] << 25; t[2] = (uint64_t)acc; acc >>= 64;
    acc += (u128)a.v[4] << 12; t[3] = (uint64_t)acc;         // < 2^256: fits (a valid loosely reduced Fp)
    Fp r; memcpy(r.v, t, 32); return r;
}
static inline Fe fe_add(const Fe &a, const Fe &b) { Fe r; for (int i = 0; i < 5; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
static inline Fe fe_sub(const Fe &a, const Fe &b) {          // a - b + 4p; b limbs < 2^53
    Fe r;
    r.v[0] = a.v[0] + 0x1fffffffffffb4ULL - b.v[0];
    for (int i = 1; i < 5; i++) r.v[i] = a.v[i] + 0x1ffffffffffffcULL - b.v[i];
    return r;
}
static inline Fe fe_mul(const Fe &a, const Fe &b) {          // limbs < 2^54 in, < 2^51 + 2^13 out
    uint64_t r0 = b.v[0], r1 = b.v[1], r2 = b.v[2], r3 = b.v[3], r4 = b.v[4];
    const uint64_t s0 = a.v[0], s1 = a.v[1], s2 = a.v[2], s3 = a.v[3], s4 = a.v[4];
    u128 t0 = (u128)r0 * s0;
    u128 t1 = (u128)r0 * s1 + (u128)r1 * s0;
    u128 t2 = (u128)r0 * s2 + (u128)r2 * s0 + (u128)r1 * s1;
    u128 t3 = (u128)r0 * s3 + (u128)r3 * s0 + (u128)r1 * s2 + (u128)r2 * s1;
    u128 t4 = (u128)r0 * s4 + (u128)r4 * s0 + (u128)r3 * s1 + (u128)r1 * s3 + (u128)r2 * s2;
    r1 *= 19; r2 *= 19; r3 *= 19; r4 *= 19;
    t0 += (u128)r4 * s1 + (u128)r1 * s4 + (u128)r2 * s3 + (u128)r3 * s2;
    t1 += (u128)r4 * s2 + (u128)r2 * s4 + (u128)r3 * s3;
    t2 += (u128)r4 * s3 + (u128)r3 * s4;
    t3 += (u128)r4 * s4;
    Fe o; uint64_t c;
    o.v[0] = (uint64_t)t0 & M51; c = (uint64_t)(t0 >> 51);
    t1 += c; o.v[1] = (uint64_t)t1 & M51; c = (uint64_t)(t1 >> 51);
    t2 += c; o.v[2] = (uint64_t)t2 & M51; c = (uint64_t)(t2 >> 51);
    t3 += c; o.v[3] = (uint64_t)t3 & M51; c = (uint64_t)(t3 >> 51);
    t4 += c; o.v[4] = (uint64_t)t4 & M51; c = (uint64_t)(t4 >> 51);
    o.v[0] += c * 19; c = o.v[0] >> 51; o.v[0] &= M51; o.v[1] += c;
    return o;
}
static inline Fe fe_sqr(const Fe &a) {
    const uint64_t r0 = a.v[0], r1 = a.v[1], r2 = a.v[2], r3 = a.v[3], r4 = a.v[4];
    const uint64_t d0 = r0 * 2, d1 = r1 * 2, d2 = r2 * 2 * 19, d419 = r4 * 19, d4 = d419 * 2;
    u128 t0 = (u128)r0 * r0 + (u128)d4 * r1 + (u128)d2 * r3;
    u128 t1 = (u128)d0 * r1 + (u128)d4 * r2 + (u128)r3 * (r3 * 19);
    u128 t2 = (u128)d0 * r2 + (u128)r1 * r1 + (u128)d4 * r3;
    u128 t3 = (u128)d0 * r3 + (u128)d1 * r2 + (u128)r4 * d419;
    u128 t4 = (u128)d0 * r4 + (u128)d1 * r3 + (u128)r2 * r2;
    Fe o; uint64_t c;
    o.v[0] = (uint64_t)t0 & M51; c = (uint64_t)(t0 >> 51);
    t1 += c; o.v[1] = (uint64_t)t1 & M51; c = (uint64_t)(t1 >> 51);
    t2 += c; o.v[2] = (uint64_t)t2 & M51; c = (uint64_t)(t2 >> 51);
    t3 += c; o.v[3] = (uint64_t)t3 & M51; c = (uint64_t)(t3 >> 51);
    t4 += c; o.v[4] = (uint64_t)t4 & M51; c = (uint64_t)(t4 >> 51);
    o.v[0] += c * 19; c = o.v[0] >> 51; o.v[0] &= M51; o.v[1] += c;
    return o;
}
static inline Fe fe_sqr_n(Fe a, int n) { for (int i = 0; i < n; i++) a = fe_sqr(a); return a; }
static Fe fe_pow22523(const Fe &a) {                         // a^((p-5)/8) = a^(2^252 - 3)
    Fe z2 = fe_sqr(a), z9 = fe_mul(fe_sqr_n(z2, 2), a), z11 = fe_mul(z9, z2);
    Fe t5 = fe_mul(fe_sqr(z11), z9);
    Fe t10 = fe_mul(fe_sqr_n(t5, 5), t5), t20 = fe_mul(fe_sqr_n(t10, 10), t10), t40 = fe_mul(fe_sqr_n(t20, 20), t20);
    Fe t50 = fe_mul(fe_sqr_n(t40, 10), t10), t100 = fe_mul(fe_sqr_n(t50, 50), t50), t200 = fe_mul(fe_sqr_n(t100, 100), t100);
    Fe t250 = fe_mul(fe_sqr_n(t200, 50), t50);
    return fe_mul(fe_sqr_n(t250, 2), a);
}

PtFe ptfe_from(const Pt &p) { PtFe r; r.X = fe_from_fp(p.X); r.Y = fe_from_fp(p.Y); r.Z = fe_from_fp(p.Z); r.T = fe_from_fp(p.T); return r; }
Pt ptfe_to(const PtFe &p) { Pt r; r.X = fe_to_fp(p.X); r.Y = fe_to_fp(p.Y); r.Z = fe_to_fp(p.Z); r.T = fe_to_fp(p.T); return r; }
NielsFe nielsfe_from(const Niels &n) { NielsFe r; r.yplusx = fe_from_fp(n.yplusx); r.yminusx = fe_from_fp(n.yminusx); r.xy2d = fe_from_fp(n.xy2d); return r; }
// add-2008-hwcd-3 mixed addition (7M), as pt_madd / pt_msub of point.h.  Coordinates stay < 2^51 + 2^13 per limb (products).
void ptfe_madd(PtFe &p, const NielsFe &q, bool negate) {
    const Fe a = fe_mul(fe_sub(p.Y, p.X), negate ? q.yplusx : q.yminusx);
    const Fe b = fe_mul(fe_add(p.Y, p.X), negate ? q.yminusx : q.yplusx);
    const Fe c = fe_mul(p.T, q.xy2d);
    const Fe d = fe_add(p.Z, p.Z);
    const Fe e = fe_sub(b, a), h = fe_add(b, a);
    const Fe f = negate ? fe_add(d, c) : fe_sub(d, c), g = negate ? fe_sub(d, c) : fe_add(d, c);
    p.X = fe_mul(e, f); p.Y = fe_mul(g, h); p.T = fe_mul(e, h); p.Z = fe_mul(f, g);
}

PtFe ptfe_identity() { static const PtFe id = ptfe_from(pt_identity()); return id; }
// add-2008-hwcd-3 with both operands extended: the same formula as pt_add (point.h)
void ptfe_add(PtFe &p, const PtFe &q) {
    static const Fe d2 = fe_from_fp(fp_2D());
    const Fe a = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    const Fe b = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    const Fe c = fe_mul(fe_mul(p.T, d2), q.T);
    const Fe zz = fe_mul(p.Z, q.Z), d = fe_add(zz, zz);
    const Fe e = fe_sub(b, a), f = fe_sub(d, c), g = fe_add(d, c), h = fe_add(b, a);
    p.X = fe_mul(e, f); p.Y = fe_mul(g, h); p.T = fe_mul(e, h); p.Z = fe_mul(f, g);
}

CachedFe ptfe_cache(const PtFe &p) {
    static const Fe d2 = fe_from_fp(fp_2D());
    CachedFe c; c.yplusx = fe_add(p.Y, p.X); c.yminusx = fe_sub(p.Y, p.X); c.z2 = fe_add(p.Z, p.Z); c.t2d = fe_mul(p.T, d2);
    fe_carry(c.yplusx); fe_carry(c.yminusx); fe_carry(c.z2);              // every limb back under 2^52: the additions below start from there
    return c;
}
// dbl-2008-hwcd with a = -1
void ptfe_dbl(PtFe &p) {
    const Fe A = fe_sqr(p.X), B = fe_sqr(p.Y), zz = fe_sqr(p.Z), C = fe_add(zz, zz);
    Fe xy = fe_add(p.X, p.Y);
    const Fe E = fe_sub(fe_sqr(xy), fe_add(A, B));                        // (X + Y)^2 - A - B
    Fe G = fe_sub(B, A); fe_carry(G);                                     // D + B with D = -A
    const Fe F = fe_sub(G, C), H = fe_sub(fe_from_fp(fp_zero()), fe_add(A, B));   // D - B
    p.X = fe_mul(E, F); p.Y = fe_mul(G, H); p.T = fe_mul(E, H); p.Z = fe_mul(F, G);
}
void ptfe_add_cached(PtFe &p, const CachedFe &q, bool negate) {
    const Fe a = fe_mul(fe_sub(p.Y, p.X), negate ? q.yplusx : q.yminusx);
    const Fe b = fe_mul(fe_add(p.Y, p.X), negate ? q.yminusx : q.yplusx);
    const Fe c = fe_mul(p.T, q.t2d), d = fe_mul(p.Z, q.z2);
    const Fe e = fe_sub(b, a), h = fe_add(b, a);
    const Fe f = negate ? fe_add(d, c) : fe_sub(d, c), g = negate ? fe_sub(d, c) : fe_add(d, c);
    p.X = fe_mul(e, f); p.Y = fe_mul(g, h); p.T = fe_mul(e, h); p.Z = fe_mul(f, g);
}

bool pt_decode_fast(Pt &o, const uint8_t b[32]) {
    static const Fe sqrt_m1 = fe_from_fp(fp_SQRT_M1()), dconst = fe_from_fp(fp_D()), one = fe_from_fp(fp_one()), zero = fe_from_fp(fp_zero());
    if (!fp_bytes_canonical(b) || (b[0] & 1)) return false;
    const Fe s = fe_from_fp(fp_from_bytes(b)), ss = fe_sqr(s);
    Fe u1 = fe_sub(one, ss), u2 = fe_add(one, ss); fe_carry(u1);
    const Fe u2s = fe_sqr(u2);
    Fe v = fe_sub(fe_sub(zero, fe_mul(dconst, fe_sqr(u1))), u2s); fe_carry(v);     // -d u1^2 - u2^2
    // SQRT_RATIO_M1(1, w), w = v u2^2: r = w^3 (w^7)^((p-5)/8); the root's sign from what w r^2 turns out to be
    const Fe w = fe_mul(v, u2s);
    const Fe w3 = fe_mul(fe_sqr(w), w), w7 = fe_mul(fe_sqr(w3), w);
    Fe r = fe_mul(w3, fe_pow22523(w7));
    const Fp check = fe_to_fp(fe_mul(w, fe_sqr(r)));
    const Fp neg_one = fp_neg(fp_one());
    const bool correct = fp_eq(check, fp_one()), flipped = fp_eq(check, neg_one), flipped_i = fp_eq(check, fp_mul(neg_one, fp_SQRT_M1()));
    if (flipped || flipped_i) r = fe_mul(r, sqrt_m1);
    const Fe inv = fe_from_fp(fp_abs(fe_to_fp(r)));
    const Fe dx = fe_mul(inv, u2), dy = fe_mul(fe_mul(inv, dx), v);
    Fp x = fe_to_fp(fe_mul(s, dx)); x = fp_abs(fp_add(x, x));
    const Fe xe = fe_from_fp(x), ye = fe_mul(u1, dy);
    const Fp y = fe_to_fp(ye), t = fe_to_fp(fe_mul(xe, ye));
    if (!(correct || flipped) || fp_is_negative(t) || fp_is_zero(y)) return false;
    o.X = x; o.Y = y; o.Z = fp_one(); o.T = t; return true;
}

void pt_encode_fast(uint8_t out[32], const Pt &p) { pt_encode_fe(out, ptfe_from(p)); }
void pt_encode_fe(uint8_t out[32], const PtFe &p) {
    static const Fe sqrt_m1 = fe_from_fp(fp_SQRT_M1()), invsqrt_a_minus_d = fe_from_fp(fp_INVSQRT_A_MINUS_D());
    const Fe &X = p.X, &Y = p.Y, &Z = p.Z, &T = p.T;
    const Fe u1 = fe_mul(fe_add(Z, Y), fe_sub(Z, Y)), u2 = fe_mul(X, Y);
    // SQRT_RATIO_M1(1, v), v = u1 * u2^2  (RFC 9496 4.2): r = v^3 (v^7)^((p-5)/8); fix the sign of the root by what v r^2 turns out to be
    const Fe v = fe_mul(u1, fe_sqr(u2));
    const Fe v3 = fe_mul(fe_sqr(v), v), v7 = fe_mul(fe_sqr(v3), v);
    Fe r = fe_mul(v3, fe_pow22523(v7));
    const Fp check = fe_to_fp(fe_mul(v, fe_sqr(r)));
    const Fp neg_one = fp_neg(fp_one());
    if (fp_eq(check, neg_one) || fp_eq(check, fp_mul(neg_one, fp_SQRT_M1()))) r = fe_mul(r, sqrt_m1);
    const Fe inv = fe_from_fp(fp_abs(fe_to_fp(r)));
    const Fe den1 = fe_mul(inv, u1), den2 = fe_mul(inv, u2);
    const Fe zinv = fe_mul(fe_mul(den1, den2), T);
    const bool rotate = fp_is_negative(fe_to_fp(fe_mul(T, zinv)));
    Fe x = X, y = Y, deninv = den2;
    if (rotate) { x = fe_mul(Y, sqrt_m1); y = fe_mul(X, sqrt_m1); deninv = fe_mul(den1, invsqrt_a_minus_d); }
    Fe zy = fe_sub(Z, y);
    if (fp_is_negative(fe_to_fp(fe_mul(x, zinv)))) zy = fe_add(Z, y);           // y -> -y
    fp_to_bytes(out, fp_abs(fe_to_fp(fe_mul(deninv, zy))));
}


// ------------------------------------------------------------------------------------------------ fr_inv_fast
namespace {
typedef __int128 i128;
constexpr int64_t kM62 = (int64_t)(((uint64_t)1 << 62) - 1);
struct S62 { int64_t v[5]; };                               // sum v[i] 2^(62 i); limbs 0..3 in [0, 2^62), limb 4 signed
struct T2x2 { int64_t u, v, q, r; };
// l = 2^252 + 27742317777372353535851937790883648493 in signed-62 limbs, and l^-1 mod 2^62
struct ModL { S62 m; uint64_t inv62; };
const ModL &mod_l() {
    static const ModL M = [] {
        const uint64_t w[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
        ModL r;
        r.m.v[0] = (int64_t)(w[0] & (uint64_t)kM62); r.m.v[1] = (int64_t)(((w[0] >> 62) | (w[1] << 2)) & (uint64_t)kM62); r.m.v[2] = (int64_t)(((w[1] >> 60) | (w[2] << 4)) & (uint64_t)kM62);
        r.m.v[3] = (int64_t)(((w[2] >> 58) | (w[3] << 6)) & (uint64_t)kM62); r.m.v[4] = (int64_t)(w[3] >> 56);
        uint64_t x = 1; for (int i = 0; i < 6; i++) x *= 2 - w[0] * x;        // Newton: l^-1 mod 2^64
        r.inv62 = x & (uint64_t)kM62;
        return r;
    }();
    return M;
}
// 62 division steps on the low words (variable time: runs of zero bits of g in one go), eta = -delta
int64_t divsteps_62_var(int64_t eta, uint64_t f0, uint64_t g0, T2x2 &t) {
    uint64_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 62;
    for (;;) {
        const int zeros = __builtin_ctzll(g | (~(uint64_t)0 << i));       // a sentinel bit stops the count at i
        g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {                                       // delta > 0 and g odd: (f, g) <- (g, -f)
            eta = -eta;
            uint64_t tmp = f; f = g; g = 0 - tmp;
            tmp = u; u = q; q = 0 - tmp;
            tmp = v; v = r; r = 0 - tmp;
        }
        // g odd here: one step g <- g + f (f odd), the run of zeros it creates is taken at the top of the loop
        g += f; q += u; r += v;
    }
    t.u = (int64_t)u; t.v = (int64_t)v; t.q = (int64_t)q; t.r = (int64_t)r;
    return eta;
}
// (f, g) <- t (f, g) / 2^62 (exact)
void update_fg(S62 &f, S62 &g, const T2x2 &t) {
    i128 cf = (i128)t.u * f.v[0] + (i128)t.v * g.v[0], cg = (i128)t.q * f.v[0] + (i128)t.r * g.v[0];
    cf >>= 62; cg >>= 62;
    for (int i = 1; i < 5; i++) {
        cf += (i128)t.u * f.v[i] + (i128)t.v * g.v[i]; cg += (i128)t.q * f.v[i] + (i128)t.r * g.v[i];
        f.v[i - 1] = (int64_t)cf & kM62; cf >>= 62; g.v[i - 1] = (int64_t)cg & kM62; cg >>= 62;
    }
    f.v[4] = (int64_t)cf; g.v[4] = (int64_t)cg;
}
// (d, e) <- t (d, e) / 2^62 mod l: multiples of l are added so that the division is exact; d, e stay in (-2 l, l)
void update_de(S62 &d, S62 &e, const T2x2 &t, const ModL &M) {
    const int64_t sd = d.v[4] >> 63, se = e.v[4] >> 63;
    int64_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
    i128 cd = (i128)t.u * d.v[0] + (i128)t.v * e.v[0], ce = (i128)t.q * d.v[0] + (i128)t.r * e.v[0];
    md -= (int64_t)((M.inv62 * (uint64_t)cd + (uint64_t)md) & (uint64_t)kM62);
    me -= (int64_t)((M.inv62 * (uint64_t)ce + (uint64_t)me) & (uint64_t)kM62);
    cd += (i128)M.m.v[0] * md; ce += (i128)M.m.v[0] * me;
    cd >>= 62; ce >>= 62;
    for (int i = 1; i < 5; i++) {
        cd += (i128)t.u * d.v[i] + (i128)t.v * e.v[i]; ce += (i128)t.q * d.v[i] + (i128)t.r * e.v[i];
        if (M.m.v[i]) { cd += (i128)M.m.v[i] * md; ce += (i128)M.m.v[i] * me; }
        d.v[i - 1] = (int64_t)cd & kM62; cd >>= 62; e.v[i - 1] = (int64_t)ce & kM62; ce >>= 62;
    }
    d.v[4] = (int64_t)cd; e.v[4] = (int64_t)ce;
}
}  // namespace
bool fr_inv_fast_try(const Fr &a, Fr &out) {
    const ModL &M = mod_l();
    uint64_t w[4]; memcpy(w, a.v, 32);
    out = fr_zero();
    if (!(w[0] | w[1] | w[2] | w[3])) return true;
    S62 f = M.m, g, d, e;
    g.v[0] = (int64_t)(w[0] & (uint64_t)kM62); g.v[1] = (int64_t)(((w[0] >> 62) | (w[1] << 2)) & (uint64_t)kM62); g.v[2] = (int64_t)(((w[1] >> 60) | (w[2] << 4)) & (uint64_t)kM62);
    g.v[3] = (int64_t)(((w[2] >> 58) | (w[3] << 6)) & (uint64_t)kM62); g.v[4] = (int64_t)(w[3] >> 56);
    for (int i = 0; i < 5; i++) { d.v[i] = 0; e.v[i] = 0; }
    e.v[0] = 1;
    int64_t eta = -1; bool done = false;
    for (int it = 0; it < 24 && !done; it++) {               // (12 rounds of 62 steps suffice for 256 bits; twice that before giving up)
        T2x2 t;
        eta = divsteps_62_var(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], t);
        update_de(d, e, t, M);
        update_fg(f, g, t);
        done = !(g.v[0] | g.v[1] | g.v[2] | g.v[3] | g.v[4]);
    }
    bool ok = false;
    if (done) {
        // f = +-1: the inverse of the integer a' = a R is +-d; into [0, l), then times R^3 / R: (a R)^-1 R^2 = a^-1 R
        const bool neg = f.v[4] < 0;
        i128 cy = 0; int64_t r[5];
        for (int i = 0; i < 5; i++) { cy += neg ? -(i128)d.v[i] : (i128)d.v[i]; if (i < 4) { r[i] = (int64_t)cy & kM62; cy >>= 62; } else r[4] = (int64_t)cy; }      // the top limb keeps the sign
        for (int pass = 0; pass < 3 && r[4] < 0; pass++) { cy = 0; for (int i = 0; i < 5; i++) { cy += (i128)r[i] + M.m.v[i]; if (i < 4) { r[i] = (int64_t)cy & kM62; cy >>= 62; } else r[4] = (int64_t)cy; } }
        if (r[4] >= 0) {
            const uint64_t x[5] = {(uint64_t)r[0], (uint64_t)r[1], (uint64_t)r[2], (uint64_t)r[3], (uint64_t)r[4]};
            uint64_t o[4] = {x[0] | (x[1] << 62), (x[1] >> 2) | (x[2] << 60), (x[2] >> 4) | (x[3] << 58), (x[3] >> 6) | (x[4] << 56)};
            uint32_t w32[8]; memcpy(w32, o, 32);
            const Fr inv_int = fr_cond_sub_l(w32, 0);          // below 2 l at worst
            out = fr_mul(inv_int, fr_R3());
            ok = fr_eq(fr_mul(out, a), fr_one());
        }
    }
    return ok;
}
Fr fr_inv_fast(const Fr &a) { Fr r; return fr_inv_fast_try(a, r) ? r : fr_inv(a); }

}  // namespace otti
