// ristretto255 group arithmetic (RFC 9496) on edwards25519 extended coordinates, shared by host and gfx950 device code.
// Replaces, for this path, upstream libspartan `src/group.rs` (GroupElement = dalek RistrettoPoint, CompressedGroup)
// [RECALL; /root/reference/Spartan is an empty submodule].
#pragma once
#include "field.h"

namespace otti {

struct Pt { Fp X, Y, Z, T; };                 // extended twisted Edwards, a = -1; 128 B
struct Niels { Fp yplusx, yminusx, xy2d; };   // affine precomputed form for mixed addition; 96 B
// one entry of the fixed-base window table (k_msm.hip).  OTTI_TABLE_ALIGN128 pads it to a 128-byte line of its own.
#ifdef OTTI_TABLE_ALIGN128
struct alignas(128) TabEntry { Niels n; uint32_t pad[8]; };
#else
struct TabEntry { Niels n; };
#endif

HD Pt pt_identity() { Pt p; p.X = fp_zero(); p.Y = fp_one(); p.Z = fp_one(); p.T = fp_zero(); return p; }
HD Niels niels_identity() { Niels n; n.yplusx = fp_one(); n.yminusx = fp_one(); n.xy2d = fp_zero(); return n; }

HD Pt pt_add(const Pt &p, const Pt &q) {      // 9M
    Fp a = fp_mul(fp_sub(p.Y, p.X), fp_sub(q.Y, q.X));
    Fp b = fp_mul(fp_add(p.Y, p.X), fp_add(q.Y, q.X));
    Fp c = fp_mul(fp_mul(p.T, q.T), fp_2D());
    Fp d = fp_mul(p.Z, q.Z); d = fp_add(d, d);
    Fp e = fp_sub(b, a), f = fp_sub(d, c), g = fp_add(d, c), h = fp_add(b, a);
    Pt r; r.X = fp_mul(e, f); r.Y = fp_mul(g, h); r.T = fp_mul(e, h); r.Z = fp_mul(f, g); return r;
}
HD Pt pt_madd(const Pt &p, const Niels &q) {  // 7M
    Fp a = fp_mul(fp_sub(p.Y, p.X), q.yminusx);
    Fp b = fp_mul(fp_add(p.Y, p.X), q.yplusx);
    Fp c = fp_mul(p.T, q.xy2d);
    Fp d = fp_add(p.Z, p.Z);
    Fp e = fp_sub(b, a), f = fp_sub(d, c), g = fp_add(d, c), h = fp_add(b, a);
    Pt r; r.X = fp_mul(e, f); r.Y = fp_mul(g, h); r.T = fp_mul(e, h); r.Z = fp_mul(f, g); return r;
}
HD Pt pt_msub(const Pt &p, const Niels &q) {
    Fp a = fp_mul(fp_sub(p.Y, p.X), q.yplusx);
    Fp b = fp_mul(fp_add(p.Y, p.X), q.yminusx);
    Fp c = fp_mul(p.T, q.xy2d);
    Fp d = fp_add(p.Z, p.Z);
    Fp e = fp_sub(b, a), f = fp_add(d, c), g = fp_sub(d, c), h = fp_add(b, a);
    Pt r; r.X = fp_mul(e, f); r.Y = fp_mul(g, h); r.T = fp_mul(e, h); r.Z = fp_mul(f, g); return r;
}
HD Pt pt_neg(const Pt &p) { Pt r; r.X = fp_neg(p.X); r.Y = p.Y; r.Z = p.Z; r.T = fp_neg(p.T); return r; }
HD Pt pt_sub(const Pt &p, const Pt &q) { return pt_add(p, pt_neg(q)); }
HD Pt pt_dbl(const Pt &p) {                   // 4S + 4M
    Fp a = fp_sqr(p.X), b = fp_sqr(p.Y), c = fp_sqr(p.Z); c = fp_add(c, c);
    Fp xy = fp_add(p.X, p.Y);
    Fp e = fp_sub(fp_sub(fp_sqr(xy), a), b);
    Fp g = fp_sub(b, a), f = fp_sub(g, c), h = fp_neg(fp_add(a, b));
    Pt r; r.X = fp_mul(e, f); r.Y = fp_mul(g, h); r.T = fp_mul(e, h); r.Z = fp_mul(f, g); return r;
}
// affine Niels form given 1/Z
HD Niels pt_to_niels(const Pt &p, const Fp &zinv) {
    Fp x = fp_mul(p.X, zinv), y = fp_mul(p.Y, zinv);
    Niels n; n.yplusx = fp_add(y, x); n.yminusx = fp_sub(y, x); n.xy2d = fp_mul(fp_mul(x, y), fp_2D()); return n;
}
HD Pt niels_to_pt(const Niels &n) { return pt_madd(pt_identity(), n); }

// RFC 9496 4.3.2 Encode.  On the host the inverse square root runs in the five-limb form of hostfast.h (pt_encode_fast: same bytes,
// a third of the time; it is on the prover's sequential path); pt_encode_ref is this generic code on either side.
#if !defined(__HIP_DEVICE_COMPILE__)
void pt_encode_fast(uint8_t out[32], const Pt &p);
#endif
HD void pt_encode_ref(uint8_t out[32], const Pt &p) {
    Fp u1 = fp_mul(fp_add(p.Z, p.Y), fp_sub(p.Z, p.Y));
    Fp u2 = fp_mul(p.X, p.Y);
    Fp inv; fp_sqrt_ratio_m1(inv, fp_one(), fp_mul(u1, fp_sqr(u2)));
    Fp den1 = fp_mul(inv, u1), den2 = fp_mul(inv, u2);
    Fp zinv = fp_mul(fp_mul(den1, den2), p.T);
    Fp ix = fp_mul(p.X, fp_SQRT_M1()), iy = fp_mul(p.Y, fp_SQRT_M1());
    Fp ench = fp_mul(den1, fp_INVSQRT_A_MINUS_D());
    bool rotate = fp_is_negative(fp_mul(p.T, zinv));
    Fp x = rotate ? iy : p.X, y = rotate ? ix : p.Y, deninv = rotate ? ench : den2;
    if (fp_is_negative(fp_mul(x, zinv))) y = fp_neg(y);
    Fp s = fp_abs(fp_mul(deninv, fp_sub(p.Z, y)));
    fp_to_bytes(out, s);
}
HD void pt_encode(uint8_t out[32], const Pt &p) {
#if defined(__HIP_DEVICE_COMPILE__)
    pt_encode_ref(out, p);
#else
    pt_encode_fast(out, p);
#endif
}
// RFC 9496 4.3.1 Decode; false = DecompressionError
HD bool pt_decode(Pt &o, const uint8_t b[32]) {
    if (!fp_bytes_canonical(b) || (b[0] & 1)) return false;
    Fp s = fp_from_bytes(b), ss = fp_sqr(s);
    Fp u1 = fp_sub(fp_one(), ss), u2 = fp_add(fp_one(), ss), u2s = fp_sqr(u2);
    Fp v = fp_sub(fp_neg(fp_mul(fp_D(), fp_sqr(u1))), u2s);
    Fp inv; bool was_square = fp_sqrt_ratio_m1(inv, fp_one(), fp_mul(v, u2s));
    Fp dx = fp_mul(inv, u2), dy = fp_mul(fp_mul(inv, dx), v);
    Fp x = fp_mul(s, dx); x = fp_abs(fp_add(x, x));
    Fp y = fp_mul(u1, dy), t = fp_mul(x, y);
    if (!was_square || fp_is_negative(t) || fp_is_zero(y)) return false;
    o.X = x; o.Y = y; o.Z = fp_one(); o.T = t; return true;
}
// RFC 9496 4.3.4 MAP (Elligator 2)
HD Pt pt_elligator(const Fp &t0) {
    Fp r = fp_mul(fp_sqr(t0), fp_SQRT_M1());
    Fp u = fp_mul(fp_add(r, fp_one()), fp_ONE_MINUS_D_SQ());
    Fp v = fp_mul(fp_neg(fp_add(fp_mul(r, fp_D()), fp_one())), fp_add(r, fp_D()));
    Fp s; bool was_square = fp_sqrt_ratio_m1(s, u, v);
    Fp sp = fp_neg(fp_abs(fp_mul(s, t0)));
    Fp c = fp_neg(fp_one());
    if (!was_square) { s = sp; c = r; }
    Fp n = fp_sub(fp_mul(fp_mul(c, fp_sub(r, fp_one())), fp_D_MINUS_ONE_SQ()), v);
    Fp w0 = fp_mul(s, v); w0 = fp_add(w0, w0);
    Fp w1 = fp_mul(n, fp_SQRT_AD_MINUS_ONE());
    Fp s2 = fp_sqr(s), w2 = fp_sub(fp_one(), s2), w3 = fp_add(fp_one(), s2);
    Pt p; p.X = fp_mul(w0, w3); p.Y = fp_mul(w2, w1); p.Z = fp_mul(w1, w3); p.T = fp_mul(w0, w2); return p;
}
HD Pt pt_from_uniform_bytes(const uint8_t b[64]) { return pt_add(pt_elligator(fp_from_bytes(b)), pt_elligator(fp_from_bytes(b + 32))); }
// ristretto equality (RFC 9496 4.3.3)
HD bool pt_eq(const Pt &a, const Pt &b) {
    if (fp_eq(fp_mul(a.X, b.Y), fp_mul(a.Y, b.X))) return true;
    return fp_eq(fp_mul(a.Y, b.Y), fp_mul(a.X, b.X));
}

// signed radix-2^c digits of a canonical scalar (raw words): d_w in [-2^(c-1), 2^(c-1)], sum d_w 2^(cw) = s
HD int scalar_window(const uint32_t raw[8], int pos, int c) {
    if (pos >= 256) return 0;
    int limb = pos >> 5, off = pos & 31;
    uint64_t x = raw[limb];
    if (limb < 7) x |= (uint64_t)raw[limb + 1] << 32;
    return (int)((x >> off) & ((1u << c) - 1));
}

}  // namespace otti
