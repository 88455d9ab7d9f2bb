// Host build of the nine-limb field arithmetic (otti_amd/csrc/fr9.h, fp9.h: the C bodies; the device uses generated asm blocks that
// tools/limbbench and the GPU parity tests cover) checked against the 8 x u32 arithmetic of field.h / point.h on random and edge inputs.
// Built and run by tests/test_limb9_host.py (g++, no GPU).
#include "fr9.h"
#include "fp9.h"
#include <stdio.h>
#include <random>
using namespace otti;
static std::mt19937_64 rng(20261004);
static Fr rnd_fr() { Fr t; for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)rng(); t.v[7] &= 0x0fffffffu; uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = t.v[i]; return fr_cond_sub_l(w, 0); }
static Fp rnd_fp() { Fp t; for (int i = 0; i < 8; i++) t.v[i] = (uint32_t)rng(); return t; }
static Fr l_minus(uint32_t k) { Fr a; for (int i = 0; i < 8; i++) a.v[i] = fr_L(i); a.v[0] -= k; return a; }
#define CHECK(cond, what) do { if (!(cond)) { if (bad < 10) printf("FAIL %s at iteration %d\n", what, it); bad++; } } while (0)
int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        Fr a = rnd_fr(), b = rnd_fr(), c = rnd_fr();
        if (it == 0) a = fr_zero();
        if (it == 1) { a = l_minus(1); b = a; c = a; }
        if (it == 2) { a = fr_one(); b = l_minus(1); }
        const Fr9 a9 = fr9_unpack(a), b9 = fr9_unpack(b), c5 = fr9_unpack5(c);
        // products, the radix correction on either operand, computed operands shifted afterwards
        const Fr9 p = fr9_mul(fr9_unpack5(a), b9);
        const Fr ab = fr_mul(a, b);
        CHECK(fr_eq(fr9_pack_lt2l(p), ab), "mul");
        CHECK(fr_eq(fr9_pack_lt2l(fr9_mul(fr9_shl5(p), fr9_unpack(c))), fr_mul(ab, c)), "shl5");
        CHECK(fr_eq(fr9_canon(fr9_mul(fr9_unpack_s<10>(a), fr9_mul(b9, fr9_unpack(c)))), fr_mul(a, fr_mul(b, c))), "unpack_s<10>");
        // differences with offsets, folds
        CHECK(fr_eq(fr9_canon(fr9_norm(fr9_sub_kl<2>(a9, b9))), fr_sub(a, b)), "sub 2l");
        CHECK(fr_eq(fr9_canon(fr9_norm(fr9_sub_kl<4>(a9, b9))), fr_sub(a, b)), "sub 4l");
        CHECK(fr_eq(fr9_canon(fr9_norm(fr9_sub_kl<16>(a9, b9))), fr_sub(a, b)), "sub 16l");
        CHECK(fr_eq(fr9_canon(fr9_norm(fr9_sub_kl<128>(fr9_unpack5(a), fr9_unpack5(b)))), fr_mul(fr_sub(a, b), fr_from_u64(32))), "sub 128l");
        const Fr9 f = fr9_norm(fr9_add(a9, fr9_mul(c5, fr9_sub_kl<2>(b9, a9))));
        CHECK(fr_eq(fr9_pack_lt3l(f), fr_add(a, fr_mul(c, fr_sub(b, a)))), "fold");
        // running sums carried down every fourth item, any normalised value back to the canonical word
        if (it % 16 == 0) {
            Fr9 acc = fr9_zero(); Fr want = fr_zero();
            for (int k = 0; k < 70; k++) { const Fr x = rnd_fr(); acc = fr9_add(acc, fr9_mul(c5, fr9_sub_kl<16>(fr9_unpack(x), a9))); want = fr_add(want, fr_mul(c, fr_sub(x, a))); if ((k & 3) == 3) acc = fr9_norm(acc); }
            CHECK(fr_eq(fr9_canon(fr9_norm(acc)), want), "accumulate");
        }
        // GF(2^255 - 19)
        const Fp x = it == 3 ? fp_from_words(~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u) : rnd_fp(), y = rnd_fp();
        CHECK(fp_eq(f9_pack(f9_mul(f9_unpack(x), f9_unpack(y))), fp_mul(x, y)), "f9_mul");
        CHECK(fp_eq(f9_pack(f9_sub(f9_unpack(x), f9_unpack(y))), fp_sub(x, y)), "f9_sub");
        CHECK(fp_eq(f9_pack(f9_add(f9_unpack(x), f9_unpack(y))), fp_add(x, y)), "f9_add");
    }
    // a walk of mixed additions / subtractions against pt_madd / pt_msub (the formulas are polynomial identities: any field elements serve as operands)
    Pt P; P.X = rnd_fp(); P.Y = rnd_fp(); P.Z = rnd_fp(); P.T = rnd_fp();
    P9 Q = p9_unpack(P);
    for (int it = 0; it < iters; it++) {
        Niels n; n.yplusx = rnd_fp(); n.yminusx = rnd_fp(); n.xy2d = rnd_fp();
        const bool neg = rng() & 1;
        N9 e = n9_unpack(n); if (neg) e = n9_negate(e);
        Q = p9_madd(Q, e); P = neg ? pt_msub(P, n) : pt_madd(P, n);
        for (int i = 0; i < 8; i++) CHECK(Q.X.v[i] < (1u << 29) + (1u << 18) && Q.Y.v[i] < (1u << 29) + (1u << 18) && Q.Z.v[i] < (1u << 29) + (1u << 18) && Q.T.v[i] < (1u << 29) + (1u << 18), "reduced bound");
        if (it % 61 == 0 || it < 40) { const Pt q = p9_pack(Q); CHECK(fp_eq(q.X, P.X) && fp_eq(q.Y, P.Y) && fp_eq(q.Z, P.Z) && fp_eq(q.T, P.T), "p9_madd"); }
    }
    printf("limb9 check: %d failures\n", bad);
    return bad != 0;
}
