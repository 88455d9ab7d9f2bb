"""Kernel-level parity for the kernels the PROVER launches (phase one with the eq table factored out, the eq pyramids, the evaluation
proof's bound, the bullet-reduction rounds on the original generators) and for the device-pointer / caller's-stream form of the
kernel ABI — each against the matching restatement in oracle/ (bit-exact: integer arithmetic).  A whole proof that differs from the
oracle's says nothing about where; these do.  Shapes include the slices a sharded prover runs (short tables, tau suffixes)."""
import numpy as np
import pytest

import otti_amd as oa
import orc

pytestmark = pytest.mark.gpu
K, KD = oa.kernels, oa.kernels_dev


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


def fr_inv(x):
    return orc.fr_from_ints([pow(v, -1, orc.L_ORDER) for v in orc.fr_to_ints(x)])


# ------------------------------------------------------------------------------------------------ eq pyramids (k_eq_tree: two small pyramids in LDS, the large levels as outer products)
@pytest.mark.parametrize("n", [0, 1, 2, 5, 6, 7, 8, 12, 13])
def test_eq_pyramid_levels_are_eq_tables_of_the_suffixes(rng, n):
    r = orc.rand_fr(rng, n)
    pyr = K.eq_pyramid(r)
    for k in range(n + 1):                                     # level k: eq over the LAST k variables
        assert eq(pyr[(1 << k) - 1: (2 << k) - 1], orc.eq_evals(r[n - k:])), k


# ------------------------------------------------------------------------------------------------ phase one, three tables (k_sc_cubic3_*)
def _cubic3_want(E, B, C, D):
    return orc.sc_cubic_evals(np.concatenate([E, E]), B, C, D)   # an eq table that is constant in the bound variable


@pytest.mark.parametrize("n", [2, 4, 64, 1 << 10, 1 << 13, 1 << 14, 1 << 17])
def test_cubic3_round_equals_four_table_round_with_constant_eq(rng, n):
    """2^13 / 2^14 straddle the switch from one pyramid to the (hi x lo) product of two; 2^17 runs a multi-workgroup grid."""
    B, C, D = (orc.rand_fr(rng, n) for _ in range(3)); tau = orc.rand_fr(rng, n.bit_length() - 2)
    got, _ = K.sc_cubic3_round(B, C, D, tau)
    assert eq(got, _cubic3_want(orc.eq_evals(tau), B, C, D))


@pytest.mark.parametrize("n", [4, 8, 128, 1 << 11, 1 << 14, 1 << 15, 1 << 17])
def test_cubic3_fused_fold_round(rng, n):
    B, C, D = (orc.rand_fr(rng, n) for _ in range(3)); r = orc.rand_fr(rng, 1); tau = orc.rand_fr(rng, n.bit_length() - 3)
    fb, fc, fd = (orc.fold_top(x, r) for x in (B, C, D))
    out, e, _ = K.sc_cubic3_fold_round(B, C, D, r, tau)
    assert eq(out[0], fb) and eq(out[1], fc) and eq(out[2], fd)
    assert eq(e, _cubic3_want(orc.eq_evals(tau), fb, fc, fd))


def _edge_tables(rng, n, count):
    """tables made of the values that sit at the ends of the nine-limb forms' ranges: 0, 1, l - 1, l - 2, 2^252, and a few random ones"""
    l = orc.L_ORDER
    pool = [0, 1, 2, l - 1, l - 2, 1 << 252, (1 << 252) - 1, (l - 1) // 2, (1 << 29) - 1, 1 << 29, (1 << 232) - 1, 1 << 232]
    out = []
    for t in range(count):
        kind = t % 4
        if kind == 0:
            vals = [l - 1] * n
        elif kind == 1:
            vals = [pool[int(x)] for x in rng.integers(0, len(pool), n)]
        elif kind == 2:
            vals = [0] * n
        else:
            vals = [pool[int(x)] if x < len(pool) else int(rng.integers(0, 2 ** 62)) ** 4 % l for x in rng.integers(0, 2 * len(pool), n)]
        out.append(orc.fr_from_ints(vals))
    return out


@pytest.mark.parametrize("n,shift", [(8, 0), (1 << 11, 1), (1 << 15, 2), (1 << 17, 3)])
def test_round_kernels_on_values_at_the_ends_of_the_field(rng, n, shift):
    """every round kernel of both sum-checks (three- and four-table cubic, quad; plain and fused with the fold) on tables full of l - 1, 0, 1,
    2^252, limb-boundary values and mixtures — the nine-limb arithmetic's offsets, carries and accumulations at their extremes"""
    l = orc.L_ORDER
    T = _edge_tables(rng, n, 8)
    T = T[shift:] + T[:shift]
    A, B, C, D = T[0], T[1], T[2], T[3]
    for r_int in (l - 1, 1, 0, int(rng.integers(0, 2 ** 62)) ** 4 % l):
        r = orc.fr_from_ints([r_int])
        assert eq(K.sc_cubic_round(A, B, C, D)[0], orc.sc_cubic_evals(A, B, C, D))
        assert eq(K.sc_quad_round(A, B)[0], orc.sc_quad_evals(A, B))
        fa, fb, fc, fd = (orc.fold_top(x, r) for x in (A, B, C, D))
        out, e, _ = K.sc_cubic_fold_round(A, B, C, D, r)
        assert all(eq(out[k], f) for k, f in enumerate((fa, fb, fc, fd))) and eq(e, orc.sc_cubic_evals(fa, fb, fc, fd))
        out2, e2, _ = K.sc_quad_fold_round(A, B, r)
        assert eq(out2[0], fa) and eq(out2[1], fb) and eq(e2, orc.sc_quad_evals(fa, fb))
        for tau_int in (l - 1, 1):
            tau = orc.fr_from_ints([tau_int] * (n.bit_length() - 2))
            assert eq(K.sc_cubic3_round(B, C, D, tau)[0], _cubic3_want(orc.eq_evals(tau), B, C, D))
            if n >= 8:
                out3, e3, _ = K.sc_cubic3_fold_round(B, C, D, r, tau[1:])
                assert eq(out3[0], fb) and eq(out3[1], fc) and eq(out3[2], fd) and eq(e3, _cubic3_want(orc.eq_evals(tau[1:]), fb, fc, fd))
        A, B, C, D = B, C, D, A


def test_cubic3_rounds_chain_like_phase_one(rng):
    """the prover's loop: evaluate, then fold+evaluate per challenge, with E_j = eq(tau[j+1:]); the scalar factors the host applies
    (c_j and the bound variable's eq factor) turn S_t into upstream's four-table sums"""
    s, l = 9, orc.L_ORDER
    n = 1 << s
    tau = orc.rand_fr(rng, s); ti = orc.fr_to_ints(tau)
    T = [orc.rand_fr(rng, n) for _ in range(3)]
    full = [orc.eq_evals(tau)] + [t.copy() for t in T]       # upstream's four tables
    cj = 1
    e, _ = K.sc_cubic3_round(T[0], T[1], T[2], tau[1:])
    for j in range(s):
        S = orc.fr_to_ints(e)
        w0 = (1 - ti[j]) % l; dw = (2 * ti[j] - 1) % l
        mine = [cj * w0 * S[0] % l, cj * (w0 + 2 * dw) * S[1] % l, cj * (w0 + 3 * dw) * S[2] % l]
        assert mine == orc.fr_to_ints(orc.sc_cubic_evals(*full)), j
        r = orc.rand_fr(rng, 1); ri = orc.fr_to_ints(r)[0]
        full = [orc.fold_top(t, r) for t in full]
        cj = cj * (ti[j] * ri + (1 - ti[j]) * (1 - ri)) % l
        if len(T[0]) >= 4:
            out, e, _ = K.sc_cubic3_fold_round(T[0], T[1], T[2], r, tau[j + 2:])
            T = [out[0], out[1], out[2]]
            assert all(eq(T[k], full[k + 1]) for k in range(3))
        else:
            break


# ------------------------------------------------------------------------------------------------ DensePolynomial::bound (k_poly_bound_*)
@pytest.mark.parametrize("L,R", [(1, 2), (2, 2), (16, 32), (64, 64), (100, 256), (512, 1024), (8, 1024)])
def test_poly_bound(rng, L, R):
    """(8, 1024): the 1/g row block of a sharded evaluation proof"""
    Z, Lv = orc.rand_fr(rng, L * R), orc.rand_fr(rng, L)
    assert eq(K.poly_bound(Z, L, R, Lv)[0], orc.poly_bound(Z, L, R, Lv))


# ------------------------------------------------------------------------------------------------ bullet reduction rounds (k_msm_rows<1> bullet mode, k_bullet_step)
@pytest.mark.parametrize("lgn", [2, 5, 8, 10])
def test_bullet_rounds_on_original_generators_equal_the_folding_reduction(rng, lgn):
    """Every round's L, R and folded a, b against nizk/bullet.rs as restated by the oracle (which folds the generator vector); after
    the last fold, sum_j s[j] P[j] must be the oracle's folded generator g_hat."""
    V = 1 << (2 * lgn); n = 1 << lgn
    gens, og = oa.NIZKGens.new(V, V, 1), orc.OGens(V, V, 1)
    assert og.R == n
    a, b = orc.rand_fr(rng, n), orc.rand_fr(rng, n)
    blinds, us = orc.rand_fr(rng, 2 * lgn), orc.rand_fr(rng, lgn); uis = fr_inv(us)
    one = orc.fr_from_ints([1])
    s = np.repeat(one, n, axis=0)
    ca, cb, cur = a, b, n
    for k in range(lgn):
        fold = k > 0
        LR, ca, cb, s, _ = K.bullet_round(gens, cur, ca, cb, s, blinds[2 * k: 2 * k + 2], us[k - 1: k] if fold else None, uis[k - 1: k] if fold else None)
        wLR, wa, wb, _ = orc.bullet_reduce(og, a, b, blinds, us[: k + 1])
        assert eq(LR, wLR[k]), ("L/R of round", k)
        if fold:
            pa, pb = orc.bullet_reduce(og, a, b, blinds, us[:k])[1:3]
            assert eq(ca, pa) and eq(cb, pb), ("state before round", k)
        cur //= 2
    # closing fold (2 -> 1)
    a1, b1, s = K.bullet_last_fold(np.concatenate([ca[:2]]), np.concatenate([cb[:2]]), s, us[lgn - 1: lgn], uis[lgn - 1: lgn])
    _, wa, wb, wG = orc.bullet_reduce(og, a, b, blinds, us)
    assert eq(a1, wa) and eq(b1, wb)
    g_hat, _ = K.msm_rows(gens, s, 1, n, orc.fr_from_ints([0]))
    assert eq(g_hat, wG)


# ------------------------------------------------------------------------------------------------ device pointers + caller's stream
def test_device_pointer_entry_points_match_the_staged_ones(rng):
    stream = KD.stream_create()
    try:
        for st in (None, stream):
            n = 1 << 12
            A, B, C, D = (orc.rand_fr(rng, n) for _ in range(4)); r = orc.rand_fr(rng, 1)
            dA, dB, dC, dD = (oa.DeviceArray.from_host(x) for x in (A, B, C, D))
            assert eq(KD.sc_cubic_round(dA, dB, dC, dD, n, st), orc.sc_cubic_evals(A, B, C, D))
            assert eq(KD.sc_quad_round(dA, dB, n, st), orc.sc_quad_evals(A, B))
            e = KD.sc_cubic_fold_round(dA, dB, dC, dD, n, r, st)
            fa, fb, fc, fd = (orc.fold_top(x, r) for x in (A, B, C, D))
            assert eq(e, orc.sc_cubic_evals(fa, fb, fc, fd)) and eq(dA.to_host(n // 2), fa) and eq(dD.to_host(n // 2), fd)
            r2 = orc.rand_fr(rng, 1)
            e2 = KD.sc_quad_fold_round(dA, dB, n // 2, r2, st)
            ga, gb = orc.fold_top(fa, r2), orc.fold_top(fb, r2)
            assert eq(e2, orc.sc_quad_evals(ga, gb)) and eq(dB.to_host(n // 4), gb)
            KD.fold_top(dC, n // 2, r2, st)
            out = oa.DeviceArray(n // 4); KD.fold_bot(dD, out, n // 2, r2, st)
            if st is not None:
                KD.stream_sync(st)
            assert eq(dC.to_host(n // 4), orc.fold_top(fc, r2)) and eq(out.to_host(), orc.fold_bot(fd, r2))
            rr = orc.rand_fr(rng, 14); dE = oa.DeviceArray(1 << 14)
            KD.eq_evals(rr, dE, st)
            assert eq(dE.to_host(), orc.eq_evals(rr))
            # sparse products and the commitment on an instance
            m = 1 << 10
            R1 = oa.synth_r1cs_compiler_like(m, 4, 3)
            inst = oa.Instance.new(R1["num_cons"], R1["num_vars"], R1["num_inputs"], R1["A"], R1["B"], R1["C"])
            oi = orc.OInstance(R1["num_cons"], R1["num_vars"], R1["num_inputs"], R1["A"], R1["B"], R1["C"])
            z = orc.rand_fr(rng, 2 * m); dz = oa.DeviceArray.from_host(z); o = [oa.DeviceArray(m) for _ in range(3)]
            KD.multiply_vec(inst, dz, o[0], o[1], o[2], st)
            ex = orc.rand_fr(rng, m); dex = oa.DeviceArray.from_host(ex); rabc = orc.rand_fr(rng, 3); dout = oa.DeviceArray(2 * m)
            KD.eval_table_sparse(inst, dex, rabc, dout, st)
            gens, og = oa.NIZKGens.new(m, m, 4), orc.OGens(m, m, 4)
            Rr = og.R; Lr = m // Rr
            Zm, bl = orc.rand_fr(rng, m), orc.rand_fr(rng, Lr)
            dZ, dbl, dpts = oa.DeviceArray.from_host(Zm), oa.DeviceArray.from_host(bl), oa.DeviceArray(Lr)
            KD.msm_rows(gens, dZ, Lr, Rr, dbl, dpts, st)
            if st is not None:
                KD.stream_sync(st)
            want = orc.multiply_vec(oi, z)
            assert all(eq(o[k].to_host(), want[k]) for k in range(3))
            ea, eb, ec = orc.eval_table_sparse(oi, ex); l = orc.L_ORDER
            ra, rb, rc = orc.fr_to_ints(rabc)
            comb = [(ra * x + rb * y + rc * w) % l for x, y, w in zip(orc.fr_to_ints(ea), orc.fr_to_ints(eb), orc.fr_to_ints(ec))]
            assert orc.fr_to_ints(dout.to_host()) == comb
            assert eq(dpts.to_host(), orc.commit_rows(og, Zm, Lr, Rr, bl))
    finally:
        KD.stream_destroy(stream)


# ------------------------------------------------------------------------------------------------ armed launches (device.h)
@pytest.mark.parametrize("n,hold_us", [(8, 0), (1 << 10, 300), (1 << 16, 2000)])
def test_armed_round_released_aborted_and_reused(rng, n, hold_us):
    """A round queued before its challenge is known: released after the kernel has waited hold_us, aborted (tables untouched, stream
    drains, nothing delivered), and armed again afterwards.  The library compares the three ways among themselves; the plain run is
    checked against the oracle here.  n = 2^16 folds in 64 workgroups (the widest grid the provers arm): the non-leader path."""
    A, B = orc.rand_fr(rng, n), orc.rand_fr(rng, n); r = orc.rand_fr(rng, 1)
    out, e = K.armed_selftest(A, B, r, hold_us)
    fa, fb = orc.fold_top(A, r), orc.fold_top(B, r)
    assert eq(out[0], fa) and eq(out[1], fb) and eq(e, orc.sc_quad_evals(fa, fb))


# ------------------------------------------------------------------------------------------------ verifier: variable-base row sums (k_decode_niels, k_msm_var)
@pytest.mark.parametrize("n,lgv", [(256, 16), (512, 18), (1024, 20), (2048, 22), (4096, 24)])
def test_row_sum_on_device_equals_the_oracles_commitment(rng, n, lgv):
    """PolyEvalProof::verify's C_LZ = sum_i L[i] * C_i on the device: batch decompression, LDS-bucket Pippenger (one and several splits,
    c = 6 ... 9), host window recombination.  Points with a known answer: the generator stream itself, whose commitment the oracle forms
    its own way (orc_commit_rows, one row, blind 0); scalars uniform with the special shapes the recoding must survive."""
    og = orc.OGens(1 << lgv, 1 << lgv, 1)
    assert og.R == n
    C = og.points()[:n]
    s = orc.rand_fr(rng, n)
    special = orc.fr_from_ints([0, 1, orc.L_ORDER - 1, 255, 256, 2 ** 252, int("80" * 31, 16), 2 ** 128 - 1])
    s[:len(special)] = special
    want = orc.commit_rows(og, s, 1, n, orc.fr_from_ints([0]))
    assert eq(K.row_sum(C, s), want[0])
    z = orc.fr_from_ints([0] * n)                              # all-zero scalars: the identity
    assert eq(K.row_sum(C, z), np.zeros(32, dtype=np.uint8))
    bad = C.copy(); bad[n // 3] = np.frombuffer(bytes([1] + [0] * 31), dtype=np.uint8)     # s = 1 is odd: not a ristretto255 encoding
    with pytest.raises(oa.ProofVerifyError):
        K.row_sum(bad, s)
    bad = C.copy(); bad[n - 1, 31] |= 0x80                     # non-canonical field element
    with pytest.raises(oa.ProofVerifyError):
        K.row_sum(bad, s)
