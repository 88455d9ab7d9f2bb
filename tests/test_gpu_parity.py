"""GPU parity tests proper: every HIP kernel and the whole proof, called through the C ABI, against the CPU oracle
(oracle/, plain C) on the same seeded inputs.  Bit-exact: all arithmetic on this path is integer (GF(l), GF(2^255-19))."""
import hashlib
import numpy as np
import pytest

import otti_amd as oa
import orc

pytestmark = pytest.mark.gpu
K = oa.kernels


def setup_module(module):
    assert oa.device_count() >= 1, "no MI355X visible"


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


# ------------------------------------------------------------------------------------------------ field ops
@pytest.mark.parametrize("n", [1, 63, 64, 1000, 1 << 16])
def test_fr_ops(rng, n):
    a, b = orc.rand_fr(rng, n), orc.rand_fr(rng, n)
    xa, xb = orc.fr_to_ints(a), orc.fr_to_ints(b)
    for op, f in (("mul", lambda x, y: x * y), ("add", lambda x, y: x + y), ("sub", lambda x, y: x - y)):
        out, _ = K.fr_op(op, a, b)
        assert orc.fr_to_ints(out) == [f(x, y) % orc.L_ORDER for x, y in zip(xa, xb)], op


def test_fr_edge_values():
    l = orc.L_ORDER
    vals = [0, 1, 2, l - 1, l - 2, (l - 1) // 2, 2 ** 252, 2 ** 128, 2 ** 32 - 1, 2 ** 64 - 1]
    a = orc.fr_from_ints([x for x in vals for _ in vals]); b = orc.fr_from_ints([y for _ in vals for y in vals])
    xa, xb = orc.fr_to_ints(a), orc.fr_to_ints(b)
    for op, f in (("mul", lambda x, y: x * y), ("add", lambda x, y: x + y), ("sub", lambda x, y: x - y)):
        out, _ = K.fr_op(op, a, b)
        assert orc.fr_to_ints(out) == [f(x, y) % l for x, y in zip(xa, xb)]


def test_canonical_roundtrip(rng):
    a = orc.rand_fr(rng, 500)
    canon = K.to_canonical(a)
    assert [int.from_bytes(c.tobytes(), "little") for c in canon] == orc.fr_to_ints(a)
    assert eq(K.from_canonical(canon), a)


# ------------------------------------------------------------------------------------------------ K2 eq tables
@pytest.mark.parametrize("ell", [0, 1, 2, 5, 6, 7, 10, 12, 13, 14, 16, 19, 25])
def test_eq_evals(rng, ell):
    r = orc.rand_fr(rng, ell)
    out, _ = K.eq_evals(r)
    assert eq(out, orc.eq_evals(r))


# ------------------------------------------------------------------------------------------------ K4/K5 folds, K3/K7 round sums
@pytest.mark.parametrize("n", [2, 4, 8, 1 << 10, 1 << 15])
def test_fold_top_bot(rng, n):
    Z, r = orc.rand_fr(rng, n), orc.rand_fr(rng, 1)
    assert eq(K.fold_top(Z, r)[0], orc.fold_top(Z, r))
    assert eq(K.fold_bot(Z, r)[0], orc.fold_bot(Z, r))


@pytest.mark.parametrize("n", [2, 4, 64, 1 << 10, 1 << 16])
def test_sumcheck_round_sums(rng, n):
    A, B, C, D = (orc.rand_fr(rng, n) for _ in range(4))
    assert eq(K.sc_cubic_round(A, B, C, D)[0], orc.sc_cubic_evals(A, B, C, D))
    assert eq(K.sc_quad_round(A, B)[0], orc.sc_quad_evals(A, B))


@pytest.mark.parametrize("n", [4, 8, 1 << 10, 1 << 16])
def test_sumcheck_fused_fold_round(rng, n):
    A, B, C, D = (orc.rand_fr(rng, n) for _ in range(4)); r = orc.rand_fr(rng, 1)
    fa, fb, fc, fd = (orc.fold_top(x, r) for x in (A, B, C, D))
    out, e, _ = K.sc_cubic_fold_round(A, B, C, D, r)
    assert eq(out[0], fa) and eq(out[1], fb) and eq(out[2], fc) and eq(out[3], fd)
    assert eq(e, orc.sc_cubic_evals(fa, fb, fc, fd))
    out2, e2, _ = K.sc_quad_fold_round(A, B, r)
    assert eq(out2[0], fa) and eq(out2[1], fb)
    assert eq(e2, orc.sc_quad_evals(fa, fb))


# ------------------------------------------------------------------------------------------------ K1/K6 sparse products
def _random_instance(rng, nc, nv, ni, nnz_per_row, heavy_rows=()):
    """ragged random matrices (not satisfiable — only multiply_vec / eval tables are exercised)"""
    mats = []
    for _ in range(3):
        rows, cols = [], []
        for r in range(nc):
            k = int(rng.integers(0, nnz_per_row + 1))
            if r in heavy_rows:
                k = 300
            rows += [r] * k; cols += list(rng.integers(0, nv + 1 + ni, size=k))
        e = np.zeros(len(rows), dtype=oa.ENTRY_DTYPE)
        e["row"], e["col"] = rows, cols
        vals = rng.integers(0, 256, size=(len(rows), 32), dtype=np.uint8); vals[:, 31] &= 0x0f
        e["val"] = vals
        mats.append(e)
    return mats


@pytest.mark.parametrize("nc,nv,ni,nnz,heavy", [(8, 8, 2, 3, ()), (1000, 700, 10, 4, (5, 77)), (1 << 12, 1 << 12, 10, 2, (0,))])
def test_multiply_vec_and_eval_table(rng, nc, nv, ni, nnz, heavy):
    A, B, C = _random_instance(rng, nc, nv, ni, nnz, heavy)
    inst, oinst = oa.Instance.new(nc, nv, ni, A, B, C), orc.OInstance(nc, nv, ni, A, B, C)
    ncp, nvp, _ = inst.dims
    assert (ncp, nvp) == (oinst.num_cons, oinst.num_vars)
    z = orc.rand_fr(rng, 2 * nvp)
    ga, gb, gc, _ = K.multiply_vec(inst, z)
    oa_, ob_, oc_ = orc.multiply_vec(oinst, z)
    assert eq(ga, oa_) and eq(gb, ob_) and eq(gc, oc_)
    eqrx, coef = orc.rand_fr(rng, ncp), orc.rand_fr(rng, 3)
    got, _ = K.eval_table_sparse(inst, eqrx, coef)
    eA, eB, eC = (orc.fr_to_ints(x) for x in orc.eval_table_sparse(oinst, eqrx))
    c = orc.fr_to_ints(coef)
    want = [(c[0] * a + c[1] * b + c[2] * d) % orc.L_ORDER for a, b, d in zip(eA, eB, eC)]
    assert orc.fr_to_ints(got) == want


# ------------------------------------------------------------------------------------------------ K8 MSM rows (Pedersen commitments)
@pytest.mark.parametrize("lg", [2, 6, 10])
def test_msm_rows(rng, lg):
    nv = 1 << lg
    gens, ogens = oa.NIZKGens.new(nv, nv, 1), orc.OGens(nv, nv, 1)
    R = ogens.R; L = nv // R
    assert eq(gens.points(R + 2), ogens.points())
    Z, blinds = orc.rand_fr(rng, L * R), orc.rand_fr(rng, L)
    # scalars with special shapes: zero, one, l-1, small, all-ones digits
    special = orc.fr_from_ints([0, 1, orc.L_ORDER - 1, 255, 256, 2 ** 252, int("80" * 31, 16)])
    Z[: min(len(special), L * R)] = special[: min(len(special), L * R)]
    got, _ = K.msm_rows(gens, Z, L, R, blinds)
    assert eq(got, orc.commit_rows(ogens, Z, L, R, blinds))


@pytest.mark.parametrize("lg,window,small_share", [(16, None, 0.0), (16, None, 0.9), (17, "12", 0.0), (17, "12", 0.9), (16, "9", 0.95), (16, "16", 1.0)])
def test_msm_rows_bulk_variants(rng, monkeypatch, lg, window, small_share):
    """The commitment-sized launches: k_msm_rows<MSM_BULK> (one lane per (term, window) pair) and <MSM_BULK_SPARSE> (compacted work
    list, picked when the scalars are mostly small), for several window widths, odd/even log sizes and ragged last chunks."""
    if window:
        monkeypatch.setenv("OTTI_MSM_WINDOW", window)
    nv = 1 << lg
    gens, ogens = oa.NIZKGens.new(nv, nv, 1), orc.OGens(nv, nv, 1)
    R = ogens.R; L = nv // R
    big = orc.fr_to_ints(orc.rand_fr(rng, L * R))
    kinds = rng.random(L * R)
    vals = []
    for x, k in zip(big, kinds):
        if k >= small_share:
            vals.append(x)
        elif k < 0.3 * small_share:
            vals.append(int(x) & 1)                                      # bits
        elif k < 0.6 * small_share:
            vals.append(int(x) % (1 << 64))
        elif k < 0.8 * small_share:
            vals.append(int(x) % (1 << 127))
        else:
            vals.append(0)
    vals[:4] = [orc.L_ORDER - 1, 1, 0, (1 << 128) - 1]
    Z, blinds = orc.fr_from_ints(vals), orc.rand_fr(rng, L)
    got, _ = K.msm_rows(gens, Z, L, R, blinds)
    assert eq(got, orc.commit_rows(ogens, Z, L, R, blinds))


def test_window_table_falls_back_when_hbm_is_short(rng):
    """The widest window the budget allows needs a 51.6 GB table at 2^20; with most of the HBM taken the library must settle for a
    narrower window instead of failing, and the commitments must not change."""
    import ctypes
    oa.device_count()
    # the HIP runtime the library itself is linked against (a second runtime in the process, e.g. torch's bundled one, may not be
    # able to open the device once the first one has)
    free, total, hip = ctypes.c_size_t(), ctypes.c_size_t(), None
    for path in sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}):
        cand = ctypes.CDLL(path)
        if cand.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0:
            hip = cand
            break
    assert hip is not None, "no loaded HIP runtime sees the device"
    hog = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(hog), ctypes.c_size_t(max(1 << 20, free.value - (30 << 30)))) == 0      # leave ~30 GB
    try:
        nv = 1 << 20
        gens = oa.NIZKGens.new(nv, nv, 1)
        gens_small, ogens = oa.NIZKGens.new(1 << 12, 1 << 12, 1), orc.OGens(nv, nv, 1)
        oa.lib.otti_prepare_device(None, gens._h)
        c, nbytes = gens.table_info
        assert 8 <= c < 16 and nbytes < (30 << 30), (c, nbytes)
        del gens_small
        R = ogens.R; L = 8                                                               # a few rows are enough to pin the table contents
        Z, blinds = orc.rand_fr(rng, L * R), orc.rand_fr(rng, L)
        got, _ = K.msm_rows(gens, Z, L, R, blinds)
        assert eq(got, orc.commit_rows(ogens, Z, L, R, blinds))
    finally:
        hip.hipFree(hog)


def test_window_table_can_be_released_and_comes_back(rng):
    """otti_gens_release_device frees the table (a process moving between instance sizes: two wide tables do not fit one card); the next
    use builds it again — same width, same commitments — and otti_gens_build_ms says what the build took."""
    nv = 1 << 14
    gens, ogens = oa.NIZKGens.new(nv, nv, 1), orc.OGens(nv, nv, 1)
    assert gens.table_info == (0, 0) and gens.build_ms == (0.0, 0.0)
    R = ogens.R; L = 4
    Z, blinds = orc.rand_fr(rng, L * R), orc.rand_fr(rng, L)
    want = orc.commit_rows(ogens, Z, L, R, blinds)
    assert eq(K.msm_rows(gens, Z, L, R, blinds)[0], want)
    c, nbytes = gens.table_info
    assert c >= 8 and nbytes > 0 and gens.build_ms[1] > 0
    gens.release_device()
    assert gens.table_info == (0, 0)
    assert eq(K.msm_rows(gens, Z, L, R, blinds)[0], want)
    assert gens.table_info == (c, nbytes)


# ------------------------------------------------------------------------------------------------ whole proof
def _prove_both(n, ni, seed=b"\x2a" * 32, label=b"nizk_example"):
    r = oa.synth_r1cs(n, ni, 1)
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    vars_, inputs = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    proof = oa.NIZK.prove(inst, vars_, inputs, gens, label, seed)
    oinst = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    ogens = orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
    oproof, _ = orc.nizk_prove(oinst, r["vars"], r["inputs"], ogens, label, seed)
    return r, inst, gens, inputs, proof, oinst, ogens, oproof


@pytest.mark.parametrize("n,ni", [(2, 0), (4, 1), (16, 3), (64, 10), (1 << 10, 10), (1 << 14, 10)])
def test_proof_bytes_identical_to_oracle(n, ni):
    r, inst, gens, inputs, proof, oinst, ogens, oproof = _prove_both(n, ni)
    assert proof.bytes == oproof, f"GPU proof differs from oracle (sha {hashlib.sha256(proof.bytes).hexdigest()[:16]} vs {hashlib.sha256(oproof).hexdigest()[:16]})"
    proof.verify(inst, inputs, gens)                                        # product verifier accepts
    assert orc.nizk_verify(oinst, r["inputs"], ogens, proof.bytes) == 0     # oracle verifier accepts the GPU proof


def test_proof_nonsquare_and_padded():
    # num_vars not a power of two, fewer constraints than variables: exercises padding and the column shift of Instance::new
    r = oa.synth_r1cs(24, 5, 3)
    nv = 40
    vars_pad = np.zeros((nv, 32), dtype=np.uint8); vars_pad[:24] = r["vars"]
    A, B, C = r["A"].copy(), r["B"].copy(), r["C"].copy()
    for m in (A, B, C):
        m["col"] = np.where(m["col"] >= 24, m["col"] + (nv - 24), m["col"])
    inst = oa.Instance.new(24, nv, 5, A, B, C)
    assert inst.dims == (32, 64, 5)
    gens = oa.NIZKGens.new(24, nv, 5)
    v, i = oa.VarsAssignment.new(vars_pad), oa.InputsAssignment.new(r["inputs"])
    assert inst.is_sat(v, i)
    proof = oa.NIZK.prove(inst, v, i, gens, b"pad", b"\x07" * 32)
    proof.verify(inst, i, gens, b"pad")
    oinst, ogens = orc.OInstance(24, nv, 5, A, B, C), orc.OGens(24, nv, 5)
    oproof, _ = orc.nizk_prove(oinst, vars_pad, r["inputs"], ogens, b"pad", b"\x07" * 32)
    assert proof.bytes == oproof


def test_resident_witness_and_determinism():
    r = oa.synth_r1cs(1 << 12, 10, 1)
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    w = oa.Witness(inst, v, i)
    p1 = oa.NIZK.prove(inst, w, None, gens, seed=b"\x01" * 32)
    p2 = oa.NIZK.prove(inst, w, None, gens, seed=b"\x01" * 32)
    p3 = oa.NIZK.prove(inst, v, i, gens, seed=b"\x01" * 32)
    assert p1.bytes == p2.bytes == p3.bytes
    p4 = oa.NIZK.prove(inst, w, None, gens, seed=b"\x02" * 32)
    assert p4.bytes != p1.bytes
    p4.verify(inst, i, gens)
    p5 = oa.NIZK.prove(inst, w, None, gens)       # OS entropy
    p5.verify(inst, i, gens)


def test_tampered_proof_and_wrong_input_rejected():
    r, inst, gens, inputs, proof, *_ = _prove_both(256, 10)
    bad = bytearray(proof.bytes); bad[len(bad) // 3] ^= 0x10
    with pytest.raises(oa.ProofVerifyError):
        oa.NIZK(bytes(bad)).verify(inst, inputs, gens)
    wrong = r["inputs"].copy(); wrong[0, 0] ^= 1
    with pytest.raises(oa.ProofVerifyError):
        proof.verify(inst, oa.InputsAssignment.new(wrong), gens)
    with pytest.raises(oa.ProofVerifyError):
        proof.verify(inst, inputs, gens, b"another label")


def test_unsatisfying_witness_yields_rejected_proof():
    r = oa.synth_r1cs(64, 4, 2)
    inst = oa.Instance.new(64, 64, 4, r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(64, 64, 4)
    bad_vars = r["vars"].copy(); bad_vars[3, 0] ^= 1
    v, i = oa.VarsAssignment.new(bad_vars), oa.InputsAssignment.new(r["inputs"])
    assert not inst.is_sat(v, i)
    proof = oa.NIZK.prove(inst, v, i, gens)
    with pytest.raises(oa.ProofVerifyError):
        proof.verify(inst, i, gens)


# ------------------------------------------------------------------------------------------------ compiler-like instance, CLI
def _compiler_like_r1cs(rng, nc, nv, ni):
    """satisfiable R1CS with 1..8 non-zeros per row, small coefficients, a heavily used constant column and one very long row"""
    L = orc.L_ORDER
    size_z = nv + 1 + ni
    z = [int(rng.integers(0, 2 ** 63)) if rng.random() < 0.9 else int.from_bytes(rng.bytes(40), "little") % L for _ in range(size_z)]
    z[nv] = 1
    ents = {k: [] for k in "ABC"}

    def lin(row, k_max, force_const):
        cols = set(int(c) for c in rng.integers(0, size_z, size=int(rng.integers(1, k_max + 1))))
        if force_const:
            cols.add(nv)
        terms = [(c, int(rng.integers(1, 1000)) if rng.random() < 0.8 else L - int(rng.integers(1, 1000))) for c in sorted(cols)]
        return terms, sum(v * z[c] for c, v in terms) % L

    for row in range(nc):
        ta, a = lin(row, 300 if row == 3 else 8, rng.random() < 0.5)
        tb, b = lin(row, 8, rng.random() < 0.5)
        tc, c = lin(row, 6, False)
        # fix C with one extra term on a non-zero variable so that <C,z> = a*b
        fix = next(cc for cc in range(size_z) if z[cc] % L and cc not in dict(tc))
        tc.append((fix, (a * b - c) * pow(z[fix], -1, L) % L))
        for name, terms in (("A", ta), ("B", tb), ("C", tc)):
            ents[name] += [(row, col, v) for col, v in terms]
    mats = []
    for name in "ABC":
        e = np.zeros(len(ents[name]), dtype=oa.ENTRY_DTYPE)
        e["row"] = [t[0] for t in ents[name]]; e["col"] = [t[1] for t in ents[name]]
        e["val"] = np.array([np.frombuffer(t[2].to_bytes(32, "little"), dtype=np.uint8) for t in ents[name]])
        mats.append(e)
    to32 = lambda xs: np.array([np.frombuffer((x % L).to_bytes(32, "little"), dtype=np.uint8) for x in xs]).reshape(-1, 32)
    return mats, to32(z[:nv]), to32(z[nv + 1:])


def test_compiler_like_instance_with_heavy_rows_and_columns(rng):
    nc, nv, ni = 700, 500, 6
    (A, B, C), vars32, inputs32 = _compiler_like_r1cs(rng, nc, nv, ni)
    inst = oa.Instance.new(nc, nv, ni, A, B, C); gens = oa.NIZKGens.new(nc, nv, ni)
    v, i = oa.VarsAssignment.new(vars32), oa.InputsAssignment.new(inputs32)
    assert inst.is_sat(v, i)
    proof = oa.NIZK.prove(inst, v, i, gens, b"circ", b"\x33" * 32)
    proof.verify(inst, i, gens, b"circ")
    oi, og = orc.OInstance(nc, nv, ni, A, B, C), orc.OGens(nc, nv, ni)
    oproof, _ = orc.nizk_prove(oi, vars32, inputs32, og, b"circ", b"\x33" * 32)
    assert proof.bytes == oproof


def test_spzk_cli_end_to_end(tmp_path):
    import os
    import subprocess
    spzk = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "otti_amd", "spzk")
    pre = str(tmp_path / "syn")
    assert subprocess.run([spzk, "synth", "300", pre, "7", "4"], capture_output=True).returncode == 0
    # the reference's argv [REF run.py:100], plus the additive reproducibility options
    res = subprocess.run([spzk, "verify", "--nizk", pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif", "--seed", "2a" * 32, "--proof-out", pre + ".proof"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert "Verification successful" in res.stdout and "prove_sc_phase_one" in res.stdout
    r = oa.zkif_load(pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif")
    oi = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    og = orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
    oproof, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og, b"nizk_example", b"\x2a" * 32)
    assert open(pre + ".proof", "rb").read() == oproof
    # a corrupted witness must make the process fail (run.py checks the exit status on the LP path)
    wit = bytearray(open(pre + ".wit.zkif", "rb").read()); wit[-40] ^= 1
    open(pre + ".wit.zkif", "wb").write(bytes(wit))
    res = subprocess.run([spzk, "verify", "--nizk", pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif"], capture_output=True, text=True)
    assert res.returncode != 0 and "Verification successful" not in res.stdout


def test_spzk_prove_and_verify_as_separate_processes(tmp_path):
    """`spzk prove … --proof-out` and `spzk verify … --proof-in` (the verifier gets no witness file)."""
    import os
    import subprocess
    spzk = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "otti_amd", "spzk")
    pre = str(tmp_path / "syn")
    assert subprocess.run([spzk, "synth", "1000", pre, "5", "9"], capture_output=True).returncode == 0
    res = subprocess.run([spzk, "prove", "--nizk", pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif", "--proof-out", pre + ".proof", "--seed", "07" * 32],
                         capture_output=True, text=True)
    assert res.returncode == 0 and "Proof written" in res.stdout and "Verification" not in res.stdout, res.stderr
    res = subprocess.run([spzk, "verify", "--nizk", pre + ".zkif", pre + ".inp.zkif", "--proof-in", pre + ".proof"], capture_output=True, text=True)
    assert res.returncode == 0 and "Verification successful" in res.stdout, res.stdout + res.stderr
    r = oa.zkif_load(pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif")
    oi, og = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"]), orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
    oproof, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og, b"nizk_example", b"\x07" * 32)
    assert open(pre + ".proof", "rb").read() == oproof
    # a flipped proof byte, a truncated proof and a different label are all rejected by the separate verifier
    pf = bytearray(oproof); pf[len(pf) // 2] ^= 4
    open(pre + ".bad", "wb").write(bytes(pf))
    assert subprocess.run([spzk, "verify", "--nizk", pre + ".zkif", pre + ".inp.zkif", "--proof-in", pre + ".bad"], capture_output=True).returncode != 0
    open(pre + ".bad", "wb").write(oproof[:-5])
    assert subprocess.run([spzk, "verify", "--nizk", pre + ".zkif", pre + ".inp.zkif", "--proof-in", pre + ".bad"], capture_output=True).returncode != 0
    assert subprocess.run([spzk, "verify", "--nizk", pre + ".zkif", pre + ".inp.zkif", "--proof-in", pre + ".proof", "--label", "other"],
                          capture_output=True).returncode != 0
    assert subprocess.run([spzk, "prove", "--nizk", pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif"], capture_output=True).returncode == 2   # no --proof-out


@pytest.mark.parametrize("window", ["10", None])
def test_fresh_process_growing_instances(window):
    """Regression: a result buffer of the device context that grew while an earlier launch's results were still in flight."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    if window:
        env["OTTI_MSM_WINDOW"] = window
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fresh_process_check.py")], env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr


def test_concurrent_provers_share_instance_and_table():
    """Several prover threads of one process (own device context each) on shared instance / generator / witness handles, plus one
    thread on a different instance: every proof must be the oracle's, whatever the interleaving on the device."""
    import threading
    jobs, inputs_of = [], []
    for gen, n, ni in ((oa.synth_r1cs, 1 << 12, 4), (oa.synth_r1cs_compiler_like, 1 << 13, 3), (oa.synth_r1cs, 300, 2)):
        r = gen(n, ni, 11)
        inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
        gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
        wit = oa.Witness(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
        oi, og = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"]), orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
        want, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og, b"mt", b"\x19" * 32)
        jobs.append((inst, gens, wit, want)); inputs_of.append(oa.InputsAssignment.new(r["inputs"]))
    results, errors = [], []

    def run(job, reps):
        try:
            inst, gens, wit, want = job
            for _ in range(reps):
                results.append(oa.NIZK.prove(inst, wit, None, gens, b"mt", b"\x19" * 32).bytes == want)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=run, args=(jobs[k % 3], 6)) for k in range(5)]      # 5 threads: jobs 0 and 1 are shared by two each
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(results) == 30 and all(results)
    # verification from several threads at once (each call fans its deferred checks out over host threads of its own)
    verdicts = []

    def check(job, tamper):
        try:
            inst, gens, wit, want = job
            pf = bytearray(want)
            if tamper:
                pf[len(pf) // 3] ^= 1
            try:
                oa.NIZK(bytes(pf)).verify(inst, job_inputs[id(inst)], gens, b"mt"); verdicts.append(not tamper)
            except oa.ProofVerifyError:
                verdicts.append(tamper)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)

    job_inputs = {id(j[0]): inp for j, inp in zip(jobs, inputs_of)}
    threads = [threading.Thread(target=check, args=(jobs[k % 3], k % 2 == 1)) for k in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(verdicts) == 8 and all(verdicts)


@pytest.mark.parametrize("shape", ["many_vars", "many_cons", "one_var_block"])
def test_rectangular_instances(rng, shape):
    """num_cons != num_vars: the two sum-checks then run over tables of different lengths, and the commitment matrix is not the
    square of the constraint count.  Built from a satisfiable square instance by adding unused variables or repeating constraints."""
    n, ni = 64, 3
    r = oa.synth_r1cs(n, ni, 21)
    A, B, C, vars_, inputs = r["A"].copy(), r["B"].copy(), r["C"].copy(), r["vars"], r["inputs"]
    if shape == "many_vars":                                   # 64 constraints over 2048 + 64 variables
        extra = 2048
        for M in (A, B, C):
            M["col"] = np.where(M["col"] >= n, M["col"] + extra, M["col"])
        canon = rng.integers(0, 256, size=(extra, 32), dtype=np.uint8); canon[:, 31] &= 0x0f
        vars_ = np.concatenate([vars_, canon]); nc, nv = n, n + extra
    elif shape == "many_cons":                                 # 4096 constraints (each original one 64 times) over 64 variables
        reps = 64
        def rep(M):
            out = np.tile(M, reps); out["row"] = np.concatenate([M["row"] + k * n for k in range(reps)]); return out
        A, B, C = rep(A), rep(B), rep(C); nc, nv = n * reps, n
    else:                                                      # 2 constraints, 64 variables: num_cons is padded to the minimum of 2
        keep = lambda M: M[M["row"] < 2]
        A, B, C = keep(A), keep(B), keep(C); nc, nv = 2, n
    inst, gens = oa.Instance.new(nc, nv, ni, A, B, C), oa.NIZKGens.new(nc, nv, ni)
    assert inst.is_sat(oa.VarsAssignment.new(vars_), oa.InputsAssignment.new(inputs))
    proof = oa.NIZK.prove(inst, oa.VarsAssignment.new(vars_), oa.InputsAssignment.new(inputs), gens, b"rect", b"\x11" * 32)
    oi, og = orc.OInstance(nc, nv, ni, A, B, C), orc.OGens(nc, nv, ni)
    want, _ = orc.nizk_prove(oi, vars_, inputs, og, b"rect", b"\x11" * 32)
    assert proof.bytes == want
    proof.verify(inst, oa.InputsAssignment.new(inputs), gens, b"rect")
    assert orc.nizk_verify(oi, inputs, og, proof.bytes, b"rect") == 0


def test_golden_proof_digests_on_gpu():
    import json
    import os
    golden = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "proofs.json")))
    for g in golden:
        r = oa.synth_r1cs(g["n"], g["num_inputs"], g["instance_seed"])
        inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
        gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
        p = oa.NIZK.prove(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]), gens, g["label"].encode(), bytes.fromhex(g["tape_seed"]))
        assert len(p.bytes) == g["proof_len"] and hashlib.sha256(p.bytes).hexdigest() == g["proof_sha256"], g["n"]


def test_very_long_row_and_column_are_segmented(rng):
    # the constant-1 column appears in every row of A (8192 entries in the transposed copy: 4 segments), and row 7 of B spans 5000 columns
    nc = nv = 1 << 13; ni = 3
    A, B, C = _random_instance(rng, nc, nv, ni, 2)
    extra = np.zeros(nc, dtype=oa.ENTRY_DTYPE); extra["row"] = np.arange(nc); extra["col"] = nv
    extra["val"] = rng.integers(0, 256, size=(nc, 32), dtype=np.uint8); extra["val"][:, 31] &= 0x0f
    A = np.concatenate([A, extra])
    longrow = np.zeros(5000, dtype=oa.ENTRY_DTYPE); longrow["row"] = 7; longrow["col"] = rng.permutation(nv)[:5000]
    longrow["val"] = rng.integers(0, 256, size=(5000, 32), dtype=np.uint8); longrow["val"][:, 31] &= 0x0f
    B = np.concatenate([B, longrow])
    inst, oinst = oa.Instance.new(nc, nv, ni, A, B, C), orc.OInstance(nc, nv, ni, A, B, C)
    z = orc.rand_fr(rng, 2 * nv)
    ga, gb, gc, _ = K.multiply_vec(inst, z)
    oa_, ob_, oc_ = orc.multiply_vec(oinst, z)
    assert eq(ga, oa_) and eq(gb, ob_) and eq(gc, oc_)
    eqrx, coef = orc.rand_fr(rng, nc), orc.rand_fr(rng, 3)
    got, _ = K.eval_table_sparse(inst, eqrx, coef)
    eA, eB, eC = (orc.fr_to_ints(x) for x in orc.eval_table_sparse(oinst, eqrx))
    c = orc.fr_to_ints(coef)
    assert orc.fr_to_ints(got) == [(c[0] * a + c[1] * b + c[2] * d) % orc.L_ORDER for a, b, d in zip(eA, eB, eC)]


@pytest.mark.parametrize("lg", [8, 13])
def test_compiler_like_generator_proofs_match_oracle(lg):
    n, ni = 1 << lg, 10
    r = oa.synth_r1cs_compiler_like(n, ni, 5)
    inst = oa.Instance.new(n, n, ni, r["A"], r["B"], r["C"]); gens = oa.NIZKGens.new(n, n, ni)
    v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    assert inst.is_sat(v, i)
    proof = oa.NIZK.prove(inst, v, i, gens, b"circ2", b"\x44" * 32)
    proof.verify(inst, i, gens, b"circ2")
    oi, og = orc.OInstance(n, n, ni, r["A"], r["B"], r["C"]), orc.OGens(n, n, ni)
    oproof, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og, b"circ2", b"\x44" * 32)
    assert proof.bytes == oproof
