"""CPU tests of the product's host side (no compute calls that need a GPU): the C-ABI library loads and exports every symbol
include/otti_spartan.h declares; Instance::new validation and padding; generator derivation; zkInterface ingest; the product
verifier against oracle-made proofs; the u64-lane transport used by the multi-GPU sum-check; loud failure without a device."""
import hashlib
import json
import os
import re
import subprocess
import sys
import numpy as np
import pytest

import otti_amd as oa
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "primitives.json")))
L = orc.L_ORDER


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "otti_spartan.h")).read()
    names = sorted(set(re.findall(r"\b(otti_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 35
    for n in names:
        assert hasattr(oa.lib, n), f"libottispartan.so does not export {n}"
    out = subprocess.check_output(["nm", "-D", "--defined-only", oa.lib_path]).decode()
    exported = set(re.findall(r" T (otti_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_library_carries_gfx950_code_object_only():
    blob = open(oa.lib_path, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"gfx1100", b"sm_90", b"nvptx"):
        assert other not in blob


def test_generators_match_libsodium_fixture():
    g = oa.NIZKGens.new(1 << 10, 1 << 10, 10)
    pts = g.points(34)
    assert [p.tobytes().hex() for p in pts] == GOLD["gens_r1cs_sat"][:34]
    assert [p.tobytes().hex() for p in pts[:3]] == GOLD["appendix_b"]["gens_first3"]


def test_synthetic_instance_is_reproducible_and_satisfiable():
    r = oa.synth_r1cs(64, 10, 1)
    assert r["num_cons"] == 64 and r["num_vars"] == 64 and r["num_inputs"] == 10
    # Z[k] = from_bytes_wide(SHAKE256("otti-synth" || seed_le64 || k_le64, 64))
    z0 = int.from_bytes(hashlib.shake_256(b"otti-synth" + (1).to_bytes(8, "little") + (0).to_bytes(8, "little")).digest(64), "little") % L
    assert int.from_bytes(r["vars"][0].tobytes(), "little") == z0
    zin = int.from_bytes(hashlib.shake_256(b"otti-synth" + (1).to_bytes(8, "little") + (65).to_bytes(8, "little")).digest(64), "little") % L
    assert int.from_bytes(r["inputs"][0].tobytes(), "little") == zin
    inst = oa.Instance.new(64, 64, 10, r["A"], r["B"], r["C"])
    v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    assert inst.is_sat(v, i)
    bad = r["vars"].copy(); bad[1, 0] ^= 1
    assert not inst.is_sat(oa.VarsAssignment.new(bad), i)
    assert orc.OInstance(64, 64, 10, r["A"], r["B"], r["C"]).is_sat(r["vars"], r["inputs"])


def test_instance_new_padding_and_errors():
    r = oa.synth_r1cs(5, 2, 4)
    inst = oa.Instance.new(5, 5, 2, r["A"], r["B"], r["C"])
    assert inst.dims == (8, 8, 2)
    one = np.zeros(1, dtype=oa.ENTRY_DTYPE)
    assert oa.Instance.new(1, 1, 0, one, one[:0], one[:0]).dims == (2, 1, 0)                  # at least two constraints
    bad = r["A"].copy(); bad["row"][0] = 5
    with pytest.raises(oa.R1CSError) as e:
        oa.Instance.new(5, 5, 2, bad, r["B"], r["C"])
    assert e.value.code == -6
    bad = r["A"].copy(); bad["col"][0] = 5 + 1 + 2
    with pytest.raises(oa.R1CSError) as e:
        oa.Instance.new(5, 5, 2, bad, r["B"], r["C"])
    assert e.value.code == -6
    bad = r["A"].copy(); bad["val"][0] = np.frombuffer(L.to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(oa.R1CSError) as e:
        oa.Instance.new(5, 5, 2, bad, r["B"], r["C"])
    assert e.value.code == -5
    with pytest.raises(oa.R1CSError) as e:
        oa.VarsAssignment.new(np.frombuffer(L.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))
    assert e.value.code == -5
    oa.VarsAssignment.new(np.frombuffer((L - 1).to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))
    v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"][:1])
    with pytest.raises(oa.R1CSError) as e:
        inst.is_sat(v, i)
    assert e.value.code == -3


def test_zkif_write_load_roundtrip(tmp_path):
    r = oa.synth_r1cs(37, 3, 6)
    p = [str(tmp_path / n) for n in ("c.zkif", "c.inp.zkif", "c.wit.zkif")]
    oa.zkif_write(r, *p)
    assert open(p[0], "rb").read()[8:12] == b"zkif"
    q = oa.zkif_load(*p)
    assert (q["num_cons"], q["num_vars"], q["num_inputs"]) == (37, 37, 3)
    for k in "ABC":
        a = np.sort(r[k], order=["row", "col"]); b = np.sort(q[k], order=["row", "col"])
        assert np.array_equal(a["row"], b["row"]) and np.array_equal(a["col"], b["col"]) and np.array_equal(a["val"], b["val"])
    assert np.array_equal(q["vars"], r["vars"]) and np.array_equal(q["inputs"], r["inputs"])
    # circuit alone (no assignment files) still gives the matrices
    q2 = oa.zkif_load(p[0])
    assert q2["num_cons"] == 37 and q2["A"].size == r["A"].size
    # truncated / corrupted files are refused, not crashed on
    blob = open(p[0], "rb").read()
    open(p[0], "wb").write(blob[: len(blob) // 2])
    with pytest.raises(oa.SpartanError) as e:
        oa.zkif_load(*p)
    assert e.value.code == -22
    open(p[0], "wb").write(blob[:4] + b"\xff\xff\xff\x7f" + blob[8:])
    with pytest.raises(oa.SpartanError):
        oa.zkif_load(*p)
    with pytest.raises(oa.SpartanError):
        oa.zkif_load(str(tmp_path / "missing.zkif"))


def test_spzk_cli_contract(tmp_path):
    spzk = os.path.join(ROOT, "otti_amd", "spzk")
    assert os.path.exists(spzk)
    out = subprocess.run([spzk, "synth", "16", str(tmp_path / "t"), "2", "5"], capture_output=True, text=True)
    assert out.returncode == 0 and os.path.exists(tmp_path / "t.wit.zkif")
    assert subprocess.run([spzk], capture_output=True).returncode == 2
    assert subprocess.run([spzk, "verify", str(tmp_path / "t.zkif")], capture_output=True).returncode == 2      # --nizk + three files required
    if oa.device_count() == 0:
        # reference argv [REF run.py:100]; with no GPU the binary must fail loudly, never fall back
        res = subprocess.run([spzk, "verify", "--nizk", str(tmp_path / "t.zkif"), str(tmp_path / "t.inp.zkif"), str(tmp_path / "t.wit.zkif")],
                             capture_output=True, text=True)
        assert res.returncode != 0 and "Verification successful" not in res.stdout


def test_product_verifier_accepts_oracle_proofs_and_rejects_tampering():
    for n, ni in ((2, 0), (8, 3), (256, 10), (1 << 11, 10)):
        r = oa.synth_r1cs(n, ni, 1)
        inst = oa.Instance.new(n, n, ni, r["A"], r["B"], r["C"]); gens = oa.NIZKGens.new(n, n, ni)
        oi, og = orc.OInstance(n, n, ni, r["A"], r["B"], r["C"]), orc.OGens(n, n, ni)
        pf, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og, b"lbl", b"\x11" * 32)
        inputs = oa.InputsAssignment.new(r["inputs"])
        oa.NIZK(pf).verify(inst, inputs, gens, b"lbl")
        with pytest.raises(oa.ProofVerifyError):
            oa.NIZK(pf).verify(inst, inputs, gens, b"lbx")
        for pos in (40, len(pf) // 2, len(pf) - 5):
            bad = bytearray(pf); bad[pos] ^= 2
            with pytest.raises(oa.ProofVerifyError):
                oa.NIZK(bytes(bad)).verify(inst, inputs, gens, b"lbl")
        with pytest.raises(oa.ProofVerifyError) as e:
            oa.NIZK(pf[:-7]).verify(inst, inputs, gens, b"lbl")
        assert e.value.code == -12


def test_lanes_transport_of_field_sums(rng):
    # 8 ranks' partial sums, each an Fr in Montgomery form, summed as 8 x u64 lanes and normalised once
    parts = [orc.rand_fr(rng, 5) for _ in range(8)]
    lanes = sum(oa.lanes_pack(p).astype(np.uint64) for p in parts)
    got = orc.fr_to_ints(oa.lanes_unpack(lanes))
    want = [sum(orc.fr_to_ints(p)[k] for p in parts) % L for k in range(5)]
    assert got == want
    # worst case: 2^32 - 1 in every limb, 8 times
    top = np.full(8, 8 * (2 ** 32 - 1), dtype=np.uint64)
    assert int.from_bytes(oa.lanes_unpack(top)[0].tobytes(), "little") == (8 * (2 ** 256 - 1)) % L


@pytest.mark.skipif(oa.device_count() > 0, reason="only meaningful without a GPU")
def test_prove_fails_loudly_without_device():
    r = oa.synth_r1cs(8, 2, 1)
    inst = oa.Instance.new(8, 8, 2, r["A"], r["B"], r["C"]); gens = oa.NIZKGens.new(8, 8, 2)
    with pytest.raises(oa.NoDeviceError):
        oa.NIZK.prove(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]), gens)
    with pytest.raises(oa.NoDeviceError):
        oa.kernels.eq_evals(orc.rand_fr(np.random.default_rng(0), 3))


def test_product_never_links_or_imports_the_oracle():
    deps = subprocess.check_output(["ldd", oa.lib_path]).decode()
    assert "liboracle" not in deps
    for dirpath, _, files in os.walk(os.path.join(ROOT, "otti_amd")):
        if "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in src and "import orc" not in src and "oracle/" not in src.replace("the CPU oracle", ""), f


def test_zkif_reader_accepts_the_official_builders_layout(tmp_path, rng):
    """Real zkInterface producers (flatc-generated builders) write buffers back to front: children before parents, vtable in front of
    its table, shared vtables, several ConstraintSystem messages per file, values of any fixed width.  The product's own writer is
    forward-laid-out, so the reader is checked here against an independent builder of the official layout (tests/fb_reverse_builder.py)."""
    import fb_reverse_builder as fb
    L = oa.L_ORDER
    # a small hand-made circuit: x*y = z ; (x + 5)*1 = w ; z*w = out (public).  ids: 0 = one, 1 = out (instance), 2..5 witness
    x, y = 3, 4
    z, w = x * y, x + 5
    cons = [([(2, 1)], [(3, 1)], [(4, 1)]), ([(2, 1), (0, 5)], [(0, 1)], [(5, 1)]), ([(4, 1)], [(5, 1)], [(1, 1)])]
    c, i, wt = (str(tmp_path / n) for n in ("c.zkif", "c.inp.zkif", "c.wit.zkif"))
    open(c, "wb").write(fb.circuit_header([1], None, 6, L - 1) + fb.constraint_system(cons[:2], width=32) + fb.constraint_system(cons[2:], width=4))
    open(i, "wb").write(fb.circuit_header([1], [z * w], 6, L - 1))
    open(wt, "wb").write(fb.witness([2, 3, 4, 5], [x, y, z, w], width=8))
    r = oa.zkif_load(c, i, wt)
    assert (r["num_cons"], r["num_vars"], r["num_inputs"]) == (3, 4, 1)
    assert list(r["A"]["row"]) == [0, 1, 1, 2] and list(r["A"]["col"]) == [0, 0, 4, 2]          # one -> column num_vars, out -> num_vars + 1
    assert [int.from_bytes(v.tobytes(), "little") for v in r["A"]["val"]] == [1, 1, 5, 1]
    assert list(r["C"]["col"]) == [2, 3, 5]
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    assert inst.is_sat(oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
    oi, og = orc.OInstance(3, 4, 1, r["A"], r["B"], r["C"]), orc.OGens(3, 4, 1)
    proof, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og)
    oa.NIZK(proof).verify(inst, oa.InputsAssignment.new(r["inputs"]), oa.NIZKGens.new(3, 4, 1))
    # a wrong public value must not satisfy
    open(i, "wb").write(fb.circuit_header([1], [z * w + 1], 6, L - 1))
    r2 = oa.zkif_load(c, i, wt)
    assert not inst.is_sat(oa.VarsAssignment.new(r2["vars"]), oa.InputsAssignment.new(r2["inputs"]))

    # the same random instance through both layouts must load identically
    s = oa.synth_r1cs_compiler_like(200, 3, 9)
    nv, ni = s["num_vars"], s["num_inputs"]

    def vid(col):                                   # product column -> zkInterface variable id (instance ids first, as the writer does)
        return 1 + ni + col if col < nv else (0 if col == nv else col - nv)

    rows = [([], [], []) for _ in range(s["num_cons"])]
    for k, name in enumerate("ABC"):
        for e in s[name]:
            rows[int(e["row"])][k].append((vid(int(e["col"])), int.from_bytes(e["val"].tobytes(), "little")))
    c2, i2, w2 = (str(tmp_path / n) for n in ("r.zkif", "r.inp.zkif", "r.wit.zkif"))
    half = len(rows) // 2
    inst_ids = list(range(1, ni + 1))
    open(c2, "wb").write(fb.circuit_header(inst_ids, None, 1 + ni + nv, L - 1) + fb.constraint_system(rows[:half]) + fb.constraint_system(rows[half:]))
    open(i2, "wb").write(fb.circuit_header(inst_ids, [int.from_bytes(v.tobytes(), "little") for v in s["inputs"]], 1 + ni + nv, L - 1))
    open(w2, "wb").write(fb.witness(list(range(1 + ni, 1 + ni + nv)), [int.from_bytes(v.tobytes(), "little") for v in s["vars"]]))
    got = oa.zkif_load(c2, i2, w2)
    f1, f2, f3 = (str(tmp_path / n) for n in ("f.zkif", "f.inp.zkif", "f.wit.zkif"))
    oa.zkif_write(s, f1, f2, f3)
    want = oa.zkif_load(f1, f2, f3)
    for k in ("num_cons", "num_vars", "num_inputs"):
        assert got[k] == want[k]
    for k in ("A", "B", "C", "vars", "inputs"):
        assert np.array_equal(got[k], want[k]), k


def test_untrusted_input_parsers_survive_mutation_fuzzing(tmp_path):
    """.zkif files and proof bytes come from outside: every mutant must load/verify or be refused with an error code (child
    process: a crash or a hang fails the test)."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_worker.py"), str(tmp_path), "1500"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "refused" in res.stdout and "rejected" in res.stdout


def _sanitizer_inputs(tmp_path, n):
    """zkif files of a small instance plus the oracle's NIZK proof, computation commitment and SNARK proof of it"""
    s = oa.synth_r1cs_compiler_like(n, 3, 4)
    paths = [str(tmp_path / x) for x in ("s.zkif", "s.inp.zkif", "s.wit.zkif")]
    oa.zkif_write(s, *paths)
    oi, og = orc.OInstance(s["num_cons"], s["num_vars"], s["num_inputs"], s["A"], s["B"], s["C"]), orc.OGens(s["num_cons"], s["num_vars"], s["num_inputs"])
    proof, _ = orc.nizk_prove(oi, s["vars"], s["inputs"], og, b"san", b"\x05" * 32)
    open(tmp_path / "proof.bin", "wb").write(proof)
    nz = int(max(s["A"].size, s["B"].size, s["C"].size))
    sg = orc.OSnarkGens(s["num_cons"], s["num_vars"], s["num_inputs"], nz)
    oc = orc.OSnarkComm.encode(oi, sg)
    sproof, _ = orc.snark_prove(oi, oc, s["vars"], s["inputs"], sg, b"san", b"\x06" * 32)
    open(tmp_path / "comm.bin", "wb").write(oc.bytes); open(tmp_path / "sproof.bin", "wb").write(sproof)
    return [*paths, str(tmp_path / "proof.bin"), "san", str(tmp_path)], [str(tmp_path / "comm.bin"), str(tmp_path / "sproof.bin"), str(nz)]


def test_host_parsers_under_address_and_ub_sanitizers(tmp_path):
    """The host sources (zkif reader, proof parsers, both verifiers, group/field code) rebuilt for the CPU with -fsanitize=address,undefined
    and driven with mutated files and proofs (tests/san/): any out-of-bounds access, overflow or misaligned load aborts the run."""
    import subprocess
    san = os.path.join(os.path.dirname(os.path.abspath(__file__)), "san")
    build = subprocess.run(["make", "-C", san, "-s"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    head, tail = _sanitizer_inputs(tmp_path, 48)
    res = subprocess.run([os.path.join(san, "_build", "san_harness"), *head, "600", *tail],
                         capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert "sanitized run" in res.stdout


def test_host_verifiers_under_thread_sanitizer(tmp_path):
    """NIZK::verify and SNARK::verify hand their group equations, the decompression of the sum-check commitments and the split
    multiplication tables to background threads through lock-free slots (snark.h Deferred, spartan_host.cpp PreDecoded) while the calling
    thread and its spinning helpers walk the rounds: the same harness built with -fsanitize=thread must see no data race."""
    import subprocess
    san = os.path.join(os.path.dirname(os.path.abspath(__file__)), "san")
    build = subprocess.run(["make", "-C", san, "-s", "tsan"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    head, tail = _sanitizer_inputs(tmp_path, 200)
    res = subprocess.run([os.path.join(san, "_build", "tsan_harness"), *head, "24", *tail], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, OTTI_VERIFY_THREADS="4", OTTI_HOST_THREADS="3", TSAN_OPTIONS="halt_on_error=0"))
    assert "sanitized run" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
    # (gcc 11's libtsan does not intercept pthread_cond_clockwait, which std::condition_variable::wait_for uses: the pool's sleeping
    # helpers are reported as "double lock of a mutex" — a known false positive; data races are what this test is about)
    assert "data race" not in res.stderr, res.stderr[-4000:]


def test_host_fast_paths_agree_with_the_generic_field_code():
    """five-limb GF(2^255-19) point compression, the fixed-base window tables of the prover's per-round host work (hostfast.h; with
    AVX-512 IFMA where the CPU has it: hostifma.h), the verifier's split multiplication tables, the transcript's fused message operations"""
    oa.host_selftest(300)


def test_host_sumcheck_tail_forms():
    """hosttail.h: the AVX-512 IFMA form of SNARK mode's host-played sum-check rounds is compared with the scalar form inside host_selftest
    (tables of 2 .. 256 elements, 0 .. 12 product instances and 0 .. 6 triples, extremes of the field); here: the measurement hook answers for
    both forms, and refuses sizes outside its range"""
    oa.host_selftest(24)
    r = oa.host_tail_bench(12, 6, 16, threads=1, reps=3)
    assert set(r) == {"avx512ifma", "scalar"} and r["scalar"] > 0 and r["avx512ifma"] >= 0
    with pytest.raises(oa.SpartanError):
        oa.host_tail_bench(12, 6, 24, threads=1, reps=1)          # not a power of two


def test_host_paths_without_avx512_ifma():
    """OTTI_HOST_IFMA=0 (what a CPU without the instructions runs): the same self-test, and the product's host verifiers on the oracle's
    proofs of both modes — whichever of the two arithmetic paths the other tests took on this machine, this one takes the scalar one"""
    import subprocess
    code = """
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import otti_amd as oa, orc
oa.host_selftest(100)
r = oa.synth_r1cs_compiler_like(100, 3, 4)
oi, og = orc.OInstance(r['num_cons'], r['num_vars'], r['num_inputs'], r['A'], r['B'], r['C']), orc.OGens(r['num_cons'], r['num_vars'], r['num_inputs'])
proof, _ = orc.nizk_prove(oi, r['vars'], r['inputs'], og, b'noifma', bytes([5]) * 32)
inst = oa.Instance.new(r['num_cons'], r['num_vars'], r['num_inputs'], r['A'], r['B'], r['C'])
gens = oa.NIZKGens.new(r['num_cons'], r['num_vars'], r['num_inputs'])
oa.NIZK(proof).verify(inst, oa.InputsAssignment.new(r["inputs"]), gens, b"noifma")
print('ok')
""" % (ROOT, ROOT)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, OTTI_HOST_IFMA="0"))
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout[-1000:] + res.stderr[-3000:]


@pytest.mark.parametrize("n,ni,kind", [(16, 3, "uniform"), (200, 4, "compiler"), (1 << 10, 10, "uniform")])
def test_product_snark_verifier_accepts_oracle_proofs_and_rejects_tampering(n, ni, kind):
    """SNARK::verify of the product (host code, needs no GPU) on proofs and commitments made by the CPU oracle: two independently
    written implementations of the SNARK-mode protocol agree; the commitment round-trips through its wire form"""
    import orc
    r = (oa.synth_r1cs if kind == "uniform" else oa.synth_r1cs_compiler_like)(n, ni, 7)
    nz = max(r["A"].size, r["B"].size, r["C"].size)
    oi = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    og = orc.OSnarkGens(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    oc = orc.OSnarkComm.encode(oi, og)
    proof, _ = orc.snark_prove(oi, oc, r["vars"], r["inputs"], og)
    gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    comm = oa.ComputationCommitment.from_bytes(oc.bytes)
    assert comm.bytes == oc.bytes
    inputs = oa.InputsAssignment.new(r["inputs"])
    oa.SNARK(proof).verify(comm, inputs, gens)
    rng = np.random.default_rng(n)
    for pos in rng.integers(0, len(proof), 12):
        bad = bytearray(proof); bad[int(pos)] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(oa.ProofVerifyError):
            oa.SNARK(bytes(bad)).verify(comm, inputs, gens)
    with pytest.raises(oa.ProofVerifyError):
        oa.SNARK(proof).verify(comm, inputs, gens, b"another label")
    with pytest.raises(oa.SpartanError):
        oa.ComputationCommitment.from_bytes(oc.bytes[:-3])
