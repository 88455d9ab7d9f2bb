"""A plain-C program linked against include/otti_spartan.h + libottispartan.so: the compiled stand-in for the Rust shim crate
(bindings/rust/otti-spartan cannot be built here: no rustc), mirroring it call for call the way rust-circ links libspartan
in-process [REF /root/reference/run.py:147].  Also the cargo-runner script that lets the LP path's `cargo run --release -- verify ...`
[REF /root/reference/run.py:52-59] start the MI355X binary without editing run.py."""
import os
import subprocess
import pytest

import otti_amd as oa
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED, LABEL = b"\x2a" * 32, b"nizk_example"


@pytest.fixture(scope="module")
def caller(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("c_caller") / "otti_caller")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings", "c", "otti_caller.c"),
                           "-o", exe, "-L" + os.path.join(ROOT, "otti_amd"), "-lottispartan", "-Wl,-rpath," + os.path.join(ROOT, "otti_amd")])
    return exe


def _oracle_proof(n):
    r = oa.synth_r1cs(n, 10, 1)
    oi, og = orc.OInstance(n, n, 10, r["A"], r["B"], r["C"]), orc.OGens(n, n, 10)
    return orc.nizk_prove(oi, r["vars"], r["inputs"], og, LABEL, SEED)[0]


def test_c_caller_host_side(caller, tmp_path):
    """no GPU: Instance::new / is_sat / error mapping / `no device` from prove / verify of an oracle-made proof and of a tampered one"""
    pf = tmp_path / "p.bin"; pf.write_bytes(_oracle_proof(256))
    res = subprocess.run([caller, "host", str(pf)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "otti_caller ok" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_c_caller_proves_on_the_gpu_byte_identical_to_the_oracle(caller, tmp_path):
    out = tmp_path / "gpu.bin"
    res = subprocess.run([caller, "prove", "12", str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "otti_caller ok" in res.stdout, res.stdout + res.stderr
    assert out.read_bytes() == _oracle_proof(1 << 12)


def _oracle_snark(n):
    r = oa.synth_r1cs(n, 10, 1)
    nz = int(max(r["A"].size, r["B"].size, r["C"].size))
    oi, og = orc.OInstance(n, n, 10, r["A"], r["B"], r["C"]), orc.OSnarkGens(n, n, 10, nz)
    comm = orc.OSnarkComm.encode(oi, og)
    return comm.bytes, orc.snark_prove(oi, comm, r["vars"], r["inputs"], og, b"snark_example", SEED)[0]


def test_c_caller_snark_mode_host_side(caller, tmp_path):
    """no GPU: SNARK mode through the C ABI as upstream spartan-zkinterface runs it without --nizk — the verifier's half: commitment from
    bytes, SNARK::verify of the CPU oracle's proof, a tampered copy, another label, `no device` from SNARK::encode"""
    cb, pb = _oracle_snark(256)
    cf, pf = tmp_path / "c.bin", tmp_path / "p.bin"; cf.write_bytes(cb); pf.write_bytes(pb)
    res = subprocess.run([caller, "snark-host", str(cf), str(pf)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "otti_caller ok: SNARK mode" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_c_caller_snark_mode_on_the_gpu_byte_identical_to_the_oracle(caller, tmp_path):
    """SNARKGens::new / SNARK::encode / prove / verify from a compiled caller: commitment and proof equal the oracle's at 2^12"""
    cf, pf = tmp_path / "c.bin", tmp_path / "p.bin"
    res = subprocess.run([caller, "snark", "12", str(cf), str(pf)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "otti_caller ok: SNARK mode" in res.stdout, res.stdout + res.stderr
    cb, pb = _oracle_snark(1 << 12)
    assert cf.read_bytes() == cb, "computation commitment differs from the oracle's"
    assert pf.read_bytes() == pb, "SNARK proof differs from the oracle's"


def test_cargo_runner_drops_the_built_binary_and_forwards_the_arguments(tmp_path):
    rel = tmp_path / "target" / "release"; rel.mkdir(parents=True)
    (rel / "spzk").write_text("#!/bin/sh\necho the-rust-binary\n"); os.chmod(rel / "spzk", 0o755)
    (rel / "spzk-mi355x").write_text('#!/bin/sh\necho mi355x "$@"\n'); os.chmod(rel / "spzk-mi355x", 0o755)
    res = subprocess.run([os.path.join(ROOT, "bindings", "cargo", "spzk-runner"), str(rel / "spzk"), "verify", "--nizk", "a.zkif", "a.inp.zkif", "a.wit.zkif"],
                         capture_output=True, text=True)
    assert res.returncode == 0 and res.stdout.strip() == "mi355x verify --nizk a.zkif a.inp.zkif a.wit.zkif"
