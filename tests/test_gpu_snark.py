"""SNARK mode on the GPU against the CPU oracle (oracle/snark.c): the computation commitment (SNARK::encode) and the whole proof
(R1CSProof + R1CSEvalProof) must be byte-identical for the same instance, witness, label and random-tape seed; both verifiers accept."""
import numpy as np
import pytest

import otti_amd as oa
import orc

pytestmark = pytest.mark.gpu
SEED, LABEL = b"\x2a" * 32, b"snark_example"


def _case(n, ni, kind, seed=7):
    r = (oa.synth_r1cs if kind == "uniform" else oa.synth_r1cs_compiler_like)(n, ni, seed)
    nz = max(r["A"].size, r["B"].size, r["C"].size)
    return r, nz


@pytest.mark.parametrize("n,ni,kind", [(2, 0, "uniform"), (16, 3, "uniform"), (64, 10, "uniform"), (200, 4, "compiler"), (1 << 10, 10, "uniform"),
                                       (1 << 12, 10, "compiler"), (1 << 14, 10, "uniform")])
def test_snark_commitment_and_proof_bytes_identical_to_oracle(n, ni, kind):
    r, nz = _case(n, ni, kind)
    oi = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    og = orc.OSnarkGens(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    oc = orc.OSnarkComm.encode(oi, og)
    want, _ = orc.snark_prove(oi, oc, r["vars"], r["inputs"], og, LABEL, SEED)

    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    comm = oa.ComputationCommitment.encode(inst, gens)
    assert comm.bytes == oc.bytes, "computation commitment differs"
    inputs = oa.InputsAssignment.new(r["inputs"])
    proof = oa.SNARK.prove(inst, comm, oa.VarsAssignment.new(r["vars"]), inputs, gens, LABEL, SEED)
    assert len(proof.bytes) == len(want)
    assert proof.bytes == want, "SNARK proof differs from the oracle's"
    verifier_comm = oa.ComputationCommitment.from_bytes(comm.bytes)        # the verifier holds the commitment only
    proof.verify(verifier_comm, inputs, gens, LABEL)
    assert orc.snark_verify(orc.OSnarkComm.parse(comm.bytes), r["inputs"], og, proof.bytes, LABEL) == 0
    bad = bytearray(proof.bytes); bad[len(bad) // 3] ^= 1
    with pytest.raises(oa.ProofVerifyError):
        oa.SNARK(bytes(bad)).verify(verifier_comm, inputs, gens, LABEL)
    with pytest.raises(oa.SpartanError):
        oa.SNARK.prove(inst, verifier_comm, oa.VarsAssignment.new(r["vars"]), inputs, gens, LABEL, SEED)   # no decommitment in a parsed commitment


def test_snark_resident_witness_gives_the_same_proof():
    r, nz = _case(1 << 10, 10, "compiler")
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    comm = oa.ComputationCommitment.encode(inst, gens)
    v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    want = oa.SNARK.prove(inst, comm, v, i, gens, LABEL, SEED).bytes
    w = oa.Witness(inst, v, i)                                # uploaded once, proved from twice
    assert oa.SNARK.prove(inst, comm, w, None, gens, LABEL, SEED).bytes == want
    assert oa.SNARK.prove(inst, comm, w, None, gens, LABEL, SEED).bytes == want
    other = oa.synth_r1cs(1 << 9, 10, 3)
    small = oa.Instance.new(other["num_cons"], other["num_vars"], other["num_inputs"], other["A"], other["B"], other["C"])
    with pytest.raises(oa.SpartanError):                      # a witness of another instance is refused
        oa.SNARK.prove(inst, comm, oa.Witness(small, oa.VarsAssignment.new(other["vars"]), oa.InputsAssignment.new(other["inputs"])), None, gens, LABEL, SEED)


def test_snark_proofs_repeat_and_differ_by_seed():
    r, nz = _case(256, 5, "uniform")
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    comm = oa.ComputationCommitment.encode(inst, gens)
    v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    a = oa.SNARK.prove(inst, comm, v, i, gens, LABEL, SEED).bytes
    assert a == oa.SNARK.prove(inst, comm, v, i, gens, LABEL, SEED).bytes
    b = oa.SNARK.prove(inst, comm, v, i, gens, LABEL, b"\x07" * 32)
    assert b.bytes != a
    b.verify(comm, i, gens, LABEL)


def test_spzk_without_nizk_runs_snark_mode(tmp_path):
    """`spzk verify <three zkif files>` (no --nizk): SNARK::encode + prove + verify in one process; the proof equals the oracle's"""
    import os, subprocess
    spzk = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "otti_amd", "spzk")
    r = oa.synth_r1cs_compiler_like(3000, 5, 11)
    pre = str(tmp_path / "c")
    oa.zkif_write(r, pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif")
    res = subprocess.run([spzk, "verify", pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif", "--seed", "2a" * 32, "--proof-out", pre + ".proof"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "Verification successful" in res.stdout and "SNARK::prove" in res.stdout, res.stdout + res.stderr
    back = oa.zkif_load(pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif")
    nz = max(back["A"].size, back["B"].size, back["C"].size)
    oi = orc.OInstance(back["num_cons"], back["num_vars"], back["num_inputs"], back["A"], back["B"], back["C"])
    og = orc.OSnarkGens(back["num_cons"], back["num_vars"], back["num_inputs"], nz)
    want, _ = orc.snark_prove(oi, orc.OSnarkComm.encode(oi, og), back["vars"], back["inputs"], og, LABEL, SEED)
    assert open(pre + ".proof", "rb").read() == want


def _golden_snark():
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "snark_proofs.json")
    return {e["n"]: e for e in json.load(open(path))}


@pytest.mark.parametrize("lg", [12, 16, 18, 20])
def test_snark_sweep_sizes_match_committed_oracle_digests(lg):
    """The sizes SNARK mode is benchmarked and swept on (2^20 is the bench line's `snark` extra): the oracle's SNARK prover takes
    minutes there, so commitment and proof are compared with the digests tests/golden/make_golden_snark.py committed."""
    import hashlib
    g = _golden_snark()[1 << lg]
    r = oa.synth_r1cs(1 << lg, g["num_inputs"], g["instance_seed"])
    assert hashlib.sha256(r["vars"].tobytes() + r["inputs"].tobytes()).hexdigest() == g["witness_sha256"]
    assert hashlib.sha256(r["A"].tobytes() + r["B"].tobytes() + r["C"].tobytes()).hexdigest() == g["matrices_sha256"]
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], g["num_nz_entries"])
    comm = oa.ComputationCommitment.encode(inst, gens)
    assert len(comm.bytes) == g["commitment_len"] and hashlib.sha256(comm.bytes).hexdigest() == g["commitment_sha256"], "computation commitment differs from the oracle's"
    inputs = oa.InputsAssignment.new(r["inputs"])
    label, seed = g["label"].encode(), bytes.fromhex(g["tape_seed"])
    proof = oa.SNARK.prove(inst, comm, oa.VarsAssignment.new(r["vars"]), inputs, gens, label, seed)
    assert len(proof.bytes) == g["proof_len"] and hashlib.sha256(proof.bytes).hexdigest() == g["proof_sha256"], "SNARK proof differs from the oracle's"
    vc = oa.ComputationCommitment.from_bytes(comm.bytes)
    proof.verify(vc, inputs, gens, label)
    bad = bytearray(proof.bytes); bad[(2 * len(bad)) // 3] ^= 0x40
    with pytest.raises(oa.ProofVerifyError):
        oa.SNARK(bytes(bad)).verify(vc, inputs, gens, label)


@pytest.mark.parametrize("lg,env", [(12, {"OTTI_PC_TAIL": "0"}), (12, {"OTTI_PC_TAIL_CAP": "16"}), (12, {"OTTI_PC_TAIL_CAP": "128"}), (12, {"OTTI_ARMED": "0"}), (12, {}),
                                    (16, {"OTTI_DEREFS_AHEAD": "0"}), (16, {"OTTI_DEREFS_CUMASK": "0"}), (16, {"OTTI_DEREFS_FREE_CUS": "128"}), (16, {"OTTI_PC_LGT_MANY": "5", "OTTI_PC_LGT_FEW": "7"}),
                                    (16, {"OTTI_HOST_FR8": "0"}), (16, {"OTTI_PC_LGT_MANY": "4", "OTTI_PC_LGT_FEW": "8"}), (16, {"OTTI_RELAY": "0"}), (16, {"OTTI_GO_POLLERS": "4"}),
                                    (16, {"OTTI_HOST_THREADS": "1"}), (16, {"OTTI_HOST_TAIL_GRAIN": "8"}),
                                    (16, {"OTTI_HASH_FUSED": "0"}), (16, {"OTTI_HASH_AHEAD": "1"}), (16, {"OTTI_PC_PREEXPORT": "0"}), (16, {"OTTI_HASH_AHEAD": "1", "OTTI_SIDE_CUS": "64"}),
                                    (12, {"OTTI_HASH_AHEAD": "1", "OTTI_PC_PREEXPORT": "0"}),
                                    (12, {"OTTI_TEST_TAIL_DROP": "1", "OTTI_TAIL_TIMEOUT_MS": "300"})])
def test_persistent_tail_variants_give_the_oracles_proof(lg, env):
    """The layered sum-checks of R1CSEvalProof three ways — a launch per round (tail off / nothing armed), the persistent tail with
    its full LDS capacity (small instances: whole layers in one launch), and with a shrunken capacity (the tail then takes over tables
    that earlier launches folded in HBM, as it does at 2^16 and beyond) — must all produce the oracle's bytes (committed digests, 2^12 and
    2^16).  Likewise the row half of the derefs commitment running ahead on the helper's CU-masked stream (2^16: on by default there),
    switched off, without the mask, with another split of the CUs, and the host tail of the sum-checks at other lengths, in its scalar form
    (OTTI_HOST_FR8=0), single-threaded and with every vector pair handed to a helper thread; the armed launches' hand-over in its older form
    (OTTI_RELAY=0) and with four pollers in the leader workgroup; the hash layer's evaluations and bounds as separate passes (OTTI_HASH_FUSED=0), the
    fused pass on a second stream beside the memory circuits' sum-check (OTTI_HASH_AHEAD=1), plain or CU-masked, the host-only layers exported one launch at a time."""
    import os, subprocess, sys
    g = _golden_snark()[1 << lg]
    e = dict(os.environ); e.update(env)
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "snark_tail_worker.py"), str(lg)], env=e,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("DIGEST")][-1].split()
    assert line[1] == g["commitment_sha256"] and line[2] == g["proof_sha256"], (env, res.stderr)
    if "OTTI_TEST_TAIL_DROP" in env:
        # half of the persistent tail's workgroups never run (what a grid that is not resident as a whole looks like to the host): the wait
        # gives up, the launch is aborted, and SNARK::prove repeats the proof with a launch per round — same bytes, and it says so once
        assert "repeats the proof with one launch per sum-check round" in res.stderr, res.stderr


def test_concurrent_snark_and_nizk_provers_in_one_process_match_the_oracle():
    """Three prover threads in one process — two in SNARK mode (2^14: the size from which a lone proof commits its dereferenced rows
    ahead of time on a helper thread's CU-masked stream and arms its launches; with company it must do neither) and one in NIZK mode —
    each proving several times while the others start and finish around it: every proof equals the oracle's, whatever the number of
    proofs in flight was when it started."""
    import threading
    cases = []
    for k, (n, kind) in enumerate([(1 << 14, "uniform"), (1 << 14, "compiler"), (1 << 13, "uniform")]):
        r, nz = _case(n, 10, kind, seed=11 + k)
        oi = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
        inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
        if k < 2:
            og, gens = orc.OSnarkGens(r["num_cons"], r["num_vars"], r["num_inputs"], nz), oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
            oc, comm = orc.OSnarkComm.encode(oi, og), oa.ComputationCommitment.encode(inst, gens)
            assert comm.bytes == oc.bytes
            want = [orc.snark_prove(oi, oc, r["vars"], r["inputs"], og, LABEL, bytes([s]) * 32)[0] for s in (1, 2)]
            cases.append(("snark", r, inst, gens, comm, want))
        else:
            og, gens = orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"]), oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
            want = [orc.nizk_prove(oi, r["vars"], r["inputs"], og, LABEL, bytes([s]) * 32)[0] for s in (1, 2)]
            cases.append(("nizk", r, inst, gens, None, want))
    bad, go = [], threading.Event()

    def worker(idx):
        mode, r, inst, gens, comm, want = cases[idx]
        v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
        go.wait()
        try:
            for rep in range(6):
                s = 1 + rep % 2
                p = (oa.SNARK.prove(inst, comm, v, i, gens, LABEL, bytes([s]) * 32) if mode == "snark" else oa.NIZK.prove(inst, v, i, gens, LABEL, bytes([s]) * 32))
                if p.bytes != want[s - 1]:
                    bad.append((idx, rep, "bytes differ"))
        except Exception as ex:                                    # noqa: BLE001 — reported by the main thread
            bad.append((idx, -1, repr(ex)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in ths:
        t.start()
    go.set()
    for t in ths:
        t.join()
    assert not bad, bad
    # and alone again afterwards: the lone-proof path (ahead-of-time rows, armed launches, persistent tail) on the contexts the threads left behind
    mode, r, inst, gens, comm, want = cases[0]
    assert oa.SNARK.prove(inst, comm, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]), gens, LABEL, b"\x01" * 32).bytes == want[0]


def _degenerate(name):
    base = oa.synth_r1cs(64, 3, 5)
    empty = base["A"][:0]
    if name == "all matrices empty":
        return dict(base, A=empty, B=empty, C=empty)
    if name == "A and C empty":                                   # 0 * (B z) = 0
        return dict(base, A=empty, C=empty)
    if name == "B and C empty":
        return dict(base, B=empty, C=empty)
    if name == "one constraint, no inputs":
        return oa.synth_r1cs(1, 0, 3)
    if name == "three constraints":
        return oa.synth_r1cs(3, 2, 3)
    return oa.synth_r1cs(1 << 12, 0, 3)                           # "no inputs"


@pytest.mark.parametrize("name", ["all matrices empty", "A and C empty", "B and C empty", "one constraint, no inputs", "three constraints", "no inputs"])
def test_empty_and_degenerate_instances_in_both_modes(name):
    """matrices without entries (SNARK mode pads each to two operations), a single constraint, no public inputs: commitment and proofs
    of both modes against the oracle, both verifiers accept"""
    r = _degenerate(name)
    nz = int(max(r["A"].size, r["B"].size, r["C"].size, 1))
    oi = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    og, gens = orc.OSnarkGens(r["num_cons"], r["num_vars"], r["num_inputs"], nz), oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    oc, comm = orc.OSnarkComm.encode(oi, og), oa.ComputationCommitment.encode(inst, gens)
    assert comm.bytes == oc.bytes
    p = oa.SNARK.prove(inst, comm, v, i, gens, LABEL, SEED)
    assert p.bytes == orc.snark_prove(oi, oc, r["vars"], r["inputs"], og, LABEL, SEED)[0]
    p.verify(oa.ComputationCommitment.from_bytes(comm.bytes), i, gens, LABEL)
    ng, ong = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"]), orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
    pn = oa.NIZK.prove(inst, v, i, ng, LABEL, SEED)
    assert pn.bytes == orc.nizk_prove(oi, r["vars"], r["inputs"], ong, LABEL, SEED)[0]
    pn.verify(inst, i, ng, LABEL)
