"""One proof sharded over g processes: proof bytes must equal the single-GPU proof and the CPU oracle's, for
g in {2, 4} (8c: "proof_bytes(GPU, g GPUs) == proof_bytes(CPU restatement)").  The GPU box has ONE card, so the g ranks share it —
each rank still runs the full per-rank code path (its own row block of the commitment, its low-bit slice of every table, its
row/column slices of the instance) and the exchange between real processes; only the physical placement differs."""
import os
import uuid
import numpy as np
import pytest

import otti_amd as oa
import orc
from test_shard_cpu import run_ranks
from shard_worker import SEED, LABEL

pytestmark = pytest.mark.gpu


def _reference_proofs(lg, dist, ni):
    from shard_worker import make_r1cs
    r = make_r1cs(lg, dist, ni)
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    single = oa.NIZK.prove(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]), gens, LABEL, SEED)
    single.verify(inst, oa.InputsAssignment.new(r["inputs"]), gens, LABEL)
    oinst = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    ogens = orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
    oracle, _ = orc.nizk_prove(oinst, r["vars"], r["inputs"], ogens, LABEL, SEED)
    return single.bytes, oracle


@pytest.mark.parametrize("world,lg,dist,ni", [(2, 10, "uniform", 4), (4, 12, "uniform", 10), (4, 11, "compiler", 3), (2, 4, "uniform", 2),
                                               (4, 6, "uniform", 1), (4, 8, "many_cons", 3), (2, 7, "many_vars", 2), (4, 5, "many_vars", 1)])
def test_sharded_proof_is_byte_identical(tmp_path, world, lg, dist, ni):
    single, oracle = _reference_proofs(lg, dist, ni)
    assert single == oracle
    seg = "otti-test-" + uuid.uuid4().hex
    env_dev = os.environ.get("OTTI_DEVICE")
    os.environ["OTTI_DEVICE"] = "0"
    try:
        run_ranks(lambda r: ["prove", seg, str(r), str(world), str(tmp_path / ("p%d.bin" % r)), str(lg), dist, str(ni)], world, timeout=500)
    finally:
        if env_dev is None:
            os.environ.pop("OTTI_DEVICE")
        else:
            os.environ["OTTI_DEVICE"] = env_dev
    for r in range(world):
        got = open(tmp_path / ("p%d.bin" % r), "rb").read()
        assert got == single, "rank %d of %d returned different proof bytes" % (r, world)


@pytest.mark.parametrize("world,lg,dist,ni", [(2, 10, "uniform", 4), (4, 12, "uniform", 10), (4, 11, "compiler", 3), (2, 6, "uniform", 2)])
def test_sharded_snark_proof_is_byte_identical(tmp_path, world, lg, dist, ni):
    """SNARK mode over g rank processes (sharing the box's one card): the R1CS part sharded as in NIZK mode, the rows of the derefs commitment
    dealt out over the ranks and gathered; commitment and proof bytes of every rank equal the single-GPU prover's and the CPU oracle's."""
    from shard_worker import make_r1cs
    r = make_r1cs(lg, dist, ni)
    nz = int(max(r["A"].size, r["B"].size, r["C"].size))
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    comm = oa.ComputationCommitment.encode(inst, gens)
    inputs = oa.InputsAssignment.new(r["inputs"])
    single = oa.SNARK.prove(inst, comm, oa.VarsAssignment.new(r["vars"]), inputs, gens, LABEL, SEED)
    single.verify(oa.ComputationCommitment.from_bytes(comm.bytes), inputs, gens, LABEL)
    oi, og = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"]), orc.OSnarkGens(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    oc = orc.OSnarkComm.encode(oi, og)
    assert oc.bytes == comm.bytes
    assert orc.snark_prove(oi, oc, r["vars"], r["inputs"], og, LABEL, SEED)[0] == single.bytes
    seg = "otti-test-" + uuid.uuid4().hex
    saved = {k: os.environ.get(k) for k in ("OTTI_DEVICE", "OTTI_TRACE")}
    os.environ["OTTI_DEVICE"] = "0"; os.environ["OTTI_TRACE"] = "1"
    try:
        outs = run_ranks(lambda k: ["snark", seg, str(k), str(world), str(tmp_path / ("s%d.bin" % k)), str(lg), dist, str(ni)], world, timeout=900)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    # the product circuits are split by residue classes whenever they are large enough for it (64 elements per rank): every case here but the 2^6 one
    assert all(("product circuits split by residue classes over %d ranks" % world in o) == (lg >= 10) for o in outs), outs[0][-2000:]
    for k in range(world):
        got = open(tmp_path / ("s%d.bin" % k), "rb").read()
        nc = int.from_bytes(got[:8], "little")
        assert got[8:8 + nc] == comm.bytes, "rank %d of %d: another computation commitment" % (k, world)
        assert got[8 + nc:] == single.bytes, "rank %d of %d returned different SNARK proof bytes" % (k, world)


def test_too_many_ranks_for_the_instance_is_refused():
    r = oa.synth_r1cs(4, 1, 5)                                                      # 2 x 2 witness matrix: cannot give 4 ranks a row each
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    wit = oa.Witness(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
    with pytest.raises(oa.SpartanError):
        oa.NIZK.prove_sharded(inst, wit, gens, LABEL, SEED)                         # no otti_shard_init


def test_sharded_proof_over_the_rccl_transport_with_a_world_of_one(tmp_path):
    """The per-round sums of a sharded proof through the RCCL transport's device path (allreduce_fr_device: the round kernel's totals packed into
    u64 lanes where it left them in HBM, scaled by the rank's eq factor, ncclAllReduce, one download) — RCCL refuses two ranks on one
    card, so a world of one: the proof must equal the unsharded one."""
    single, oracle = _reference_proofs(10, "uniform", 4)
    seg = "otti-test-" + uuid.uuid4().hex
    saved = {k: os.environ.get(k) for k in ("OTTI_DEVICE", "OTTI_SHARD_TRANSPORT")}
    os.environ["OTTI_DEVICE"] = "0"; os.environ["OTTI_SHARD_TRANSPORT"] = "rccl"
    try:
        run_ranks(lambda r: ["prove", seg, "0", "1", str(tmp_path / "p0.bin"), "10", "uniform", "4"], 1, timeout=500)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert open(tmp_path / "p0.bin", "rb").read() == single == oracle


def test_rccl_lane_transport_of_the_round_sums_on_one_card(tmp_path):
    """OTTI_SHARD_TRANSPORT=rccl: allreduce_fr goes pack -> ncclAllReduce(ncclUint64, ncclSum) on the GPU -> normalise mod l.  RCCL
    refuses two ranks on one card, so this box can only run a world of one — which still exercises the whole path (bootstrap of the
    communicator through the mailbox, lanes on the device, the collective call, carry propagation and reduction); the arithmetic of
    summing lanes across ranks is rehearsed with gloo in tests/test_dist_cpu.py."""
    import subprocess, sys
    out = tmp_path / "x.npz"
    env = dict(os.environ, OTTI_SHARD_TRANSPORT="rccl", OTTI_DEVICE="0")
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shard_worker.py"), "exchange",
                          "otti-test-" + uuid.uuid4().hex, "0", "1", str(out)], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    got = np.load(out)
    assert bytes(got["transport"]).decode() == "rccl"
    rng = np.random.default_rng(1234)
    for nbytes in [1, 32, 96, 4096, 1 << 20, 7, 96, 96, 96]:
        rng.integers(0, 256, nbytes, dtype=np.uint8)
    for it, n in enumerate([1, 3, 1000, 40000]):
        vals = [int(x) for x in rng.integers(0, 2 ** 62, n)]
        vals[0] = oa.L_ORDER - 1
        assert np.array_equal(got["r%d" % it], oa.fr_from_ints(vals)), it     # a world of one: the sum is the rank's own values
