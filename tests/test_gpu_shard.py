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


def test_too_many_ranks_for_the_instance_is_refused():
    r = oa.synth_r1cs(4, 1, 5)                                                      # 2 x 2 witness matrix: cannot give 4 ranks a row each
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    wit = oa.Witness(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
    with pytest.raises(oa.SpartanError):
        oa.NIZK.prove_sharded(inst, wit, gens, LABEL, SEED)                         # no otti_shard_init
