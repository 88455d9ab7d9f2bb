"""Randomised differential soak: proofs of random instance shapes / sizes / input counts / labels / seeds from the GPU prover
against the CPU oracle, for a given number of seconds.  usage (GPU box): python3 tools/soak.py [seconds] [rng seed] [nizk|snark]
(snark: computation commitment and SNARK proof bytes, sizes up to 2^11)  [lo:hi] as a fourth argument pins log2 of the uniform / compiler
sizes to lo..hi-1 (e.g. snark 14:17 — the sizes at which SNARK::prove commits the dereferenced rows ahead of time on a helper thread)"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import otti_amd as oa  # noqa: E402
import orc  # noqa: E402
from shard_worker import make_r1cs  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mode = sys.argv[3] if len(sys.argv) > 3 else "nizk"
orc.set_threads(min(16, os.cpu_count() or 1))
lg_range = tuple(int(x) for x in sys.argv[4].split(":")) if len(sys.argv) > 4 else None
t_end, n_ok, n_bad = time.time() + budget, 0, 0
while time.time() < t_end:
    dist = str(rng.choice(["uniform", "compiler"] if lg_range else ["uniform", "compiler", "many_cons", "many_vars", "odd"]))
    lg = int(rng.integers(3 if dist == "compiler" else 1, (12 if mode == "snark" else 14) if dist in ("uniform", "compiler") else 9))
    if lg_range:
        lg = int(rng.integers(lg_range[0], lg_range[1]))
    ni = int(rng.integers(0, min(12, (1 << lg) - 1) + 1))
    if dist == "odd":                                         # sizes that are not powers of two: exercises the padding rules
        n = int(rng.integers(2, 3000)); ni = int(rng.integers(0, min(12, n - 1) + 1))
        r = oa.synth_r1cs(n, ni, int(rng.integers(1, 1 << 30)))
    else:
        r = make_r1cs(lg, dist, ni)
    label, seed = bytes(rng.integers(97, 123, int(rng.integers(1, 20)), dtype=np.uint8)), bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    if mode == "snark":
        nz = int(max(r["A"].size, r["B"].size, r["C"].size, 1))
        oi, og = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"]), orc.OSnarkGens(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
        oc = orc.OSnarkComm.encode(oi, og)
        want, _ = orc.snark_prove(oi, oc, r["vars"], r["inputs"], og, label, seed)
        gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
        comm = oa.ComputationCommitment.encode(inst, gens)
        inputs = oa.InputsAssignment.new(r["inputs"])
        p = oa.SNARK.prove(inst, comm, oa.VarsAssignment.new(r["vars"]), inputs, gens, label, seed)
        ok = comm.bytes == oc.bytes and p.bytes == want
        if ok:
            p.verify(oa.ComputationCommitment.from_bytes(comm.bytes), inputs, gens, label)
        n_ok += ok; n_bad += (not ok)
        if not ok:
            print("MISMATCH", dist, lg, r["num_cons"], r["num_vars"], ni, label, seed.hex())
        continue
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    p = oa.NIZK.prove(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]), gens, label, seed)
    oi, og = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"]), orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
    want, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og, label, seed)
    ok = p.bytes == want
    if ok:
        p.verify(inst, oa.InputsAssignment.new(r["inputs"]), gens, label)
    n_ok += ok; n_bad += (not ok)
    if not ok:
        print("MISMATCH", dist, lg, r["num_cons"], r["num_vars"], ni, label, seed.hex())
print("soak (%s mode): %d proofs identical to the oracle's, %d mismatches" % (mode, n_ok, n_bad))
sys.exit(1 if n_bad else 0)
