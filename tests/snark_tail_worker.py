"""Worker of tests/test_gpu_snark.py::test_persistent_tail_variants...: one SNARK::prove of the synthetic 2^lg instance in a fresh process
(the sum-check tail's switches are read from the environment once per process); prints the SHA-256 of commitment and proof."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa  # noqa: E402

lg = int(sys.argv[1])
r = oa.synth_r1cs(1 << lg, 10, 1)
nz = int(max(r["A"].size, r["B"].size, r["C"].size))
inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
comm = oa.ComputationCommitment.encode(inst, gens)
inputs = oa.InputsAssignment.new(r["inputs"])
proof = oa.SNARK.prove(inst, comm, oa.VarsAssignment.new(r["vars"]), inputs, gens, b"snark_example", b"\x2a" * 32)
again = oa.SNARK.prove(inst, comm, oa.VarsAssignment.new(r["vars"]), inputs, gens, b"snark_example", b"\x2a" * 32)
assert proof.bytes == again.bytes
proof.verify(oa.ComputationCommitment.from_bytes(comm.bytes), inputs, gens, b"snark_example")
print("DIGEST", hashlib.sha256(comm.bytes).hexdigest(), hashlib.sha256(proof.bytes).hexdigest(), flush=True)
