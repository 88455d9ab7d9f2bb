"""Run in a FRESH process by test_gpu_parity.py::test_fresh_process_growing_instances: proofs of growing size, first thing after the
library is loaded, so that every result buffer of the device context starts small and has to grow between (never during) proofs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import otti_amd as oa  # noqa: E402
import orc  # noqa: E402

bad = 0
for n, ni in ((6, 1), (300, 7), (1000, 5), (64, 3), (5000, 2), (40000, 9)):
    r = oa.synth_r1cs(n, ni, 4)
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    p = oa.NIZK.prove(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]), gens, b"nizk_example", b"\x2a" * 32)
    oi, og = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"]), orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
    op, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og, b"nizk_example", b"\x2a" * 32)
    ok = p.bytes == op
    bad += 0 if ok else 1
    print(n, "same" if ok else "DIFFERENT")
sys.exit(1 if bad else 0)
