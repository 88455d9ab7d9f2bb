import os, sys, time
sys.path.insert(0, '.')
os.environ["OTTI_TRACE"] = "1"
import otti_amd as oa
lg = 20
r = oa.synth_r1cs(1 << lg, 10, 1)
inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
p = oa.NIZK.prove(inst, v, i, gens, b"x", b"\x01" * 32)
for k in range(3):
    t = time.perf_counter(); p.verify(inst, i, gens, b"x"); print("verify %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr)
