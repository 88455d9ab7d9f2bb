"""Per-proof stage times plus the sum-check loop's own split (OTTI_TRACE: wait / begin / launch / finish) for a resident witness."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OTTI_TRACE"] = "1"
import otti_amd as oa

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
r = oa.synth_r1cs(1 << lg, 10, 1)
inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
vars_, inputs = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
inst.prepare_device(gens)
w = oa.Witness(inst, vars_, inputs)
for i in range(8):
    t = time.perf_counter()
    p = oa.NIZK.prove(inst, w, None, gens, b"x", bytes([i + 1]) * 32)
    dt = (time.perf_counter() - t) * 1e3
    print("prove %.3f ms  " % dt + "  ".join("%s %.3f" % kv for kv in p.stage_ms.items()), flush=True)
