import sys, time
sys.path.insert(0, '.')
import otti_amd as oa
r = oa.synth_r1cs(1 << 20, 10, 1)
inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
inst.prepare_device(gens)
for k in range(6):
    t = time.perf_counter(); p = oa.NIZK.prove(inst, v, i, gens, b"x", bytes([k + 1]) * 32); dt = (time.perf_counter() - t) * 1e3
    print("host-pointer prove %.2f ms (prove stages total %.2f)" % (dt, p.stage_ms["total"]))
