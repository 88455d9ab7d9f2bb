import sys, os, time
sys.path.insert(0, os.getcwd())
import otti_amd as oa
for lg in (20, 22):
    n = 1 << lg
    r = oa.synth_r1cs(n, 10, 1)
    inst = oa.Instance.new(n, n, 10, r["A"], r["B"], r["C"]); gens = oa.NIZKGens.new(n, n, 10)
    inst.prepare_device(gens)
    w = oa.Witness(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
    for i in range(6):
        t0 = time.perf_counter(); p = oa.NIZK.prove(inst, w, None, gens, b"x", b"\x2a" * 32); t1 = time.perf_counter()
        print("2^%d wall %.3f ms, stage total %.3f, gap %.3f" % (lg, 1e3 * (t1 - t0), p.stage_ms["total"], 1e3 * (t1 - t0) - p.stage_ms["total"]), flush=True)
    del w, inst; gens.release_device()
