"""Where NIZK::verify / SNARK::verify spend their time at 2^lg: wall time per call (the first call allocates the verifier's device buffers),
the verifier's own stage laps (OTTI_TRACE=1) and the device time of its kernel classes (decode, msm_var, msm_small, spmv, eq)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
n = 1 << lg
r = oa.synth_r1cs(n, 10, 1)
inst = oa.Instance.new(n, n, 10, r["A"], r["B"], r["C"])
v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
if mode in ("nizk", "both"):
    gens = oa.NIZKGens.new(n, n, 10)
    p = oa.NIZK.prove(inst, v, i, gens, b"x", b"\x01" * 32)
    for k in range(4):
        if k == 3:
            oa.stats_enable(True)
        t = time.perf_counter(); p.verify(inst, i, gens, b"x"); print("NIZK::verify %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr, flush=True)
    print("  device time of the last call by kernel class (launches, ms):", {k: (c, round(ms, 3)) for k, (c, ms) in oa.stats_read().items() if c}, file=sys.stderr)
    oa.stats_enable(False)
if mode in ("snark", "both"):
    nz = int(max(r["A"].size, r["B"].size, r["C"].size))
    sg = oa.SNARKGens.new(n, n, 10, nz)
    comm = oa.ComputationCommitment.encode(inst, sg)
    sp = oa.SNARK.prove(inst, comm, v, i, sg, b"s", b"\x02" * 32)
    vc = oa.ComputationCommitment.from_bytes(comm.bytes)
    for k in range(4):
        if k == 3:
            oa.stats_enable(True)
        t = time.perf_counter(); sp.verify(vc, i, sg, b"s"); print("SNARK::verify %.2f ms" % ((time.perf_counter() - t) * 1e3), file=sys.stderr, flush=True)
    print("  device time of the last call by kernel class (launches, ms):", {k: (c, round(ms, 3)) for k, (c, ms) in oa.stats_read().items() if c}, file=sys.stderr)
