"""SNARK mode timings on the GPU: SNARK::encode once, then SNARK::prove (stage split) and SNARK::verify, synthetic instance of 2^lg constraints."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 1 << lg
r = oa.synth_r1cs(n, 10, 1)
nz = max(r["A"].size, r["B"].size, r["C"].size)
inst = oa.Instance.new(n, n, 10, r["A"], r["B"], r["C"])
t0 = time.perf_counter(); gens = oa.SNARKGens.new(n, n, 10, nz); t1 = time.perf_counter()
comm = oa.ComputationCommitment.encode(inst, gens); t2 = time.perf_counter()
print(f"2^{lg}: SNARKGens::new {1e3*(t1-t0):.1f} ms, SNARK::encode {1e3*(t2-t1):.1f} ms (includes building the window table), commitment {len(comm.bytes)} bytes", flush=True)
v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
for k in range(reps):
    t0 = time.perf_counter(); p = oa.SNARK.prove(inst, comm, v, i, gens, b"snark_example", b"\x2a" * 32); t1 = time.perf_counter()
    print(f"  prove {1e3*(t1-t0):.2f} ms  ({n/(t1-t0)/1e6:.1f} M constraints/s), proof {len(p.bytes)} bytes; stages: " + ", ".join(f"{k} {x:.2f}" for k, x in p.stage_ms.items()), flush=True)
vc = oa.ComputationCommitment.from_bytes(comm.bytes)
for k in range(3):                                              # the first call allocates the verifier's device buffers (kept with the context)
    t0 = time.perf_counter(); p.verify(vc, i, gens, b"snark_example"); t1 = time.perf_counter()
    print(f"  verify {1e3*(t1-t0):.1f} ms", flush=True)
# device objects go before the interpreter starts taking modules apart (under rocprofv3 a fault in the process's teardown was seen otherwise)
del p, comm, vc, gens, inst, v, i
import gc
gc.collect()
