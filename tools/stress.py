"""Repeat the 2^lg proof of one instance for a number of seconds (fixed seed: every proof must have the same bytes, equal to the oracle's
committed digest when there is one) — looks for rare hand-over faults in the armed launches, the persistent sum-check tail and the
ahead-of-time derefs commitment.  usage (GPU box): python3 tools/stress.py [lg [seconds [nizk|snark]]]"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import otti_amd as oa  # noqa: E402

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
mode = sys.argv[3] if len(sys.argv) > 3 else "nizk"
n = 1 << lg
r = oa.synth_r1cs(n, 10, 1)
inst = oa.Instance.new(n, n, 10, r["A"], r["B"], r["C"])
v, i = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
seed = b"\x2a" * 32
gold = None
try:
    for e in json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json" if mode == "nizk" else "snark_proofs.json"))):
        if e["n"] == n and e["num_inputs"] == 10 and e["instance_seed"] == 1:
            gold = e["proof_sha256"]
except Exception:
    pass
if mode == "snark":
    nz = int(max(r["A"].size, r["B"].size, r["C"].size))
    gens = oa.SNARKGens.new(n, n, 10, nz); comm = oa.ComputationCommitment.encode(inst, gens)
    w = oa.Witness(inst, v, i)
    prove = lambda: oa.SNARK.prove(inst, comm, w, None, gens, b"snark_example", seed)
else:
    gens = oa.NIZKGens.new(n, n, 10); inst.prepare_device(gens); w = oa.Witness(inst, v, i)
    prove = lambda: oa.NIZK.prove(inst, w, None, gens, b"nizk_example", seed)
first = hashlib.sha256(prove().bytes).hexdigest()
ok = gold is None or first == gold
t_end, count, bad, worst, t_sum = time.time() + budget, 0, 0, 0.0, 0.0
while time.time() < t_end:
    t0 = time.perf_counter(); p = prove(); dt = time.perf_counter() - t0
    count += 1; t_sum += dt; worst = max(worst, dt)
    if hashlib.sha256(p.bytes).hexdigest() != first:
        bad += 1
print("stress (%s, 2^%d): %d proofs, %d differing from the first, first %s the committed oracle digest; mean %.3f ms, slowest %.3f ms"
      % (mode, lg, count, bad, "equals" if (gold and ok) else ("DIFFERS from" if gold else "has no counterpart in"), 1e3 * t_sum / max(count, 1), 1e3 * worst))
sys.exit(1 if (bad or not ok) else 0)
