"""Which stage do the slow proofs lose their time in?  N proofs of the 2^lg instance, one after the other (witness resident, as bench.py's timed loop);
prints the distribution and, for every proof slower than 1.3 x the median, its stage split.  usage (GPU box): python3 tools/outlier_probe.py [lg [N]]"""
import gc, os, sys, time
if os.environ.get("PROBE_TORCH") == "1":
    import torch
    torch.cuda.is_available()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = 1 << lg
r = oa.synth_r1cs(n, 10, 1)
inst = oa.Instance.new(n, n, 10, r["A"], r["B"], r["C"]); gens = oa.NIZKGens.new(n, n, 10)
inst.prepare_device(gens)
wit = oa.Witness(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
seed = b"\x07" * 32
for _ in range(5): oa.NIZK.prove(inst, wit, None, gens, b"probe", seed)
for mode in ("run 1", "run 2"):
    ms, st = [], []
    for _ in range(N):
        t0 = time.perf_counter(); p = oa.NIZK.prove(inst, wit, None, gens, b"probe", seed); ms.append(1e3 * (time.perf_counter() - t0)); st.append(dict(p.stage_ms))
    s = sorted(ms); med = s[len(s) // 2]
    print(f"{mode}: {N} proofs: mean {sum(ms)/N:.3f} ms, p50 {med:.3f}, p90 {s[int(.9*N)]:.3f}, p99 {s[int(.99*N)]:.3f}, max {s[-1]:.3f}; slower than 1.3 x p50: {sum(1 for x in ms if x > 1.3*med)}")
    for i, x in enumerate(ms):
        if x > 1.3 * med and os.environ.get("PROBE_VERBOSE") == "1": print(f"   proof {i}: {x:.3f} ms  " + ", ".join(f"{k} {v:.2f}" for k, v in st[i].items()))
