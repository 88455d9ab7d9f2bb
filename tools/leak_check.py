"""Device-memory leak check: free HBM must not drift over hundreds of proofs of changing size (run on a GPU box: python3 tools/leak_check.py)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch, otti_amd as oa
def prep(n):
    r = oa.synth_r1cs(n, 4, 3)
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    return r, inst, gens
free0 = None
for rep in range(6):
    for n in (1 << 10, 1 << 14, 1 << 12):
        r, inst, gens = prep(n)
        for _ in range(40):
            p = oa.NIZK.prove(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]), gens, b"x", b"\x01" * 32)
        p.verify(inst, oa.InputsAssignment.new(r["inputs"]), gens, b"x")
        del inst, gens
    torch.cuda.synchronize()
    free, tot = torch.cuda.mem_get_info()
    if free0 is None: free0 = free
    print("rep", rep, "free GB", round(free / 2**30, 3), "delta MB vs rep0", round((free0 - free) / 2**20, 1))
