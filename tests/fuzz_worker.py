"""Mutation fuzzing of the two untrusted-input parsers (run as a child process so that a crash is a test failure, not a dead test
runner): .zkif files and proof bytes.  Every mutant must either load/verify or fail with an ordinary error code.
usage: fuzz_worker.py <workdir> <iterations>"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import otti_amd as oa  # noqa: E402
import orc  # noqa: E402
import fb_reverse_builder as fb  # noqa: E402


def mutate(rng, data):
    b = bytearray(data)
    kind = rng.integers(0, 5)
    if kind == 0 and len(b) > 1:                       # truncate
        del b[rng.integers(0, len(b)):]
    elif kind == 1:                                    # flip bits
        for _ in range(int(rng.integers(1, 6))):
            b[rng.integers(0, len(b))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 2:                                    # overwrite a 4-byte word (offsets / lengths) with an extreme value
        p = int(rng.integers(0, max(1, len(b) - 4)))
        b[p:p + 4] = int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x80000000, len(b), len(b) - 1])).to_bytes(4, "little")
    elif kind == 3:                                    # duplicate a slice
        p, q = sorted(int(x) for x in rng.integers(0, len(b), 2))
        b[p:p] = b[p:q]
    else:                                              # random garbage tail
        b += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
    return bytes(b)


def main():
    work, iters = sys.argv[1], int(sys.argv[2])
    rng = np.random.default_rng(20261003)
    s = oa.synth_r1cs_compiler_like(40, 3, 2)
    paths = [os.path.join(work, n) for n in ("f.zkif", "f.inp.zkif", "f.wit.zkif")]
    oa.zkif_write(s, *paths)
    cons = [([(2, 1)], [(3, 1)], [(4, 1)]), ([(2, 1), (0, 5)], [(0, 1)], [(5, 1)])]
    rev = [fb.circuit_header([1], None, 6, oa.L_ORDER - 1) + fb.constraint_system(cons), fb.circuit_header([1], [96], 6, oa.L_ORDER - 1),
           fb.witness([2, 3, 4, 5], [3, 4, 12, 8])]
    originals = [[open(p, "rb").read() for p in paths], rev]
    loaded = failed = 0
    tmp = [os.path.join(work, n) for n in ("m.zkif", "m.inp.zkif", "m.wit.zkif")]
    for it in range(iters):
        src = originals[it % 2]
        which = int(rng.integers(0, 3))
        for k in range(3):
            open(tmp[k], "wb").write(mutate(rng, src[k]) if k == which else src[k])
        try:
            r = oa.zkif_load(*tmp)
            oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
            loaded += 1
        except oa.SpartanError:
            failed += 1
    # proofs
    r = oa.synth_r1cs(64, 3, 1)
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"]); gens = oa.NIZKGens.new(64, 64, 3)
    oi, og = orc.OInstance(64, 64, 3, r["A"], r["B"], r["C"]), orc.OGens(64, 64, 3)
    proof, _ = orc.nizk_prove(oi, r["vars"], r["inputs"], og)
    inputs = oa.InputsAssignment.new(r["inputs"])
    oa.NIZK(proof).verify(inst, inputs, gens)
    accepted = rejected = 0
    for it in range(iters):
        m = mutate(rng, proof)
        try:
            oa.NIZK(m).verify(inst, inputs, gens)
            accepted += 1
            assert m[:len(proof)] == proof or m == proof, "a modified proof was accepted"
        except oa.ProofVerifyError:
            rejected += 1
    print("zkif: %d loaded, %d refused; proofs: %d accepted, %d rejected" % (loaded, failed, accepted, rejected))


if __name__ == "__main__":
    main()
