#!/usr/bin/env python3
"""Generates tests/golden/snark_proofs.json: SHA-256 digests of the CPU oracle's SNARK-mode outputs (computation commitment and
whole proof) on the synthetic instances of SURVEY.md 8(d), for the sizes SNARK mode is benchmarked and swept on.

  python tests/golden/make_golden_snark.py [log2 ...]        default: 16 18 20

The oracle (oracle/snark.c) is the checker; its SNARK prover takes minutes at 2^20, so the GPU tests and bench.py compare against
these committed digests instead of running it.  Like tests/golden/proofs.json these are regression pins for oracle and GPU alike —
the reference ships no golden proofs (/root/reference/Spartan is an empty submodule): parity with upstream's bytes stays unpinned.
Entries of sizes not asked for are kept as they are.
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    import otti_amd as oa          # instance generator only (host code of the library; no GPU needed)
    import orc
    lgs = [int(a) for a in sys.argv[1:]] or [16, 18, 20]
    orc.set_threads(os.cpu_count() or 1)
    path = os.path.join(HERE, "snark_proofs.json")
    entries = {e["n"]: e for e in json.load(open(path))} if os.path.exists(path) else {}
    label, seed, ni = b"snark_example", b"\x2a" * 32, 10
    for lg in lgs:
        n = 1 << lg
        r = oa.synth_r1cs(n, ni, 1)
        nz = int(max(r["A"].size, r["B"].size, r["C"].size))
        oi = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
        og = orc.OSnarkGens(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
        t0 = time.perf_counter(); oc = orc.OSnarkComm.encode(oi, og); t1 = time.perf_counter()
        pf, _ = orc.snark_prove(oi, oc, r["vars"], r["inputs"], og, label, seed); t2 = time.perf_counter()
        cb = oc.bytes
        assert orc.snark_verify(orc.OSnarkComm.parse(cb), r["inputs"], og, pf, label) == 0
        entries[n] = {"n": n, "num_inputs": ni, "num_nz_entries": nz, "instance_seed": 1, "tape_seed": "2a" * 32, "label": "snark_example",
                      "commitment_len": len(cb), "commitment_sha256": hashlib.sha256(cb).hexdigest(),
                      "proof_len": len(pf), "proof_sha256": hashlib.sha256(pf).hexdigest(),
                      "witness_sha256": hashlib.sha256(r["vars"].tobytes() + r["inputs"].tobytes()).hexdigest(),
                      "matrices_sha256": hashlib.sha256(r["A"].tobytes() + r["B"].tobytes() + r["C"].tobytes()).hexdigest()}
        print("oracle SNARK 2^%d: encode %.1f s, prove %.1f s, verify ok" % (lg, t1 - t0, t2 - t1), flush=True)
        del r, oi, og, oc, pf
        with open(path, "w") as f:
            json.dump([entries[k] for k in sorted(entries)], f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
