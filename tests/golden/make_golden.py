#!/usr/bin/env python3
"""Generates tests/golden/*.json.  Run in the build container only (needs /opt/conda/lib/libsodium.so.23, which the GPU box's
tests never touch).  Sources of truth:
  * libsodium 1.0.18 ristretto255 / scalar API  (RFC 9496 conformant) — points, scalar mults, Pedersen commitments;
  * Python big integers — GF(l) arithmetic;
  * hashlib — SHA3 / SHAKE256;
  * SURVEY.md App. B — STROBE / Merlin known answers (copied as constants);
  * the CPU oracle itself — whole-proof SHA-256 digests (regression pins for oracle and GPU alike; the reference ships no
    golden proofs: /root/reference/Spartan is an empty submodule).
"""
import ctypes
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
S = ctypes.CDLL("/opt/conda/lib/libsodium.so.23")
assert S.sodium_init() >= 0
L = 2 ** 252 + 27742317777372353535851937790883648493
rnd = random.Random(20261003)


def buf(n):
    return ctypes.create_string_buffer(n)


def from_hash(u):
    o = buf(32); assert S.crypto_core_ristretto255_from_hash(o, u) == 0; return o.raw


def smul(x, p):
    o = buf(32); assert S.crypto_scalarmult_ristretto255(o, (x % L).to_bytes(32, "little"), p) == 0; return o.raw


def padd(a, b):
    o = buf(32); assert S.crypto_core_ristretto255_add(o, a, b) == 0; return o.raw


def gens(label, count):
    base = bytes.fromhex("e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76")
    stream = hashlib.shake_256(label + base).digest(64 * count)
    return [from_hash(stream[64 * i: 64 * i + 64]) for i in range(count)]


def main():
    prim = {}
    prim["from_uniform_bytes"] = []
    for _ in range(24):
        u = rnd.randbytes(64); prim["from_uniform_bytes"].append({"in": u.hex(), "out": from_hash(u).hex()})
    prim["scalarmul"] = []
    for _ in range(12):
        p = from_hash(rnd.randbytes(64)); x = rnd.randrange(1, L)
        prim["scalarmul"].append({"point": p.hex(), "scalar": x.to_bytes(32, "little").hex(), "out": smul(x, p).hex()})
    prim["point_add"] = []
    for _ in range(8):
        a, b = from_hash(rnd.randbytes(64)), from_hash(rnd.randbytes(64))
        prim["point_add"].append({"a": a.hex(), "b": b.hex(), "out": padd(a, b).hex()})
    prim["fr"] = []
    for _ in range(32):
        x, y = rnd.randrange(L), rnd.randrange(L); w = rnd.randbytes(64)
        prim["fr"].append({"x": x.to_bytes(32, "little").hex(), "y": y.to_bytes(32, "little").hex(), "mul": (x * y % L).to_bytes(32, "little").hex(),
                           "add": ((x + y) % L).to_bytes(32, "little").hex(), "sub": ((x - y) % L).to_bytes(32, "little").hex(),
                           "inv": pow(x, -1, L).to_bytes(32, "little").hex(), "wide": w.hex(),
                           "wide_reduced": (int.from_bytes(w, "little") % L).to_bytes(32, "little").hex()})
    # generator stream (commitments.rs MultiCommitGens::new) and Pedersen commitments over it
    G = gens(b"gens_r1cs_sat", 40)
    prim["gens_r1cs_sat"] = [g.hex() for g in G]
    prim["pedersen"] = []
    for n in (1, 3, 4, 34):
        xs = [rnd.randrange(L) for _ in range(n)]; blind = rnd.randrange(L)
        acc = smul(blind, G[n])          # h = point after the n bases: MultiCommitGens::new(n, label)
        for x, g in zip(xs, G):
            if x:
                acc = padd(acc, smul(x, g))
        prim["pedersen"].append({"n": n, "scalars": [x.to_bytes(32, "little").hex() for x in xs], "blind": blind.to_bytes(32, "little").hex(), "out": acc.hex()})
    # SURVEY App. B constants
    prim["appendix_b"] = {
        "strobe_prf1": "b48e645ca17c667fd5206ba57a6a228d72d8e1903814d3f17f622996d7cfefb0",
        "strobe_prf2": "07e45cce8078cee259e3e375bb85d75610e2d1e1201c5f645045a194edd49ff8",
        "merlin_challenge": "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615",
        "basepoint": "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
        "basepoint_x2": "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919",
        "espresso_map": "3066f82a1a747d45120d1740f14358531a8f04bbffe6a819f86dfe50f44a0a46",
        "wide_00_3f": "7a3c6282f02d37a05023b60d5428e6cc5961d4c31221937adae0b574e4d07205",
        "gens_first3": ["f8dad3b0fba18ec2a61684952cbfd51372cbdcca26b05e5b0b4637157c98ca43",
                        "da819f7228eaa0de8b0112cc7520a7367292513556bd70d3f7b68cf86e962d23",
                        "b08bef187b662be947be60e24e92240b4100e7fb26197837098c11a131845f05"],
        "pedersen_kat_msm": "78f853ecdff2c5075dd183838d785fbea76802feb32e259cbc2d152575398d70",
    }
    with open(os.path.join(HERE, "primitives.json"), "w") as f:
        json.dump(prim, f, indent=1)

    # whole-proof digests from the oracle on the synthetic instance (needs the product library only for the instance generator)
    import otti_amd as oa
    import orc
    proofs = []
    sizes = [(2, 0), (4, 1), (16, 3), (64, 10), (1 << 10, 10), (1 << 12, 10), (1 << 14, 10)]
    # BASELINE.json's sweep sizes (configs[1], the 2^20 headline, 2^22, and configs[4]'s 2^24): minutes of oracle time and ~20 GB of
    # host memory at 2^24, so only with --large; without it the entries already in proofs.json are kept as they are
    large = [(1 << 16, 10), (1 << 18, 10), (1 << 20, 10), (1 << 22, 10), (1 << 24, 10)]
    path = os.path.join(HERE, "proofs.json")
    keep = []
    if "--large" in sys.argv:
        sizes += large
    elif os.path.exists(path):
        keep = [e for e in json.load(open(path)) if (e["n"], e["num_inputs"]) in large]
    for n, ni in sizes:
        r = oa.synth_r1cs(n, ni, 1)
        oinst = orc.OInstance(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
        ogens = orc.OGens(r["num_cons"], r["num_vars"], r["num_inputs"])
        pf, _ = orc.nizk_prove(oinst, r["vars"], r["inputs"], ogens, b"nizk_example", b"\x2a" * 32)
        assert orc.nizk_verify(oinst, r["inputs"], ogens, pf) == 0
        proofs.append({"n": n, "num_inputs": ni, "instance_seed": 1, "tape_seed": "2a" * 32, "label": "nizk_example", "proof_len": len(pf),
                       "proof_sha256": hashlib.sha256(pf).hexdigest(),
                       "witness_sha256": hashlib.sha256(r["vars"].tobytes() + r["inputs"].tobytes()).hexdigest(),
                       "matrices_sha256": hashlib.sha256(r["A"].tobytes() + r["B"].tobytes() + r["C"].tobytes()).hexdigest()})
        print("oracle proof 2^%d done" % (n.bit_length() - 1), flush=True)
        del r, oinst, ogens, pf
    with open(path, "w") as f:
        json.dump(proofs + keep, f, indent=1)
    print("wrote primitives.json, proofs.json")


if __name__ == "__main__":
    main()
