"""One rank of a sharded proof (tests/test_gpu_shard.py and tests/test_shard_cpu.py start several of these as child processes).
usage: shard_worker.py exchange <segment> <rank> <world> <out.npz>
       shard_worker.py prove    <segment> <rank> <world> <out.bin> <log2 n> <uniform|compiler> <num_inputs>
       shard_worker.py snark    <segment> <rank> <world> <out.bin> <log2 n> <uniform|compiler> <num_inputs>   (SNARK mode: commitment bytes + proof bytes)"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa  # noqa: E402

SEED = bytes([0x2A]) * 32
LABEL = b"shard-test"


def exchange(seg, rank, world, out):
    oa.shard_init(seg, rank, world)
    rng = np.random.default_rng(1234 + rank)
    got = {}
    for it, nbytes in enumerate([1, 32, 96, 4096, 1 << 20, 7, 96, 96, 96]):       # many epochs: both slot parities reused
        mine = rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes()
        got["g%d" % it] = np.frombuffer(oa.shard_allgather(mine, world), dtype=np.uint8)
    for it, n in enumerate([1, 3, 1000, 40000]):                                   # 40000 * 32 B > one slot: chunked
        vals = [int(x) for x in rng.integers(0, 2 ** 62, n)]
        vals[0] = oa.L_ORDER - 1 - rank
        got["r%d" % it] = oa.shard_allreduce(oa.fr_from_ints(vals))
    got["transport"] = np.frombuffer(oa.shard_info()[2].encode(), dtype=np.uint8)
    oa.shard_finalize()
    np.savez(out, **got)


def make_r1cs(lg, dist, ni):
    """uniform / compiler: square synthetic instances; many_cons / many_vars: rectangular ones derived from a 2^lg square instance"""
    n = 1 << lg
    if dist in ("uniform", "compiler"):
        return (oa.synth_r1cs_compiler_like if dist == "compiler" else oa.synth_r1cs)(n, ni, 5)
    r = oa.synth_r1cs(n, ni, 5)
    A, B, C = r["A"].copy(), r["B"].copy(), r["C"].copy()
    if dist == "many_cons":                                     # every constraint 16 times: 16 n constraints over n variables
        def rep(M):
            out = np.tile(M, 16); out["row"] = np.concatenate([M["row"] + k * n for k in range(16)]); return out
        return dict(r, A=rep(A), B=rep(B), C=rep(C), num_cons=16 * n)
    extra = 15 * n                                              # many_vars: n constraints over 16 n variables (the new ones unused)
    for M in (A, B, C):
        M["col"] = np.where(M["col"] >= n, M["col"] + extra, M["col"])
    pad = np.random.default_rng(77).integers(0, 256, size=(extra, 32), dtype=np.uint8); pad[:, 31] &= 0x0f
    return dict(r, A=A, B=B, C=C, vars=np.concatenate([r["vars"], pad]), num_vars=n + extra)


def prove(seg, rank, world, out, lg, dist, ni):
    r = make_r1cs(lg, dist, ni)
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    wit = oa.Witness(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
    oa.shard_init(seg, rank, world)
    for _ in range(2):                                                             # twice: the exchange state carries over between proofs
        pf = oa.NIZK.prove_sharded(inst, wit, gens, LABEL, SEED)
    oa.shard_finalize()
    open(out, "wb").write(pf.bytes)


def snark(seg, rank, world, out, lg, dist, ni):
    r = make_r1cs(lg, dist, ni)
    nz = int(max(r["A"].size, r["B"].size, r["C"].size))
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
    comm = oa.ComputationCommitment.encode(inst, gens)          # every rank encodes (once per circuit; the verifier's bytes are the same everywhere)
    wit = oa.Witness(inst, oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"]))
    oa.shard_init(seg, rank, world)
    for _ in range(2):
        pf = oa.SNARK.prove_sharded(inst, comm, wit, gens, LABEL, SEED)
    oa.shard_finalize()
    open(out, "wb").write(len(comm.bytes).to_bytes(8, "little") + comm.bytes + pf.bytes)


if __name__ == "__main__":
    mode, seg, rank, world, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    if mode == "exchange":
        exchange(seg, rank, world, out)
    elif mode == "snark":
        snark(seg, rank, world, out, int(sys.argv[6]), sys.argv[7], int(sys.argv[8]))
    else:
        prove(seg, rank, world, out, int(sys.argv[6]), sys.argv[7], int(sys.argv[8]))
