"""world_size-2 (and 4) rehearsal, on CPU with gloo, of the multi-GPU layout of SURVEY.md 8(e) / DESIGN.md "Multi-GPU":
tables sharded by the LOW index bits (so bound_poly_var_top never crosses a rank until the last log2(g) rounds), per-round partial
sums exchanged as 8 x u64 lanes per field element with a plain integer SUM all-reduce (what RCCL ncclSum/ncclUint64 does over
xGMI), commitment rows sharded round-robin and all-gathered.  The per-rank table work is done by the CPU oracle (checker role);
the lane packing / normalisation is the product's host code (otti_lanes_pack / otti_lanes_unpack)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _allreduce_fr(parts):
    """sum field elements across ranks: pack -> int64 SUM all-reduce -> normalise"""
    import otti_amd as oa
    lanes = torch.from_numpy(oa.lanes_pack(parts).astype(np.int64))
    dist.all_reduce(lanes, op=dist.ReduceOp.SUM)
    return oa.lanes_unpack(lanes.numpy().astype(np.uint64))


def _worker(rank, world, port, lg_n, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import otti_amd as oa
        import orc
        n, ni = 1 << lg_n, 4
        lg_g = world.bit_length() - 1
        r = oa.synth_r1cs(n, ni, 7)
        oinst = orc.OInstance(n, n, ni, r["A"], r["B"], r["C"])
        rng = np.random.default_rng(99)                                   # same stream on every rank: "the transcript"
        tau = orc.rand_fr(rng, lg_n)
        vars_m = orc.fr_from_ints([int.from_bytes(v.tobytes(), "little") for v in r["vars"]])
        z = np.zeros((2 * n, 32), dtype=np.uint8); z[:n] = vars_m; z[n] = orc.fr_from_ints([1])[0]
        z[n + 1: n + 1 + ni] = orc.fr_from_ints([int.from_bytes(v.tobytes(), "little") for v in r["inputs"]])
        full = [orc.eq_evals(tau)] + orc.multiply_vec(oinst, z)           # eq(tau), Az, Bz, Cz  (every rank could build only its shard)
        local = [t[rank::world].copy() for t in full]                     # low-bit sharding: global i = i' * g + rank
        ref = [t.copy() for t in full]
        ok = True
        for j in range(lg_n - lg_g):
            part = orc.sc_cubic_evals(*local)                             # this rank's (e0, e2, e3)
            tot = _allreduce_fr(part)
            want = orc.sc_cubic_evals(*ref)
            ok &= np.array_equal(tot, want)
            rj = orc.rand_fr(rng, 1)                                      # challenge: identical on every rank
            local = [orc.fold_top(t, rj) for t in local]
            ref = [orc.fold_top(t, rj) for t in ref]
        # last log2(g) rounds: each rank now holds ONE entry per table; gather them to every rank (rank 0 suffices)
        gathered = []
        for t in local:
            buf = [torch.zeros(32, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(buf, torch.from_numpy(t[0].copy()))
            gathered.append(np.stack([b.numpy() for b in buf]))           # entry of rank k = global index k
        ok &= all(np.array_equal(g, f) for g, f in zip(gathered, ref))
        # commitment rows: row i on rank i mod g, all-gather of the compressed points
        lg_v = lg_n; Lsz, Rsz = 1 << (lg_v // 2), 1 << (lg_v - lg_v // 2)
        ogens = orc.OGens(n, n, ni)
        blinds = orc.rand_fr(rng, Lsz)
        mine = list(range(rank, Lsz, world))
        Zl = np.concatenate([vars_m[i * Rsz:(i + 1) * Rsz] for i in mine])
        pts = orc.commit_rows(ogens, Zl, len(mine), Rsz, blinds[mine])
        buf = [torch.zeros((len(mine), 32), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(buf, torch.from_numpy(pts))
        allpts = np.zeros((Lsz, 32), dtype=np.uint8)
        for k in range(world):
            allpts[k::world] = buf[k].numpy()
        ok &= np.array_equal(allpts, orc.commit_rows(ogens, vars_m, Lsz, Rsz, blinds))
        q.put((rank, bool(ok)))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))


@pytest.mark.parametrize("world,lg_n", [(2, 6), (4, 8)])
def test_sharded_sumcheck_and_commit_rows_match_single_rank(world, lg_n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, lg_n, q)) for rk in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(rk, True) for rk in range(world)], res
