"""The node-local exchange a sharded proof uses (otti_amd/csrc/shard.h: shared-memory mailbox between the per-GPU processes),
exercised with 2 and 4 real processes and no GPU: all-gather of byte payloads and exact GF(l) all-reduce."""
import os
import subprocess
import sys
import uuid
import numpy as np
import pytest

import otti_amd as oa

HERE = os.path.dirname(os.path.abspath(__file__))


def run_ranks(args_of_rank, world, timeout=300):
    env = dict(os.environ, OTTI_SHARD_TIMEOUT_S="60")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py")] + args_of_rank(r), env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r])
    return outs


@pytest.mark.parametrize("world", [2, 4])
def test_exchange_between_processes(tmp_path, world):
    seg = "otti-test-" + uuid.uuid4().hex
    run_ranks(lambda r: ["exchange", seg, str(r), str(world), str(tmp_path / ("x%d.npz" % r))], world)
    res = [np.load(tmp_path / ("x%d.npz" % r)) for r in range(world)]
    # what every rank posted, regenerated
    rngs = [np.random.default_rng(1234 + r) for r in range(world)]
    for it, nbytes in enumerate([1, 32, 96, 4096, 1 << 20, 7, 96, 96, 96]):
        want = np.concatenate([g.integers(0, 256, nbytes, dtype=np.uint8) for g in rngs])
        for r in range(world):
            assert np.array_equal(res[r]["g%d" % it], want), (it, r)
    for it, n in enumerate([1, 3, 1000, 40000]):
        cols = []
        for r, g in enumerate(rngs):
            v = [int(x) for x in g.integers(0, 2 ** 62, n)]
            v[0] = oa.L_ORDER - 1 - r
            cols.append(v)
        want = [sum(c[i] for c in cols) % oa.L_ORDER for i in range(n)]
        for r in range(world):
            assert oa.fr_to_ints(res[r]["r%d" % it]) == want, (it, r)
    assert not os.path.exists("/dev/shm/" + seg)                                   # the name is gone once all ranks attached


def test_world_of_one_and_bad_arguments():
    seg = "otti-test-" + uuid.uuid4().hex
    oa.shard_init(seg, 0, 1)
    assert oa.shard_allgather(b"abc", 1) == b"abc"
    a = oa.fr_from_ints([5, oa.L_ORDER - 1])
    assert oa.fr_to_ints(oa.shard_allreduce(a)) == [5, oa.L_ORDER - 1]
    oa.shard_finalize()
    with pytest.raises(oa.SpartanError):
        oa.shard_allgather(b"abc", 1)                                              # no communicator any more
    for rank, world in ((0, 3), (2, 2), (0, 128)):
        with pytest.raises(oa.SpartanError):
            oa.shard_init("otti-test-" + uuid.uuid4().hex, rank, world)
    with pytest.raises(oa.SpartanError):
        oa.shard_init("bad/name", 0, 1)
