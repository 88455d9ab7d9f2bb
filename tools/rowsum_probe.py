"""Device time of the verifier's variable-base row sum (k_decode_niels + k_msm_var) by size, through otti_k_row_sum."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import otti_amd as oa
K = oa.kernels
rng = np.random.default_rng(1)
for n, lgv in ((256, 16), (512, 18), (1024, 20), (2048, 22), (4096, 24)):
    g = oa.NIZKGens.new(1 << lgv, 1 << lgv, 1)
    C = g.points(n)
    raw = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    s = oa.fr_from_ints([int.from_bytes(raw[k].tobytes(), "little") % oa.L_ORDER for k in range(n)])
    K.row_sum(C, s)
    oa.stats_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        K.row_sum(C, s)
    dt = (time.perf_counter() - t0) / 5
    st = oa.stats_read(); oa.stats_enable(False)
    print("n=%5d  wall %.3f ms per call; device: decode %.3f ms, msm_var %.3f ms per launch" % (n, 1e3 * dt, st["decode"][1] / max(1, st["decode"][0]), st["msm_var"][1] / max(1, st["msm_var"][0])), flush=True)
