"""Achieved HBM bandwidth of the streaming kernels through the kernel-level C ABI (HIP-event times of the kernels alone)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import otti_amd as oa
K = oa.kernels
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n = 1 << lg
rng = np.random.default_rng(1)
def rnd(m):
    a = rng.integers(0, 256, size=(m, 32), dtype=np.uint8); a[:, 31] &= 0x0f; return a
A, B, C, D = rnd(n), rnd(n), rnd(n), rnd(n); r = rnd(1); tau = rnd(lg)
for name, f, nbytes in [
    ("fold_top", lambda: K.fold_top(A, r)[1], 32 * n * 1.5),
    ("fold_bot", lambda: K.fold_bot(A, r)[1], 32 * n * 1.5),
    ("sc_cubic_round", lambda: K.sc_cubic_round(A, B, C, D)[1], 4 * 32 * n),
    ("sc_cubic_fold_round", lambda: K.sc_cubic_fold_round(A, B, C, D, r)[2], 4 * 32 * n * 1.5),
    ("sc_cubic3_round", lambda: K.sc_cubic3_round(B, C, D, tau[1:])[1], 3 * 32 * n),
    ("sc_cubic3_fold_round", lambda: K.sc_cubic3_fold_round(B, C, D, r, tau[2:])[2], 3 * 32 * n * 1.5),
    ("sc_quad_round", lambda: K.sc_quad_round(A, B)[1], 2 * 32 * n),
    ("sc_quad_fold_round", lambda: K.sc_quad_fold_round(A, B, r)[2], 2 * 32 * n * 1.5),
    ("fr_mul", lambda: K.fr_op("mul", A, B)[1], 3 * 32 * n),
    ("fr_add", lambda: K.fr_op("add", A, B)[1], 3 * 32 * n),
    ("eq_evals", lambda: K.eq_evals(rnd(lg))[1], 32 * n),
]:
    f(); ms = min(f() for _ in range(3))
    print(f"{name:22s} n=2^{lg} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s")
