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
# reference points from the runtime's own kernels over the same number of bytes: a pure write (hipMemsetAsync) and a device-to-device
# copy — what a write-only kernel such as eq_evals can be compared with
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def chk(rc):
    if rc != 0: raise RuntimeError(f"HIP error {rc}")
x, y = ctypes.c_void_p(), ctypes.c_void_p()
chk(hip.hipMalloc(ctypes.byref(x), ctypes.c_size_t(32 * n))); chk(hip.hipMalloc(ctypes.byref(y), ctypes.c_size_t(32 * n)))
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
chk(hip.hipEventCreate(ctypes.byref(e0))); chk(hip.hipEventCreate(ctypes.byref(e1)))
def timed(fn):
    chk(hip.hipEventRecord(e0, None)); chk(fn()); chk(hip.hipEventRecord(e1, None)); chk(hip.hipEventSynchronize(e1))
    ms = ctypes.c_float(); chk(hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)); return ms.value
for name, fn, nbytes in [("runtime_fill (write only)", lambda: hip.hipMemsetAsync(x, 0, ctypes.c_size_t(32 * n), None), 32 * n),
                         ("runtime_copy (read + write)", lambda: hip.hipMemcpyAsync(y, x, ctypes.c_size_t(32 * n), 3, None), 64 * n)]:
    timed(fn); ms = min(timed(fn) for _ in range(5))
    print(f"{name:28s} n=2^{lg} {ms*1e3:9.1f} us  {nbytes/ms/1e6:8.1f} GB/s")
hip.hipFree(x); hip.hipFree(y)
