"""ctypes wrapper of oracle/liboracle.so — the CPU restatement used ONLY as the checker in tests, smoke() and the
cpu_baseline leg of bench.py.  Nothing under otti_amd/ imports this."""
import ctypes
import os
import subprocess
import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_DIR, "liboracle.so")

L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493
_R = (1 << 256) % L_ORDER
_RINV = pow(_R, -1, L_ORDER)


def build():
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return _SO


def _load():
    if not os.path.exists(_SO):
        build()
    return ctypes.CDLL(_SO)


lib = _load()
_sz, _vp = ctypes.c_size_t, ctypes.c_void_p
lib.orc_gens_new.restype = _vp
lib.orc_gens_new.argtypes = [_sz, _sz, _sz]
lib.orc_instance_new.argtypes = [_sz, _sz, _sz, _vp, _sz, _vp, _sz, _vp, _sz, ctypes.POINTER(_vp)]
lib.orc_nizk_prove.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp, ctypes.c_char_p, _sz, ctypes.c_char_p, ctypes.POINTER(_vp), ctypes.POINTER(_sz),
                               ctypes.POINTER(ctypes.c_double)]
lib.orc_nizk_verify.argtypes = [_vp, _vp, _sz, _vp, ctypes.c_char_p, _sz, _vp, _sz]
lib.orc_instance_is_sat.argtypes = [_vp, _vp, _sz, _vp, _sz, ctypes.POINTER(ctypes.c_int)]
lib.orc_buf_free.argtypes = [_vp]
lib.orc_instance_free.argtypes = [_vp]
lib.orc_gens_free.argtypes = [_vp]
lib.orc_gens_points.argtypes = [_vp, _vp]
lib.orc_eq_evals.argtypes = [_vp, _sz, _vp]
lib.orc_multiply_vec.argtypes = [_vp] * 5
lib.orc_eval_table_sparse.argtypes = [_vp] * 5
lib.orc_fold_top.argtypes = [_vp, _sz, _vp]
lib.orc_fold_bot.argtypes = [_vp, _sz, _vp]
lib.orc_sc_cubic_evals.argtypes = [_vp, _vp, _vp, _vp, _sz, _vp]
lib.orc_sc_quad_evals.argtypes = [_vp, _vp, _sz, _vp]
lib.orc_commit_rows.argtypes = [_vp, _sz, _sz, _vp, _vp, _vp]
lib.orc_poly_bound.argtypes = [_vp, _sz, _sz, _vp, _vp]
lib.orc_set_threads.argtypes = [ctypes.c_int]
lib.orc_bullet_reduce.argtypes = [_vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp]


lib.orc_snark_gens_new.restype = _vp
lib.orc_snark_gens_new.argtypes = [_sz, _sz, _sz, _sz]
lib.orc_snark_gens_free.argtypes = [_vp]
lib.orc_snark_encode.restype = _vp
lib.orc_snark_encode.argtypes = [_vp, _vp]
lib.orc_snark_comm_bytes.argtypes = [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz)]
lib.orc_snark_comm_parse.restype = _vp
lib.orc_snark_comm_parse.argtypes = [_vp, _sz]
lib.orc_snark_comm_free.argtypes = [_vp]
lib.orc_snark_prove.argtypes = [_vp, _vp, _vp, _sz, _vp, _sz, _vp, ctypes.c_char_p, _sz, ctypes.c_char_p, ctypes.POINTER(_vp), ctypes.POINTER(_sz), ctypes.POINTER(ctypes.c_double)]
lib.orc_snark_verify.argtypes = [_vp, _vp, _sz, _vp, ctypes.c_char_p, _sz, _vp, _sz]


def _p(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint8).reshape(-1, 32)


def set_threads(n):
    lib.orc_set_threads(int(n))


class OInstance:
    def __init__(self, num_cons, num_vars, num_inputs, A, B, C):
        self._keep = [np.ascontiguousarray(m) for m in (A, B, C)]
        h = _vp()
        rc = lib.orc_instance_new(num_cons, num_vars, num_inputs, _p(self._keep[0]), self._keep[0].size, _p(self._keep[1]), self._keep[1].size,
                                  _p(self._keep[2]), self._keep[2].size, ctypes.byref(h))
        if rc:
            raise ValueError(f"orc_instance_new rc={rc}")
        self.h = h
        # struct orc_instance starts with three size_t: padded cons, padded vars, inputs
        dims = (ctypes.c_size_t * 3).from_address(h.value)
        self.num_cons, self.num_vars, self.num_inputs = dims[0], dims[1], dims[2]

    def is_sat(self, vars32, inputs32):
        v, i = _c(vars32), _c(inputs32); sat = ctypes.c_int()
        rc = lib.orc_instance_is_sat(self.h, _p(v), v.shape[0], _p(i), i.shape[0], ctypes.byref(sat))
        if rc:
            raise ValueError(f"rc={rc}")
        return bool(sat.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_instance_free(self.h); self.h = None


class OGens:
    def __init__(self, num_cons, num_vars, num_inputs):
        self.h = _vp(lib.orc_gens_new(num_cons, num_vars, num_inputs))
        # struct orc_gens starts with orc_mcgens pc_n = {size_t n; ge_t *G; ge_t h}
        self.R = ctypes.c_size_t.from_address(self.h.value).value

    def points(self):
        out = np.zeros((self.R + 2, 32), dtype=np.uint8)
        lib.orc_gens_points(self.h, _p(out))
        return out

    @property
    def pc_n(self):
        return _vp(self.h.value)   # &gens->pc_n is the first member

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_gens_free(self.h); self.h = None


class OSnarkGens:
    """lib.rs SNARKGens::new(num_cons, num_vars, num_inputs, num_nz_entries)"""

    def __init__(self, num_cons, num_vars, num_inputs, num_nz_entries):
        self.h = _vp(lib.orc_snark_gens_new(num_cons, num_vars, num_inputs, num_nz_entries))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_snark_gens_free(self.h); self.h = None


class OSnarkComm:
    """SNARK::encode: computation commitment (with its decommitment when made by encode; commitment only when parsed from bytes)"""

    def __init__(self, h):
        if not h:
            raise ValueError("orc_snark_encode / parse failed")
        self.h = _vp(h)

    @classmethod
    def encode(cls, inst, gens):
        return cls(lib.orc_snark_encode(inst.h, gens.h))

    @classmethod
    def parse(cls, data):
        buf = np.frombuffer(data, dtype=np.uint8)
        return cls(lib.orc_snark_comm_parse(_p(buf), buf.size))

    @property
    def bytes(self):
        p, n = _vp(), _sz()
        lib.orc_snark_comm_bytes(self.h, ctypes.byref(p), ctypes.byref(n))
        data = ctypes.string_at(p, n.value); lib.orc_buf_free(p); return data

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_snark_comm_free(self.h); self.h = None


def snark_prove(inst, comm, vars32, inputs32, gens, label=b"snark_example", seed=b"\x2a" * 32):
    v, i = _c(vars32), _c(inputs32)
    p, n, ms = _vp(), _sz(), (ctypes.c_double * 9)()
    rc = lib.orc_snark_prove(inst.h, comm.h, _p(v), v.shape[0], _p(i), i.shape[0], gens.h, label, len(label), seed, ctypes.byref(p), ctypes.byref(n), ms)
    if rc:
        raise ValueError(f"orc_snark_prove rc={rc}")
    data = ctypes.string_at(p, n.value)
    lib.orc_buf_free(p)
    return data, list(ms)


def snark_verify(comm, inputs32, gens, proof, label=b"snark_example"):
    i = _c(inputs32); buf = np.frombuffer(proof, dtype=np.uint8)
    return lib.orc_snark_verify(comm.h, _p(i), i.shape[0], gens.h, label, len(label), _p(buf), buf.size)


def nizk_prove(inst, vars32, inputs32, gens, label=b"nizk_example", seed=b"\x2a" * 32):
    v, i = _c(vars32), _c(inputs32)
    p, n, ms = _vp(), _sz(), (ctypes.c_double * 7)()
    rc = lib.orc_nizk_prove(inst.h, _p(v), v.shape[0], _p(i), i.shape[0], gens.h, label, len(label), seed, ctypes.byref(p), ctypes.byref(n), ms)
    if rc:
        raise ValueError(f"orc_nizk_prove rc={rc}")
    data = ctypes.string_at(p, n.value)
    lib.orc_buf_free(p)
    return data, list(ms)


def nizk_verify(inst, inputs32, gens, proof, label=b"nizk_example"):
    i = _c(inputs32); buf = np.frombuffer(proof, dtype=np.uint8)
    return lib.orc_nizk_verify(inst.h, _p(i), i.shape[0], gens.h, label, len(label), _p(buf), buf.size)


# ---- kernel-level restatements; arrays are (n,32) uint8 Montgomery-form (same bytes as the device layout and as oracle fr_t)
def eq_evals(r):
    r = _c(r); out = np.zeros((1 << r.shape[0], 32), dtype=np.uint8)
    lib.orc_eq_evals(_p(r), r.shape[0], _p(out)); return out


def multiply_vec(inst, z):
    z = _c(z); o = [np.zeros((inst.num_cons, 32), dtype=np.uint8) for _ in range(3)]
    lib.orc_multiply_vec(inst.h, _p(z), _p(o[0]), _p(o[1]), _p(o[2])); return o


def eval_table_sparse(inst, eq_rx):
    e = _c(eq_rx); o = [np.zeros((2 * inst.num_vars, 32), dtype=np.uint8) for _ in range(3)]
    lib.orc_eval_table_sparse(inst.h, _p(e), _p(o[0]), _p(o[1]), _p(o[2])); return o


def fold_top(Z, r):
    Z = _c(Z).copy(); r = _c(r); lib.orc_fold_top(_p(Z), Z.shape[0], _p(r)); return Z[: Z.shape[0] // 2]


def fold_bot(Z, r):
    Z = _c(Z).copy(); r = _c(r); lib.orc_fold_bot(_p(Z), Z.shape[0], _p(r)); return Z[: Z.shape[0] // 2]


def sc_cubic_evals(A, B, C, D):
    A, B, C, D = (_c(x) for x in (A, B, C, D)); e = np.zeros((3, 32), dtype=np.uint8)
    lib.orc_sc_cubic_evals(_p(A), _p(B), _p(C), _p(D), A.shape[0], _p(e)); return e


def sc_quad_evals(A, B):
    A, B = _c(A), _c(B); e = np.zeros((2, 32), dtype=np.uint8)
    lib.orc_sc_quad_evals(_p(A), _p(B), A.shape[0], _p(e)); return e


def commit_rows(gens, Z, L, R, blinds):
    Z, blinds = _c(Z), _c(blinds); out = np.zeros((L, 32), dtype=np.uint8)
    lib.orc_commit_rows(_p(Z), L, R, _p(blinds), gens.pc_n, _p(out)); return out


def poly_bound(Z, L, R, Lv):
    Z, Lv = _c(Z), _c(Lv); out = np.zeros((R, 32), dtype=np.uint8)
    lib.orc_poly_bound(_p(Z), L, R, _p(Lv), _p(out)); return out


def bullet_reduce(gens, a, b, blinds, us):
    """nizk/bullet.rs reduction with the challenges `us` given; blinds: (2 * rounds, 32) as (bL, bR) per round.
    Returns (LR (rounds, 2, 32), a', b', compressed folded generators)."""
    a, b, blinds, us = _c(a), _c(b), _c(blinds), _c(us); n, rounds = a.shape[0], us.shape[0]
    LR = np.zeros((rounds, 2, 32), dtype=np.uint8); m = n >> rounds
    ao, bo, G = (np.zeros((m, 32), dtype=np.uint8) for _ in range(3))
    lib.orc_bullet_reduce(gens.h, _p(a), _p(b), n, _p(blinds), _p(us), rounds, _p(LR), _p(ao), _p(bo), _p(G))
    return LR, ao, bo, G


def fr_from_ints(xs):
    out = np.zeros((len(xs), 32), dtype=np.uint8)
    for k, x in enumerate(xs):
        out[k] = np.frombuffer(((x % L_ORDER) * _R % L_ORDER).to_bytes(32, "little"), dtype=np.uint8)
    return out


def fr_to_ints(a):
    a = _c(a)
    return [int.from_bytes(a[k].tobytes(), "little") * _RINV % L_ORDER for k in range(a.shape[0])]


def rand_fr(rng, n):
    """n uniformly random Montgomery-form elements (rejection-free: 64 random bytes mod l)"""
    raw = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    return fr_from_ints([int.from_bytes(raw[k].tobytes(), "little") % L_ORDER for k in range(n)])
