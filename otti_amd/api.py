"""ctypes binding of libottispartan.so — same names, argument meaning and error behaviour as upstream libspartan's
`src/lib.rs` for the NIZK path [RECALL], which is what `spzk verify --nizk` [REF /root/reference/run.py:58,100] and
rust-circ `--action spartan` [REF /root/reference/run.py:147] call."""
import ctypes
import os
import numpy as np

L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493
_R = (1 << 256) % L_ORDER
_RINV = pow(_R, -1, L_ORDER)

ENTRY_DTYPE = np.dtype([("row", "<u8"), ("col", "<u8"), ("val", "u1", (32,))])   # otti_entry

_HERE = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_HERE, "libottispartan.so")


class SpartanError(Exception):
    def __init__(self, code, msg=""):
        super().__init__(f"{type(self).__name__}({code}): {msg}")
        self.code = code


class R1CSError(SpartanError):
    """upstream R1CSError: InvalidNumberOfInputs(-3) InvalidNumberOfVars(-4) InvalidScalar(-5) InvalidIndex(-6)"""


class ProofVerifyError(SpartanError):
    """upstream ProofVerifyError: InternalError(-10) DecompressionError(-11); -12 = malformed proof bytes"""


class NoDeviceError(SpartanError):
    """no gfx950 device or HIP failure — the proving path has no CPU fallback"""


def _load():
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C otti_amd/csrc). otti_amd has no pure-Python or CPU fallback.")
    return ctypes.CDLL(lib_path)


lib = _load()
_sz, _u64, _i32, _vp = ctypes.c_size_t, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p
_u8p = ctypes.POINTER(ctypes.c_uint8)


class _R1CS(ctypes.Structure):
    _fields_ = [("num_cons", _u64), ("num_vars", _u64), ("num_inputs", _u64),
                ("A", _vp), ("B", _vp), ("C", _vp), ("nA", _sz), ("nB", _sz), ("nC", _sz),
                ("vars32", _vp), ("nvars", _sz), ("inputs32", _vp), ("ninputs", _sz)]


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_sig("otti_last_error", _sz, ctypes.c_char_p, _sz)
_sig("otti_buf_free", None, _vp)
_sig("otti_device_count", _i32)
_sig("otti_host_selftest", _i32, ctypes.c_uint32)
_sig("otti_host_microbench", _i32, ctypes.POINTER(ctypes.c_double))
_sig("otti_host_tail_bench", _i32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double))
_sig("otti_instance_new", _i32, _u64, _u64, _u64, _vp, _sz, _vp, _sz, _vp, _sz, ctypes.POINTER(_vp))
_sig("otti_instance_free", None, _vp)
_sig("otti_instance_dims", _i32, _vp, ctypes.POINTER(_u64), ctypes.POINTER(_u64), ctypes.POINTER(_u64))
_sig("otti_instance_is_sat", _i32, _vp, _vp, _sz, _vp, _sz, ctypes.POINTER(_i32))
_sig("otti_gens_new", _i32, _u64, _u64, _u64, ctypes.POINTER(_vp))
_sig("otti_gens_free", None, _vp)
_sig("otti_gens_points", _i32, _vp, _vp, _sz)
_sig("otti_gens_table_info", _i32, _vp, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(_u64))
_sig("otti_nizk_prove", _i32, _vp, _vp, _sz, _vp, _sz, _vp, ctypes.c_char_p, _sz, _vp, ctypes.c_uint32,
     ctypes.POINTER(_vp), ctypes.POINTER(_sz), ctypes.POINTER(ctypes.c_double))
_sig("otti_witness_upload", _i32, _vp, _vp, _sz, _vp, _sz, ctypes.POINTER(_vp))
_sig("otti_witness_free", None, _vp)
_sig("otti_nizk_prove_resident", _i32, _vp, _vp, _vp, ctypes.c_char_p, _sz, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz),
     ctypes.POINTER(ctypes.c_double))
_sig("otti_shard_init", _i32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32)
_sig("otti_shard_finalize", _i32)
_sig("otti_shard_info", _i32, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32))
_sig("otti_shard_allgather", _i32, _vp, _sz, _vp)
_sig("otti_shard_allreduce", _i32, _vp, _sz)
_sig("otti_nizk_prove_sharded", _i32, _vp, _vp, _vp, ctypes.c_char_p, _sz, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz),
     ctypes.POINTER(ctypes.c_double))
_sig("otti_nizk_verify", _i32, _vp, _vp, _sz, _vp, ctypes.c_char_p, _sz, _vp, _sz)
_sig("otti_prepare_device", _i32, _vp, _vp)
_sig("otti_snark_gens_new", _i32, _u64, _u64, _u64, _u64, ctypes.POINTER(_vp))
_sig("otti_snark_gens_free", None, _vp)
_sig("otti_snark_encode", _i32, _vp, _vp, ctypes.POINTER(_vp))
_sig("otti_comp_comm_bytes", _i32, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz))
_sig("otti_comp_comm_from_bytes", _i32, _vp, _sz, ctypes.POINTER(_vp))
_sig("otti_comp_comm_free", None, _vp)
_sig("otti_snark_prove", _i32, _vp, _vp, _vp, _sz, _vp, _sz, _vp, ctypes.c_char_p, _sz, _vp, ctypes.c_uint32,
     ctypes.POINTER(_vp), ctypes.POINTER(_sz), ctypes.POINTER(ctypes.c_double))
_sig("otti_snark_prove_resident", _i32, _vp, _vp, _vp, _vp, ctypes.c_char_p, _sz, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz), ctypes.POINTER(ctypes.c_double))
_sig("otti_snark_prove_sharded", _i32, _vp, _vp, _vp, _vp, ctypes.c_char_p, _sz, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_sz), ctypes.POINTER(ctypes.c_double))
_sig("otti_snark_verify", _i32, _vp, _vp, _sz, _vp, ctypes.c_char_p, _sz, _vp, _sz)
_sig("otti_zkif_load", _i32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.POINTER(_R1CS)))
_sig("otti_zkif_write", _i32, ctypes.POINTER(_R1CS), ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p)
_sig("otti_r1cs_free", None, ctypes.POINTER(_R1CS))
_sig("otti_synth_r1cs", _i32, _u64, _u64, _u64, ctypes.POINTER(ctypes.POINTER(_R1CS)))
_sig("otti_synth_r1cs_compiler_like", _i32, _u64, _u64, _u64, ctypes.POINTER(ctypes.POINTER(_R1CS)))
_sig("otti_bench_madd_peak", _i32, ctypes.POINTER(ctypes.c_double))
_sig("otti_stats_enable", _i32, _i32)
_sig("otti_stats_select", _i32, ctypes.c_char_p)
_sig("otti_armed_launches_on", _i32, ctypes.POINTER(_i32))
_sig("otti_gens_release_device", _i32, _vp)
_sig("otti_gens_build_ms", _i32, _vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))
_sig("otti_bench_fr_mul_peak", _i32, ctypes.POINTER(ctypes.c_double))
_sig("otti_stats_read", _i32, ctypes.c_char_p, ctypes.POINTER(_u64), ctypes.POINTER(ctypes.c_double))
_sig("otti_lanes_pack", None, _vp, _sz, _vp)
_sig("otti_lanes_unpack", None, _vp, _sz, _vp)
_fp = ctypes.POINTER(ctypes.c_float)
_sig("otti_k_fr_op", _i32, _i32, _vp, _vp, _vp, _sz, _fp)
_sig("otti_k_fr_from_canonical", _i32, _vp, _vp, _sz)
_sig("otti_k_fr_to_canonical", _i32, _vp, _vp, _sz)
_sig("otti_k_multiply_vec", _i32, _vp, _vp, _vp, _vp, _vp, _fp)
_sig("otti_k_eval_table_sparse", _i32, _vp, _vp, _vp, _vp, _fp)
_sig("otti_k_eq_evals", _i32, _vp, _sz, _vp, _fp)
_sig("otti_k_fold_top", _i32, _vp, _sz, _vp, _vp, _fp)
_sig("otti_k_fold_bot", _i32, _vp, _sz, _vp, _vp, _fp)
_sig("otti_k_sc_cubic_round", _i32, _vp, _vp, _vp, _vp, _sz, _vp, _fp)
_sig("otti_k_sc_cubic_fold_round", _i32, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _fp)
_sig("otti_k_sc_quad_round", _i32, _vp, _vp, _sz, _vp, _fp)
_sig("otti_k_sc_quad_fold_round", _i32, _vp, _vp, _sz, _vp, _vp, _vp, _fp)
_sig("otti_k_armed_selftest", _i32, _vp, _vp, _sz, _vp, ctypes.c_uint32, _vp, _vp)
_sig("otti_k_msm_rows", _i32, _vp, _vp, _sz, _sz, _vp, _vp, _fp)
_sig("otti_k_row_sum", _i32, _vp, _sz, _vp, _vp)
_sig("otti_k_eq_pyramid", _i32, _vp, _sz, _vp)
_sig("otti_k_sc_cubic3_round", _i32, _vp, _vp, _vp, _sz, _vp, _vp, _fp)
_sig("otti_k_sc_cubic3_fold_round", _i32, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _fp)
_sig("otti_k_poly_bound", _i32, _vp, _sz, _sz, _vp, _vp, _fp)
_sig("otti_k_bullet_round", _i32, _vp, _sz, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _fp)
_sig("otti_k_bullet_last_fold", _i32, _sz, _vp, _vp, _vp, _vp, _vp)
_sig("otti_kd_multiply_vec", _i32, _vp, _vp, _vp, _vp, _vp, _vp)
_sig("otti_kd_eval_table_sparse", _i32, _vp, _vp, _vp, _vp, _vp)
_sig("otti_kd_eq_evals", _i32, _vp, _sz, _vp, _vp)
_sig("otti_kd_fold_top", _i32, _vp, _sz, _vp, _vp)
_sig("otti_kd_fold_bot", _i32, _vp, _vp, _sz, _vp, _vp)
_sig("otti_kd_sc_cubic_round", _i32, _vp, _vp, _vp, _vp, _sz, _vp, _vp)
_sig("otti_kd_sc_cubic_fold_round", _i32, _vp, _vp, _vp, _vp, _sz, _vp, _vp, _vp)
_sig("otti_kd_sc_quad_round", _i32, _vp, _vp, _sz, _vp, _vp)
_sig("otti_kd_sc_quad_fold_round", _i32, _vp, _vp, _sz, _vp, _vp, _vp)
_sig("otti_kd_msm_rows", _i32, _vp, _vp, _sz, _sz, _vp, _vp, _vp)
_sig("otti_dev_alloc", _i32, _sz, ctypes.POINTER(_vp))
_sig("otti_dev_free", _i32, _vp)
_sig("otti_dev_upload", _i32, _vp, _vp, _sz)
_sig("otti_dev_download", _i32, _vp, _vp, _sz)
_sig("otti_dev_stream_create", _i32, ctypes.POINTER(_vp))
_sig("otti_dev_stream_sync", _i32, _vp)
_sig("otti_dev_stream_destroy", _i32, _vp)


def _last_error():
    buf = ctypes.create_string_buffer(512)
    lib.otti_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def _check(rc):
    if rc == 0:
        return
    msg = _last_error()
    if rc in (-10, -11, -12):
        raise ProofVerifyError(rc, msg)
    if rc == -20:
        raise NoDeviceError(rc, msg)
    if -6 <= rc <= -1:
        raise R1CSError(rc, msg)
    raise SpartanError(rc, msg)


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


def _scalars(a, what):
    """(n,32) uint8 array of canonical little-endian scalars"""
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim == 1:
        a = a.reshape(-1, 32)
    if a.ndim != 2 or a.shape[1] != 32:
        raise ValueError(f"{what}: expected an (n, 32) uint8 array")
    return a


def device_count():
    return int(lib.otti_device_count())


def host_selftest(iterations=200):
    """host-side fast paths of the prover (five-limb field, fixed-base tables) against the generic code; raises on a mismatch"""
    _check(lib.otti_host_selftest(iterations))


HOST_OPS = ("fixed_base_mul", "compress", "keccak_f1600", "append_point+challenge", "fr_mul", "fr_inv", "helper_thread_handoff",
            "zk_round_begin", "zk_round_finish", "threads")


def host_microbench():
    """nanoseconds per host-side primitive on this machine (no GPU needed); the last entry is the thread count used"""
    out = (ctypes.c_double * 10)()
    _check(lib.otti_host_microbench(out))
    return dict(zip(HOST_OPS, out))


def host_tail_bench(np_, nd, T, threads=1, reps=200):
    """microseconds per layer of SNARK mode's host-played sum-check rounds: {'avx512ifma': .., 'scalar': ..} (0 = form not available)"""
    out = (ctypes.c_double * 2)()
    _check(lib.otti_host_tail_bench(np_, nd, T, threads, reps, out))
    return {"avx512ifma": out[0], "scalar": out[1]}


# ---------------------------------------------------------------------------------------------- libspartan mirror
class Instance:
    """Instance::new(num_cons, num_vars, num_inputs, &A, &B, &C) -> Result<Instance, R1CSError>"""

    def __init__(self, handle, entries):
        self._h = handle
        self._keep = entries

    @classmethod
    def new(cls, num_cons, num_vars, num_inputs, A, B, C):
        ents = [np.ascontiguousarray(m, dtype=ENTRY_DTYPE) for m in (A, B, C)]
        h = _vp()
        _check(lib.otti_instance_new(num_cons, num_vars, num_inputs, _ptr(ents[0]), ents[0].size, _ptr(ents[1]), ents[1].size,
                                     _ptr(ents[2]), ents[2].size, ctypes.byref(h)))
        return cls(h, ents)

    @property
    def dims(self):
        a, b, c = _u64(), _u64(), _u64()
        _check(lib.otti_instance_dims(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def is_sat(self, vars_, inputs):
        v, i = _scalars(vars_.assignment, "vars"), _scalars(inputs.assignment, "inputs")
        sat = _i32()
        _check(lib.otti_instance_is_sat(self._h, _ptr(v), v.shape[0], _ptr(i), i.shape[0], ctypes.byref(sat)))
        return bool(sat.value)

    def prepare_device(self, gens=None):
        _check(lib.otti_prepare_device(self._h, gens._h if gens is not None else None))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.otti_instance_free(self._h)
            self._h = None


def _validate_scalars(a):
    # Scalar::from_bytes: canonical iff < l  (upstream returns Err(R1CSError::InvalidScalar))
    if a.shape[0] == 0:
        return
    words = a.view("<u8").reshape(-1, 4)
    lw = [(L_ORDER >> (64 * k)) & (2 ** 64 - 1) for k in range(4)]
    bad = np.zeros(a.shape[0], dtype=bool)
    undecided = np.ones(a.shape[0], dtype=bool)
    for k in (3, 2, 1, 0):
        gt = undecided & (words[:, k] > np.uint64(lw[k]))
        lt = undecided & (words[:, k] < np.uint64(lw[k]))
        bad |= gt
        undecided &= ~(gt | lt)
    bad |= undecided   # equal to l
    if bad.any():
        raise R1CSError(-5, "InvalidScalar")


class VarsAssignment:
    """VarsAssignment::new(&[[u8; 32]]) -> Result<_, R1CSError::InvalidScalar>"""

    def __init__(self, assignment):
        self.assignment = assignment

    @classmethod
    def new(cls, assignment):
        a = _scalars(assignment, "assignment")
        _validate_scalars(a)
        return cls(a)


class InputsAssignment(VarsAssignment):
    """InputsAssignment::new(&[[u8; 32]])"""


class NIZKGens:
    """NIZKGens::new(num_cons, num_vars, num_inputs)"""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def new(cls, num_cons, num_vars, num_inputs):
        h = _vp()
        _check(lib.otti_gens_new(num_cons, num_vars, num_inputs, ctypes.byref(h)))
        return cls(h)

    @property
    def table_info(self):
        """(window bits, bytes) of the device-side fixed-base table; (0, 0) before it has been built"""
        c, b = ctypes.c_uint32(), _u64()
        _check(lib.otti_gens_table_info(self._h, ctypes.byref(c), ctypes.byref(b)))
        return c.value, b.value

    @property
    def build_ms(self):
        """(allocations, upload + kernels) of the last window-table build, in ms"""
        a, k = ctypes.c_double(), ctypes.c_double()
        _check(lib.otti_gens_build_ms(self._h, ctypes.byref(a), ctypes.byref(k)))
        return a.value, k.value

    def release_device(self):
        """free the device-side window table (rebuilt by the next prepare_device / proof)"""
        _check(lib.otti_gens_release_device(self._h))

    def points(self, count):
        out = np.zeros((count, 32), dtype=np.uint8)
        _check(lib.otti_gens_points(self._h, _ptr(out), count))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib.otti_gens_free(self._h)
            self._h = None


class Witness:
    """Assignment resident in HBM (z = vars || 1 || inputs || 0..): upload once, prove many times."""

    def __init__(self, inst, vars_, inputs):
        v, i = _scalars(vars_.assignment, "vars"), _scalars(inputs.assignment, "inputs")
        h = _vp()
        _check(lib.otti_witness_upload(inst._h, _ptr(v), v.shape[0], _ptr(i), i.shape[0], ctypes.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.otti_witness_free(self._h)
            self._h = None


STAGES = ("polycommit", "multiply_vec", "sc_phase_one", "eval_table_sparse", "sc_phase_two", "polyeval", "total")


class NIZK:
    """NIZK::prove(&inst, vars, &inputs, &gens, &mut Transcript::new(label)) / NIZK::verify(...)"""

    def __init__(self, proof_bytes, stage_ms=None):
        self.bytes = proof_bytes
        self.stage_ms = stage_ms

    @staticmethod
    def _take(ptr, n):
        data = ctypes.string_at(ptr, n.value)
        lib.otti_buf_free(ptr)
        return data

    @classmethod
    def prove(cls, inst, vars_, inputs, gens, transcript_label=b"nizk_example", seed=None):
        label = bytes(transcript_label)
        if isinstance(vars_, Witness):
            w = vars_
            p, n, ms = _vp(), _sz(), (ctypes.c_double * 8)()
            _check(lib.otti_nizk_prove_resident(inst._h, w._h, gens._h, label, len(label), _seed(seed), ctypes.byref(p), ctypes.byref(n), ms))
            return cls(cls._take(p, n), dict(zip(STAGES, ms)))
        v, i = _scalars(vars_.assignment, "vars"), _scalars(inputs.assignment, "inputs")
        p, n, ms = _vp(), _sz(), (ctypes.c_double * 8)()
        _check(lib.otti_nizk_prove(inst._h, _ptr(v), v.shape[0], _ptr(i), i.shape[0], gens._h, label, len(label), _seed(seed), 1,
                                   ctypes.byref(p), ctypes.byref(n), ms))
        return cls(cls._take(p, n), dict(zip(STAGES, ms)))

    @classmethod
    def prove_sharded(cls, inst, witness, gens, transcript_label=b"nizk_example", seed=None):
        """This rank's part of ONE proof spread over the GPUs of a node (collective over the ranks of ``shard_init``; every rank
        passes the same instance, resident witness, generators, label and 32-byte seed and gets the same proof bytes back)."""
        label = bytes(transcript_label)
        p, n, ms = _vp(), _sz(), (ctypes.c_double * 8)()
        _check(lib.otti_nizk_prove_sharded(inst._h, witness._h, gens._h, label, len(label), _seed(seed), ctypes.byref(p), ctypes.byref(n), ms))
        return cls(cls._take(p, n), dict(zip(STAGES, ms)))

    def verify(self, inst, inputs, gens, transcript_label=b"nizk_example"):
        """Ok(()) -> None; Err(ProofVerifyError) -> raises"""
        label = bytes(transcript_label)
        i = _scalars(inputs.assignment, "inputs")
        buf = np.frombuffer(self.bytes, dtype=np.uint8)
        _check(lib.otti_nizk_verify(inst._h, _ptr(i), i.shape[0], gens._h, label, len(label), _ptr(buf), buf.size))


# ---------------------------------------------------------------------------------------------- SNARK mode (lib.rs SNARKGens / SNARK)
class SNARKGens:
    """SNARKGens::new(num_cons, num_vars, num_inputs, num_nz_entries)"""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def new(cls, num_cons, num_vars, num_inputs, num_nz_entries):
        h = _vp()
        _check(lib.otti_snark_gens_new(num_cons, num_vars, num_inputs, num_nz_entries, ctypes.byref(h)))
        return cls(h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.otti_snark_gens_free(self._h)
            self._h = None


class ComputationCommitment:
    """SNARK::encode(&inst, &gens) -> (ComputationCommitment, ComputationDecommitment); `from_bytes` gives the verifier's copy"""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def encode(cls, inst, gens):
        h = _vp()
        _check(lib.otti_snark_encode(inst._h, gens._h, ctypes.byref(h)))
        return cls(h)

    @classmethod
    def from_bytes(cls, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        h = _vp()
        _check(lib.otti_comp_comm_from_bytes(_ptr(buf), buf.size, ctypes.byref(h)))
        return cls(h)

    @property
    def bytes(self):
        p, n = _vp(), _sz()
        _check(lib.otti_comp_comm_bytes(self._h, ctypes.byref(p), ctypes.byref(n)))
        return NIZK._take(p, n)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.otti_comp_comm_free(self._h)
            self._h = None


SNARK_STAGES = ("polycommit", "multiply_vec", "sc_phase_one", "eval_table_sparse", "sc_phase_two", "polyeval", "derefs_commit", "product_circuits", "hash_layer", "total")


class SNARK:
    """SNARK::prove(&inst, &comm, &decomm, vars, &inputs, &gens, &mut Transcript::new(label)) / SNARK::verify(&comm, &inputs, ..)"""

    def __init__(self, proof_bytes, stage_ms=None):
        self.bytes = proof_bytes
        self.stage_ms = stage_ms

    @classmethod
    def prove(cls, inst, comm, vars_, inputs, gens, transcript_label=b"snark_example", seed=None):
        label = bytes(transcript_label)
        p, n, ms = _vp(), _sz(), (ctypes.c_double * 10)()
        if isinstance(vars_, Witness):                       # assignment already resident in HBM
            _check(lib.otti_snark_prove_resident(inst._h, comm._h, vars_._h, gens._h, label, len(label), _seed(seed), ctypes.byref(p), ctypes.byref(n), ms))
            return cls(NIZK._take(p, n), dict(zip(SNARK_STAGES, ms)))
        v, i = _scalars(vars_.assignment, "vars"), _scalars(inputs.assignment, "inputs")
        _check(lib.otti_snark_prove(inst._h, comm._h, _ptr(v), v.shape[0], _ptr(i), i.shape[0], gens._h, label, len(label), _seed(seed), 1,
                                    ctypes.byref(p), ctypes.byref(n), ms))
        return cls(NIZK._take(p, n), dict(zip(SNARK_STAGES, ms)))

    @classmethod
    def prove_sharded(cls, inst, comm, witness, gens, transcript_label=b"snark_example", seed=None):
        """This rank's part of ONE SNARK::prove spread over the GPUs of a node (collective over the ranks of ``shard_init``)."""
        label = bytes(transcript_label)
        p, n, ms = _vp(), _sz(), (ctypes.c_double * 10)()
        _check(lib.otti_snark_prove_sharded(inst._h, comm._h, witness._h, gens._h, label, len(label), _seed(seed), ctypes.byref(p), ctypes.byref(n), ms))
        return cls(NIZK._take(p, n), dict(zip(SNARK_STAGES, ms)))

    def verify(self, comm, inputs, gens, transcript_label=b"snark_example"):
        label = bytes(transcript_label)
        i = _scalars(inputs.assignment, "inputs")
        buf = np.frombuffer(self.bytes, dtype=np.uint8)
        _check(lib.otti_snark_verify(comm._h, _ptr(i), i.shape[0], gens._h, label, len(label), _ptr(buf), buf.size))


def shard_init(segment_name, rank, world):
    """Join the node-local exchange of a sharded proof (collective; ``segment_name`` must be fresh and the same on every rank)."""
    _check(lib.otti_shard_init(str(segment_name).encode(), rank, world))


def shard_finalize():
    _check(lib.otti_shard_finalize())


def shard_info():
    """(rank, world, transport) of the sharded-proof exchange; transport is 'mailbox' or 'rccl' (OTTI_SHARD_TRANSPORT)"""
    r, w, t = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    _check(lib.otti_shard_info(ctypes.byref(r), ctypes.byref(w), ctypes.byref(t)))
    return r.value, w.value, ("mailbox", "rccl")[t.value]


def shard_allgather(mine, world):
    """Exchange primitive: every rank's ``mine`` (bytes of equal length), concatenated in rank order."""
    mine = bytes(mine)
    out = ctypes.create_string_buffer(len(mine) * world)
    _check(lib.otti_shard_allgather(ctypes.cast(ctypes.c_char_p(mine), _vp), len(mine), ctypes.cast(out, _vp)))
    return out.raw


def shard_allreduce(scalars32):
    """Exchange primitive: element-wise sum over ranks of canonical GF(l) scalars, shape (n, 32) uint8."""
    a = np.ascontiguousarray(scalars32, dtype=np.uint8).reshape(-1, 32).copy()
    _check(lib.otti_shard_allreduce(_ptr(a), a.shape[0]))
    return a


def _seed(seed):
    if seed is None:
        return None
    seed = bytes(seed)
    if len(seed) != 32:
        raise ValueError("seed must be 32 bytes")
    return ctypes.cast(ctypes.create_string_buffer(seed, 32), _vp)


# ---------------------------------------------------------------------------------------------- ingest / synthetic
def _r1cs_to_py(rp):
    r = rp.contents

    def ents(p, n):
        if not n:
            return np.zeros(0, dtype=ENTRY_DTYPE)
        a = np.empty(n, dtype=ENTRY_DTYPE)                     # one memcpy (np.ctypeslib.as_array builds a ctypes array type per shape: seconds at 2^22)
        ctypes.memmove(a.ctypes.data, ctypes.cast(p, _vp), n * ENTRY_DTYPE.itemsize)
        return a

    def bytes32(p, n):
        if not n:
            return np.zeros((0, 32), dtype=np.uint8)
        a = np.empty((n, 32), dtype=np.uint8)
        ctypes.memmove(a.ctypes.data, ctypes.cast(p, _vp), n * 32)
        return a

    out = dict(num_cons=r.num_cons, num_vars=r.num_vars, num_inputs=r.num_inputs, A=ents(r.A, r.nA), B=ents(r.B, r.nB), C=ents(r.C, r.nC),
               vars=bytes32(r.vars32, r.nvars), inputs=bytes32(r.inputs32, r.ninputs))
    lib.otti_r1cs_free(rp)
    return out


def synth_r1cs(n, num_inputs=10, seed=1):
    """Synthetic satisfiable R1CS of SURVEY.md 8(d): num_cons = num_vars = n, one non-zero per row per matrix."""
    rp = ctypes.POINTER(_R1CS)()
    _check(lib.otti_synth_r1cs(n, num_inputs, seed, ctypes.byref(rp)))
    return _r1cs_to_py(rp)


def synth_r1cs_compiler_like(n, num_inputs=10, seed=1):
    """Second distribution of SURVEY.md 8(d): compiler-like R1CS (small witness values, ragged rows, heavy constant column)."""
    rp = ctypes.POINTER(_R1CS)()
    _check(lib.otti_synth_r1cs_compiler_like(n, num_inputs, seed, ctypes.byref(rp)))
    return _r1cs_to_py(rp)


def zkif_load(circuit, inputs=None, witness=None):
    rp = ctypes.POINTER(_R1CS)()
    enc = lambda s: None if s is None else os.fsencode(s)
    _check(lib.otti_zkif_load(enc(circuit), enc(inputs), enc(witness), ctypes.byref(rp)))
    return _r1cs_to_py(rp)


def zkif_write(r, circuit, inputs, witness):
    A, B, C = (np.ascontiguousarray(r[k], dtype=ENTRY_DTYPE) for k in "ABC")
    v, i = _scalars(r["vars"], "vars"), _scalars(r["inputs"], "inputs")
    s = _R1CS(r["num_cons"], r["num_vars"], r["num_inputs"], _ptr(A), _ptr(B), _ptr(C), A.size, B.size, C.size, _ptr(v), v.shape[0], _ptr(i), i.shape[0])
    _check(lib.otti_zkif_write(ctypes.byref(s), os.fsencode(circuit), os.fsencode(inputs), os.fsencode(witness)))


# ---------------------------------------------------------------------------------------------- field element helpers (tests/bench)
def fr_from_ints(xs):
    """python ints -> (n,32) uint8 Montgomery-form elements (the in-HBM layout)"""
    out = np.zeros((len(xs), 32), dtype=np.uint8)
    for k, x in enumerate(xs):
        out[k] = np.frombuffer(((x % L_ORDER) * _R % L_ORDER).to_bytes(32, "little"), dtype=np.uint8)
    return out


def fr_to_ints(a):
    a = _scalars(a, "fr")
    return [int.from_bytes(a[k].tobytes(), "little") * _RINV % L_ORDER for k in range(a.shape[0])]


KERNEL_CLASSES = ("msm_rows", "msm_small", "msm_finish", "sc_cubic", "sc_quad", "spmv", "eq", "reduce", "poly_bound", "bullet", "other",
                  "pc_round", "prod_layer", "hash_layer", "gather", "dot_many", "decode", "msm_var")


def stats_enable(on=True, only=None):
    _check(lib.otti_stats_enable(1 if on else 0))
    if on and only is not None:
        _check(lib.otti_stats_select(only.encode()))


def fr_mul_peak():
    """whole-chip Montgomery products in GF(l) per second (the streaming kernels' second roof), measured now"""
    v = ctypes.c_double()
    _check(lib.otti_bench_fr_mul_peak(ctypes.byref(v)))
    return v.value


def armed_launches_on():
    """whether the calling thread's next proof uses armed launches (see otti_armed_launches_on)"""
    v = _i32()
    _check(lib.otti_armed_launches_on(ctypes.byref(v)))
    return bool(v.value)


def madd_peak():
    """whole-chip mixed point additions per second (the MSM's ALU roof), measured now"""
    v = ctypes.c_double()
    _check(lib.otti_bench_madd_peak(ctypes.byref(v)))
    return v.value


def stats_read():
    """{class: (launch count, total ms)} measured with HIP events on the library's stream"""
    out = {}
    for k in KERNEL_CLASSES:
        n, ms = _u64(), ctypes.c_double()
        _check(lib.otti_stats_read(k.encode(), ctypes.byref(n), ctypes.byref(ms)))
        out[k] = (n.value, ms.value)
    return out


def lanes_pack(fr_mont):
    a = _scalars(fr_mont, "fr")
    out = np.zeros(a.shape[0] * 8, dtype=np.uint64)
    lib.otti_lanes_pack(_ptr(a), a.shape[0], _ptr(out))
    return out


def lanes_unpack(lanes):
    lanes = np.ascontiguousarray(lanes, dtype=np.uint64)
    n = lanes.size // 8
    out = np.zeros((n, 32), dtype=np.uint8)
    lib.otti_lanes_unpack(_ptr(lanes), n, _ptr(out))
    return out


class kernels:
    """Kernel-level entry points (otti_k_*): host (n,32) uint8 Montgomery arrays in, arrays out, plus kernel milliseconds."""

    @staticmethod
    def _ms():
        return ctypes.c_float(0)

    @staticmethod
    def fr_op(op, a, b):
        a, b = _scalars(a, "a"), _scalars(b, "b")
        out, ms = np.zeros_like(a), ctypes.c_float(0)
        _check(lib.otti_k_fr_op({"mul": 0, "add": 1, "sub": 2}[op], _ptr(a), _ptr(b), _ptr(out), a.shape[0], ctypes.byref(ms)))
        return out, ms.value

    @staticmethod
    def from_canonical(a):
        a = _scalars(a, "a"); out = np.zeros_like(a)
        _check(lib.otti_k_fr_from_canonical(_ptr(a), _ptr(out), a.shape[0])); return out

    @staticmethod
    def to_canonical(a):
        a = _scalars(a, "a"); out = np.zeros_like(a)
        _check(lib.otti_k_fr_to_canonical(_ptr(a), _ptr(out), a.shape[0])); return out

    @staticmethod
    def multiply_vec(inst, z):
        z = _scalars(z, "z"); nc, nv, _ = inst.dims
        assert z.shape[0] == 2 * nv
        o = [np.zeros((nc, 32), dtype=np.uint8) for _ in range(3)]; ms = ctypes.c_float(0)
        _check(lib.otti_k_multiply_vec(inst._h, _ptr(z), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), ctypes.byref(ms)))
        return o[0], o[1], o[2], ms.value

    @staticmethod
    def eval_table_sparse(inst, eq_rx, rABC):
        eq_rx, rABC = _scalars(eq_rx, "eq_rx"), _scalars(rABC, "rABC"); nc, nv, _ = inst.dims
        assert eq_rx.shape[0] == nc and rABC.shape[0] == 3
        out = np.zeros((2 * nv, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_eval_table_sparse(inst._h, _ptr(eq_rx), _ptr(rABC), _ptr(out), ctypes.byref(ms)))
        return out, ms.value

    @staticmethod
    def eq_evals(r):
        r = _scalars(r, "r"); ell = r.shape[0]
        out = np.zeros((1 << ell, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_eq_evals(_ptr(r), ell, _ptr(out), ctypes.byref(ms)))
        return out, ms.value

    @staticmethod
    def fold_top(Z, r):
        Z, r = _scalars(Z, "Z"), _scalars(r, "r"); out = np.zeros((Z.shape[0] // 2, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_fold_top(_ptr(Z), Z.shape[0], _ptr(r), _ptr(out), ctypes.byref(ms)))
        return out, ms.value

    @staticmethod
    def fold_bot(Z, r):
        Z, r = _scalars(Z, "Z"), _scalars(r, "r"); out = np.zeros((Z.shape[0] // 2, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_fold_bot(_ptr(Z), Z.shape[0], _ptr(r), _ptr(out), ctypes.byref(ms)))
        return out, ms.value

    @staticmethod
    def sc_cubic_round(A, B, C, D):
        A, B, C, D = (_scalars(x, "t") for x in (A, B, C, D)); e = np.zeros((3, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_sc_cubic_round(_ptr(A), _ptr(B), _ptr(C), _ptr(D), A.shape[0], _ptr(e), ctypes.byref(ms)))
        return e, ms.value

    @staticmethod
    def sc_cubic_fold_round(A, B, C, D, r):
        A, B, C, D, r = (_scalars(x, "t") for x in (A, B, C, D, r)); n = A.shape[0]
        out = np.zeros((4, n // 2, 32), dtype=np.uint8); e = np.zeros((3, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_sc_cubic_fold_round(_ptr(A), _ptr(B), _ptr(C), _ptr(D), n, _ptr(r), _ptr(out), _ptr(e), ctypes.byref(ms)))
        return out, e, ms.value

    @staticmethod
    def sc_quad_round(A, B):
        A, B = _scalars(A, "A"), _scalars(B, "B"); e = np.zeros((2, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_sc_quad_round(_ptr(A), _ptr(B), A.shape[0], _ptr(e), ctypes.byref(ms)))
        return e, ms.value

    @staticmethod
    def sc_quad_fold_round(A, B, r):
        A, B, r = (_scalars(x, "t") for x in (A, B, r)); n = A.shape[0]
        out = np.zeros((2, n // 2, 32), dtype=np.uint8); e = np.zeros((2, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_sc_quad_fold_round(_ptr(A), _ptr(B), n, _ptr(r), _ptr(out), _ptr(e), ctypes.byref(ms)))
        return out, e, ms.value

    @staticmethod
    def armed_selftest(A, B, r, hold_us=200):
        """plain / armed + released / armed + aborted runs of the quadratic fold round; returns the plain run's (folded tables, sums)"""
        A, B, r = (_scalars(x, "t") for x in (A, B, r)); n = A.shape[0]
        out = np.zeros((2, n // 2, 32), dtype=np.uint8); e = np.zeros((2, 32), dtype=np.uint8)
        _check(lib.otti_k_armed_selftest(_ptr(A), _ptr(B), n, _ptr(r), hold_us, _ptr(out), _ptr(e)))
        return out, e

    @staticmethod
    def row_sum(compressed, scalars_mont):
        """the verifier's variable-base sum on the device: compress(sum_i s[i] * decompress(C[i]))"""
        C = np.ascontiguousarray(compressed, dtype=np.uint8).reshape(-1, 32); s = _scalars(scalars_mont, "s")
        assert C.shape[0] == s.shape[0]
        out = np.zeros(32, dtype=np.uint8)
        _check(lib.otti_k_row_sum(_ptr(C), C.shape[0], _ptr(s), _ptr(out)))
        return out

    @staticmethod
    def msm_rows(gens, Z, L, R, blinds):
        Z, blinds = _scalars(Z, "Z"), _scalars(blinds, "blinds")
        assert Z.shape[0] == L * R and blinds.shape[0] == L
        out = np.zeros((L, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_msm_rows(gens._h, _ptr(Z), L, R, _ptr(blinds), _ptr(out), ctypes.byref(ms)))
        return out, ms.value

    # ---- the kernels the prover itself launches (phase one without the eq table, evaluation proof, bullet reduction)
    @staticmethod
    def eq_pyramid(r):
        r = _scalars(r, "r"); n = r.shape[0]
        out = np.zeros(((2 << n) - 1, 32), dtype=np.uint8)
        _check(lib.otti_k_eq_pyramid(_ptr(r), n, _ptr(out)))
        return out

    @staticmethod
    def sc_cubic3_round(B, C, D, tau):
        B, C, D, tau = (_scalars(x, "t") for x in (B, C, D, tau)); e = np.zeros((3, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        assert (1 << (tau.shape[0] + 1)) == B.shape[0]
        _check(lib.otti_k_sc_cubic3_round(_ptr(B), _ptr(C), _ptr(D), B.shape[0], _ptr(tau), _ptr(e), ctypes.byref(ms)))
        return e, ms.value

    @staticmethod
    def sc_cubic3_fold_round(B, C, D, r, tau):
        B, C, D, r, tau = (_scalars(x, "t") for x in (B, C, D, r, tau)); n = B.shape[0]
        assert (1 << (tau.shape[0] + 2)) == n
        out = np.zeros((3, n // 2, 32), dtype=np.uint8); e = np.zeros((3, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_sc_cubic3_fold_round(_ptr(B), _ptr(C), _ptr(D), n, _ptr(r), _ptr(tau), _ptr(out), _ptr(e), ctypes.byref(ms)))
        return out, e, ms.value

    @staticmethod
    def poly_bound(Z, L, R, Lv):
        Z, Lv = _scalars(Z, "Z"), _scalars(Lv, "Lv"); assert Z.shape[0] == L * R and Lv.shape[0] == L
        out = np.zeros((R, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_poly_bound(_ptr(Z), L, R, _ptr(Lv), _ptr(out), ctypes.byref(ms)))
        return out, ms.value

    @staticmethod
    def bullet_round(gens, n_cur, a, b, s, blinds2, u=None, uinv=None):
        """one bullet-reduction round on the original generators; (u, uinv) given => the previous challenge is applied first"""
        a, b, s, blinds2 = (_scalars(x, "t") for x in (a, b, s, blinds2)); R = s.shape[0]
        fold = u is not None
        uu = _scalars(u, "u") if fold else None; ui = _scalars(uinv, "uinv") if fold else None
        ao, bo, so = (np.zeros((k, 32), dtype=np.uint8) for k in (n_cur, n_cur, R)); LR = np.zeros((2, 32), dtype=np.uint8); ms = ctypes.c_float(0)
        _check(lib.otti_k_bullet_round(gens._h, n_cur, 1 if fold else 0, _ptr(uu), _ptr(ui), _ptr(a), _ptr(b), _ptr(s), _ptr(blinds2), _ptr(ao), _ptr(bo), _ptr(so),
                                       _ptr(LR), ctypes.byref(ms)))
        return LR, ao, bo, so, ms.value

    @staticmethod
    def bullet_last_fold(a2, b2, s, u, uinv):
        a2, b2, s, u, uinv = (_scalars(x, "t").copy() for x in (a2, b2, s, u, uinv))
        _check(lib.otti_k_bullet_last_fold(s.shape[0], _ptr(u), _ptr(uinv), _ptr(a2), _ptr(b2), _ptr(s)))
        return a2[:1], b2[:1], s


class DeviceArray:
    """(n, 32) field elements in HBM (otti_dev_alloc): operand of the device-pointer kernel entry points (otti_kd_*)"""

    def __init__(self, n, nbytes_each=32):
        self.n, self.each = n, nbytes_each
        p = _vp()
        _check(lib.otti_dev_alloc(max(1, n * nbytes_each), ctypes.byref(p)))
        self.ptr = p

    @classmethod
    def from_host(cls, a):
        a = _scalars(a, "a"); d = cls(a.shape[0])
        _check(lib.otti_dev_upload(d.ptr, _ptr(a), a.nbytes))
        return d

    def to_host(self, n=None):
        n = self.n if n is None else n
        out = np.zeros((n, self.each), dtype=np.uint8)
        _check(lib.otti_dev_download(_ptr(out), self.ptr, out.nbytes))
        return out

    def __del__(self):
        if getattr(self, "ptr", None):
            lib.otti_dev_free(self.ptr); self.ptr = None


class kernels_dev:
    """otti_kd_*: the same kernels on device pointers and a caller's stream (None = the library's stream of this thread)"""

    @staticmethod
    def stream_create():
        s = _vp(); _check(lib.otti_dev_stream_create(ctypes.byref(s))); return s

    @staticmethod
    def stream_sync(s):
        _check(lib.otti_dev_stream_sync(s))

    @staticmethod
    def stream_destroy(s):
        _check(lib.otti_dev_stream_destroy(s))

    @staticmethod
    def multiply_vec(inst, z, Az, Bz, Cz, stream=None):
        _check(lib.otti_kd_multiply_vec(inst._h, z.ptr, Az.ptr, Bz.ptr, Cz.ptr, stream))

    @staticmethod
    def eval_table_sparse(inst, eq_rx, rABC, out, stream=None):
        rABC = _scalars(rABC, "rABC"); _check(lib.otti_kd_eval_table_sparse(inst._h, eq_rx.ptr, _ptr(rABC), out.ptr, stream))

    @staticmethod
    def eq_evals(r, out, stream=None):
        r = _scalars(r, "r"); _check(lib.otti_kd_eq_evals(_ptr(r), r.shape[0], out.ptr, stream))

    @staticmethod
    def fold_top(Z, length, r, stream=None):
        r = _scalars(r, "r"); _check(lib.otti_kd_fold_top(Z.ptr, length, _ptr(r), stream))

    @staticmethod
    def fold_bot(Z, out, length, r, stream=None):
        r = _scalars(r, "r"); _check(lib.otti_kd_fold_bot(Z.ptr, out.ptr, length, _ptr(r), stream))

    @staticmethod
    def sc_cubic_round(A, B, C, D, length, stream=None):
        e = np.zeros((3, 32), dtype=np.uint8); _check(lib.otti_kd_sc_cubic_round(A.ptr, B.ptr, C.ptr, D.ptr, length, _ptr(e), stream)); return e

    @staticmethod
    def sc_cubic_fold_round(A, B, C, D, length, r, stream=None):
        r = _scalars(r, "r"); e = np.zeros((3, 32), dtype=np.uint8)
        _check(lib.otti_kd_sc_cubic_fold_round(A.ptr, B.ptr, C.ptr, D.ptr, length, _ptr(r), _ptr(e), stream)); return e

    @staticmethod
    def sc_quad_round(A, B, length, stream=None):
        e = np.zeros((2, 32), dtype=np.uint8); _check(lib.otti_kd_sc_quad_round(A.ptr, B.ptr, length, _ptr(e), stream)); return e

    @staticmethod
    def sc_quad_fold_round(A, B, length, r, stream=None):
        r = _scalars(r, "r"); e = np.zeros((2, 32), dtype=np.uint8)
        _check(lib.otti_kd_sc_quad_fold_round(A.ptr, B.ptr, length, _ptr(r), _ptr(e), stream)); return e

    @staticmethod
    def msm_rows(gens, Z, L, R, blinds, out32, stream=None):
        _check(lib.otti_kd_msm_rows(gens._h, Z.ptr, L, R, blinds.ptr, out32.ptr, stream))
