"""otti_amd — MI355X-native Spartan NIZK proving path for Otti.

Python mirror of the libspartan interface the reference's callers use [RECALL upstream `src/lib.rs`:
Instance, VarsAssignment, InputsAssignment, NIZKGens, NIZK::{prove, verify}; the Spartan/ submodule is an empty
directory in /root/reference], bound over the C ABI of ``libottispartan.so`` (include/otti_spartan.h) with ctypes.
There is no CPU proving path: without the HIP library and a gfx950 device ``NIZK.prove`` raises.
"""
from .api import (  # noqa: F401
    ENTRY_DTYPE, SpartanError, R1CSError, ProofVerifyError, NoDeviceError,
    Instance, VarsAssignment, InputsAssignment, NIZKGens, NIZK, Witness, SNARKGens, ComputationCommitment, SNARK,
    synth_r1cs, synth_r1cs_compiler_like, zkif_load, zkif_write, device_count, host_selftest, host_microbench, host_tail_bench, lib, lib_path,
    fr_from_ints, fr_to_ints, kernels, kernels_dev, DeviceArray, lanes_pack, lanes_unpack, L_ORDER, stats_enable, stats_read, madd_peak, fr_mul_peak, armed_launches_on, KERNEL_CLASSES,
    shard_init, shard_finalize, shard_info, shard_allgather, shard_allreduce,
)
