#!/usr/bin/env python3
"""Benchmark of the MI355X Spartan proving path (NIZK mode = the headline; SNARK mode, sweep and transports as extras).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Metric (BASELINE.json): R1CS constraints proved per second = N_constraints / wall time of NIZK::prove, on the
synthetic satisfiable 2^20-constraint R1CS; instance, generators (with their window table) and witness resident in HBM before timing.
A "step" is ONE NIZK::prove of that workload, one proof at a time, exactly as a library caller runs it (armed launches on: the HIP
events of the timed region bracket only the dominant kernel class, which has no armed launches); W untimed warm-up steps, then exactly
K timed steps bracketed by barrier + device synchronisation; time = max over ranks; rank 0 prints ONE JSON line.

  N = 1   value = 2^20 * K / elapsed  (single-proof latency, what SURVEY 8(d) defines)
  N > 1   ALL ranks prove the SAME proof together (commitment rows, sparse matrices and sum-check tables are sharded;
          the witness is replicated beforehand by a torch.distributed broadcast over RCCL/xGMI; per-round sums cross ranks through
          the node-local mailbox or, with OTTI_SHARD_TRANSPORT=rccl, an ncclUint64/ncclSum all-reduce of u64 lanes); value =
          2^20 * K / elapsed of that one sharded proof; "scaling": "strong".  --replicas makes the primary line independent proofs per
          GPU instead (weak scaling).

Extras in the same line (all measured in this run, none of them the headline):
  roofline      N = 1: the kernel class with the largest summed device time of one instrumented proof, over ALL classes; N > 1: the
                same class as at N = 1 (msm_rows, the witness commitment) with this rank's share of the scalars, so the lines of a
                scaling run are comparable.  Algorithmic bytes per launch / average launch duration (HIP events on the library's
                stream over the timed region) against the HBM peak; for the MSM classes additionally mixed point additions per
                second against the ALU roof measured in this run AND against the instruction roof (v_mad_u64_u32 rate / mads per addition)
  cpu_baseline  the CPU oracle (the reference Rust prover cannot be built here) on the same workload, host cores stated; on rank 0
                for every N
  transports    N > 1: the same sharded proof timed over the node-local mailbox and over the RCCL u64-lane all-reduce, in this run
  sweep         the other sizes of BASELINE.json's range (2^18, 2^22 = north_star's target size, 2^24), sharded like the headline
                when N > 1, each checked against the committed oracle digest; 2^22 with the CPU oracle beside it
  in_flight     throughput with B independent proofs in flight per GPU (B prover threads, each with its OWN instance, witness and
                random-tape seed; they share only the generator table); summed over the GPUs (= the replicas figure for N > 1)
  snark         SNARK mode on the same workload: SNARK::encode once (untimed: preprocessing of the circuit), then SNARK::prove = the
                headline's R1CSProof + R1CSEvalProof; its own roofline (dominant class over SNARK::prove's kernels); commitment and
                proof compared with the oracle digests committed in tests/golden/snark_proofs.json; CPU oracle on a bounded 2^16 sample
                (N > 1: one independent SNARK::prove per GPU between barriers — SNARK mode is not sharded — aggregate over the ranks, "scaling": "weak")
  spzk_e2e      the path run.py actually executes: `spzk verify --nizk` on a zkInterface triple of the workload, one process
Every timed proof is checked: identical bytes across steps (fixed seed), accepted by the verifier, equal to the oracle's committed
digest for the size; rank 0 also compares a 2^12 proof with the CPU oracle (checker only, outside the timed region) and, when the CPU
baseline runs the same workload, the full-size proof.
OTTI_BENCH_REHEARSE=1 puts every rank on GPU 0 with the gloo backend (how the sharded mode is rehearsed on a one-GPU box).
"""
import argparse
import gc
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md); ~6300 achievable
MAD_U64_LANE_OPS = 32.6e12      # v_mad_u64_u32 lane-operations per second, whole chip (tools/fmabench.hip, profiles/r1_microbench.txt)
MADS_PER_MADD = 700             # 7 multiplications in GF(2^255-19) x 100 limb products per mixed point addition (fp10.h)
F = 32                          # bytes per field element
KERNEL_NAMES = {"msm_rows": "k_msm_rows<0>", "msm_small": "k_msm_small", "sc_cubic": "k_sc_cubic3_fold_eval", "sc_quad": "k_sc_quad_fold_eval",
                "spmv": "k_spmv3_light / k_spmv3_quad (one lane or one quad per row, by the matrix's entries per row)", "msm_finish": "k_encode_points", "eq": "k_eq_expand",
                "poly_bound": "k_poly_bound_slab", "pc_round": "k_pc_round / k_pc_tail", "prod_layer": "k_prod_layer", "hash_layer": "k_hash_ops / k_hash_mem",
                "gather": "k_gather", "dot_many": "k_dot_many / k_sum3"}


def algorithmic_bytes(N, V, nnz):
    """compulsory HBM traffic of one proof, W = 80*nnz + 704*N + 736*V bytes."""
    return 80 * nnz + 704 * N + 736 * V


def class_bytes_per_proof(cls, N, V, nnz):
    """algorithmic bytes one NIZK proof moves in the launches of a kernel class (SURVEY 8(d)'s per-stage figures)"""
    ell = V.bit_length() - 1
    L, R = 1 << (ell // 2), 1 << (ell - ell // 2)
    nrx, nry, lgR = N.bit_length() - 1, (2 * V).bit_length() - 1, R.bit_length() - 1
    return {
        "msm_rows": F * V,                                                     # the witness commitment reads V scalars once
        # the one/two-row launches: 32 B per term summed — blinds, the tape-only points of every round (4 rows x 6 bases), Cx, delta,
        # the bullet rounds (2 rows x (R/2 + 2) terms each)
        "msm_small": F * (L + 24 * (nrx + nry) + 2 * (R + 1) + lgR * (R + 4)),
        "sc_cubic": 384 * N,                                                   # three tables (eq factored out), fold fused with the next round's sums
        "sc_quad": 512 * V,
        "spmv": 80 * nnz + 160 * V + 128 * N,
        "eq": F * N + 2 * F * (L + R),
        "poly_bound": F * V + 2 * F * R,
        "msm_finish": (128 + 32) * L,                                          # row sums in, compressed points out
        "bullet": 3 * F * R, "reduce": 4 * F, "other": 0,
    }.get(cls, 0)


def snark_class_bytes_per_proof(cls, N, V, nnz, nz, M):
    """the same for SNARK::prove = R1CSProof + R1CSEvalProof: nz = padded non-zeros per matrix, M = memory cells.
    pc_round: every table of a batch is read once by the layer's first launch; launch k >= 1 reads the table as launch k - 1 left it
    and writes the folded half (the eq table is never stored).  Product circuits: 12 over nz + 6 dot-product triples over nz / 2 at
    the input layer, 4 over M."""
    def batch(n_elems, circuits, dotp_tables):
        total, nl = 0, max(1, n_elems.bit_length() - 1)
        for nr in range(nl):                                               # layer with 2^nr elements per table, nr rounds
            tables = 2 * circuits + (dotp_tables if nr == nl - 1 else 0)
            h, rounds = 1 << nr, nr
            elems = h                                                      # first launch: every element once
            for k in range(1, rounds + 1):
                elems += (h >> (k - 1)) + (h >> k)                         # read what the previous launch left, write the folded half
            total += tables * elems * F
        return total
    base = class_bytes_per_proof(cls, N, V, nnz)
    return {
        "msm_rows": F * V + F * 8 * nz,                                    # witness commitment + the 8 nz dereferenced values
        "pc_round": batch(nz, 12, 18) + batch(M, 4, 0),
        "prod_layer": 3 * F * (12 * nz + 4 * M),                           # every layer reads two halves and writes one: 3 x the input vector in all
        "hash_layer": 6 * 5 * F * nz + 2 * 4 * F * M,                      # 6 launches over nz (3 reads, 2 writes), 2 over M (2 reads, 2 writes)
        "gather": 6 * nz * (4 + 2 * F),                                    # index, gathered element, stored element
        "dot_many": F * (22 * nz + 3 * M + 18 * nz // 2),                  # 21 polynomials + eq over nz, 2 + eq over M, the dot-product triples
        "eq": base + 2 * F * M + F * (nz + M),                             # eq(rx), eq(ry) over M; eq(rand_ops), eq(rand_mem)
    }.get(cls, base)


def usable_cores():
    """cores this process may actually burn: affinity mask, capped by the cgroup CPU quota (a GPU box grants a share of the host)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(txt[0]) // int(txt[1])))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except Exception:
            pass
    return n


def bind_rank_to_l3_groups(local_rank, local_world):
    """N > 1: every rank process on L3 groups of its own (the launcher's job where there is one — mpirun --bind-to l3cache; torch.distributed.run
    has no such option).  The library keeps a prover thread's spinning helper threads on cores that share the L3 with it (pool.h); two
    ranks whose prover threads the scheduler happened to put into one group would put their helpers on the same cores.  Returns the
    CPUs bound to, or None when the topology cannot be read or the mask cannot be set."""
    try:
        allowed = sorted(os.sched_getaffinity(0)); groups, seen = [], set()
        for cpu in allowed:
            if cpu in seen:
                continue
            txt = open("/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list" % cpu).read().strip()
            members = set()
            for part in txt.split(","):
                lo, _, hi = part.partition("-")
                members.update(range(int(lo), int(hi or lo) + 1))
            members &= set(allowed); seen |= members
            if members:
                groups.append(sorted(members))
        per = len(groups) // local_world
        if per < 1:
            return None
        mine = sorted(c for g in groups[local_rank * per:(local_rank + 1) * per] for c in g)
        os.sched_setaffinity(0, mine)
        return mine
    except Exception:                                          # noqa: BLE001 — no topology files, a cpuset that refuses: leave the process where it is
        return None


def golden_digest(kind, n):
    """committed oracle digests (tests/golden/): kind 'nizk' -> proofs.json, 'snark' -> snark_proofs.json; None when the size is not there"""
    try:
        for e in json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json" if kind == "nizk" else "snark_proofs.json"))):
            if e["n"] == n and e["num_inputs"] == 10 and e["instance_seed"] == 1:
                return e
    except Exception:
        pass
    return None


def main():
    t_start = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2-constraints", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the one-shot `spzk verify --nizk` figure")
    ap.add_argument("--no-snark", action="store_true", help="skip the SNARK-mode figure (SNARK::encode once, then SNARK::prove)")
    ap.add_argument("--no-transports", action="store_true", help="N > 1: skip timing the sharded proof over the other transport")
    ap.add_argument("--sweep", default="18,22,24", help="log2 sizes of the `sweep` extra (comma separated; '' or --no-sweep = none)")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--sweep-budget-s", type=float, default=240.0, help="a sweep size is skipped (and says so) once the run has used this much wall time")
    ap.add_argument("--dist", choices=("uniform", "compiler"), default="uniform", help="synthetic instance distribution (SURVEY 8d); the metric is quoted on 'uniform'")
    ap.add_argument("--cpu-log2", type=int, default=None, help="size of the CPU-baseline sample (default: same workload)")
    ap.add_argument("--replicas", action="store_true", help="N > 1: primary line = one independent proof per GPU (weak scaling) instead of one proof sharded over all GPUs")
    ap.add_argument("--shard", action="store_true", help="(default for N > 1; kept for older command lines)")
    ap.add_argument("--in-flight", type=int, default=0, help="extra figure: proofs in flight per GPU (prover threads).  0 = what this rank's host cores feed "
                    "(2 host threads per prover, at most 6); -1 = skip")
    ap.add_argument("--concurrent", type=int, default=None, help="alias of --in-flight (older command lines)")
    args = ap.parse_args()
    if args.concurrent is not None:
        args.in_flight = args.concurrent if args.concurrent != 1 else -1

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = bool(os.environ.get("OTTI_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    os.environ.setdefault("OTTI_DEVICE", str(local_rank))
    # OTTI_FORCE_SHARD (with OTTI_FORCE_DIST): the sharded code path — witness broadcast, exchange, transports, sharded sweep — with a world of
    # ONE over the real RCCL backend: how the N > 1 path is exercised on a one-GPU box beyond the gloo rehearsal
    shard = (world > 1 or bool(os.environ.get("OTTI_FORCE_SHARD"))) and not args.replicas
    lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    node_cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None   # before any binding: what the CPU baseline of an N > 1 line may use
    node_cores = usable_cores()
    bound_cpus = bind_rank_to_l3_groups(local_rank, lws) if (lws > 1 and os.environ.get("OTTI_BENCH_BIND", "1") != "0") else None
    cores_here = max(1, usable_cores() if bound_cpus else usable_cores() // lws)   # host cores of this rank (a bound rank's mask is already its share)
    conc = 0 if args.in_flight < 0 else (args.in_flight if args.in_flight > 0 else max(1, min(6, cores_here // 2)))
    # host threads of the single-proof prover (itself + spinning helpers for the per-round sigma-protocol work)
    os.environ.setdefault("OTTI_HOST_THREADS", str(max(1, min(4, cores_here))))
    if conc > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")         # one hardware queue per prover stream (HIP's default is 4)
    # stdout carries ONE line, rank 0's JSON: native libraries print banners there (gloo's "Rank 0 is connected to ...", librccl's version
    # line), so file descriptor 1 points at stderr for the whole run and the line goes out through the descriptor saved here
    sys.stdout.flush()
    line_fd = os.dup(1); os.dup2(2, 1)
    dist = None
    if world > 1 or os.environ.get("OTTI_FORCE_DIST"):      # OTTI_FORCE_DIST: exercise the RCCL path on a one-GPU box
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    xdev = "cpu" if rehearse else "cuda"                     # where tensors handed to torch.distributed live
    import numpy as np
    import otti_amd as oa

    if oa.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible; the proving path has no CPU fallback")

    lg = args.log2_constraints
    n, ni, label, seed = 1 << lg, 10, b"nizk_example", b"\x2a" * 32
    gen = oa.synth_r1cs if args.dist == "uniform" else oa.synth_r1cs_compiler_like

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def same_on_every_rank(blob, what):
        if dist is None:
            return
        import torch
        dg = torch.frombuffer(bytearray(hashlib.sha256(blob).digest()), dtype=torch.uint8).to(xdev)
        every = [torch.zeros_like(dg) for _ in range(world)]
        dist.all_gather(every, dg)
        assert all(bool((e == dg).all()) for e in every), what

    shard_seq = [0]

    def shard_join(transport_name=None):
        """(re)join the node-local exchange of the sharded proofs; returns None, or the error text when the transport cannot be set up
        on EVERY rank (then nobody stays joined)"""
        if transport_name:
            os.environ["OTTI_SHARD_TRANSPORT"] = transport_name
        name = [("otti-bench-%d-%d-%d" % (os.getpid(), time.time_ns(), shard_seq[0])) if rank == 0 else None]
        shard_seq[0] += 1
        dist.broadcast_object_list(name, src=0)
        err = None
        try:
            oa.shard_init(name[0], rank, world)
        except Exception as e:                                   # noqa: BLE001 - e.g. RCCL refusing two ranks on one card (rehearsal)
            err = "%s: %s" % (type(e).__name__, e)
        errs = [None] * world
        dist.all_gather_object(errs, err)
        bad = [e for e in errs if e]
        if bad:
            if err is None:
                oa.shard_finalize()
            return bad[0]
        return None

    r = gen(n, ni, 1 if (shard or world == 1) else 1 + rank)   # --replicas: each rank its own instance
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    vars_, inputs = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    transport = None
    if shard:
        # the instance is public (every rank builds it); the WITNESS exists on rank 0 only and reaches the other GPUs over RCCL/xGMI
        import torch
        wbuf = torch.from_numpy(np.ascontiguousarray(r["vars"])).to(xdev) if rank == 0 else torch.zeros((r["vars"].shape[0], 32), dtype=torch.uint8, device=xdev)
        ibuf = torch.from_numpy(np.ascontiguousarray(r["inputs"])).to(xdev) if rank == 0 else torch.zeros((ni, 32), dtype=torch.uint8, device=xdev)
        dist.broadcast(wbuf, src=0); dist.broadcast(ibuf, src=0)
        vars_, inputs = oa.VarsAssignment.new(wbuf.cpu().numpy()), oa.InputsAssignment.new(ibuf.cpu().numpy())
        e = shard_join()
        if e:
            raise SystemExit("bench.py: cannot join the sharded-proof exchange: " + e)
        transport = oa.shard_info()[2]
    t0 = time.perf_counter()
    inst.prepare_device(gens)                              # CSR upload + generator window table: resident before timing
    t_prepare = time.perf_counter() - t0
    cbits, table_bytes = gens.table_info
    table_alloc_ms, table_kernels_ms = gens.build_ms
    t0 = time.perf_counter()
    wit = oa.Witness(inst, vars_, inputs)                  # witness resident in HBM before timing
    t_upload = time.perf_counter() - t0
    N, V, _ = inst.dims
    nnz = int(r["A"].size + r["B"].size + r["C"].size)

    def prove_once(i_=None, w_=None, g_=None):
        i_, w_, g_ = i_ or inst, w_ or wit, g_ or gens
        return oa.NIZK.prove_sharded(i_, w_, g_, label, seed) if shard else oa.NIZK.prove(i_, w_, None, g_, label, seed)

    proofs = [prove_once() for _ in range(args.warmup)]
    # one untimed, fully instrumented proof: per-class kernel time -> the dominant kernel class, chosen over ALL classes at N = 1; a
    # sharded run keeps the N = 1 choice (the witness commitment) so that the roofline of every line of a scaling run is the same kernel
    oa.stats_enable(True)
    proofs.append(prove_once())
    breakdown = oa.stats_read()
    dom = max(breakdown, key=lambda k: breakdown[k][1])
    if shard and breakdown.get("msm_rows", (0, 0))[0]:
        dom = "msm_rows"
    # timed region: HIP events only around the dominant class (two event records per launch would otherwise tax every round, and a
    # class with armed launches cannot be timed with them on); everything else runs exactly as a library caller runs it
    oa.stats_enable(True, only=dom)
    armed_on = oa.armed_launches_on()
    barrier()
    t0 = time.perf_counter()
    stage_acc = {}
    step_ms = []
    for _ in range(args.steps):
        t_step = time.perf_counter()
        p = prove_once()                                       # returns after the library's stream has been synchronised
        step_ms.append(1e3 * (time.perf_counter() - t_step))
        proofs.append(p)
        for k, v in p.stage_ms.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    stats = dict(oa.stats_read())
    oa.stats_enable(False)
    madd_peak = oa.madd_peak()                                 # the MSM's ALU roof, measured in this run (after the timed region)
    frmul_peak = oa.fr_mul_peak()                              # the streaming kernels' second roof: Montgomery products in GF(l) per second

    # correctness of what was timed
    digests = {hashlib.sha256(p.bytes).hexdigest() for p in proofs}
    assert len(digests) == 1, "proofs of the same inputs and seed differ between steps"
    gold = golden_digest("nizk", n) if (args.dist == "uniform" and (shard or world == 1)) else None
    digest_ok = None if gold is None else (gold["proof_sha256"] in digests)
    assert digest_ok is not False, "the timed proof differs from the oracle's committed digest for this size"
    t_verify_calls = []
    for _ in range(3):                                       # the first call allocates the verifier's device buffers (kept with the context afterwards)
        t0 = time.perf_counter()
        proofs[-1].verify(inst, inputs, gens, label)
        t_verify_calls.append(time.perf_counter() - t0)
    t_verify = min(t_verify_calls)
    elapsed = max_over_ranks(elapsed)
    if shard:
        same_on_every_rank(proofs[-1].bytes, "ranks of a sharded proof returned different bytes")
    final_proof = proofs[-1].bytes
    del proofs

    # ---- extra (N > 1): the same sharded proof over the other transport of the per-round sums, in this run
    transports = None
    if shard and not args.no_transports:
        ksteps = max(2, min(args.steps, 10))
        transports = {transport: {"ms_per_proof": round(1e3 * elapsed / max(1, args.steps), 3), "steps": args.steps, "primary": True}}
        other = "rccl" if transport == "mailbox" else "mailbox"
        oa.shard_finalize()
        sys.stdout.flush()
        saved_stdout = os.dup(1); os.dup2(2, 1)                 # librccl prints a version banner on stdout when a communicator comes up: the line rank 0 prints must stay the only one
        try:
            err = shard_join(other)
        finally:
            os.dup2(saved_stdout, 1); os.close(saved_stdout)
        if err:
            transports[other] = {"error": err}
        else:
            pw = [prove_once() for _ in range(2)]
            barrier(); t0 = time.perf_counter()
            for _ in range(ksteps):
                pw.append(prove_once())
            barrier(); el = max_over_ranks(time.perf_counter() - t0)
            assert {hashlib.sha256(p.bytes).hexdigest() for p in pw} == digests, "the proof depends on the transport of the per-round sums"
            transports[other] = {"ms_per_proof": round(1e3 * el / ksteps, 3), "steps": ksteps, "primary": False}
            oa.shard_finalize()
        err = shard_join(transport)                             # back to the primary transport for what follows
        if err:
            raise SystemExit("bench.py: cannot rejoin the sharded-proof exchange: " + err)
        transports["note"] = ("mailbox = node-local shared-memory post/flag/spin of the pinned per-round sums; rccl = ncclAllReduce(ncclUint64, ncclSum) of 8 u64 lanes per "
                              "field element on the GPUs (xGMI between cards) + one normalisation mod l; same proof bytes either way")

    # ---- extra: B independent proofs in flight per GPU, each prover thread with its own instance, witness and seed
    in_flight = None
    if conc >= 1:
        isteps = max(2, min(2 * args.steps, 40))              # 40 proofs per thread by default: a tenth of a second of proving said little (the figure moved by 2 x between runs)
        jobs = []
        for t in range(conc):
            if t == 0 and not shard:
                jobs.append((inst, wit, seed))
                continue
            rt = gen(n, ni, 1000 + 16 * rank + t)
            it = oa.Instance.new(rt["num_cons"], rt["num_vars"], rt["num_inputs"], rt["A"], rt["B"], rt["C"])
            it.prepare_device(gens)
            jobs.append((it, oa.Witness(it, oa.VarsAssignment.new(rt["vars"]), oa.InputsAssignment.new(rt["inputs"])), bytes([t + 1]) * 32))
            del rt
        gate, errors, lat = threading.Barrier(conc + 1), [], []

        def prover(job):
            try:
                it, wt, sd = job
                oa.NIZK.prove(it, wt, None, gens, label, sd)          # warm-up: this thread's device context and workspace
                gate.wait()
                for _ in range(isteps):
                    t1 = time.perf_counter(); oa.NIZK.prove(it, wt, None, gens, label, sd); lat.append(time.perf_counter() - t1)
            except BaseException as e:                              # noqa: BLE001 - reported after the join
                errors.append(e)
                gate.abort()

        ths = [threading.Thread(target=prover, args=(j,)) for j in jobs]
        for th in ths:
            th.start()
        barrier()
        gate.wait()
        t0 = time.perf_counter()
        for th in ths:
            th.join()
        barrier()
        el = time.perf_counter() - t0
        if errors:
            raise errors[0]
        el = max_over_ranks(el)
        in_flight = {"proofs_in_flight_per_gpu": conc, "n_gpus": world, "value": round(world * conc * isteps * n / el, 1), "unit": "constraints/s",
                     "proofs_per_thread": isteps, "latency_ms_per_proof": round(1e3 * sum(lat) / max(1, len(lat)), 3),
                     "note": "independent proofs: every prover thread has its own instance, witness and random-tape seed (only the generator window table is shared); "
                             "each GPU works for itself (for n_gpus > 1 this is the replicas figure)"}
        del jobs, ths

    steps = max(1, args.steps)
    ms_per_step = 1e3 * elapsed / steps
    value = (world if (world > 1 and not shard) else 1) * n * steps / elapsed

    import orc                                                  # the oracle: checker and cpu_baseline leg only

    def cpu_oracle_nizk(cr, cores):
        """one NIZK::prove (+ verify) of the instance by the plain-C oracle on `cores` host threads: (proof, seconds, stage ms, prove + verify ms)"""
        ci, cg = orc.OInstance(cr["num_cons"], cr["num_vars"], cr["num_inputs"], cr["A"], cr["B"], cr["C"]), orc.OGens(cr["num_cons"], cr["num_vars"], cr["num_inputs"])
        if lws > 1 and bound_cpus and node_cpus:
            os.sched_setaffinity(0, node_cpus)                  # the OpenMP team is created below, by a thread that may run anywhere on the node
        orc.set_threads(cores)
        tiny = oa.synth_r1cs(256, ni, 1)
        orc.nizk_prove(orc.OInstance(256, 256, ni, tiny["A"], tiny["B"], tiny["C"]), tiny["vars"], tiny["inputs"], orc.OGens(256, 256, ni))   # spin up the OpenMP team
        t0_ = time.perf_counter()
        cp, cms = orc.nizk_prove(ci, cr["vars"], cr["inputs"], cg, label, seed)
        ct = time.perf_counter() - t0_
        t0_ = time.perf_counter()
        assert orc.nizk_verify(ci, cr["inputs"], cg, cp) == 0
        te = 1e3 * (ct + time.perf_counter() - t0_)
        if lws > 1 and bound_cpus and node_cpus:
            os.sched_setaffinity(0, bound_cpus)
        return cp, ct, cms, te

    # N = 1: the box's CPU share (16 cores for one GPU).  N > 1: north_star wants the reference "on the node's own host cores" — every core
    # the launcher's process may use (rank 0 steps out of its L3 binding for that leg; the other ranks are idle at their barrier by then)
    cpu_cores = int(os.environ.get("OTTI_CPU_THREADS", min(usable_cores(), 16) if lws == 1 else node_cores))

    # ---- extra: the other sizes of BASELINE.json's range, sharded like the headline when N > 1 (collective: every rank takes part)
    sweep = None
    sweep_sizes = [] if (args.no_sweep or not args.sweep.strip() or args.dist != "uniform" or (world > 1 and not shard)) else \
        [int(x) for x in args.sweep.split(",") if x.strip() and int(x) != lg]
    if sweep_sizes:
        sweep = {}
        ssteps = 4
        for slg in sweep_sizes:
            over = max_over_ranks(time.perf_counter() - t_start) > args.sweep_budget_s
            if over:
                sweep["2^%d" % slg] = {"skipped": "time budget of the default run (--sweep-budget-s) used up before this size"}
                continue
            sn = 1 << slg
            sr = oa.synth_r1cs(sn, ni, 1)                         # public and deterministic: every rank derives it (the headline broadcasts its witness)
            si = oa.Instance.new(sr["num_cons"], sr["num_vars"], sr["num_inputs"], sr["A"], sr["B"], sr["C"])
            sg = gens if slg == lg else oa.NIZKGens.new(sr["num_cons"], sr["num_vars"], sr["num_inputs"])
            if sg is not gens and gens.table_info[1]:              # two wide window tables do not fit one card: the headline's makes room, and is built again after the sweep
                gens.release_device()
            si.prepare_device(sg)
            sw = oa.Witness(si, oa.VarsAssignment.new(sr["vars"]), oa.InputsAssignment.new(sr["inputs"]))
            sp = [prove_once(si, sw, sg), prove_once(si, sw, sg)]      # two warm-ups: the second proof of a size still ran 5 % slow (tools/gap_probe.py)
            barrier(); t0 = time.perf_counter(); s_each = []
            for _ in range(ssteps):
                t1 = time.perf_counter(); sp.append(prove_once(si, sw, sg)); s_each.append(1e3 * (time.perf_counter() - t1))
            barrier(); el = max_over_ranks(time.perf_counter() - t0)
            sd = {hashlib.sha256(p.bytes).hexdigest() for p in sp}
            assert len(sd) == 1
            if shard:
                same_on_every_rank(sp[-1].bytes, "ranks of a sharded sweep proof returned different bytes")
            sp[-1].verify(si, oa.InputsAssignment.new(sr["inputs"]), sg, label)
            g_ = golden_digest("nizk", sn)
            ok = None if g_ is None else (g_["proof_sha256"] in sd)
            assert ok is not False, "sweep proof 2^%d differs from the oracle's committed digest" % slg
            sc, sb = sg.table_info
            ent = {"ms_per_proof": round(1e3 * el / ssteps, 3), "ms_per_proof_p50": round(sorted(s_each)[len(s_each) // 2], 3), "ms_per_proof_max": round(max(s_each), 3),
                   "value": round(sn * ssteps / el, 1), "unit": "constraints/s", "steps": ssteps,
                   "hbm_frac_whole_proof": round(algorithmic_bytes(*si.dims[:2], 3 * sn) / (el / ssteps) / 1e9 / (HBM_PEAK_GBPS * (world if shard else 1)), 6),
                   "msm_window_bits": sc, "msm_table_GB": round(sb / 1e9, 2), "stage_ms": {k: round(v, 3) for k, v in sp[-1].stage_ms.items()},
                   "proof_sha256": next(iter(sd)), "equals_oracle_digest": ok}
            if rank == 0 and slg == 22 and not args.no_cpu_baseline:    # north_star's target size: the CPU figure beside it, same run
                cp, ct, _, _ = cpu_oracle_nizk(sr, cpu_cores)
                ent["cpu_baseline"] = {"value": round(sn / ct, 1), "unit": "constraints/s", "cores": cpu_cores, "kind": "port",
                                       "sample": f"one NIZK::prove of the 2^{slg} instance by the plain-C oracle (OpenMP, {cpu_cores} threads), {ct:.2f} s",
                                       "proof_equals_gpu_proof": cp == sp[-1].bytes}
                ent["vs_cpu_baseline"] = round(ent["value"] / ent["cpu_baseline"]["value"], 1)
            sweep["2^%d" % slg] = ent
            del sr, si, sw, sp
            if sg is not gens:
                del sg
            gc.collect()
        if not gens.table_info[1]:
            t0 = time.perf_counter(); inst.prepare_device(gens); t_reprepare = time.perf_counter() - t0
            assert gens.table_info[0] == cbits, "the headline's window table came back with another width"
        sweep["note"] = ("NIZK::prove on the synthetic instance of each size, %s; instance, generator table and witness resident; %d timed proofs after two warm-ups; "
                         "every proof compared with tests/golden/proofs.json" % (("one proof sharded over %d GPUs" % world) if shard else "one GPU", ssteps))

    # SNARK mode at N > 1: SNARK::prove is not sharded — every GPU proves the same instance on its own (replicas), between barriers; the
    # figure is world x n / (the slowest rank's best proof).  No collective sits inside a try block: a rank that fails still reaches them.
    snark_multi = None
    if world > 1 and not args.no_snark and (not rehearse or os.environ.get("OTTI_BENCH_REHEARSE_SNARK")):
        import torch
        s_err, s_ms, s_dig, s_stage = None, float("inf"), None, None
        try:
            nz_m = nnz // 3 if args.dist == "uniform" else int(max(r["A"].size, r["B"].size, r["C"].size))
            t0 = time.perf_counter()
            sg_m = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz_m)
            sc_m = oa.ComputationCommitment.encode(inst, sg_m)
            t_enc_m = time.perf_counter() - t0
            oa.SNARK.prove(inst, sc_m, wit, None, sg_m, b"snark_example", seed)
        except Exception as ex:                               # noqa: BLE001 — reported in the line, never raised past the collectives
            s_err = repr(ex)
        barrier()
        if s_err is None:
            try:
                for _ in range(3):
                    t0 = time.perf_counter(); sp_m = oa.SNARK.prove(inst, sc_m, wit, None, sg_m, b"snark_example", seed); s_ms = min(s_ms, 1e3 * (time.perf_counter() - t0))
                s_dig, s_stage = hashlib.sha256(sp_m.bytes).hexdigest(), {k: round(v, 3) for k, v in sp_m.stage_ms.items()}
                gold_m = golden_digest("snark", n) if args.dist == "uniform" else None
                if gold_m is not None and (gold_m["proof_sha256"] != s_dig or gold_m["commitment_sha256"] != hashlib.sha256(sc_m.bytes).hexdigest()):
                    s_err = "SNARK commitment / proof differ from the oracle's committed digests"
            except Exception as ex:                           # noqa: BLE001
                s_err = repr(ex)
        worst = torch.tensor([s_ms if s_err is None else 1e30], dtype=torch.float64, device=xdev)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        worst_ms = float(worst.item())
        if worst_ms < 1e29:
            snark_multi = {"value": round(world * n / (worst_ms * 1e-3), 1), "unit": "constraints/s", "ms_per_proof": round(worst_ms, 3), "n_gpus": world, "scaling": "weak",
                           "parallelism": "replicas: one independent SNARK::prove per GPU (SNARK mode is not sharded)", "encode_ms": round(1e3 * t_enc_m, 1),
                           "stage_ms": s_stage, "proof_sha256": s_dig, "equals_oracle_digest": (golden_digest("snark", n) is not None and args.dist == "uniform") or None,
                           "note": "every rank proves the N = 1 line's instance against its own computation commitment; ms_per_proof = the slowest rank's best of three, between barriers"}
            # ... and ONE SNARK::prove over all the GPUs (otti_snark_prove_sharded: R1CS part sharded, derefs commitment rows dealt out over the
            # ranks); every rank got this far without an error (the all_reduce above), so the collective inside the library has all its parties
            if shard:
                sh_ms, sh_err, sh_dig = float("inf"), None, None
                try:
                    oa.SNARK.prove_sharded(inst, sc_m, wit, sg_m, b"snark_example", seed)
                    for _ in range(3):
                        barrier(); t0 = time.perf_counter()
                        sp_s = oa.SNARK.prove_sharded(inst, sc_m, wit, sg_m, b"snark_example", seed)
                        barrier(); sh_ms = min(sh_ms, 1e3 * max_over_ranks(time.perf_counter() - t0))
                    sh_dig = hashlib.sha256(sp_s.bytes).hexdigest()
                    same_on_every_rank(sp_s.bytes, "ranks of a sharded SNARK proof returned different bytes")
                    if sh_dig != s_dig:
                        sh_err = "the sharded SNARK proof differs from the single-GPU proof"
                except Exception as ex:                       # noqa: BLE001
                    sh_err = repr(ex)
                replicas = snark_multi
                if sh_err is None:
                    snark_multi = {"value": round(n / (sh_ms * 1e-3), 1), "unit": "constraints/s", "ms_per_proof": round(sh_ms, 3), "n_gpus": world, "scaling": "strong",
                                   "parallelism": "1 SNARK::prove sharded over %d GPUs: R1CS proof as the headline, derefs commitment rows dealt out over the ranks, product circuits by residue classes (per-round sums over the exchange); host rounds and evaluation proofs on every rank alike" % world,
                                   "stage_ms": {k: round(v, 3) for k, v in sp_s.stage_ms.items()}, "proof_sha256": sh_dig, "equals_oracle_digest": replicas["equals_oracle_digest"],
                                   "encode_ms": replicas["encode_ms"], "replicas": replicas,
                                   "note": "ms_per_proof = best of three between barriers, max over ranks; `replicas` = the weak-scaling figure (one independent proof per GPU)"}
                else:
                    snark_multi = dict(replicas, sharded_error=sh_err)
        else:
            snark_multi = {"error": s_err or "a peer rank failed"}
        del worst
    if shard:
        oa.shard_finalize()
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # oracle cross-check of the GPU path (checker only; small size, outside the timed region)
    rs = oa.synth_r1cs(1 << 12, ni, 1)
    si = oa.Instance.new(rs["num_cons"], rs["num_vars"], rs["num_inputs"], rs["A"], rs["B"], rs["C"]); sg = oa.NIZKGens.new(1 << 12, 1 << 12, ni)
    sp = oa.NIZK.prove(si, oa.VarsAssignment.new(rs["vars"]), oa.InputsAssignment.new(rs["inputs"]), sg, label, seed)
    oi, og = orc.OInstance(1 << 12, 1 << 12, ni, rs["A"], rs["B"], rs["C"]), orc.OGens(1 << 12, 1 << 12, ni)
    op, _ = orc.nizk_prove(oi, rs["vars"], rs["inputs"], og, label, seed)
    parity_ok = sp.bytes == op
    del si, sg, sp

    def traffic_of(kernel_name, snark_run=False):
        """HBM bytes of the kernel's largest launch from a SEPARATE rocprofv3 --pmc pass kept under profiles/ (never measured in this run)"""
        for cand in (("r4_snark_pmc_traffic.json", "r3_snark_pmc_traffic.json") if snark_run else ("r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json")):
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", cand)))
                names = [k_.strip() for k_ in (kernel_name or "").split("/")]                     # a class may name two kernels: the first that was profiled
                hit = next((k_ for k_ in names if k_ in pm["kernels"]), None)
                if hit is None and names and names[0] + "<true>" in pm["kernels"]:
                    hit = names[0] + "<true>"
                kernel_name = hit
                if lg == pm.get("log2_constraints", 20) and cbits == pm.get("msm_window_bits", 12) and kernel_name in pm["kernels"]:
                    return pm["kernels"][kernel_name]["traffic_bytes_corrected"], ("profiles/%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, largest launch "
                                                                                     "of the kernel (not measured in this run)" % cand)
            except Exception:
                pass
        return None, None

    def roofline_of(dom_, stats_, breakdown_, bytes_fn, share, snark_run=False):
        cnt, tot_ms = stats_[dom_]
        per_proof = breakdown_[dom_][0]                             # launches of that class in one proof
        if not (cnt and per_proof):
            return None
        avg_ms = tot_ms / cnt
        bytes_per_launch = bytes_fn(dom_) / per_proof / share
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = traffic_of(KERNEL_NAMES.get(dom_), snark_run)
        rf = {"bound": "hbm", "kernel": dom_, "kernel_name": KERNEL_NAMES.get(dom_, dom_), "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
              "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": traffic_src, "launches": cnt,
              "launches_per_proof": per_proof, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
              "chosen_from": {k: round(v[1], 3) for k, v in breakdown_.items() if v[0]}}
        if dom_ in ("msm_rows", "msm_small"):
            W = 253 // cbits + 1
            adds = bytes_fn(dom_) // F * W / per_proof / share           # one mixed addition per scalar and window
            rate = adds / (avg_ms * 1e-3)
            isa = MAD_U64_LANE_OPS / MADS_PER_MADD
            rf["alu"] = {"bound": "integer ALU (v_mad_u64_u32)", "achieved": round(rate / 1e9, 3), "peak": round(madd_peak / 1e9, 3),
                         "unit": "G mixed additions/s", "frac": round(rate / madd_peak, 4),
                         "peak_isa": round(isa / 1e9, 3), "frac_isa": round(rate / isa, 4),
                         "note": "peak = otti_bench_madd_peak measured in this run: the kernel's own 7-multiplication mixed addition, operands in registers, every CU busy; "
                                 "peak_isa = the hardware's v_mad_u64_u32 rate (32.6 T lane-ops/s, tools/fmabench.hip) / 700 multiply-adds per mixed addition: the "
                                 "roof no implementation of this formula on these limbs can pass (carry sweeps, additions and the table gather all come on top)"}
        return rf

    roofline = roofline_of(dom, stats, breakdown, lambda c_: class_bytes_per_proof(c_, N, V, nnz), world if shard else 1)
    if roofline is not None and world > 1:
        roofline["note"] = "N > 1: fixed to the class that dominates at N = 1 (the witness commitment), priced on this rank's 1/%d share of the scalars" % world
    # ---- the second roof of the "streaming" classes: they carry 7-13 Montgomery products per 192 bytes, so the multiplier binds before HBM does.
    # products per proof by the kernels' own arithmetic (k_sumcheck.hip, k_sparse.hip): phase one 7 per pair in the first round and 13 per
    # quad in the fused fold + sums rounds (10 N in all); phase two 2 per pair, then 6 per quad (8 V over tables of 2 V); the two sparse
    # products one per non-zero entry (+ 3 per output row for the fused combination); eq tables one per element written
    products = {"sc_cubic": 10 * N, "sc_quad": 8 * (2 * V) // 2, "spmv": 2 * nnz + 3 * 2 * V, "eq": N + 2 * V}
    field_mul = {"peak": round(frmul_peak / 1e9, 2), "unit": "G Montgomery products/s", "classes": {},
                 "note": "peak = otti_bench_fr_mul_peak measured in this run: the nine-limb product of fr9.h (round 4; operands unpacked in registers, two chains per lane, every CU "
                         "busy): 135 v_mad_u64_u32 + 17 64-bit shifts + 9 v_mul_lo_u32 ('heavy': ~4.3 nominal cycles per wave instruction each) + ~27 masks and moves (~2.4) in one asm block; "
                         "the 8 x u32 form it replaces (104 multiply-adds + 104 carry adds, 143-147 G/s) is timed beside it by tools/limbbench (profiles/r4_limbbench.txt); "
                         "classes: products per proof / summed device time of the class in one instrumented proof "
                         "(most launches of a sum-check are rounds too short to fill the chip, so the whole-proof rate sits far below the roof; profiles/*kbw* has the rates at size)"}
    for cls, nprod in products.items():
        if breakdown.get(cls, (0, 0))[0]:
            rate = nprod / (world if shard else 1) / (breakdown[cls][1] * 1e-3)
            field_mul["classes"][cls] = {"products_per_proof": int(nprod), "ms_per_proof": round(breakdown[cls][1], 3), "achieved": round(rate / 1e9, 2), "frac": round(rate / frmul_peak, 4),
                                         "hbm_frac": round(class_bytes_per_proof(cls, N, V, nnz) / (world if shard else 1) / (breakdown[cls][1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
    whole = algorithmic_bytes(N, V, nnz)
    proof_gbps = whole / (ms_per_step * 1e-3) / 1e9

    cpu_baseline, cpu_e2e_ms = None, None
    if not args.no_cpu_baseline:
        cores = cpu_cores
        clg = args.cpu_log2 if args.cpu_log2 is not None else lg
        cr = r if clg == lg else gen(1 << clg, ni, 1)
        cp, ct, cms, cpu_e2e_ms = cpu_oracle_nizk(cr, cores)
        same = (cp == final_proof) if (clg == lg and (shard or world == 1)) else None
        # SURVEY 8(d) also asks for the single-thread figure: one proof of a 2^16 instance on one core (a bounded sample)
        slg = min(clg, 16)
        sr = gen(1 << slg, ni, 1)
        si_, sg_ = orc.OInstance(sr["num_cons"], sr["num_vars"], sr["num_inputs"], sr["A"], sr["B"], sr["C"]), orc.OGens(sr["num_cons"], sr["num_vars"], sr["num_inputs"])
        orc.set_threads(1)
        t0 = time.perf_counter()
        orc.nizk_prove(si_, sr["vars"], sr["inputs"], sg_, label, seed)
        st = time.perf_counter() - t0
        orc.set_threads(cores)
        cpu_baseline = {"value": round((1 << clg) / ct, 1), "unit": "constraints/s", "cores": cores, "kind": "port",
                        "single_thread": {"value": round((1 << slg) / st, 1), "unit": "constraints/s", "cores": 1, "sample": f"one proof of the 2^{slg} instance, {st:.2f} s"},
                        "sample": f"one NIZK::prove of the synthetic 2^{clg}-constraint R1CS by the plain-C oracle (OpenMP, {cores} threads), {ct:.2f} s; "
                                  "reference Spartan (Rust) is not buildable here" + ("; timed on rank 0's host cores after the ranks' timed region" if world > 1 else ""),
                        "proof_equals_gpu_proof": same, "stage_ms": [round(x, 1) for x in cms]}

    # ---- the one-shot path run.py executes: spzk verify --nizk <three zkif files>, one process
    spzk_e2e = None
    if not args.no_e2e and world == 1:
        spzk = os.path.join(ROOT, "otti_amd", "spzk")
        with tempfile.TemporaryDirectory(prefix="otti-bench-") as td:
            pre = os.path.join(td, "w")
            oa.zkif_write(r, pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif")
            fsize = sum(os.path.getsize(pre + e) for e in (".zkif", ".inp.zkif", ".wit.zkif"))
            # what ANY process pays on this box for starting the HIP runtime and launching one kernel (tools/hipfloor.hip): the floor under the one-shot figure
            floor_ms, hipfloor = None, os.path.join(ROOT, "tools", "hipfloor.bin")
            if os.path.exists(hipfloor):
                for _ in range(3):
                    t0 = time.perf_counter()
                    if subprocess.run([hipfloor], capture_output=True).returncode == 0:
                        dt = 1e3 * (time.perf_counter() - t0); floor_ms = dt if floor_ms is None else min(floor_ms, dt)
            best, lines = None, None
            for _ in range(3):                                  # later runs: page cache warm, as in a pipeline that has just written the files
                t0 = time.perf_counter()
                res = subprocess.run([spzk, "verify", "--nizk", pre + ".zkif", pre + ".inp.zkif", pre + ".wit.zkif", "--seed", "2a" * 32], capture_output=True, text=True)
                dt = 1e3 * (time.perf_counter() - t0)
                assert res.returncode == 0 and "Verification successful" in res.stdout, res.stdout + res.stderr
                if best is None or dt < best:
                    best, lines = dt, res.stdout
            stages, before_main, in_main = {}, None, None
            for ln in lines.splitlines():
                if "process start to main()" in ln:
                    import re
                    m_ = re.search(r"main\(\) (-?[0-9.]+) ms .*main\(\) ([0-9.]+) ms", ln)
                    if m_:
                        before_main, in_main = float(m_.group(1)), float(m_.group(2))
                    continue
                parts = ln.strip("* ").rsplit(" ", 2)
                if len(parts) == 3 and parts[2] == "ms":
                    try:
                        stages[parts[0].strip()] = float(parts[1])
                    except ValueError:
                        pass
            spzk_e2e = {"ms": round(best, 1), "process": "otti_amd/spzk verify --nizk c.zkif i.inp.zkif w.wit.zkif (process start to exit, files in the page cache)",
                        "zkif_bytes": fsize, "stages_ms": stages,
                        "accounting_ms": None if in_main is None else {
                            "exec_to_main": before_main, "main": in_main, "after_main_and_spawn": round(best - (before_main or 0.0) - in_main, 1),
                            "stages_sum": round(sum(v for k, v in stages.items() if k in ("zkif_load", "setup (Instance::new, NIZKGens::new, device tables)", "NIZK::prove", "NIZK::verify")), 1),
                            "note": "exec_to_main: the dynamic loader mapping libamdhip64 and its dependencies before main() (10 ms resolution); main: everything the program does, ending in "
                                    "_exit (no runtime teardown, no frees); after_main_and_spawn: what is left of the parent's wall clock — fork/exec on the parent's side and process exit"},
                        "hip_process_floor_ms": None if floor_ms is None else round(floor_ms, 1),
                        "floor_note": "tools/hipfloor.bin in this run, best of 3: a process that only starts the HIP runtime, allocates, launches one kernel and copies 256 bytes back",
                        "cpu_prove_plus_verify_ms": None if cpu_e2e_ms is None else round(cpu_e2e_ms, 1),
                        "cpu_note": "CPU oracle NIZK::prove + NIZK::verify of the same instance on the host cores above, instance already parsed (no zkif reader in the oracle)"}

    # ---- SNARK mode (BASELINE.json's metric names it; run.py itself only invokes --nizk): SNARK::encode once, then SNARK::prove of the same workload
    snark = None
    if not args.no_snark and world == 1:
        slabel = b"snark_example"
        nz = nnz // 3 if args.dist == "uniform" else int(max(r["A"].size, r["B"].size, r["C"].size))
        # the headline's NIZK window table (97 GB) makes room: a SNARK prover holds the two tables of SNARKGens and no third one; with the card to
        # themselves they come out as c = 16 / 16 (206 + 52 GB) instead of 15 / 16 beside it.  This is the last leg: nothing needs the table again.
        if gens.table_info[1] and os.environ.get("OTTI_BENCH_SNARK_KEEP_NIZK_TABLE") != "1":      # (=1: A/B of the two table widths)
            gens.release_device()
        t0 = time.perf_counter()
        sgens = oa.SNARKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"], nz)
        scomm = oa.ComputationCommitment.encode(inst, sgens)
        t_encode = time.perf_counter() - t0
        sp_list = [oa.SNARK.prove(inst, scomm, wit, None, sgens, slabel, seed)]        # warms up (workspace allocation); witness resident in HBM, as for the headline
        oa.stats_enable(True)                                   # one instrumented proof: the dominant class over SNARK::prove's kernels
        sp_list.append(oa.SNARK.prove(inst, scomm, wit, None, sgens, slabel, seed))
        s_break = oa.stats_read()
        s_dom = max(s_break, key=lambda k: s_break[k][1])
        # the timed proofs run as a caller runs them (armed launches and the persistent sum-check tail on).  A class that has armed
        # launches cannot be timed with them on (a waiting kernel's duration includes the host), so when such a class dominates its
        # launch durations come from the instrumented proof above (nothing armed there), not from the timed proofs.
        s_timed_stats = s_dom not in ("pc_round", "msm_small", "sc_cubic", "sc_quad")
        oa.stats_enable(s_timed_stats, only=s_dom if s_timed_stats else None)
        s_armed = oa.armed_launches_on()
        sms = []
        for _ in range(5):
            t0 = time.perf_counter(); spf = oa.SNARK.prove(inst, scomm, wit, None, sgens, slabel, seed); sms.append(1e3 * (time.perf_counter() - t0)); sp_list.append(spf)
        s_stats = dict(oa.stats_read()) if s_timed_stats else dict(s_break)
        oa.stats_enable(False)
        sdig = {hashlib.sha256(p.bytes).hexdigest() for p in sp_list}
        assert len(sdig) == 1
        vcomm = oa.ComputationCommitment.from_bytes(scomm.bytes); sv = []
        for _ in range(3):
            t0 = time.perf_counter(); sp_list[-1].verify(vcomm, inputs, sgens, slabel); sv.append(time.perf_counter() - t0)
        t_sverify = min(sv)
        best = min(sms)
        sgold = golden_digest("snark", n) if args.dist == "uniform" else None
        s_ok = None if sgold is None else (sgold["proof_sha256"] in sdig and sgold["commitment_sha256"] == hashlib.sha256(scomm.bytes).hexdigest())
        assert s_ok is not False, "SNARK commitment / proof differ from the oracle's committed digests for this size"
        Nz = 1 << max(1, (nz - 1).bit_length()); Mm = 1 << max((N - 1).bit_length(), (2 * V - 1).bit_length())
        s_roof = roofline_of(s_dom, s_stats, s_break, lambda c_: snark_class_bytes_per_proof(c_, N, V, nnz, Nz, Mm), 1, snark_run=True)
        if s_roof is not None:
            s_roof.pop("alu", None)                            # (two window widths in one class here: the addition rate is the headline's figure)
            s_roof["launch_durations_from"] = "the timed proofs" if s_timed_stats else "one instrumented proof with nothing armed (per-round launches, no persistent tail): the class has armed launches"
        # oracle parity at 2^12 and the CPU figure on a bounded sample (2^16: the oracle's SNARK prover is half a minute at 2^20)
        rs2 = oa.synth_r1cs(1 << 12, ni, 1)
        si2 = oa.Instance.new(1 << 12, 1 << 12, ni, rs2["A"], rs2["B"], rs2["C"]); sg2 = oa.SNARKGens.new(1 << 12, 1 << 12, ni, 1 << 12)
        sp2 = oa.SNARK.prove(si2, oa.ComputationCommitment.encode(si2, sg2), oa.VarsAssignment.new(rs2["vars"]), oa.InputsAssignment.new(rs2["inputs"]), sg2, slabel, seed)
        oi2 = orc.OInstance(1 << 12, 1 << 12, ni, rs2["A"], rs2["B"], rs2["C"]); og2 = orc.OSnarkGens(1 << 12, 1 << 12, ni, 1 << 12)
        op2, _ = orc.snark_prove(oi2, orc.OSnarkComm.encode(oi2, og2), rs2["vars"], rs2["inputs"], og2, slabel, seed)
        cpu_s = None
        if not args.no_cpu_baseline:
            slg2 = min(lg, 16)
            rs3 = gen(1 << slg2, ni, 1); nz3 = int(max(rs3["A"].size, rs3["B"].size, rs3["C"].size))
            oi3 = orc.OInstance(rs3["num_cons"], rs3["num_vars"], rs3["num_inputs"], rs3["A"], rs3["B"], rs3["C"]); og3 = orc.OSnarkGens(rs3["num_cons"], rs3["num_vars"], rs3["num_inputs"], nz3)
            oc3 = orc.OSnarkComm.encode(oi3, og3)
            t0 = time.perf_counter(); orc.snark_prove(oi3, oc3, rs3["vars"], rs3["inputs"], og3, slabel, seed); ct3 = time.perf_counter() - t0
            cpu_s = {"value": round((1 << slg2) / ct3, 1), "unit": "constraints/s", "cores": orc.lib.orc_get_threads(), "kind": "port",
                     "sample": f"one SNARK::prove of the 2^{slg2} instance by the plain-C oracle, {ct3:.2f} s"}
        snark = {"value": round(n / (best * 1e-3), 1), "unit": "constraints/s", "ms_per_proof": round(best, 3), "ms_per_proof_mean": round(sum(sms) / len(sms), 3), "timed_proofs": len(sms), "encode_ms": round(1e3 * t_encode, 1),
                 "verify_ms": round(1e3 * t_sverify, 2), "verify_first_call_ms": round(1e3 * sv[0], 2), "proof_bytes": len(sp_list[-1].bytes), "commitment_bytes": len(scomm.bytes),
                 "stage_ms": {k: round(v, 3) for k, v in sp_list[-1].stage_ms.items()}, "roofline": s_roof, "armed_launches": s_armed,
                 "kernel_ms_per_proof": {k: round(v[1], 3) for k, v in s_break.items() if v[0]}, "kernel_launches_per_proof": {k: v[0] for k, v in s_break.items() if v[0]},
                 "proof_sha256": next(iter(sdig)), "equals_oracle_digest": s_ok, "oracle_parity_2^12": sp2.bytes == op2, "cpu_baseline": cpu_s,
                 "note": "ms_per_proof = the best of the timed proofs (two warm-ups before them), ms_per_proof_mean their mean.  SNARK::prove = R1CSProof (the headline's NIZK path) + R1CSEvalProof against the computation commitment made once by SNARK::encode (encode_ms includes building the "
                         "second generator window table); witness resident in HBM (otti_snark_prove_resident), as for the headline; equals_oracle_digest: commitment and proof against "
                         "tests/golden/snark_proofs.json (the CPU oracle's SNARK::encode / prove of this very instance).  The headline's NIZK window table is released before this leg "
                         "(a SNARK prover holds SNARKGens' two tables and no third: with the card to themselves they are built with c = 16 / 16, 206 + 52 GB)"}

    out = {
        "metric": "R1CS constraints/sec proved (Spartan NIZK) at 2^%d" % lg, "value": round(value, 1), "unit": "constraints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        # scalars a reader of a truncated record wants early: the slowest timed step beside the mean, and SNARK mode's figure (details in `snark`)
        "ms_per_step_p50": round(sorted(step_ms)[len(step_ms) // 2], 3), "ms_per_step_p99": round(sorted(step_ms)[min(len(step_ms) - 1, (99 * len(step_ms)) // 100)], 3),
        "ms_per_step_max": round(max(step_ms), 3),
        "snark_ms_per_proof": (snark or {}).get("ms_per_proof") if world == 1 else (snark_multi or {}).get("ms_per_proof"),
        "snark_value": (snark or {}).get("value") if world == 1 else (snark_multi or {}).get("value"),
        "scaling": "weak" if (world == 1 or not shard) else "strong", "vs_baseline": None, "dtype": "u256 (GF(l) / GF(2^255-19), 8 x u32 limbs)", "data": "synthetic",
        "config": {"workload": (f"synthetic satisfiable R1CS, 2^{lg} constraints = variables, 10 inputs, 1 nnz/row/matrix, uniform GF(l) witness "
                                if args.dist == "uniform" else
                                f"synthetic compiler-like R1CS, 2^{lg} constraints = variables, 10 inputs, 1..8 nnz/row/matrix, 90% of the witness < 2^64, heavy constant column ")
                               + "(SURVEY 8d); witness/instance/generators resident in HBM; one step = one NIZK::prove, one proof at a time",
                   "parallelism": ("1 proof sharded over %d GPUs (%s exchange of the per-round sums)" % (world, transport)) if shard else
                                  ("1 independent proof per GPU" if world > 1 else "single GPU"),
                   "msm_window_bits": cbits, "msm_table_GB": round(table_bytes / 1e9, 2), "armed_launches": armed_on,
                   "rank_cpu_binding": ("each rank on %d L3 groups of its own (rank 0: %d logical CPUs)" % (len(bound_cpus) // 16 or 1, len(bound_cpus))) if bound_cpus else None},
        "roofline": roofline,
        "field_mul": field_mul,
        "cpu_baseline": cpu_baseline,
        "transports": transports,
        "sweep": sweep,
        "in_flight": in_flight,
        "snark": snark if world == 1 else snark_multi,
        "spzk_e2e": spzk_e2e,
        "stage_ms": {k: round(v / steps, 3) for k, v in stage_acc.items()},
        "kernel_ms_per_proof": {k: round(v[1], 3) for k, v in breakdown.items() if v[0]},
        "kernel_launches_per_proof": {k: v[0] for k, v in breakdown.items() if v[0]},
        "whole_proof_algorithmic_GBps": round(proof_gbps, 2), "whole_proof_hbm_frac": round(proof_gbps / (HBM_PEAK_GBPS * (world if shard else 1)), 6),
        "prepare_device_ms": round(1e3 * t_prepare, 1),
        "prepare_device_breakdown_ms": {"window_table_allocations": round(table_alloc_ms, 1), "window_table_upload_and_kernels": round(table_kernels_ms, 1),
                                        "instance_upload_and_csr_build": round(1e3 * t_prepare - table_alloc_ms - table_kernels_ms, 1),
                                        "note": "hipMalloc of the table (tens of GB: the driver maps and clears the pages) is the part that varies between boxes; the kernels scale with the table"}, "witness_upload_ms": round(1e3 * t_upload, 2), "verify_ms": round(1e3 * t_verify, 2), "verify_first_call_ms": round(1e3 * t_verify_calls[0], 2),
        "proof_bytes": len(final_proof), "proof_sha256": next(iter(digests)), "equals_oracle_digest": digest_ok, "oracle_parity_2^12": parity_ok,
        "wall_s": round(time.perf_counter() - t_start, 1),
    }
    sys.stdout.flush()
    os.write(line_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
