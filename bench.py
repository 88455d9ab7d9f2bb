#!/usr/bin/env python3
"""Benchmark of the MI355X Spartan NIZK proving path.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one batch of B independent NIZK::prove calls of the workload in flight on a GPU at once (B prover threads of the rank's
process, each with its own stream and workspace, sharing the instance, the generator window table and the resident witness: a single
proof is a chain of ~50 strictly sequential Fiat-Shamir rounds that leave most of the chip idle, so a prover that serves a stream of
proofs keeps several in flight).  B = --concurrent (default: what this rank's host cores feed, at most 6; --concurrent 1 = one proof
at a time, whose latency is reported as single_proof_ms either way).  Instance, generators (and their window table) and witness are
resident in HBM before timing; W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device sync; rank 0 prints
ONE JSON line; value = constraints of all proofs of all ranks / elapsed.
metric = BASELINE.json's "R1CS constraints/sec proved" on the synthetic 2^20-constraint R1CS of SURVEY.md 8(d).
Every timed proof is checked: all K proofs of a rank are byte-identical (fixed random-tape seed) and the product verifier
accepts them; rank 0 additionally compares a 2^12 proof with the CPU oracle (checker only, outside the timed region).

N > 1, default: each rank proves its own independent instance of the same size — no data-path collective; scaling = weak
(proofs are independent objects; this is how a node serves a stream of Otti proofs).
N > 1 with --shard: ALL ranks prove ONE instance together (commitment rows, sum-check tables and the sparse
matrices are sharded; per-round sums cross ranks through the node-local mailbox of otti_amd/csrc/shard.h; the witness is
replicated beforehand with a torch.distributed broadcast over RCCL/xGMI); value = N / time of that one proof; scaling = strong.
OTTI_BENCH_REHEARSE=1 puts every rank on GPU 0 with the gloo backend (how the sharded mode is rehearsed on a one-GPU box).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (guide: MI355X_MICROARCH.md); ~6300 achievable
F = 32                          # bytes per field element
MADD_PEAK_G = 24.07             # measured: 10-limb mixed point additions per second (x1e9), tools/mulbench.hip on MI355X


def algorithmic_bytes(N, V, nnz):
    """compulsory HBM traffic of one proof, W = 80*nnz + 704*N + 736*V bytes."""
    return 80 * nnz + 704 * N + 736 * V


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2-constraints", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist", choices=("uniform", "compiler"), default="uniform", help="synthetic instance distribution (SURVEY 8d); the metric is quoted on 'uniform'")
    ap.add_argument("--cpu-log2", type=int, default=None, help="size of the CPU-baseline sample (default: same workload)")
    ap.add_argument("--shard", action="store_true", help="N > 1: all ranks prove ONE instance together (strong scaling) instead of one proof per GPU")
    ap.add_argument("--concurrent", type=int, default=0, help="prover threads per GPU: a step is then that many proofs of the workload in flight at once "
                    "(each thread has its own stream/workspace; instance, window table and witness are shared).  0 = as many as the host "
                    "cores of this rank feed (2 host threads per prover, at most 6); 1 = one proof at a time (latency mode)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearse = bool(os.environ.get("OTTI_BENCH_REHEARSE"))
    if rehearse:
        local_rank = 0
    os.environ.setdefault("OTTI_DEVICE", str(local_rank))
    shard = bool(args.shard and world > 1)
    # how many proofs this rank keeps in flight, and the environment that goes with it — before anything initialises the HIP runtime
    lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    cores_here = max(1, usable_cores() // lws)                  # host cores of this rank
    conc = 1 if shard else (args.concurrent if args.concurrent > 0 else max(1, min(6, cores_here // 2)))
    # host threads per prover (itself + spinning helpers for the per-round sigma-protocol work): share this rank's cores fairly
    os.environ.setdefault("OTTI_HOST_THREADS", str(max(1, min(4, cores_here // conc))))
    if conc > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")         # one hardware queue per prover stream (HIP's default is 4)
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
    import threading
    import numpy as np
    import otti_amd as oa

    if oa.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible; the proving path has no CPU fallback")

    lg = args.log2_constraints
    n, ni, label, seed = 1 << lg, 10, b"nizk_example", b"\x2a" * 32
    gen = oa.synth_r1cs if args.dist == "uniform" else oa.synth_r1cs_compiler_like
    r = gen(n, ni, 1 if shard else 1 + rank)               # one proof per GPU: each rank its own instance; --shard: the same one
    inst = oa.Instance.new(r["num_cons"], r["num_vars"], r["num_inputs"], r["A"], r["B"], r["C"])
    gens = oa.NIZKGens.new(r["num_cons"], r["num_vars"], r["num_inputs"])
    vars_, inputs = oa.VarsAssignment.new(r["vars"]), oa.InputsAssignment.new(r["inputs"])
    if shard:
        # the instance is public (every rank builds it); the WITNESS exists on rank 0 only and reaches the other GPUs over RCCL/xGMI
        import torch
        wbuf = torch.from_numpy(np.ascontiguousarray(r["vars"])).to(xdev) if rank == 0 else torch.zeros((r["vars"].shape[0], 32), dtype=torch.uint8, device=xdev)
        ibuf = torch.from_numpy(np.ascontiguousarray(r["inputs"])).to(xdev) if rank == 0 else torch.zeros((ni, 32), dtype=torch.uint8, device=xdev)
        dist.broadcast(wbuf, src=0); dist.broadcast(ibuf, src=0)
        vars_, inputs = oa.VarsAssignment.new(wbuf.cpu().numpy()), oa.InputsAssignment.new(ibuf.cpu().numpy())
        name = [("otti-bench-%d-%d" % (os.getpid(), time.time_ns())) if rank == 0 else None]
        dist.broadcast_object_list(name, src=0)
        oa.shard_init(name[0], rank, world)
    t0 = time.perf_counter()
    inst.prepare_device(gens)                              # CSR upload + generator window table: resident before timing
    t_prepare = time.perf_counter() - t0
    cbits, table_bytes = gens.table_info
    t0 = time.perf_counter()
    wit = oa.Witness(inst, vars_, inputs)                  # witness resident in HBM before timing
    t_upload = time.perf_counter() - t0
    N, V, _ = inst.dims
    nnz = int(r["A"].size + r["B"].size + r["C"].size)

    def barrier():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def prove_once():
        return oa.NIZK.prove_sharded(inst, wit, gens, label, seed) if shard else oa.NIZK.prove(inst, wit, None, gens, label, seed)

    proofs = []
    for _ in range(args.warmup):
        proofs.append(prove_once())
    # latency of ONE proof with the GPU to itself (untimed region; the timed region below may keep several proofs in flight)
    single_ms = []
    for _ in range(3):
        t0 = time.perf_counter(); proofs.append(prove_once()); single_ms.append(1e3 * (time.perf_counter() - t0))
    # one untimed, fully instrumented proof: per-class kernel time -> picks the dominant kernel class
    oa.stats_enable(True)
    prove_once()
    breakdown = oa.stats_read()
    # dominant kernel = the streaming/ALU kernel class with the largest summed time.  "msm_small" is the same k_msm_rows kernel in
    # its one/two-row launches of the bullet reduction (latency-bound by construction); it is reported in kernel_ms_per_step only.
    dom = max(("msm_rows", "sc_cubic", "sc_quad", "spmv"), key=lambda k: breakdown[k][1])
    # timed region: HIP events only around the dominant class (two event records per launch would otherwise tax every round)
    oa.stats_enable(True, only=dom)
    # --concurrent B: B - 1 more prover threads (own device context each), warmed up and parked on a barrier
    # the K steps x B proofs of the timed region are handed out from one counter, so no thread idles while another still has work
    gate, others, other_proofs, other_stats, errors = threading.Barrier(conc), [], [], [], []
    todo, todo_lock = [conc * args.steps], threading.Lock()

    def take():
        with todo_lock:
            if todo[0] <= 0:
                return False
            todo[0] -= 1
            return True

    def extra_prover():
        try:
            oa.stats_enable(True, only=dom)                     # kernel timing state is per prover thread; its event pool is built on first use
            mine = [prove_once() for _ in range(max(1, args.warmup))]
            oa.stats_enable(True, only=dom)                     # again: resets the counters for the timed region
            gate.wait()
            while take():
                mine.append(prove_once())
            other_stats.append(oa.stats_read()[dom])
            oa.stats_enable(False)
            other_proofs.extend(mine)
        except BaseException as e:                              # noqa: BLE001 - reported after the join
            errors.append(e)
            gate.abort()

    for _ in range(conc - 1):
        th = threading.Thread(target=extra_prover); th.start(); others.append(th)
    barrier()
    gate.wait()
    t0 = time.perf_counter()
    stage_acc, mine_n = {}, 0
    while take():
        p = prove_once()                                       # returns after the library's stream has been synchronised
        proofs.append(p); mine_n += 1
        for k, v in p.stage_ms.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    for th in others:
        th.join()
    barrier()
    elapsed = time.perf_counter() - t0
    if errors:
        raise errors[0]
    proofs += other_proofs
    stats = dict(oa.stats_read())
    oa.stats_enable(False)
    for cnt_, ms_ in other_stats:                               # the dominant kernel's launches of every prover thread in the timed region
        stats[dom] = (stats[dom][0] + cnt_, stats[dom][1] + ms_)

    # correctness of what was timed
    digests = {hashlib.sha256(p.bytes).hexdigest() for p in proofs}
    assert len(digests) == 1, "proofs of the same inputs and seed differ between steps"
    t0 = time.perf_counter()
    proofs[-1].verify(inst, inputs, gens, label)
    t_verify = time.perf_counter() - t0

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if shard:                                               # every rank must hold the same proof
            dg = torch.frombuffer(bytearray(hashlib.sha256(proofs[-1].bytes).digest()), dtype=torch.uint8).to(xdev)
            every = [torch.zeros_like(dg) for _ in range(world)]
            dist.all_gather(every, dg)
            assert all(bool((e == dg).all()) for e in every), "ranks of a sharded proof returned different bytes"
            oa.shard_finalize()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    steps = max(1, args.steps)
    ms_per_step = 1e3 * elapsed / steps
    value = (1 if shard else world) * conc * n * steps / elapsed

    # oracle cross-check of the GPU path (checker only; small size, outside the timed region)
    import orc
    rs = oa.synth_r1cs(1 << 12, ni, 1)
    si = oa.Instance.new(rs["num_cons"], rs["num_vars"], rs["num_inputs"], rs["A"], rs["B"], rs["C"]); sg = oa.NIZKGens.new(1 << 12, 1 << 12, ni)
    sp = oa.NIZK.prove(si, oa.VarsAssignment.new(rs["vars"]), oa.InputsAssignment.new(rs["inputs"]), sg, label, seed)
    oi, og = orc.OInstance(1 << 12, 1 << 12, ni, rs["A"], rs["B"], rs["C"]), orc.OGens(1 << 12, 1 << 12, ni)
    op, _ = orc.nizk_prove(oi, rs["vars"], rs["inputs"], og, label, seed)
    parity_ok = sp.bytes == op

    # dominant kernel: the one with the largest summed HIP-event time inside the timed region
    cnt, tot_ms = stats[dom]
    ell = V.bit_length() - 1
    Lsz, Rsz = 1 << (ell // 2), 1 << (ell - ell // 2)
    roofline = None
    if cnt:
        avg_ms = tot_ms / cnt
        if dom == "msm_rows":
            # the witness commitment: one launch per proof reads V scalars once (SURVEY 8d "commit 32*V"); the window-table gathers
            # (96 B per mixed addition) are not compulsory traffic and are not counted
            bytes_per_launch = F * V
        elif dom == "sc_cubic":
            bytes_per_launch = 384 * N / max(1, (N.bit_length() - 1))          # phase one streams three tables (eq factored out): 384*N over log2(N) launches
        elif dom == "sc_quad":
            bytes_per_launch = 512 * V / max(1, ((2 * V).bit_length() - 1))
        elif dom == "spmv":
            bytes_per_launch = (80 * nnz + 160 * V + 128 * N) / 2.0
        else:
            bytes_per_launch = algorithmic_bytes(N, V, nnz) / max(1, cnt / steps)
        if shard:
            bytes_per_launch /= world                      # each rank's launch covers 1/world of the rows / table
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = None
        try:   # HBM traffic of the dominant kernel's largest launch, from a separate rocprofv3 --pmc pass (profiles/, see its note)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            kname = {"msm_rows": "k_msm_rows<0>", "sc_cubic": "k_sc_cubic3_fold_eval", "sc_quad": "k_sc_quad_fold_eval", "spmv": "k_spmv3_light"}.get(dom)
            if kname and lg == pm.get("log2_constraints", 20) and cbits == pm.get("msm_window_bits", 12):
                traffic = pm["kernels"][kname]["traffic_bytes_corrected"]
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "launches": cnt, "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_bytes_per_launch": int(bytes_per_launch)}
        if conc > 1 and breakdown[dom][0]:
            # the same kernel with the GPU to itself (the instrumented single proof above): under `conc` proofs in flight a launch shares
            # the CUs with other streams' kernels and its wall duration is no longer the kernel's own speed
            u_ms = breakdown[dom][1] / breakdown[dom][0]
            roofline["uncontended"] = {"avg_launch_ms": round(u_ms, 4), "achieved": round(bytes_per_launch / (u_ms * 1e-3) / 1e9, 3),
                                       "frac": round(bytes_per_launch / (u_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)}
        if dom == "msm_rows":
            W = 253 // cbits + 1
            adds = V * W // (world if shard else 1)        # one mixed addition (7 multiplications in GF(2^255-19)) per scalar and window
            rate = adds / (avg_ms * 1e-3)
            roofline["alu"] = {"bound": "integer ALU (v_mad_u64_u32)", "achieved": round(rate / 1e9, 3), "peak": MADD_PEAK_G, "unit": "G mixed additions/s",
                               "frac": round(rate / 1e9 / MADD_PEAK_G, 4),
                               "frac_uncontended": round(adds / (roofline["uncontended"]["avg_launch_ms"] * 1e-3) / 1e9 / MADD_PEAK_G, 4) if "uncontended" in roofline else None,
                               "note": "peak = tools/mulbench.hip p10_madd throughput on this chip, all CUs busy, operands in registers"}
    whole = algorithmic_bytes(N, V, nnz)
    proof_gbps = conc * whole / (ms_per_step * 1e-3) / 1e9

    cpu_baseline = None
    if not args.no_cpu_baseline:
        cores = int(os.environ.get("OTTI_CPU_THREADS", min(usable_cores(), 16)))   # a 1-GPU box's CPU share is 16 cores
        clg = args.cpu_log2 if args.cpu_log2 is not None else lg
        cr = r if (clg == lg and rank == 0) else gen(1 << clg, ni, 1)
        ci, cg = orc.OInstance(cr["num_cons"], cr["num_vars"], cr["num_inputs"], cr["A"], cr["B"], cr["C"]), orc.OGens(cr["num_cons"], cr["num_vars"], cr["num_inputs"])
        orc.set_threads(cores)
        orc.nizk_prove(orc.OInstance(256, 256, ni, *[oa.synth_r1cs(256, ni, 1)[k] for k in "ABC"]), oa.synth_r1cs(256, ni, 1)["vars"],
                       oa.synth_r1cs(256, ni, 1)["inputs"], orc.OGens(256, 256, ni))       # spin up the OpenMP team
        t0 = time.perf_counter()
        cp, cms = orc.nizk_prove(ci, cr["vars"], cr["inputs"], cg, label, seed)
        ct = time.perf_counter() - t0
        same = (cp == proofs[-1].bytes) if clg == lg else None
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
                                  "reference Spartan (Rust) is not buildable here",
                        "proof_equals_gpu_proof": same, "stage_ms": [round(x, 1) for x in cms]}

    out = {
        "metric": "R1CS constraints/sec proved (Spartan NIZK) at 2^%d" % lg, "value": round(value, 1), "unit": "constraints/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "u256 (GF(l) / GF(2^255-19), 8 x u32 limbs)", "data": "synthetic",
        "config": {"workload": (f"synthetic satisfiable R1CS, 2^{lg} constraints = variables, 10 inputs, 1 nnz/row/matrix, uniform GF(l) witness "
                                if args.dist == "uniform" else
                                f"synthetic compiler-like R1CS, 2^{lg} constraints = variables, 10 inputs, 1..8 nnz/row/matrix, 90% of the witness < 2^64, heavy constant column ")
                               + "(SURVEY 8d); witness/instance/generators resident in HBM; one step = "
                               + ("one NIZK::prove" if conc == 1 else "%d independent NIZK::prove calls of that workload in flight on the GPU (one prover thread each)" % conc),
                   "parallelism": (("1 proof sharded over %d GPUs" % world if shard else "1 proof per GPU") if world > 1 else "single GPU")
                                  + ("" if conc == 1 else ", %d proofs in flight per GPU (one prover thread each)" % conc), "proofs_in_flight_per_gpu": conc, "msm_window_bits": cbits, "msm_table_GB": round(table_bytes / 1e9, 2)},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "stage_ms": {k: round(v / max(1, mine_n), 3) for k, v in stage_acc.items()},
        "kernel_ms_per_step": {k: round(v[1], 3) for k, v in breakdown.items() if v[0]},
        "whole_proof_algorithmic_GBps": round(proof_gbps, 2), "whole_proof_hbm_frac": round(proof_gbps / (HBM_PEAK_GBPS * (world if shard else 1)), 6),
        "prepare_device_ms": round(1e3 * t_prepare, 1), "single_proof_ms": round(min(single_ms), 3), "witness_upload_ms": round(1e3 * t_upload, 2), "verify_ms": round(1e3 * t_verify, 2), "proof_bytes": len(proofs[-1].bytes), "proof_sha256": next(iter(digests)),
        "oracle_parity_2^12": parity_ok,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
