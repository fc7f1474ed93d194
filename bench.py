#!/usr/bin/env python3
"""Headline benchmark: PCG iterations/s (and kernel-matvec GB/s) on the n = 200 000 RBF KRR
system of BASELINE.json, row-sharded over N MI355X GPUs.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload: X = randn(n, 64)/sqrt(64) (seed 0, drawn on the CPU in
float64, cast to float32), RBF lengthscale 1, B = randn(n, 1), reg = 1e-6 n, Nystrom rank 512
with an injected Gaussian test matrix (seed 1), adaptive damping, float32.  One "step" is one
iteration of the LinSys loop: a block-PCG step (kernel shard product with the "+ reg P" term,
Nystrom inverse apply, Gram blocks, k x k solves, fused updates, collectives) plus - every
``callback_freq`` = 10 iterations, like the reference's loop - the true-residual evaluation.
Setup (materialising the kernel shard, Nystrom assembly) is outside the timed region and is
reported separately; inputs are resident in HBM when the timed region starts.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
``roofline`` for the dominant kernel - the HBM-streaming kernel product: for symmetric data
(A1 is A2, the KRR case) ``rk_symm_matmat`` over the strip-packed upper triangle, algorithmic
bytes = packed triangle + V + Y; otherwise ``rk_matmat_skinny`` over the row-major shard,
(n_loc n + n k + n_loc k) * 4 bytes - duration from HIP events on the launch stream inside the
timed region (the row-major kernel is additionally timed on a 1/8 shard: ``rowmajor_shard_product``) and ``cpu_baseline`` (the oracle's C restatement of the
matrix-free kernel product timed on the host cores over a bounded row sample, rank 0, N = 1).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver only supports dmabuf IPC (RCCL across processes fails without this)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 is the measured copy ceiling


def make_problem(n, d, r, dt, seed_x=0, seed_om=1):
    g = torch.Generator().manual_seed(seed_x)
    X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).to(dt)
    B = torch.randn(n, 1, dtype=torch.float64, generator=g).to(dt)
    g = torch.Generator().manual_seed(seed_om)
    Om = torch.empty(n, r, dtype=dt)
    for c0 in range(0, r, 64):  # randn(r, n)/sqrt(r) transposed, drawn in slabs to bound memory
        Om[:, c0 : c0 + 64] = (torch.randn(min(64, r - c0), n, dtype=torch.float64, generator=g) / r**0.5).T.to(dt)
    # on the float16 grid, as the package's own Gaussian sketch is drawn (rlaopt_amd/sketches: the sketch product then splits
    # Omega into one float16 piece); on_grid() re-attaches the tag after a copy to the device
    return X, B, Om.to(torch.float16).to(dt)


def on_grid(Om):
    from rlaopt_amd.ops import f16_grid

    return f16_grid(Om)


def es_of(dt):
    return 4 if dt == torch.float32 else 8


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(X, n, d, r, callback_freq, dev, budget_s=30.0):
    """CPU leg, two parts, both on the oracle's C/OpenMP matrix-free RBF product (the reference's CPU
    path is pykeops' matrix-free CPU backend):

    1. ``value``: the product timed on a growing row sample of the SAME n x n system, converted to PCG
       iterations/s of that system (the product is > 99 % of a CPU iteration: SURVEY 8a, row a11) -
       an extrapolation, labelled as such;
    2. ``solve``: an ACTUAL solve - Nystrom assembly + block PCG + LinSys loop of the oracle
       (oracle/krr.py), same recipe (seeds, rank, reg = 1e-6 n, rtol = 1e-6, callback_freq), same dtype,
       at the largest n of the ladder whose estimated cost fits ``budget_s`` of host work - with its
       iterations, final relative residual and wall time; ``gpu_same_system`` is the product's solve of
       that very system (same Omega) on the GPU, timed the same way."""
    from oracle import cbind, krr

    # threads = the CPUs this process is entitled to (affinity capped by the cgroup quota: a one-GPU box is a
    # 16-CPU share of a 256-thread host, and threads beyond the quota are only throttled)
    cores = min(cbind.host_cpus(), cbind.num_threads())
    cbind.set_num_threads(cores)
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    dt = X.dtype
    V = torch.ones(n, 1, dtype=dt)
    rows, spent = 256, 0.0
    cbind.kernel_matmat("rbf", X[:64], X, 1.0, V)  # warm the threads
    while spent < 4.0 and rows <= n:
        t = time.time()
        cbind.kernel_matmat("rbf", X[:rows], X, 1.0, V)
        el = time.time() - t
        spent += el
        last = (rows, el)
        rows *= 2
    pair_rate = last[0] * n / last[1]
    sec_per_iter = n * n / pair_rate * 1.1  # + the residual product every 10th iteration
    out = {
        "value": 1.0 / sec_per_iter,
        "unit": "iter/s",
        "cores": cores,
        "cpu_model": _cpu_model(),
        "kind": "port",
        "sample": f"EXTRAPOLATED: oracle/rk_oracle.c matrix-free RBF product, rows 0..{last[0]} of the n={n} "
                  f"d={d} system in {last[1]:.2f}s ({pair_rate / 1e9:.3f} Gpair/s on {cores} threads) -> one "
                  f"full product x 1.1 per iteration; the measured solve is under 'solve'",
        "gpair_per_s": pair_rate / 1e9,
    }
    # ---- the measured solve ---------------------------------------------------------------------------------
    # The float32 kernel matrix is written once by the C closed forms and every product streams it (C/OpenMP rows;
    # MKL gemm for the sketch): the FASTER of the two host paths (the matrix-free product the reference's pykeops CPU backend takes
    # would spend n^2 (d + 24) flops per iteration instead of streaming n^2 words).  A first rung (n = 6 000) is
    # timed; the solve reported is the largest rung of the ladder whose cost, scaled by n^2 from the first, fits
    # the budget (n = 50 000 is BASELINE.json's configs[1]).
    max_iters = 200

    def host_solve(nc):
        r_c = min(r, nc // 40)
        Xc, Bc, Omc = make_problem(nc, d, r_c, dt)
        t0 = time.time()
        ref = krr.solve(Xc, Bc, Omc, 1e-6 * nc, rtol=1e-6, atol=0.0, max_iters=max_iters,
                        callback_freq=callback_freq, dense=True)
        return nc, r_c, Xc, Bc, Omc, ref, time.time() - t0

    run = host_solve(min(6000, n))
    for cand in (50000, 30000, 20000, 12000):
        if cand <= n and cand * cand * es_of(dt) <= 16e9 and run[-1] * (cand / run[0]) ** 2 <= budget_s:
            run = host_solve(cand)
            break
    n_cpu, r_c, Xc, Bc, Omc, ref, wall = run
    reg = 1e-6 * n_cpu
    out["solve"] = {
        "workload": f"rbf_krr_pcg n={n_cpu} d={d} k=1 nystrom_rank={r_c} reg=1e-6*n rtol=1e-6 {str(dt)[6:]}",
        "host_path": "dense K written once (oracle/rk_oracle.c closed forms, OpenMP); products by the C row-streaming "
                     "kernel (k <= 16) / MKL gemm (sketch); oracle Nystrom / block PCG / LinSys loop (oracle/krr.py)",
        "iters": int(ref["iters"]), "rel_res": float(ref["rel_res"][0]), "wall_s": round(wall, 3),
        "nystrom_build_s": round(ref["build_s"], 3), "loop_s": round(ref["loop_s"], 3),
        "iters_per_s": ref["iters"] / ref["loop_s"], "products": ref["products"], "cores": cores,
    }
    # the same system through the product, same Omega
    from rlaopt_amd import sketches
    from rlaopt_amd.kernels import KernelConfig, RBFLinOp
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.solvers import PCGConfig

    Xg = Xc.to(dev)
    torch.cuda.synchronize()
    t0 = time.time()
    A = RBFLinOp(Xg, Xg, KernelConfig(lengthscale=1.0))
    cfg = PCGConfig(device=dev, max_iters=max_iters, atol=0.0, rtol=1e-6,
                    precond_config=NystromConfig(rank=r_c, rho=reg, sketch="gauss"))
    with sketches.inject(on_grid(Omc.to(dev))):
        Wg, log = LinSys(A, Bc.to(dev), reg=reg).solve(cfg, torch.zeros(n_cpu, 1, dtype=dt, device=dev),
                                                       callback_freq=callback_freq)
    torch.cuda.synchronize()
    gwall = time.time() - t0
    torch.set_num_threads(torch_threads)
    last_it = max(log)
    out["gpu_same_system"] = {
        "iters": int(last_it), "rel_res": float(log[last_it]["metrics"]["internal_metrics"]["rel_res"][0]),
        "wall_s": round(gwall, 3), "storage": A.storage_mode,
        "speedup_wall": wall / gwall,
        "first_logged_rel_res": [float(log[i]["metrics"]["internal_metrics"]["rel_res"][0]) for i in sorted(log)[:4]],
        "cpu_first_logged_rel_res": [float(ref["log"][i]["rel_res"][0]) for i in sorted(ref["log"])[:4]],
    }
    # the measured solve, flat (the driver's parse keeps top-level keys only)
    out.update({
        "measured_solve_n": n_cpu, "measured_solve_iters": out["solve"]["iters"],
        "measured_solve_iters_per_s": out["solve"]["iters_per_s"], "measured_solve_rel_res": out["solve"]["rel_res"],
        "measured_solve_wall_s": out["solve"]["wall_s"], "gpu_same_system_wall_s": out["gpu_same_system"]["wall_s"],
        "gpu_same_system_rel_res": out["gpu_same_system"]["rel_res"],
        "speedup_wall_same_system": out["gpu_same_system"]["speedup_wall"],
    })
    return out


def committed_traffic(world, n, mode):
    """HBM bytes per shard-product launch from the committed PMC summary (profiles/*_traffic.json,
    written by tools/summarize_profile.py from separate rocprofv3 --pmc passes) when one exists
    for this exact workload; None otherwise."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("n_gpus") == world and f"n={n} " in rec.get("workload", "") and rec.get("storage") == mode:
            best = rec
    return None if best is None else best["hbm_bytes_per_launch"]



def _device_info(dev):
    """Which card this was (the streaming rate differs by up to 5 % from card to card)."""
    try:
        p = torch.cuda.get_device_properties(dev)
        return {"name": p.name, "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count,
                "clock_mhz": round(getattr(p, "clock_rate", 0) / 1e3) or None,
                "memory_clock_mhz": round(getattr(p, "memory_clock_rate", 0) / 1e3) or None,
                "memory_GB": round(p.total_memory / 1e9, 1)}
    except Exception as e:  # noqa: BLE001 - informational only
        return {"error": str(e)}


def run_schedule(step, log, warmup, steps, freq, start_timer, stop_timer):
    """The bench loop.  ``step`` = one solver step, ``log`` = what a logging iteration of LinSys.solve adds
    (metric, termination test with its host read).  Iteration i logs when i % freq == 0, as the reference's
    loop does (models/model.py:89-114).

    Before anything is timed, every code path of the loop is executed at least twice, whatever the alignment
    of ``warmup`` and ``freq``: a plain step, a logging iteration, and the step that FOLLOWS a logging
    iteration (it consumes the product the metric pass computed ahead and takes the separate Gram launch).
    Round 3's driver line (--warmup 5, callback_freq 10) met its first logging iteration inside the timed
    region and charged 81 ms of one-time code-object loads to the metric."""
    for _ in range(2):
        step()
        log()
        step()
    for i in range(1, warmup + 1):
        step()
        if i % freq == 0:
            log()
    start_timer()
    for i in range(warmup + 1, warmup + steps + 1):
        step()
        if i % freq == 0:
            log()
    return stop_timer()


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh ranks with torch.distributed.run as a
    CHILD process (this process has not touched the GPU and never will), pass rank 0's JSON line through,
    return the child's exit code.  Never exec: a process image swap after GPU initialisation takes the node
    down on this pool, and a child keeps the rule trivially true."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    env["RK_BENCH_N"] = str(args.n)  # (torch.distributed.run's parser claims --n* prefixes)
    passed, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == "--n":
            skip = True
            continue
        if a.startswith("--n="):
            continue
        passed.append(a)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + passed
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n", type=int, default=int(os.environ.get("RK_BENCH_N", "200000")),
                    help="rows of the system (RK_BENCH_N sets it under torchrun, whose parser rejects --n)")
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--rank", type=int, default=512)
    ap.add_argument("--callback-freq", type=int, default=10)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32",
                    help="f32 is the BASELINE.json configuration; f64 is the secondary run of SURVEY 8(d)")
    ap.add_argument("--storage", choices=["same", "f32"], default="same",
                    help="with --dtype f64: keep the packed kernel matrix in float32 (RK_STORAGE_DTYPE)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-solve", action="store_true", help="skip the full solve to tolerance")
    args = ap.parse_args()

    if args.storage == "f32":
        os.environ["RK_STORAGE_DTYPE"] = "float32"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one (children only; nothing here has initialised the GPU)
        sys.exit(self_launch(args, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if os.environ.get("RK_BENCH_ONE_GPU", "0") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # RK_BENCH_BACKEND=gloo + RK_BENCH_ONE_GPU=1 rehearse the multi-rank path on a one-GPU box
        backend = os.environ.get("RK_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            td.init_process_group("nccl", device_id=dev)
        else:
            td.init_process_group(backend)

    from rlaopt_amd import sketches
    from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.ops import get_backend
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.solvers import PCGConfig
    from rlaopt_amd.solvers.pcg import PCG

    be = get_backend()
    n, d, r, k = args.n, args.d, args.rank, 1
    reg = 1e-6 * n
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    es = 4 if args.dtype == "f32" else 8
    X, B, Om = make_problem(n, d, r, dt)
    Xd, Bd, Omd = X.to(dev), B.to(dev), on_grid(Om.to(dev))

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()

    # ---- setup (untimed): materialise the shard, build the preconditioner -------------------
    sync()
    t0 = time.time()
    A = DistributedRBFLinOp(Xd, Xd, KernelConfig(lengthscale=1.0), devices={dev})
    system = LinSys(A, Bd, reg=reg)
    lo, hi = A.row_range if A.is_row_sharded else (0, n)
    _ = A.local_matmat(torch.zeros(n, 1, dtype=dt, device=dev))  # forces materialisation
    sync()
    t_mat = time.time() - t0
    t0 = time.time()
    pcfg = NystromConfig(rank=r, rho=reg, sketch="gauss", damping_mode="adaptive")
    with sketches.inject(Omd):
        solver = PCG(system, torch.zeros(n, k, dtype=dt, device=dev), pcfg, dev)
    system._active_solver = solver  # what LinSys.solve does: metric and next step share a pass
    sync()
    t_pre = time.time() - t0
    # the same assembly again (discarded): the first one pays this process's one-time costs - code-object loads
    # of every kernel on its path, allocator growth; a long-lived process pays them once
    from rlaopt_amd.preconditioners.nystrom import Nystrom as _Nystrom

    t0 = time.time()
    with sketches.inject(Omd):
        _p2 = _Nystrom(NystromConfig(rank=r, rho=reg, sketch="gauss", damping_mode="adaptive"))
        _p2._update(A, dev)
        _p2._update_damping(baseline_rho=reg)
    sync()
    t_pre_warm = time.time() - t0
    del Omd, _p2
    # the same assembly with the column-sampling sketch (Y = K[:, J] written from the closed form: no n x n x r
    # product at all); discarded - the timed iterations use the Gaussian preconditioner above
    t_cs = []
    for _ in range(2):
        sync()
        t0 = time.time()
        _p3 = _Nystrom(NystromConfig(rank=r, rho=reg, sketch="colsample", damping_mode="adaptive"))
        _p3._update(A, dev)
        _p3._update_damping(baseline_rho=reg)
        sync()
        t_cs.append(time.time() - t0)
    del _p3

    def one_step():
        solver._step()

    def one_log():
        # what a logged iteration of LinSys.solve adds: the metric and the termination test with its one
        # host read (rtol = 0: never satisfied, the loop keeps its k = 1 active column)
        m = system._compute_internal_metrics(solver.W)
        system._check_termination_criteria(m, 0.0, 0.0)

    # ---- warmup, then exactly K timed steps ---------------------------------------------------
    from rlaopt_amd.utils import quiet_host

    loop_guard = quiet_host()  # what LinSys.solve holds around its loop (small CPU thread pool)
    loop_guard.__enter__()
    be.enable_timing(True)  # also during warmup: the first HIP event pair costs milliseconds
    import gc

    from rlaopt_amd import dist as _rdist

    clock = {}

    def start_timer():
        be.enable_timing(True)  # fresh table for the timed region
        gc.collect()
        gc.disable()  # no collector pauses on the host inside the timed region
        _rdist.reset_wire_stats()
        sync()
        clock["t0"] = time.perf_counter()

    def stop_timer():
        sync()
        return time.perf_counter() - clock["t0"]

    elapsed = run_schedule(one_step, one_log, args.warmup, args.steps, args.callback_freq, start_timer, stop_timer)
    wire = {"collectives_per_step": _rdist.wire_stats["collectives"] / args.steps,
            "bytes_sent_per_rank_per_step": _rdist.wire_stats["bytes_sent_per_rank"] / args.steps,
            "by_kind": dict(_rdist.wire_stats["by_kind"])}
    gc.enable()
    loop_guard.__exit__(None, None, None)
    timing = be.timing_summary()
    be.enable_timing(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        elapsed = float(tmax)
    rel_now = float(system._compute_internal_metrics(solver.W)["rel_res"][0])

    mode = A.storage_mode
    # which kernel the timed products ran (decided here: the operator is released before the line is printed)
    blk_kernel = bool(getattr(A, "_sym", None) is not None and os.environ.get("RK_SYMM_BLKREG", "1") != "0")
    mv = timing.get("kernel_shard_product", {"avg_ms": float("nan"), "bytes_per_launch": 0, "launches": 0})
    achieved = mv["bytes_per_launch"] / (mv["avg_ms"] * 1e-3) / 1e9 if mv["launches"] else float("nan")
    if world > 1:  # roofline is per GPU: report the slowest rank's kernel
        amin = torch.tensor([achieved], dtype=torch.float64, device=dev)
        td.all_reduce(amin, op=td.ReduceOp.MIN)
        achieved = float(amin)

    # ---- the row-major streaming kernel on a 1/8 row shard (SURVEY 8d formula), for reference ------
    rowmajor = None
    if rank == 0:
        rows8 = n // 8
        Ksh = be.kernel_materialize("rbf", Xd[:rows8], Xd, 1.0)
        vv = torch.randn(n, 1, dtype=dt, device=dev)
        be.matmat(Ksh, vv)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            be.matmat(Ksh, vv, diag_shift=reg, row0=0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nbytes = (rows8 * n + n + rows8) * es
        rowmajor = {"kernel": "rk_matmat_skinny (skinny_nn_kernel<T,1,8>)", "shard": f"{rows8} x {n} {args.dtype} (the p = 8 shard)",
                    "avg_launch_ms": ms, "achieved_GBps": nbytes / (ms * 1e-3) / 1e9,
                    "frac_of_8000": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        del Ksh, vv
    # ---- the storage-free alternative: the same product recomputed on the bf16 matrix cores ---------
    matrix_free = None
    if rank == 0 and world == 1 and args.dtype == "f32":
        gcen = be.gram_center(Xd, Xd, 1.0)
        vv = torch.randn(n, 1, dtype=dt, device=dev)
        be.kernel_matmat("rbf", Xd, Xd, 1.0, vv, gram_center=gcen)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            be.kernel_matmat("rbf", Xd, Xd, 1.0, vv, gram_center=gcen)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        matrix_free = {"kernel": "rk_kernel_matmat (kmat_matmat_bf16x3_kernel, symmetric evaluation): K(X, X) v "
                                 "recomputed from X, two float16 pieces per coordinate on the 16-bit matrix cores "
                                 "(three exact products), float32 accuracy",
                       "avg_launch_ms": ms, "Gpair_per_s": n * n / ms / 1e6,
                       "norm_bound_ok": gcen is not None}
        del vv
    if world > 1:
        td.barrier()

    # ---- full solve to the reference tolerance (reported, not the headline) ---------------------
    solve = {}
    if not args.no_solve:
        system2 = LinSys(A, Bd, reg=reg)
        cfg = PCGConfig(device=dev, max_iters=200, atol=0.0, rtol=1e-6,
                        precond_config=NystromConfig(rank=r, rho=reg, sketch="gauss"))
        Omd = on_grid(Om.to(dev))
        sync()
        t0 = time.time()
        with sketches.inject(Omd):
            W, log = system2.solve(cfg, torch.zeros(n, k, dtype=dt, device=dev), callback_freq=args.callback_freq)
        sync()
        solve = {
            "solve_iters": max(log),
            "solve_s": round(time.time() - t0, 3),
            "solve_rel_res": float(log[max(log)]["metrics"]["internal_metrics"]["rel_res"][0]),
        }
        del Omd

    # ---- the converging configuration of the same system: float64 iteration over float32 storage ---------
    # (the float32 solve above stalls at eps32 x condition number and runs to max_iters; with the packed
    # triangle still in float32 - same HBM traffic - and every vector and accumulation in float64 it meets
    # rtol = 1e-6.  Same X (the float32 values), same B, same Omega.)
    converging = None
    if not args.no_solve and args.dtype == "f32" and mode.startswith("sympack"):
        del solver, system
        if "system2" in dir():
            del system2
        A.shutdown()
        del A
        torch.cuda.empty_cache()
        os.environ["RK_STORAGE_DTYPE"] = "float32"
        try:
            sync()
            t0 = time.time()
            X64 = Xd.double()
            A64 = DistributedRBFLinOp(X64, X64, KernelConfig(lengthscale=1.0), devices={dev})
            sys64 = LinSys(A64, Bd.double(), reg=reg)
            cfg = PCGConfig(device=dev, max_iters=200, atol=0.0, rtol=1e-6,
                            precond_config=NystromConfig(rank=r, rho=reg, sketch="gauss"))
            Om64 = on_grid(Om.to(dev).double())
            with sketches.inject(Om64):
                W64, log64 = sys64.solve(cfg, torch.zeros(n, k, dtype=torch.float64, device=dev),
                                         callback_freq=args.callback_freq)
            sync()
            t_all = time.time() - t0
            last = max(log64)
            # iterations/s of the loop alone: re-time a few steps on the converged state's solver
            converging = {
                "config": "float64 vectors / accumulation, packed triangle stored in float32 (RK_STORAGE_DTYPE=float32)",
                "iters": int(last),
                "rel_res": float(log64[last]["metrics"]["internal_metrics"]["rel_res"][0]),
                "history": [float(log64[i]["metrics"]["internal_metrics"]["rel_res"][0]) for i in sorted(log64)],
                "wall_s_with_setup": round(t_all, 3),
                "logger_cum_time_s": round(log64[last]["cum_time"], 3) if "cum_time" in log64[last] else None,
                # iterations/s of the loop itself: the logger's device-synchronised clock between the first and the
                # last logged iteration (it excludes the residual evaluations, like the reference's logger)
                "iters_per_s": (last / (log64[last]["cum_time"] - log64[0]["cum_time"])
                                if last > 0 and log64[last]["cum_time"] > log64[0]["cum_time"] else None),
                "storage": A64.storage_mode,
            }
            del Om64, W64, sys64
            A64.shutdown()
            del A64, X64
            torch.cuda.empty_cache()
        finally:
            del os.environ["RK_STORAGE_DTYPE"]

    if rank == 0:
        out = {
            "metric": "pcg_iters_per_sec",
            "value": args.steps / elapsed,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            # everything of a step that is not the operator product (launch gaps, preconditioner apply, fused
            # updates, small solves, the logging iterations' metric and host read): ms_per_step minus the
            # HIP-event time of every product launch of the timed region / steps
            "nonproduct_ms_per_step": (1e3 * elapsed - mv["launches"] * mv["avg_ms"]) / args.steps if mv["launches"] else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"rbf_krr_pcg n={n} d={d} k={k} nystrom_rank={r} reg=1e-6*n "
                            f"(BASELINE.json configs[2]; one n x n system row-sharded over {world} GPU(s))",
                "callback_freq": args.callback_freq,
                "rows_per_gpu": hi - lo,
                "parallelism": f"row-shard x{world}",
            },
            "roofline": {
                "kernel": (("rk_symm_matmat (symm_blk_reg_kernel + finish): strip-packed upper triangle, every stored entry "
                            "used for y[i] and y[j]; work items of four strips side by side x up to 1024 rows claimed from a "
                            "device counter, a row's partial sum carried across the strips (vector units, exact "
                            + ("float32" if args.dtype == "f32" else
                               ("float64" if args.storage == "same" else "float64 over float32 storage")) + " arithmetic); "
                            if blk_kernel else
                            "rk_symm_matmat (symm_strip_kernel + finish): strip-packed upper triangle, every stored entry "
                            "used for y[i] and y[j]; ") + "algorithmic bytes = packed triangle + V + Y"
                           if mode.startswith("sympack") else
                           "rk_matmat_skinny (skinny_nn_kernel<float,1,8>): row-major shard; algorithmic "
                           "bytes = (n_loc n + n k + n_loc k) * 4"),
                "storage": mode,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": committed_traffic(world, n, mode) if args.dtype == "f32" else None,
                "bytes_per_launch": mv["bytes_per_launch"],
                "avg_launch_ms": mv["avg_ms"],
                "launches": mv["launches"],
            },
            "matvec_GBps_aggregate": achieved * world,
            "matvec_dense_equivalent_GBps_aggregate": (n * n * float(es) / (mv["avg_ms"] * 1e-3) / 1e9) if mv["launches"] else None,
            "rowmajor_shard_product": rowmajor,
            "matrix_free_product": matrix_free,
            "device": _device_info(dev),
            "setup_materialize_s": round(t_mat, 3),
            "setup_nystrom_s": round(t_pre, 3),
            "setup_nystrom_warm_s": round(t_pre_warm, 3),
            "setup_nystrom_colsample_s": round(t_cs[0], 3),
            "setup_nystrom_colsample_warm_s": round(t_cs[1], 3),
            "process_group": {"world_size_seen": td.get_world_size() if world > 1 else 1,
                              "backend": td.get_backend() if world > 1 else None, **wire},
            "rel_res_after_timed_steps": rel_now,
            "launch_timing": {op: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4)} for op, v in timing.items()},
            **solve,
            "converging_solve": converging,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, n, d, r, args.callback_freq, dev)  # same dtype as the GPU run
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
