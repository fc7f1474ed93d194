#!/usr/bin/env python3
"""BASELINE.json config 5 (ASkotch): RBF KRR n = 500 000, d = 50, k = 10 right-hand sides,
block size n/100, Nystrom rank 100, float32, accelerated SAP, on one MI355X.  Prints ms per
SAP iteration (and where the time goes) plus the residual trend."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig
from rlaopt_amd.models import LinSys
from rlaopt_amd.preconditioners import NystromConfig
from rlaopt_amd.solvers import SAPAccelConfig, SAPConfig
from rlaopt_amd.solvers.sap import SAP


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500_000)
    ap.add_argument("--d", type=int, default=50)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--rank", type=int, default=100)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--solve", action="store_true",
                    help="the configuration end to end: LinSys.solve for max_iters = 300, callback_freq = 100 "
                         "(experiments/distributed_krr_linsys_askotch_solve_test.py:45-75) - residual trajectory")
    ap.add_argument("--max-iters", type=int, default=300)
    ap.add_argument("--callback-freq", type=int, default=100)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, d, k = args.n, args.d, args.k
    g = torch.Generator().manual_seed(0)
    X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).float().to(dev)
    B = torch.randn(n, k, dtype=torch.float64, generator=g).float().to(dev)
    reg = 1e-2
    os.environ["RK_KERNEL_MODE"] = "matrix_free"  # SAP only touches the oracles
    A = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={dev})  # matrix-free: no n x n block
    system = LinSys(A, B, reg=reg, A_row_oracle=A.row_oracle, A_blk_oracle=A.blk_oracle)
    cfg = SAPConfig(device=dev, max_iters=300, atol=1e-6, rtol=1e-6,
                    precond_config=NystromConfig(rank=args.rank, rho=reg), blk_sz=n // 100,
                    accel_config=SAPAccelConfig(mu=reg, nu=100.0))
    if args.solve:
        cfg.max_iters = args.max_iters
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        W, log = system.solve(cfg, torch.zeros(n, k, device=dev), callback_freq=args.callback_freq)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        its = sorted(log)
        traj = {i: [round(float(v), 5) for v in log[i]["metrics"]["internal_metrics"]["rel_res"]] for i in its}
        solver_s = log[its[-1]]["cum_time"]
        print(json.dumps({"workload": f"askotch solve n={n} d={d} k={k} blk={cfg.blk_sz} rank={args.rank} f32 "
                                      f"mu={reg} nu=100 max_iters={args.max_iters} callback_freq={args.callback_freq}",
                          "iters": its[-1], "wall_s": round(wall, 3), "solver_s_excluding_metrics": round(solver_s, 3),
                          "ms_per_iter": round(1e3 * solver_s / max(its[-1], 1), 3),
                          "rel_res_by_iteration": traj}))
        return
    torch.manual_seed(0)
    solver = SAP(system, torch.zeros(n, k, device=dev), cfg.precond_config, dev, cfg.blk_sz, True,
                 cfg.accel_config, cfg.power_iters)
    for _ in range(3):
        solver._step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver._step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    # pieces
    blk = solver._get_blk()

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    Wk = solver.Y
    pieces = {
        "draw_block_ms": timed(solver._get_blk),
        "block_precond_ms": timed(lambda: solver._get_precond(blk)),
        "row_oracle_product_ms": timed(lambda: system.A_row_oracle(blk) @ Wk),
    }
    P = solver._get_precond(blk)
    pieces["stepsize_power_iteration_ms"] = timed(lambda: solver._get_stepsize(blk, P))
    pair_rate = cfg.blk_sz * n / (pieces["row_oracle_product_ms"] * 1e-3) / 1e9
    res = system._compute_internal_metrics(solver.W)["rel_res"]
    print(json.dumps({"workload": f"askotch n={n} d={d} k={k} blk={cfg.blk_sz} rank={args.rank} f32",
                      "ms_per_iter": round(ms, 2), "pieces": {a: round(b, 2) for a, b in pieces.items()},
                      "row_oracle_Gpair_per_s": round(pair_rate, 1),
                      "rel_res_after": [round(float(v), 4) for v in res]}))


if __name__ == "__main__":
    main()
