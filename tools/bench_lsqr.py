#!/usr/bin/env python3
"""BASELINE.json config 4: tall-skinny least squares 2M x 512, float64, sketch-and-precondition
(SkPre, sparse-sign sketch s = 2048) + LSQR on one MI355X.  Prints one JSON line with
iterations, time per iteration, achieved GB/s of the two streaming passes over A and a CPU
figure (torch-CPU passes over a row sample of A on the host cores)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.models import LeastSquares
from rlaopt_amd.ops import get_backend
from rlaopt_amd.preconditioners import IdentityConfig, SkPreConfig
from rlaopt_amd.solvers import LSQRConfig
from rlaopt_amd.utils import quiet_host
from rlaopt_amd.solvers.lsqr import LSQR


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2_000_000)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--sketch-size", type=int, default=2048)
    ap.add_argument("--sketch", default="sparse")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    be = get_backend()
    m, d = args.m, args.d
    torch.manual_seed(0)
    A = be.randn(m, d, torch.float64, dev, seed=0) * torch.logspace(0, -4, d, dtype=torch.float64, device=dev)
    b = be.matmat(A, torch.ones(d, 1, dtype=torch.float64, device=dev)) + 1e-3 * be.randn(m, 1, torch.float64, dev, seed=1)
    torch.cuda.synchronize()
    out = {"workload": f"lsq m={m} d={d} f64 skpre({args.sketch}, s={args.sketch_size}) + lsqr"}
    for name, pcfg in (("skpre", SkPreConfig(sketch_size=args.sketch_size, rho=1e-12, sketch=args.sketch)),
                       ("none", IdentityConfig())):
        system = LeastSquares(A, b, reg=0.0)
        t0 = time.time()
        solver = LSQR(system, torch.zeros(d, 1, dtype=torch.float64, device=dev), pcfg, dev)
        torch.cuda.synchronize()
        t_setup = time.time() - t0
        with quiet_host():  # what LeastSquares.solve holds around its loop
            for _ in range(3):
                solver._step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                solver._step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
        passes = 1 if solver._one_pass else 2
        out[name] = {"setup_s": round(t_setup, 3), "ms_per_iter": round(dt * 1e3, 3),
                     "passes_over_A_per_iter": passes,
                     "stream_GBps": round(passes * m * d * 8 / dt / 1e9, 1)}
        system = LeastSquares(A, b, reg=0.0)
        cfg = LSQRConfig(device=dev, max_iters=300 if name == "skpre" else 60, atol=0.0, rtol=1e-10, precond_config=pcfg)
        t0 = time.time()
        W, log = system.solve(cfg, torch.zeros(d, 1, dtype=torch.float64, device=dev), callback_freq=5)
        torch.cuda.synchronize()
        last = log[max(log)]["metrics"]["internal_metrics"]
        out[name].update({"solve_iters": max(log), "solve_s": round(time.time() - t0, 3),
                          "rel_res": float(last["rel_res"][0]),
                          "err_vs_ones": float((W[:, 0] - 1).abs().max())})
    # CPU: two passes over a row sample on the host cores
    rows = min(m, 200_000)
    Ac = A[:rows].cpu()
    x, u = torch.ones(d, 1, dtype=torch.float64), torch.ones(rows, 1, dtype=torch.float64)
    t0 = time.time()
    for _ in range(3):
        Ac @ x
        Ac.T @ u
    tc = (time.time() - t0) / 3 * (m / rows)
    out["cpu"] = {"ms_per_iter": round(tc * 1e3, 1), "threads": torch.get_num_threads(),
                  "sample": f"torch-CPU A x and A^T u on rows 0..{rows}, scaled to m"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
