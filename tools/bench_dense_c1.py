#!/usr/bin/env python3
"""BASELINE config 1 on the GPU: dense SPD 4096 x 4096 float64 system (eigenvalues i^-2), Nystrom rank 256
(gauss, adaptive damping), PCG to rtol = atol = 1e-6, callback_freq 10 - the reference's own CPU-runnable
case (36 iterations, 0.36 s on 8 CPU threads with the reference itself)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.models import LinSys
from rlaopt_amd.preconditioners import NystromConfig
from rlaopt_amd.solvers import PCGConfig

dev = torch.device("cuda:0")
n, r, reg = 4096, 256, 1e-6
torch.manual_seed(0)
Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64))
A = (Q * torch.arange(1, n + 1, dtype=torch.float64) ** -2.0) @ Q.T
A = (0.5 * (A + A.T)).to(dev)
B = (A @ torch.ones(n, 1, dtype=torch.float64, device=dev) + reg * torch.randn(n, 1, dtype=torch.float64).to(dev))
cfg = PCGConfig(device=dev, max_iters=500, atol=1e-6, rtol=1e-6,
                precond_config=NystromConfig(rank=r, rho=reg, sketch="gauss"))
out = {}
for rep in range(3):
    torch.manual_seed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    W, log = LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=10)
    torch.cuda.synchronize()
    out = {"solve_s": round(time.perf_counter() - t0, 4), "iters": max(log),
           "rel_res": float(log[max(log)]["metrics"]["internal_metrics"]["rel_res"].max())}
# per-iteration cost without the solve's setup / logging
from rlaopt_amd.solvers.pcg import PCG  # noqa: E402

s = PCG(LinSys(A, B, reg=reg), torch.zeros_like(B), cfg.precond_config, dev)
for _ in range(5):
    s._step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    s._step()
torch.cuda.synchronize()
out["ms_per_iter"] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
out["workload"] = "dense spd 4096 f64, nystrom rank 256 gauss adaptive, pcg 1e-6 (BASELINE config 1)"
print(json.dumps(out))
