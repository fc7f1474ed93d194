#!/usr/bin/env python3
"""Host/launch floor of one PCG iteration: a tiny system (n = 4096) makes every kernel trivial, so
the measured time per step is the launch + Python overhead that a many-GPU run cannot go below."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig
from rlaopt_amd.models import LinSys
from rlaopt_amd.preconditioners import NystromConfig
from rlaopt_amd.solvers.pcg import PCG

dev = torch.device("cuda:0")
for n, r in ((4096, 128), (25000, 512)):
    X = torch.randn(n, 64, device=dev) / 8
    B = torch.randn(n, 1, device=dev)
    A = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={dev})
    system = LinSys(A, B, reg=1e-6 * n)
    solver = PCG(system, torch.zeros(n, 1, device=dev), NystromConfig(rank=r, rho=1e-6 * n, sketch="gauss"), dev)
    system._active_solver = solver
    for _ in range(5):
        solver._step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        solver._step()
    t_host = time.perf_counter() - t0  # time to ENQUEUE 200 steps
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"n={n} r={r}: enqueue {t_host/200*1e3:.3f} ms/step, complete {t_all/200*1e3:.3f} ms/step", flush=True)
