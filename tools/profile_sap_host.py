#!/usr/bin/env python3
"""Host-side (Python) profile of SAP steps at the BASELINE config-5 shape: where the interpreter spends its
time between kernel launches (cProfile, sorted by own time)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig
from rlaopt_amd.models import LinSys
from rlaopt_amd.preconditioners import NystromConfig
from rlaopt_amd.solvers import SAPAccelConfig
from rlaopt_amd.solvers.sap import SAP

dev = torch.device("cuda:0")
n, d, k = 500_000, 50, 10
g = torch.Generator().manual_seed(0)
X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).float().to(dev)
B = torch.randn(n, k, dtype=torch.float64, generator=g).float().to(dev)
os.environ["RK_KERNEL_MODE"] = "matrix_free"
A = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={dev})
system = LinSys(A, B, reg=1e-2, A_row_oracle=A.row_oracle, A_blk_oracle=A.blk_oracle)
s = SAP(system, torch.zeros(n, k, device=dev), NystromConfig(rank=100, rho=1e-2), dev, n // 100, True,
        SAPAccelConfig(mu=1e-2, nu=100.0), 10)
for _ in range(5):
    s._step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    s._step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
