#!/usr/bin/env python3
"""Where one accelerated SAP iteration (BASELINE config 5) spends its wall time: every line of
SAP._step bracketed by device synchronisation."""
import os
import sys
import time

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
reg = 1e-2
os.environ["RK_KERNEL_MODE"] = "matrix_free"
A = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={dev})
system = LinSys(A, B, reg=reg, A_row_oracle=A.row_oracle, A_blk_oracle=A.blk_oracle)
s = SAP(system, torch.zeros(n, k, device=dev), NystromConfig(rank=100, rho=reg), dev, n // 100, True,
        SAPAccelConfig(mu=reg, nu=100.0), 10)
for _ in range(3):
    s._step()
acc = {}


def lap(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
    return out


R = 10
for _ in range(R):
    mask = system.mask
    cols = lap("cols", lambda: torch.nonzero(mask)[:, 0].to(s._W.device))
    blk = lap("draw_block", s._get_blk)
    P = lap("block_precond", lambda: s._get_precond(blk))
    eta = lap("stepsize", lambda: s._get_stepsize(blk, P))
    loc = lap("gather_cols", lambda: s.Y[:, cols])
    grad = lap("row_oracle_product", lambda: system.A_row_oracle(blk) @ loc)
    lap("row_oracle_again", lambda: system.A_row_oracle(blk) @ loc)
    lap("row_oracle_on_Y", lambda: system.A_row_oracle(blk) @ s.Y)
    ro = system.A_row_oracle(blk)
    lap("row_oracle_prebuilt", lambda: ro @ loc)
    from rlaopt_amd.ops import get_backend
    A1b = X[blk]
    lap("direct_kernel_matmat", lambda: get_backend().kernel_matmat("rbf", A1b, X, 1.0, loc))
    grad = lap("grad_rest", lambda: grad + reg * loc[blk, :] - system.B[:, cols][blk, :])
    dd = lap("precond_apply", lambda: P._inv @ grad)

    def nesterov():
        Wm = loc.clone()
        Wm[blk] -= eta * dd
        Vm = s.beta * s.V[:, cols] + (1 - s.beta) * loc
        Vm[blk] -= (eta * s.gamma) * dd
        s._W[:, cols] = Wm
        s.V[:, cols] = Vm
        s.Y[:, cols] = s.alpha * Vm + (1 - s.alpha) * Wm
    lap("nesterov_updates", nesterov)
tot = sum(acc.values())
for k_, v in acc.items():
    print(f"{k_:22s} {v / R:8.3f} ms")
print(f"{'sum':22s} {tot / R:8.3f} ms")
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(R):
    s._step()
torch.cuda.synchronize()
print(f"{'_step()':22s} {(time.perf_counter() - t) / R * 1e3:8.3f} ms")

# ---- the real step, host timestamps only (no added synchronisation) ------------------------
import collections
stamps = collections.defaultdict(float)


def real_step(overlap):
    t = [time.perf_counter()]

    def mark(name):
        now = time.perf_counter()
        stamps[(overlap, name)] += (now - t[0]) * 1e3
        t[0] = now
    mask = system.mask
    cols = torch.nonzero(mask)[:, 0].to(s._W.device)
    blk = s._get_blk()
    blk_d = blk.to(s._W.device)
    loc = s.Y[:, cols]
    mark("prep")
    if overlap:
        main = torch.cuda.current_stream(dev)
        side = s._side_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            grad = s._block_gradient(loc, system.B[:, cols], blk_d)
        mark("gradient(enqueue)")
    P = s._get_precond(blk)
    mark("precond")
    eta = s._get_stepsize(blk, P)
    mark("stepsize")
    if overlap:
        main.wait_stream(side)
    else:
        grad = s._block_gradient(loc, system.B[:, cols], blk_d)
        mark("gradient(enqueue)")
    dd = P._inv @ grad
    Wm = loc.clone()
    Wm[blk_d] -= eta * dd
    Vm = s.beta * s.V[:, cols] + (1 - s.beta) * loc
    Vm[blk_d] -= (eta * s.gamma) * dd
    s._W[:, cols] = Wm
    s.V[:, cols] = Vm
    s.Y[:, cols] = s.alpha * Vm + (1 - s.alpha) * Wm
    mark("updates(enqueue)")
    torch.cuda.synchronize()
    mark("drain")


from rlaopt_amd.utils import quiet_host

for ov in (False, True):
    with quiet_host():  # what SAP._step holds (no spinning OpenMP workers next to the syncs)
        for _ in range(R):
            real_step(ov)
    print(f"--- host timeline, overlap={ov}")
    tot = 0.0
    for (o, name), v in stamps.items():
        if o == ov:
            print(f"{name:22s} {v / R:8.3f} ms")
            tot += v / R
    print(f"{'total':22s} {tot:8.3f} ms")
