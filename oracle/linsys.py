"""Oracle: the LinSys driver loop.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Restates rlaopt/models/linsys.py:96-107 (metrics and termination), rlaopt/models/
model.py:89-114 (loop) and rlaopt/utils/logger.py:32-34 (logging cadence): metrics are
evaluated at iteration 0 and at every multiple of ``callback_freq``; the convergence
mask consumed by the solver is refreshed only there; the loop stops at the first logged
iteration where every column satisfies abs_res <= max(rtol * ||b||, atol).
"""

from __future__ import annotations

import torch


def residual_metrics(apply_A, B, reg, W):
    """linsys.py:96-99."""
    abs_res = torch.linalg.norm(B - (apply_A(W) + reg * W), dim=0, ord=2)
    rel_res = abs_res / torch.linalg.norm(B, dim=0, ord=2)
    return {"abs_res": abs_res, "rel_res": rel_res}


def active_mask(B, abs_res, atol, rtol):
    """linsys.py:101-107: True where the column has NOT converged."""
    tol = torch.clamp(rtol * torch.linalg.norm(B, dim=0, ord=2), min=atol)
    return abs_res > tol


def run(solver, apply_A, B, reg, max_iters, atol, rtol, callback_freq=10):
    """Drive ``solver`` (an object with ``W`` and ``step(mask)``); returns (W, log) where
    log maps logged iteration -> {"abs_res", "rel_res"}."""
    if B.ndim == 1:
        B = B[:, None]  # linsys.py:39-40
    log = {}
    m = residual_metrics(apply_A, B, reg, solver.W)
    log[0] = m
    mask = active_mask(B, m["abs_res"], atol, rtol)
    if not bool(mask.any()):
        return solver.W, log
    for i in range(1, max_iters + 1):
        solver.step(mask)
        if i % callback_freq == 0:
            m = residual_metrics(apply_A, B, reg, solver.W)
            log[i] = m
            mask = active_mask(B, m["abs_res"], atol, rtol)
            if not bool(mask.any()):
                break
    return solver.W, log
