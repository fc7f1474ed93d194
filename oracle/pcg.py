"""Oracle: masked block PCG.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Restates rlaopt/solvers/pcg.py:13-93.  This is a TRUE block CG (matrix-valued alpha and
beta coupling the right-hand sides), not k independent CGs.  ``apply_A(X)`` must return
A @ X for a 2-D X; ``apply_Pinv(X)`` the preconditioner inverse (oracle/nystrom.py).
"""

from __future__ import annotations

import torch


class BlockPCG:
    def __init__(self, apply_A, apply_Pinv, B: torch.Tensor, reg: float, W_init: torch.Tensor):
        """pcg.py:32-37: R = B - (A W + reg W); Z = P^-1 R; P = Z; RZ = R^T Z."""
        self.apply_A = apply_A
        self.apply_Pinv = apply_Pinv
        self.B = B
        self.reg = reg
        self.W = W_init.clone()
        self.R = B - (apply_A(self.W) + reg * self.W)
        self.Z = apply_Pinv(self.R)
        self.P = self.Z.clone()
        self.RZ = self.R.T @ self.Z

    def step(self, mask: torch.Tensor) -> None:
        """One iteration on the columns selected by ``mask`` (pcg.py:45-93)."""
        if not bool(mask.any()):
            return
        Pm = self.P[:, mask]
        RZm = self.RZ[mask][:, mask]
        APm = self.apply_A(Pm) + self.reg * Pm
        alpha = torch.linalg.solve(Pm.T @ APm, RZm)
        self.W[:, mask] += Pm @ alpha
        self.R[:, mask] -= APm @ alpha
        Zm = self.apply_Pinv(self.R[:, mask])
        self.Z[:, mask] = Zm
        RZ_new = self.R[:, mask].T @ Zm
        beta = torch.linalg.solve(RZm, RZ_new)
        self.P[:, mask] = Zm + Pm @ beta
        full = torch.zeros_like(self.RZ)
        full[torch.outer(mask, mask)] = RZ_new.flatten()
        self.RZ = full
