"""Oracle: sketch-and-precondition preconditioner.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Restates rlaopt/preconditioners/skpre.py:55-118, including its ordering quirk: the
factor is the LOWER Cholesky factor L of G = Y^T Y + rho I, but the operator applied
is L^T L (skpre.py:111) and its inverse L^-1 L^-T (skpre.py:116-117) - mutually
consistent, not equal to G.  A parity build must reproduce this.
"""

from __future__ import annotations

import torch


def build(A, Omega_left: torch.Tensor, rho: float) -> torch.Tensor:
    """Y = Omega A; G = Y^T Y (+ rho on the diagonal if rho != 0); L = chol(G) lower."""
    Y = Omega_left @ A
    G = Y.T @ Y
    if rho != 0:
        G = G + rho * torch.eye(G.shape[0], dtype=G.dtype, device=G.device)
    return torch.linalg.cholesky(G, upper=False)


def matmul(L: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """skpre.py:111: L^T (L x)."""
    return L.T @ (L @ x)


def inverse_matmul(L: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """skpre.py:113-118: solve with L^T (upper) first, then with L (lower)."""
    squeeze = x.ndim == 1
    X = x[:, None] if squeeze else x
    w = torch.linalg.solve_triangular(L.T, X, upper=True)
    out = torch.linalg.solve_triangular(L, w, upper=False)
    return out[:, 0] if squeeze else out
