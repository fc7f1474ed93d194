"""Oracle: Nystrom low-rank assembly and shifted-inverse apply.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Restates rlaopt/preconditioners/nystrom.py:55-152.  ``A`` is anything supporting
``A @ X`` for a 2-D X (a dense tensor or an operator object); Omega is supplied by
the caller (oracle/sketches.py) so that parity runs can inject a fixed test matrix.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch


@dataclass
class NystromState:
    U: torch.Tensor  # (n, r) orthonormal columns
    S: torch.Tensor  # (r,) clipped eigenvalues, descending
    shift: torch.Tensor  # scalar stabilising shift
    rho: float  # damping actually used by apply / inverse
    low_precision: bool  # True iff operator dtype != float64 (nystrom.py:66-67)
    L: torch.Tensor | None = None  # lazily built factor for the low-precision inverse


def build(A, Omega: torch.Tensor, rho: float) -> NystromState:
    """nystrom.py:55-96.

    Y = A Omega;  Core = Omega^T Y;  shift = eps(dtype) * trace(Core);  Core += shift I;
    L = chol(Core);  B = L^-1 Y^T;  U, sigma, _ = svd(B^T);  S = max(sigma^2 - shift, 0).
    """
    Y = A @ Omega
    core = Omega.T @ Y
    shift = torch.finfo(Y.dtype).eps * torch.trace(core)
    core = core + shift * torch.eye(core.shape[0], dtype=core.dtype, device=core.device)
    L = torch.linalg.cholesky(core, upper=False)
    B = torch.linalg.solve_triangular(L, Y.T, upper=False)
    U, sigma, _ = torch.linalg.svd(B.T, full_matrices=False)
    S = torch.clamp(sigma * sigma - shift, min=0.0)
    return NystromState(
        U=U, S=S, shift=shift, rho=float(rho), low_precision=(Y.dtype != torch.float64)
    )


def update_damping(state: NystromState, baseline_rho: float, adaptive: bool) -> NystromState:
    """nystrom.py:140-152: adaptive mode sets rho = baseline + smallest eigenvalue."""
    if adaptive:
        state.rho = float(baseline_rho + state.S[-1])
    return state


def matmul(state: NystromState, x: torch.Tensor) -> torch.Tensor:
    """P x = U diag(S) U^T x + rho x  (nystrom.py:98-108)."""
    s = state.S if x.ndim == 1 else state.S[:, None]
    return state.U @ (s * (state.U.T @ x)) + state.rho * x


def inverse_matmul(state: NystromState, x: torch.Tensor) -> torch.Tensor:
    """P^-1 x (nystrom.py:112-132).

    float64:      (x - U U^T x)/rho + U (U^T x / (S + rho))
    otherwise:    L = chol(rho diag(1/S) + U^T U) built once, then
                  (x - U L^-T L^-1 U^T x) / rho
    """
    squeeze = x.ndim == 1
    X = x[:, None] if squeeze else x
    t = state.U.T @ X
    if state.low_precision:
        if state.L is None:
            state.L = torch.linalg.cholesky(
                state.rho * torch.diag(state.S**-1) + state.U.T @ state.U
            )
        w = torch.linalg.solve_triangular(state.L, t, upper=False)
        w = torch.linalg.solve_triangular(state.L.T, w, upper=True)
        out = (1.0 / state.rho) * (X - state.U @ w)
    else:
        out = (1.0 / state.rho) * (X - state.U @ t) + state.U @ (t / (state.S + state.rho)[:, None])
    return out[:, 0] if squeeze else out
