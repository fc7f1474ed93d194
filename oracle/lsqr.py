"""Oracle: LSQR (Paige & Saunders 1982) with an optional right preconditioner and ridge term.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

PARITY UNPINNED: the reference has no least-squares model and no LSQR (SURVEY.md section 8,
row a14: ``grep lsqr`` over the tree has zero hits); BASELINE.json config 4 names it anyway.
This restatement follows the published algorithm (C. C. Paige, M. A. Saunders, "LSQR: an
algorithm for sparse linear equations and sparse least squares", ACM TOMS 8(1), 1982,
section 4) and is checked against ``scipy.sparse.linalg.lsqr`` (scipy 1.15.3) in
tests/test_oracle_lsqr.py.  The preconditioner is the upper-triangular factor R = L^T of the
reference's SkPre Gram matrix G = L L^T (rlaopt/preconditioners/skpre.py:93-98): LSQR runs on
A R^-1, whose Gram matrix is ~ I, and x = R^-1 y.

    minimise ||A x - b||^2 + reg ||x||^2      solved as LSQR on [A; sqrt(reg) I] R^-1
"""

from __future__ import annotations

import torch


def lsqr(A, b, reg=0.0, Rinv=None, x0=None, iters=100):
    """Run ``iters`` LSQR steps column-wise on b (m x k); returns (x, history of the
    normal-equation residual ||A^T (b - A x) - reg x|| per column, one row per iteration)."""
    squeeze = b.ndim == 1
    B = b[:, None] if squeeze else b
    m, d = A.shape
    k = B.shape[1]
    dt = B.dtype
    sq = reg**0.5
    Rinv = torch.eye(d, dtype=dt) if Rinv is None else Rinv
    X0 = torch.zeros(d, k, dtype=dt) if x0 is None else (x0[:, None] if x0.ndim == 1 else x0)

    def fwd(Y):  # [A; sq I] R^-1 Y
        Z = Rinv @ Y
        return A @ Z, sq * Z

    def adj(Ut, Ub):  # R^-T (A^T Ut + sq Ub)
        return Rinv.T @ (A.T @ Ut + sq * Ub)

    def norms(Ut, Ub):
        return torch.sqrt((Ut * Ut).sum(0) + (Ub * Ub).sum(0))

    Ut, Ub = B - A @ X0, -sq * X0
    beta = norms(Ut, Ub)
    Ut, Ub = Ut / beta, Ub / beta
    Vv = adj(Ut, Ub)
    alpha = torch.linalg.norm(Vv, dim=0)
    Vv = Vv / alpha
    W = Vv.clone()
    phibar, rhobar = beta.clone(), alpha.clone()
    Y = torch.zeros(d, k, dtype=dt)
    hist = []
    for _ in range(iters):
        Tt, Tb = fwd(Vv)
        Ut, Ub = Tt - alpha * Ut, Tb - alpha * Ub
        beta = norms(Ut, Ub)
        Ut, Ub = Ut / beta, Ub / beta
        Vv = adj(Ut, Ub) - beta * Vv
        alpha = torch.linalg.norm(Vv, dim=0)
        Vv = Vv / alpha
        rho = torch.sqrt(rhobar * rhobar + beta * beta)
        c, s = rhobar / rho, beta / rho
        theta = s * alpha
        rhobar = -c * alpha
        phi = c * phibar
        phibar = s * phibar
        Y = Y + (phi / rho) * W
        W = Vv - (theta / rho) * W
        Xc = X0 + Rinv @ Y
        hist.append(torch.linalg.norm(A.T @ (B - A @ Xc) - reg * Xc, dim=0))
    X = X0 + Rinv @ Y
    return (X[:, 0] if squeeze else X), torch.stack(hist)
