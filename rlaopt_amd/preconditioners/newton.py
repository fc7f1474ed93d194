"""Exact ("Newton") preconditioner P = A + rho I = L L^T (rlaopt/preconditioners/newton.py:
55-88).  The dense n x n Cholesky is a plain library factorisation (rocSOLVER through
torch.linalg) - it is not on the Krylov hot path; the triangular applies use it too."""

import torch

from ..ops import get_backend
from .preconditioner import Preconditioner


class Newton(Preconditioner):
    def __init__(self, config):
        super().__init__(config)
        self.L = None

    def _update(self, A, device):
        if isinstance(A, torch.Tensor):
            M = A.clone()  # the reference adds rho to the caller's tensor in place (newton.py:71)
        else:
            M = A @ torch.eye(A.shape[1], dtype=A.dtype, device=device)
        M.diagonal().add_(self.config.rho)
        self.L = torch.linalg.cholesky(M, upper=False)

    def _matmul(self, x):
        be = get_backend()
        return be.matmat(self.L, be.rmatmat(self.L, x))

    def _solve(self, x):
        X = x[:, None] if x.ndim == 1 else x
        W = torch.linalg.solve_triangular(self.L, X, upper=False)
        W = torch.linalg.solve_triangular(self.L.T, W, upper=True)
        return W[:, 0] if x.ndim == 1 else W

    _inverse_matmul_1d = _solve
    _inverse_matmul_2d = _solve
