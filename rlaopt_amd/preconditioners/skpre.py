"""Sketch-and-precondition (rlaopt/preconditioners/skpre.py:55-118) for tall A (m x d).

Y = Omega A (s x d) is one GEMM; G = Y^T Y + rho I is an f64-accumulated Gram product; the
d x d Cholesky runs in float64 on the host.  The reference's ordering quirk is reproduced
on purpose: with the LOWER factor L of G the operator applied is L^T L (skpre.py:111) and
its inverse L^-1 L^-T (skpre.py:116-117) - mutually consistent, not equal to G.  The
inverse is applied with the explicit d x d matrix L^-1 (two small device products per
apply instead of two triangular solves)."""

from warnings import warn

import torch

from ..ops import get_backend
from ..sketches import get_sketch
from ..utils import small_host_lapack
from .preconditioner import Preconditioner


class SkPre(Preconditioner):
    def __init__(self, config):
        super().__init__(config)
        self.Y = None
        self.L = None
        self._L64 = None
        self._Linv64 = None

    def _update(self, A, device):
        if self.config.sketch_size < A.shape[1]:
            warn(
                f"Sketch size ({self.config.sketch_size}) is smaller than the number of columns "
                f"in input matrix A ({A.shape[1]}). This may lead to a poor and/or unstable "
                "approximation."
            )
        be = get_backend()
        sk = get_sketch(self.config.sketch, "left", self.config.sketch_size, A.shape[0],
                        dtype=A.dtype, device=device)
        Y = sk._apply_left(A)
        G = be.gram_tn(Y, Y).cpu()
        del Y  # the reference frees it too (skpre.py:142-153)
        if self.config.rho != 0:
            G.diagonal().add_(self.config.rho)
        with small_host_lapack(G.shape[0]):
            L = torch.linalg.cholesky(G, upper=False)
            Linv = torch.linalg.solve_triangular(L, torch.eye(L.shape[0], dtype=torch.float64), upper=False)
        self.L = L.to(device=device, dtype=A.dtype)  # the attribute the reference exposes
        # The d x d factors are applied in float64 whatever the working dtype (the vectors they
        # act on are d-long: free), so that P (P^-1 x) = x holds to working precision even for
        # the ill-conditioned Gram matrices of a too-small sketch; results are cast back.
        self._L64 = L.to(device=device)
        self._Linv64 = Linv.to(device=device)

    def _apply64(self, fn, x):
        return fn(x.double()).to(x.dtype)

    def _matmul(self, x):
        be = get_backend()
        return self._apply64(lambda v: be.rmatmat(self._L64, be.matmat(self._L64, v)), x)  # L^T (L x)

    def _solve(self, x):
        be = get_backend()
        return self._apply64(lambda v: be.matmat(self._Linv64, be.rmatmat(self._Linv64, v)), x)

    _inverse_matmul_1d = _solve
    _inverse_matmul_2d = _solve

    # split factors for RIGHT preconditioning (LSQR on A R^-1 with R = L^T, G = L L^T):
    def _right_solve(self, x):
        """R^-1 x = L^-T x."""
        return self._apply64(lambda v: get_backend().rmatmat(self._Linv64, v), x)

    def _right_solve_t(self, x):
        """R^-T x = L^-1 x."""
        return self._apply64(lambda v: get_backend().matmat(self._Linv64, v), x)
