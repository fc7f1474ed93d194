"""Randomized Nystrom preconditioner  P = U diag(S) U^T + rho I
(rlaopt/preconditioners/nystrom.py:55-152).

Same mathematics as the reference, re-planned for MI355X:

assembly (``_update``)
    Y = A Omega is the one big product (n x n x r): a matrix-core GEMM against the
    materialised kernel shard.  The reference then takes a thin SVD of the n x r matrix
    B^T = Y L^-T on one device (nystrom.py:94-95).  Here only r x r matrices ever leave the
    GPU: Core = Omega^T Y and G = Y^T Y are f64-accumulated Gram products (``rk_gram_tn``),
    the r x r algebra (Cholesky of Core + shift I, eigen-decomposition of
    L^-1 G L^-T = B B^T) runs in float64 on the host, and U = Y (L^-T V diag(1/sigma)) is a
    second tall GEMM, followed by one symmetric re-orthonormalisation pass.  S = sigma^2 -
    shift as in the reference.  With a row-sharded operator every tall array (Y, U) stays
    row-partitioned and the two Gram products are all-reduced (one fused r x r x 2 message).
apply (``P^-1 x``)
    both reference branches (nystrom.py:112-132) have the form
    (x - U (M (U^T x))) / rho  with an r x r matrix M prepared once:
    float64: M = diag(S / (S + rho));  otherwise M = (rho diag(1/S) + U^T U)^-1, the inverse
    of the matrix whose Cholesky factor the reference keeps as ``L``.  Two HBM passes over U.
    Directions whose clipped eigenvalue is exactly 0 make the reference's ``S**-1`` infinite
    (NaNs); here they are dropped from M, which is the S -> 0 limit.
"""

from __future__ import annotations

import torch

from .. import dist as _dist
from ..ops import get_backend
from ..sketches import get_sketch
from ..utils import small_host_lapack
from .configs import NystromConfig
from .enums import _DampingMode
from .preconditioner import Preconditioner


def _inv_sqrt_near_identity(G: torch.Tensor) -> torch.Tensor:
    """G^(-1/2) of the r x r float64 Gram matrix G = U^T U of a numerically orthonormal U.  With
    E = G - I small the series I - E/2 + 3E^2/8 - 5E^3/16 + 35E^4/128 is exact to ||E||^5 - three
    r x r products instead of an eigen-decomposition (0.7 ms on the host at r = 100, the largest
    single cost of a block preconditioner).  Rank-deficient or far-from-orthonormal input takes
    the eigen-decomposition, which also drops the null directions."""
    r = G.shape[0]
    E = G - torch.eye(r, dtype=G.dtype)
    if float(torch.linalg.matrix_norm(E)) <= 1e-3:
        E2 = E @ E
        return (torch.eye(r, dtype=G.dtype) - 0.5 * E + 0.375 * E2 - 0.3125 * (E2 @ E)
                + (35.0 / 128.0) * (E2 @ E2))
    w, Q = torch.linalg.eigh(G)
    inv_sqrt = torch.where(w > 1e-8, w.clamp_min(1e-300).rsqrt(), torch.zeros_like(w))
    return (Q * inv_sqrt[None, :]) @ Q.T


class Nystrom(Preconditioner):
    def __init__(self, config: NystromConfig):
        super().__init__(config)
        self.U = None
        self.S = None
        self._S_host = None
        self.low_precision = False
        self.L = None
        self._M = None  # r x r (dense) or length-r (diagonal) apply factor
        self._sharded = False
        self._group = None

    # -- assembly ------------------------------------------------------------------------------
    def _update(self, A, device, before_host_wait=None):
        """``before_host_wait``: called once, right before the first device -> host read-back (the
        r x r Gram blocks), i.e. when everything this assembly can do without the host has been
        queued: a caller with independent device work (SAP's block gradient) launches it there, so
        that it runs while the host factorises instead of delaying the kernels queued so far."""
        be = get_backend()
        if A.dtype != torch.float64:
            self.low_precision = True
        self._sharded = bool(getattr(A, "is_row_sharded", False))
        self._group = getattr(A, "_group", None)
        rank_r = self.config.rank
        sk = get_sketch(self.config.sketch, "right", rank_r, A.shape[1], dtype=A.dtype, device=device)
        Om = sk.Omega_mat
        if self._sharded:
            lo, hi = A.row_range
            Y = A.local_matmat(Om)  # rows [lo, hi) of A @ Omega, never gathered
            Om_rows = Om[lo:hi]
        else:
            Y = sk._apply_right(A).to(device)
            Om_rows = Om
        both = torch.stack([be.gram_tn(Om_rows, Y), be.gram_tn(Y, Y)])  # float64 (2, r, r)
        if self._sharded:
            _dist.all_reduce_sum(both, self._group)
        if before_host_wait is not None:
            before_host_wait()
        core, gram = both.cpu()

        shift = torch.finfo(A.dtype).eps * torch.trace(core)  # eps of the WORKING dtype
        core = core + shift * torch.eye(rank_r, dtype=torch.float64)
        with small_host_lapack(rank_r):
            L = torch.linalg.cholesky(core, upper=False)
            H = torch.linalg.solve_triangular(L, gram, upper=False)
            H = torch.linalg.solve_triangular(L, H.T.contiguous(), upper=False)  # L^-1 G L^-T
            H = 0.5 * (H + H.T)
            evals, V = torch.linalg.eigh(H)
            evals, V = evals.flip(0), V.flip(1)  # descending, like singular values
            sig2 = evals.clamp_min(0.0)
            sigma = sig2.sqrt()
            tiny = sigma[0] * 1e-150 if sigma[0] > 0 else 0.0
            inv_sigma = torch.where(sigma > tiny, 1.0 / sigma.clamp_min(1e-300), torch.zeros_like(sigma))
            M = torch.linalg.solve_triangular(L.T.contiguous(), V, upper=True) * inv_sigma[None, :]
        U = be.gemm(Y, M.to(device=device, dtype=A.dtype))
        del Y
        # one symmetric (Loewdin) re-orthonormalisation: U <- U (U^T U)^-1/2
        G2 = be.gram_tn(U, U)
        if self._sharded:
            _dist.all_reduce_sum(G2, self._group)
        with small_host_lapack(rank_r):
            fix = _inv_sqrt_near_identity(G2.cpu())
        self.U = be.gemm(U, fix.to(device=device, dtype=A.dtype))
        # (the eigenvalues also stay on the host, rounded to the working dtype: the damping and
        # the apply factor read them without a device round trip)
        self._S_host = (sig2 - shift).clamp_min(0.0).to(A.dtype)
        self.S = self._S_host.to(device=device)
        self._M = None
        self.L = None

    def _update_damping(self, baseline_rho: float) -> None:
        """Adaptive mode: rho = baseline + smallest eigenvalue (nystrom.py:140-152)."""
        if self.config.damping_mode == _DampingMode.ADAPTIVE:
            s_min = self._S_host[-1] if getattr(self, "_S_host", None) is not None else self.S[-1]
            self.config.rho = float(baseline_rho + float(s_min))
            self._M = None
            self.L = None

    # -- apply ---------------------------------------------------------------------------------
    def _project(self, X, out_f64=False):
        """U^T X (summed over row shards)."""
        t = get_backend().rmatmat(self.U, X, out_f64=out_f64)
        if self._sharded:
            _dist.all_reduce_sum(t, self._group)
        return t

    def _matmul(self, x: torch.Tensor) -> torch.Tensor:
        be = get_backend()
        X = x[:, None] if x.ndim == 1 else x
        t = self._project(X) * self.S[:, None]
        out = be.matmat(self.U, t, alpha=1.0, beta=float(self.config.rho), C=X)
        return out[:, 0] if x.ndim == 1 else out

    def _apply_factor(self):
        if self._M is not None:
            return self._M
        rho = float(self.config.rho)
        S64 = (self._S_host if getattr(self, "_S_host", None) is not None else self.S.cpu()).double()
        dev, dt = self.U.device, self.U.dtype
        if self.low_precision:
            UtU = get_backend().gram_tn(self.U, self.U)
            if self._sharded:
                _dist.all_reduce_sum(UtU, self._group)
            UtU = UtU.cpu()
            keep = torch.nonzero(S64 > 0)[:, 0]
            r = S64.numel()
            Minv = torch.zeros((r, r), dtype=torch.float64)
            if keep.numel():
                small = rho * torch.diag(1.0 / S64[keep]) + UtU[keep][:, keep]
                with small_host_lapack(r):
                    Lk = torch.linalg.cholesky(0.5 * (small + small.T))
                    Minv[keep[:, None], keep[None, :]] = torch.cholesky_inverse(Lk)
                self.L = Lk.to(device=dev, dtype=dt)
            self._M = Minv.to(device=dev)  # float64: the r x r step is applied in float64
        else:
            self._M = (S64 / (S64 + rho)).to(device=dev)
        return self._M

    def _inverse_matmul_general(self, x: torch.Tensor, unsqueeze: bool) -> torch.Tensor:
        be = get_backend()
        X = x[:, None] if unsqueeze else x
        M = self._apply_factor()
        t = self._project(X, out_f64=True)  # r x k, float64
        t = be.matmat(M, t) if M.ndim == 2 else t * M[:, None]
        inv_rho = 1.0 / float(self.config.rho)
        out = be.matmat(self.U, t.to(X.dtype), alpha=-inv_rho, beta=inv_rho, C=X)
        return out[:, 0] if unsqueeze else out

    def _inverse_matmul_with_gram(self, X: torch.Tensor):
        """(Z, X^T Z) with Z = P^-1 X for a row-sharded n_loc x k block, in ONE collective: with
        c = U^T X,  P^-1 X = (X - U M c) / rho  and  X^T P^-1 X = (X^T X - c^T M c) / rho,  so the
        partial c and X^T X of every rank travel in the same all-reduce (block PCG needs both,
        pcg.py:79-81; separately they cost two latency-bound collectives per iteration)."""
        be = get_backend()
        M = self._apply_factor()
        r, k = self.U.shape[1], X.shape[1]
        c = be.rmatmat(self.U, X, out_f64=True)   # r x k, float64
        xx = be.rmatmat(X, X, out_f64=True)       # k x k, float64
        both = torch.cat([c.reshape(-1), xx.reshape(-1)])
        if self._sharded:
            _dist.all_reduce_sum(both, self._group)
        c, xx = both[: r * k].view(r, k), both[r * k :].view(k, k)
        t = be.matmat(M, c) if M.ndim == 2 else c * M[:, None]
        inv_rho = 1.0 / float(self.config.rho)
        Z = be.matmat(self.U, t.to(X.dtype), alpha=-inv_rho, beta=inv_rho, C=X)
        XtZ = (xx - c.T @ t) * inv_rho
        return Z, 0.5 * (XtZ + XtZ.T)

    def _inverse_matmul_1d(self, x):
        return self._inverse_matmul_general(x, unsqueeze=True)

    def _inverse_matmul_2d(self, x):
        return self._inverse_matmul_general(x, unsqueeze=False)
