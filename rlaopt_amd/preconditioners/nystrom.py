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

import os

import numpy as np
import torch

try:  # (scipy is a declared dependency of the reference, pyproject.toml:21; the torch path needs neither)
    import scipy.linalg as _sl
    from threadpoolctl import ThreadpoolController as _TPC
except ImportError:  # pragma: no cover
    _sl = None
_tpc = None


def _one_blas_thread():
    """Context: scipy's BLAS / LAPACK on ONE thread.  Its OpenBLAS pool is as wide as the host (256 threads on
    a GPU box) and spins next to torch's own pool: a 100 x 100 Cholesky + smallest eigenvalue takes 8 ms with
    the pool against 0.3 ms on one thread (entering the context: 14 us)."""
    global _tpc
    if _tpc is None:
        _tpc = _TPC()
    return _tpc.limit(limits=1, user_api="blas")

from .. import dist as _dist
from ..ops import get_backend
from ..sketches import get_sketch
from ..utils import small_host_lapack
from .configs import NystromConfig
from .enums import _DampingMode
from .preconditioner import Preconditioner


_FUSED_GRAM_BYTES = 16 << 20


def _inv_sqrt_near_identity(G: torch.Tensor) -> torch.Tensor:
    """G^(-1/2) of the r x r float64 Gram matrix G = U^T U of a numerically orthonormal U.  With
    E = G - I small the series I - E/2 + 3E^2/8 - 5E^3/16 + 35E^4/128 is exact to ||E||^5 - three
    r x r products instead of an eigen-decomposition (0.7 ms on the host at r = 100, the largest
    single cost of a block preconditioner).  Rank-deficient or far-from-orthonormal input takes
    the eigen-decomposition, which also drops the null directions."""
    # (no torch.eye here: outside a small_host_lapack context it runs on the host-wide thread pool - 17 ms per call
    # for a 512 x 512 identity on a GPU box, a third of the whole assembly)
    E = G.clone()
    E.diagonal().sub_(1.0)
    if float(torch.linalg.matrix_norm(E)) <= 1e-3:
        E2 = E @ E
        out = -0.5 * E
        out.diagonal().add_(1.0)  # I - E/2, then the remaining terms in the series' order
        return out + 0.375 * E2 - 0.3125 * (E2 @ E) + (35.0 / 128.0) * (E2 @ E2)
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
        self._raw = False  # True: self.U is the raw sketch image Y (see _update_block)
        self._rho = None   # this instance's damping once set (the config object is shared and mutable)
        self._C = None
        self._sharded = False
        self._group = None

    # -- assembly ------------------------------------------------------------------------------
    def _sketch_and_grams(self, A, device, before_host_wait=None, raw_ortho=False):
        """Y = A Omega (rows of this shard) and the two r x r Gram blocks Omega^T Y, Y^T Y on the host
        (float64).  ``before_host_wait``: called once, right before the device -> host read-back, i.e.
        when everything this assembly can do without the host has been queued: a caller with
        independent device work (SAP's block gradient) launches it there, so that it runs while the
        host factorises instead of delaying the kernels queued so far."""
        be = get_backend()
        if A.dtype != torch.float64:
            self.low_precision = True
        self._sharded = bool(getattr(A, "is_row_sharded", False))
        self._group = getattr(A, "_group", None)
        self._raw = False
        self._C = None
        self._rho = None
        rank_r = self.config.rank
        if raw_ortho:
            from ..sketches import raw_ortho as _raw

            with _raw():
                sk = get_sketch(self.config.sketch, "right", rank_r, A.shape[1], dtype=A.dtype, device=device)
        else:
            sk = get_sketch(self.config.sketch, "right", rank_r, A.shape[1], dtype=A.dtype, device=device)
        lo, hi = A.row_range if self._sharded else (0, A.shape[0])
        Ycols = sk.columns_of(A, local=self._sharded) if hasattr(sk, "columns_of") else None
        if Ycols is not None:
            # column-sampling sketch: Y = scaled columns of A (no product), Omega^T Y = scaled rows J of Y
            Y = Ycols.to(device)
            both = torch.stack([sk.rows_of(Y, lo, hi), be.gram_tn(Y, Y)])
            if self._sharded:
                _dist.all_reduce_sum(both, self._group)
            if before_host_wait is not None:
                before_host_wait()
            host = both.cpu()
            return Y, host[0], host[1], None
        Om = sk.Omega_mat
        if self._sharded:
            Y = A.local_matmat(Om)  # rows [lo, hi) of A @ Omega, never gathered
            Om_rows = Om[lo:hi]
        else:
            Y = sk._apply_right(A).to(device)
            Om_rows = Om
        if sk.raw:
            # un-orthonormalised Ortho sketch: Omega^T Omega rides along ([Omega Y]^T [Omega Y], one launch)
            Z = torch.cat([Om_rows, Y], dim=1)
            G3 = be.gram_tn(Z, Z)
            both = torch.stack([G3[:rank_r, rank_r:], G3[rank_r:, rank_r:], G3[:rank_r, :rank_r]])
        elif Y.numel() * Y.element_size() <= _FUSED_GRAM_BYTES:
            # a block preconditioner (5 000 x 100): both blocks from one launch, [Omega Y]^T Y
            both = be.gram_tn(torch.cat([Om_rows, Y], dim=1), Y).view(2, rank_r, rank_r)
        else:
            both = torch.stack([be.gram_tn(Om_rows, Y), be.gram_tn(Y, Y)])  # float64 (2, r, r)
        if self._sharded:
            _dist.all_reduce_sum(both, self._group)
        if before_host_wait is not None:
            before_host_wait()
        host = both.cpu()
        return Y, host[0], host[1], (host[2] if sk.raw else None)

    def _update(self, A, device, before_host_wait=None):
        from ..utils import quiet_host

        with quiet_host():  # (a stable, small host thread pool for the r x r algebra; re-entrant inside a solve)
            Y, core, gram, _ = self._sketch_and_grams(A, device, before_host_wait)
            self._finish_spectral(Y, core, gram, A.dtype, device)

    def _finish_spectral(self, Y, core, gram, dtype, device):
        """U, S of the rank-r approximation from the Gram blocks (the reference's thin SVD of B^T,
        nystrom.py:94-96, as r x r algebra on the host + two tall GEMMs)."""
        be = get_backend()
        rank_r = self.config.rank
        shift = torch.finfo(dtype).eps * torch.trace(core)  # eps of the WORKING dtype
        core = core.clone()
        core.diagonal().add_(shift)
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
        U = be.gemm(Y, M.to(device=device, dtype=dtype))
        del Y
        # one symmetric (Loewdin) re-orthonormalisation: U <- U (U^T U)^-1/2
        G2 = be.gram_tn(U, U)
        if self._sharded:
            _dist.all_reduce_sum(G2, self._group)
        with small_host_lapack(rank_r):
            fix = _inv_sqrt_near_identity(G2.cpu())
        self.U = be.gemm(U, fix.to(device=device, dtype=dtype))
        # (the eigenvalues also stay on the host, rounded to the working dtype: the damping and
        # the apply factor read them without a device round trip)
        self._S_host = (sig2 - shift).clamp_min(0.0).to(dtype)
        self.S = self._S_host.to(device=device)
        self._M = None
        self.L = None

    def _update_block(self, A, device, baseline_rho: float, before_host_wait=None) -> None:
        """Assembly AND damping of a preconditioner that will only ever be INVERTED (SAP's block
        preconditioner, sap.py:64-74), in ONE host round trip and without an eigen-decomposition.
        With B = Y L^-T (L L^T = Core + shift I) the reference's operator is
        P = U diag(sigma^2 - shift) U^T + rho I = B f(H) B^T + rho I,  H = B^T B = L^-1 (Y^T Y) L^-T,
        and every matrix function of H it needs is an r x r solve:
            P^-1 x = (x - Y M Y^T x) / rho,   M = L^-T (Hs + rho I)^-1 Hs H^-1 L^-1,   Hs = H - shift I,
            rho    = baseline + lambda_min(Hs)           (adaptive damping, nystrom.py:140-152)
        so Y itself serves as the basis (``self.U``; NOT orthonormal) and only the smallest eigenvalue
        is computed.  The spectral route costs the SAP step three host round trips, an r x r
        eigen-decomposition and two tall GEMMs (1.85 ms at blk = 5 000, r = 100; this one: see
        DESIGN.md).  Rank-deficient sketches (an eigenvalue the reference would clip to 0), a sharded
        operator or RK_NYSTROM_BLOCK_FAST=0 take the spectral route."""
        import os

        fast = (os.environ.get("RK_NYSTROM_BLOCK_FAST", "1") != "0"
                and not bool(getattr(A, "is_row_sharded", False)))
        Y, core, gram, SO = self._sketch_and_grams(
            A, device, before_host_wait,
            raw_ortho=fast and self.config.sketch == "ortho" and os.environ.get("RK_NYSTROM_RAW_ORTHO", "1") != "0")
        if not fast or not self._finish_woodbury(Y, core, gram, A.dtype, device, baseline_rho, SO):
            if SO is not None:
                # back to the orthonormal sketch Q = Omega R^-1 (R^T R = Omega^T Omega) the spectral route
                # assumes: Y, Core and the Gram block transform by R^-1
                with small_host_lapack(self.config.rank):
                    SOs = 0.5 * (SO + SO.T)
                    R, info = torch.linalg.cholesky_ex(SOs, upper=True)
                    if int(info) == 0:
                        Rinv = torch.linalg.solve_triangular(R, torch.eye(R.shape[0], dtype=torch.float64), upper=True)
                    else:
                        # Omega^T Omega is not numerically positive definite (a rank-deficient raw draw):
                        # orthonormalise the directions that carry weight, Omega W with W = V diag(lam^-1/2)
                        # (zero columns for the others: they contribute sigma = 0, which the spectral route
                        # clips exactly as the reference clips its own)
                        lam, Vs = torch.linalg.eigh(SOs)
                        keep = lam > lam[-1] * SOs.shape[0] * torch.finfo(torch.float64).eps
                        if not bool(torch.isfinite(lam).all()) or not bool(keep.any()):
                            raise RuntimeError("Nystrom: the sketch image is not finite / has no range")
                        Rinv = Vs * torch.where(keep, lam.clamp_min(1e-300).rsqrt(), torch.zeros_like(lam))
                    core = Rinv.T @ core @ Rinv
                    gram = Rinv.T @ gram @ Rinv
                Y = get_backend().gemm(Y, Rinv.to(device=device, dtype=A.dtype))
            self._finish_spectral(Y, 0.5 * (core + core.T), 0.5 * (gram + gram.T), A.dtype, device)
            self._update_damping(baseline_rho)

    def _finish_woodbury(self, Y, core, gram, dtype, device, baseline_rho, SO=None) -> bool:
        """In Y coordinates (C = Core + shift I, G = Y^T Y, both r x r on the host, float64):
            M = L^-T (Hs + rho I)^-1 Hs H^-1 L^-1 = N (I - shift C G^-1),   N = (G + (rho - shift) C)^-1
        so the only use of H = L^-1 G L^-T is its smallest eigenvalue (values only: ``eigvalsh``).
        ``SO`` = Omega^T Omega of an Ortho sketch left un-orthonormalised (Omega = Q R): every block
        above transforms by congruence with R (Core_Q = R^-T Core R^-1, ...), the generalized
        eigenvalues and M R-covariantly, so the orthonormal sketch's result is obtained with
        shift = eps trace(Core_Q) = eps trace(SO^-1 Core) and C = Core + shift SO - no QR."""
        r = self.config.rank
        if _sl is not None and os.environ.get("RK_NYSTROM_SCIPY", "1") != "0":
            with _one_blas_thread():
                out = self._woodbury_blocks_scipy(core, gram, dtype, baseline_rho, SO)
            if out is None:
                return False
            M, C, LG, shift, rho = out
            return self._install_woodbury(Y, M, C, LG, shift, rho, device)
        eye = torch.eye(r, dtype=torch.float64)
        with small_host_lapack(r):
            if SO is None:
                shift = torch.finfo(dtype).eps * torch.trace(core)
                C = core + shift * eye
            else:
                SO = 0.5 * (SO + SO.T)
                LS, info = torch.linalg.cholesky_ex(SO)
                if int(info) != 0:
                    return False
                shift = torch.finfo(dtype).eps * torch.trace(torch.cholesky_solve(core, LS))
                C = core + shift * SO
            C = 0.5 * (C + C.T)
            if self.config.damping_mode == _DampingMode.ADAPTIVE:
                L, info = torch.linalg.cholesky_ex(C, upper=False)
                if int(info) != 0:
                    return False
                H = torch.linalg.solve_triangular(L, gram, upper=False)
                H = torch.linalg.solve_triangular(L, H.T.contiguous(), upper=False)  # L^-1 G L^-T
                lam = torch.linalg.eigvalsh(0.5 * (H + H.T))  # ascending
                lam_min = lam[0] - shift
                if not bool(lam_min > 1e-13 * lam[-1]):
                    return False  # the reference clips this direction to 0: spectral route
                rho = float(baseline_rho + float(lam_min.to(dtype)))
            else:
                rho = float(self.config.rho)
            LG, info = torch.linalg.cholesky_ex(gram)  # (positive definite <=> every sigma^2 > 0)
            if int(info) != 0:
                return False
            LN, info = torch.linalg.cholesky_ex(gram + (rho - shift) * C)
            if int(info) != 0:
                return False
            N = torch.cholesky_inverse(LN)
            CGinv = torch.cholesky_solve(C, LG).T  # C G^-1
            M = N - shift * (N @ CGinv)
            M = 0.5 * (M + M.T)
            if self.config.damping_mode != _DampingMode.ADAPTIVE:
                # sigma^2 > shift for every direction (else the reference clips): G - shift C > 0
                if int(torch.linalg.cholesky_ex(gram - shift * C)[1]) != 0:
                    return False
        return self._install_woodbury(Y, M, C, LG, shift, rho, device)

    def _install_woodbury(self, Y, M, C, LG, shift, rho, device) -> bool:
        self.U = Y
        self._M = M.to(device=device)  # float64: the r x r step is applied in float64
        self._C = None                 # (P @ x is prepared on first use, see _matmul)
        self._host_blocks = (C, LG, shift)
        self._raw = True
        self.config.rho = rho
        self._rho = rho
        self.S = None
        self._S_host = None
        self.L = None
        return True

    def _woodbury_blocks_scipy(self, core, gram, dtype, baseline_rho, SO):
        """The r x r algebra of ``_finish_woodbury`` through scipy's LAPACK bindings on the tensors' numpy
        views: the same factorisations without torch's per-call dispatch (a Cholesky of a 100 x 100 matrix is
        0.03 ms there against 0.1 - 0.4 ms through ``torch.linalg``) and only the SMALLEST eigenvalue of H
        (``eigvalsh(subset_by_index=[0, 0])``) - 0.75 instead of 1.15 ms per block preconditioner, on the
        critical path of every SAP step.  Returns (M, C, LG, shift, rho) as float64 tensors / floats, or
        None where the torch path returns False (a factorisation fails, a direction the reference clips)."""
        r = self.config.rank
        cn = core.numpy()
        gn = gram.numpy()
        eps = float(torch.finfo(dtype).eps)
        try:
            if SO is None:
                shift = eps * float(np.trace(cn))
                C = cn + shift * np.eye(r)
            else:
                SOn = SO.numpy()
                SOn = 0.5 * (SOn + SOn.T)
                LS = _sl.cho_factor(SOn, lower=True, check_finite=False)
                shift = eps * float(np.trace(_sl.cho_solve(LS, cn, check_finite=False)))
                C = cn + shift * SOn
            C = 0.5 * (C + C.T)
            if self.config.damping_mode == _DampingMode.ADAPTIVE:
                L = _sl.cholesky(C, lower=True, check_finite=False)
                H = _sl.solve_triangular(L, gn, lower=True, check_finite=False)
                H = _sl.solve_triangular(L, H.T, lower=True, check_finite=False)  # L^-1 G L^-T
                H = 0.5 * (H + H.T)
                lam0 = float(_sl.eigvalsh(H, subset_by_index=[0, 0], check_finite=False)[0])
                lam_min = lam0 - shift
                # (lambda_max <= trace(H): at least as strict as the torch path's 1e-13 lambda_max)
                if not lam_min > 1e-13 * float(np.trace(H)):
                    return None  # the reference clips this direction to 0: spectral route
                rho = float(baseline_rho + float(torch.tensor(lam_min, dtype=torch.float64).to(dtype)))
            else:
                rho = float(self.config.rho)
            LG = _sl.cholesky(gn, lower=True, check_finite=False)  # (positive definite <=> every sigma^2 > 0)
            LN = _sl.cho_factor(gn + (rho - shift) * C, lower=True, check_finite=False)
            N = _sl.cho_solve(LN, np.eye(r), check_finite=False)
            CGinv = _sl.cho_solve((LG, True), C, check_finite=False).T  # C G^-1
            M = N - shift * (N @ CGinv)
            M = 0.5 * (M + M.T)
            if self.config.damping_mode != _DampingMode.ADAPTIVE:
                _sl.cholesky(gn - shift * C, lower=True, check_finite=False)  # sigma^2 > shift everywhere
            if not (np.isfinite(M).all() and np.isfinite(rho)):
                return None
        except (np.linalg.LinAlgError, ValueError):
            return None
        shift_t = torch.tensor(shift, dtype=torch.float64)
        return (torch.from_numpy(np.ascontiguousarray(M)), torch.from_numpy(np.ascontiguousarray(C)),
                torch.from_numpy(np.ascontiguousarray(np.tril(LG))), shift_t, rho)

    def _raw_forward_factor(self):
        """P = Y (C^-1 - shift G^-1) Y^T + rho I in the raw basis (only P @ x needs it; SAP never does)."""
        if self._C is None:
            C, LG, shift = self._host_blocks
            r = C.shape[0]
            with small_host_lapack(r):
                F = torch.linalg.inv(C) - shift * torch.cholesky_inverse(LG)
            self._C = (0.5 * (F + F.T)).to(device=self.U.device)
        return self._C

    def _update_damping(self, baseline_rho: float) -> None:
        """Adaptive mode: rho = baseline + smallest eigenvalue (nystrom.py:140-152)."""
        if getattr(self, "_raw", False):
            raise RuntimeError("Nystrom._update_block already applied the damping")
        if self.config.damping_mode == _DampingMode.ADAPTIVE:
            s_min = self._S_host[-1] if getattr(self, "_S_host", None) is not None else self.S[-1]
            self.config.rho = float(baseline_rho + float(s_min))
            self._rho = self.config.rho
            self._M = None
            self.L = None

    def _rho_now(self) -> float:
        """The damping this instance applies.  Like the reference, ``_update_damping`` writes it into
        the (shared) config (nystrom.py:151-152); SAP prepares the next block preconditioner from the
        same config object before this one has been applied, so each instance keeps its own copy."""
        return float(self._rho if self._rho is not None else self.config.rho)

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
        if getattr(self, "_raw", False):  # P = Y C Y^T + rho I (basis not orthonormal, see _update_block)
            t = be.matmat(self._raw_forward_factor(), self._project(X, out_f64=True)).to(X.dtype)
        else:
            t = self._project(X) * self.S[:, None]
        out = be.matmat(self.U, t, alpha=1.0, beta=self._rho_now(), C=X)
        return out[:, 0] if x.ndim == 1 else out

    def _apply_factor(self):
        if self._M is not None:
            return self._M
        rho = self._rho_now()
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
        if (not self._sharded and hasattr(be, "nystrom_inv_apply") and X.shape[1] <= 16 and X.is_cuda
                and M.dtype == torch.float64):
            out = be.nystrom_inv_apply(self.U, M, self._rho_now(), X)  # three launches, no host-side glue
            return out[:, 0] if unsqueeze else out
        t = self._project(X, out_f64=True)  # r x k, float64
        t = be.matmat(M, t) if M.ndim == 2 else t * M[:, None]
        inv_rho = 1.0 / self._rho_now()
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
        inv_rho = 1.0 / self._rho_now()
        Z = be.matmat(self.U, t.to(X.dtype), alpha=-inv_rho, beta=inv_rho, C=X)
        XtZ = (xx - c.T @ t) * inv_rho
        return Z, 0.5 * (XtZ + XtZ.T)

    def _inverse_matmul_1d(self, x):
        return self._inverse_matmul_general(x, unsqueeze=True)

    def _inverse_matmul_2d(self, x):
        return self._inverse_matmul_general(x, unsqueeze=False)
