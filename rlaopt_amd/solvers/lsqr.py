"""LSQR (Paige & Saunders, ACM TOMS 8(1) 1982, section 4) for ``LeastSquares`` models:

    minimise ||A w - b||^2 + reg ||w||^2     over w, column-wise for b (m x k)

run on the right-preconditioned, ridge-augmented operator  [A; sqrt(reg) I] R^-1, where R = L^T
is the triangular factor of the sketch-and-precondition Gram matrix (SkPre, rlaopt/
preconditioners/skpre.py:93-98) or the identity.  The reference has no LSQR (SURVEY.md 8a row
a14); parity is pinned against ``oracle/lsqr.py`` / scipy instead.

Device plan: for a dense A whose rows fit a wave's registers (n * sizeof <= 4 KiB, k <= 2) both
m-long products of an iteration come out of ONE sweep over A (``rk_normal_pass``: beta is not
needed inside the pass because A^T u = A^T(unnormalised u) / beta) - m d s bytes per iteration;
otherwise they are ``rk_matmat_skinny`` / ``rk_matmat_skinny_t`` (A read twice, 2 m d s bytes); the Golub-Kahan vector
updates are ``rk_axpby_cols`` with per-column coefficients held in device memory; all rotation
scalars are k-vectors on the device, so an iteration never synchronises the host.
"""

from __future__ import annotations

import os
from typing import TYPE_CHECKING

import torch

from ..ops import get_backend
from ..preconditioners import IdentityConfig, PreconditionerConfig, SkPreConfig
from ..preconditioners import _get_precond as _pf_get_precond
from ..utils import quiet_host
from .solver import Solver

if TYPE_CHECKING:
    from ..models import LeastSquares

VALID_PRECONDS = [IdentityConfig, SkPreConfig]


class LSQR(Solver):
    def __init__(self, system: "LeastSquares", W_init: torch.Tensor,
                 precond_config: PreconditionerConfig, device: torch.device):
        if type(precond_config) not in VALID_PRECONDS:
            raise TypeError(
                f"Valid preconditioner configs for LSQR are {VALID_PRECONDS}, "
                f"but received {type(precond_config)}"
            )
        self.system, self.precond_config, self.device = system, precond_config, device
        self._be = be = get_backend()
        B = system.B
        if W_init.ndim != 2 or W_init.shape != (system.A.shape[1], B.shape[1]):
            raise ValueError(
                f"W_init must have shape {(system.A.shape[1], B.shape[1])}, got {tuple(W_init.shape)}"
            )
        self._W0 = W_init.clone()
        self._sq = float(system.reg) ** 0.5
        self.P = self._get_precond()
        f64 = dict(dtype=torch.float64, device=B.device)
        # beta_1 u_1 = [b - A w0; -sqrt(reg) w0]
        self._Ut = system._residual(self._W0)
        self._Ub = (-self._sq) * self._W0 if self._sq else None
        beta = self._norms(self._Ut, self._Ub)
        self._scale(self._Ut, self._Ub, 1.0 / beta)
        # alpha_1 v_1 = Ahat^T u_1
        self._V = self._adjoint(self._Ut, self._Ub)
        self._alpha = self._norms(self._V, None)
        be.axpby_cols(self._V, 1.0 / self._alpha, self._V)
        self._Wd = self._V.clone()  # search direction "w" of the paper
        self._phibar, self._rhobar = beta.clone(), self._alpha.clone()
        self._Y = torch.zeros_like(self._V)
        self._ones = torch.ones(B.shape[1], **f64)
        # One-pass mode (dense A, shapes rk_normal_pass supports): the top block of u is kept
        # UNNORMALISED together with 1/beta per column - u = _Ut * _uinv - because then
        # beta u = A z - alpha u and A^T u both come out of a single sweep over A.
        self._uinv = self._ones.clone()
        self._one_pass = os.environ.get("RK_LSQR_ONE_PASS", "1") != "0"

    @property
    def W(self) -> torch.Tensor:
        return self._W0 + self.P._right_solve(self._Y)

    def _get_precond(self):
        P = _pf_get_precond(self.precond_config)
        P._update(self.system.A, self.device)
        return P

    # -- pieces -----------------------------------------------------------------------------
    def _norms(self, top, bot):
        be = self._be
        sq = be.rmatmat(top, top, out_f64=True).diagonal()
        if bot is not None:
            sq = sq + be.rmatmat(bot, bot, out_f64=True).diagonal()
        return sq.clamp_min(0.0).sqrt()

    def _scale(self, top, bot, coeff):
        self._be.axpby_cols(top, coeff, top)
        if bot is not None:
            self._be.axpby_cols(bot, coeff, bot)

    def _adjoint(self, top, bot):
        """R^-T (A^T top + sqrt(reg) bot)."""
        g = self.system._rmatmat(top)
        if bot is not None:
            g = g + self._sq * bot  # d x k (tiny)
        return self.P._right_solve_t(g)

    def _step(self):
        with quiet_host():  # (a no-op inside LinSys.solve, which already holds it)
            self._step_impl()

    def _step_impl(self):
        mask = self.system.mask
        if not bool(mask.any()):
            return
        be = self._be
        # beta u = Ahat v - alpha u
        Z = self.P._right_solve(self._V)
        neg_alpha = -self._alpha
        fused = self.system._normal_pass(Z, self._Ut, self._alpha * self._uinv) if self._one_pass else None
        if fused is not None:
            # one sweep over A: _Ut <- A z - alpha u (unnormalised), g = A^T _Ut, ss = |_Ut|^2
            g_raw, ss = fused
            if self._Ub is not None:
                be.axpby_cols(Z, self._sq * self._ones, self._Ub, neg_alpha)
                ss = ss + be.rmatmat(self._Ub, self._Ub, out_f64=True).diagonal()
            beta = ss.clamp_min(0.0).sqrt()
            self._uinv = 1.0 / beta
            g = (g_raw * self._uinv).to(self._V.dtype)
            if self._Ub is not None:
                be.axpby_cols(self._Ub, self._uinv, self._Ub)
                g = g + self._sq * self._Ub
            G = self.P._right_solve_t(g)
        else:
            self._one_pass = False  # (decided at the first step; _Ut stays normalised from here on)
            T = self.system._matmat(Z)
            be.axpby_cols(T, None, self._Ut, neg_alpha)
            if self._Ub is not None:
                be.axpby_cols(Z, self._sq * self._ones, self._Ub, neg_alpha)
            beta = self._norms(self._Ut, self._Ub)
            self._scale(self._Ut, self._Ub, 1.0 / beta)
            # alpha v = Ahat^T u - beta v
            G = self._adjoint(self._Ut, self._Ub)
        if (hasattr(be, "lsqr_tail") and G.is_cuda and G.dtype == self._V.dtype
                and os.environ.get("RK_LSQR_FUSED_TAIL", "1") != "0"):
            # v, alpha, the plane rotation and the y / w updates in one launch (~20 otherwise)
            beta = beta.contiguous()
            alpha = be.lsqr_tail(G, self._V, self._Wd, self._Y, beta, self._rhobar, self._phibar)
            if alpha is not None:
                self._alpha = alpha
                return
        be.axpby_cols(G, None, self._V, -beta)
        alpha = self._norms(self._V, None)
        be.axpby_cols(self._V, 1.0 / alpha, self._V)
        # plane rotation (k-vectors on the device)
        rho = torch.sqrt(self._rhobar * self._rhobar + beta * beta)
        c, s = self._rhobar / rho, beta / rho
        theta = s * alpha
        self._rhobar = -c * alpha
        phi = c * self._phibar
        self._phibar = s * self._phibar
        self._alpha = alpha
        # y += (phi / rho) w ;  w = v - (theta / rho) w
        be.axpby_cols(self._Wd, phi / rho, self._Y, self._ones)
        be.axpby_cols(self._V, None, self._Wd, -(theta / rho))
