"""``LeastSquares``: minimise ||A w - b||^2 + reg ||w||^2 for a tall operator A (m x d).

No reference counterpart; the class follows the ``LinSys`` pattern
(rlaopt/models/linsys.py:14-159): same ``solve`` signature, same log layout, same
mask / termination cadence.  Internal metrics are the normal-equation residual
``||A^T (b - A w) - reg w||`` per column (``abs_res``), its ratio to ``||A^T b||`` (``rel_res``)
and the data misfit ``||b - A w||`` (``misfit``)."""

from __future__ import annotations

import torch

from .. import dist as _dist
from ..linops import _is_linop_or_torch_tensor
from ..ops import get_backend
from ..solvers import _get_solver, _get_solver_name, _is_solver_config
from ..utils import Logger, _is_nonneg_float, _is_torch_tensor
from .linsys import _device_sync
from .model import Model

__all__ = ["LeastSquares"]


class LeastSquares(Model):
    def __init__(self, A, B: torch.Tensor, reg: float = 0.0):
        self._check_inputs(A, B, reg)
        self._A = A
        self._B = B[:, None] if B.ndim == 1 else B
        self._reg = reg
        self._mask = torch.ones(self._B.shape[1], dtype=torch.bool)
        self._g0 = None
        self._ones = None

    A = property(lambda self: self._A)
    B = property(lambda self: self._B)
    reg = property(lambda self: self._reg)
    mask = property(lambda self: self._mask)

    def _check_inputs(self, A, B, reg):
        _is_linop_or_torch_tensor(A, "A")
        _is_torch_tensor(B, "B")
        _is_nonneg_float(reg, "reg")
        if B.shape[0] != A.shape[0]:
            raise ValueError(f"B has {B.shape[0]} rows, A has {A.shape[0]}")

    # products with A: fused streaming kernels for a dense tensor, the operator protocol otherwise
    def _matmat(self, X):
        return get_backend().matmat(self._A, X) if isinstance(self._A, torch.Tensor) else self._A @ X

    def _rmatmat(self, U):
        if isinstance(self._A, torch.Tensor):
            return get_backend().rmatmat(self._A, U)
        return self._A.T @ U

    def _normal_pass(self, Z, U, coef):
        """U <- A Z - U diag(coef) and (A^T U_new, column sums of squares) from ONE pass over a
        dense A, or None when A is an operator / the shape is unsupported."""
        if not isinstance(self._A, torch.Tensor):
            return None
        fn = getattr(get_backend(), "normal_pass", None)
        return fn(self._A, Z, U, coef) if fn is not None else None

    def _residual(self, W):
        if isinstance(self._A, torch.Tensor):
            return get_backend().matmat(self._A, W, alpha=-1.0, beta=1.0, C=self._B)
        return self._B - self._A @ W

    def _col_norms(self, X):
        return get_backend().rmatmat(X, X, out_f64=True).diagonal().clamp_min(0.0).sqrt()

    def _compute_internal_metrics(self, W: torch.Tensor):
        """Normal-equations residual |A^T (b - A w) - reg w| (absolute, and relative to |A^T b|)
        and the misfit |b - A w| per column.  For a dense A both products come out of one sweep
        (rk_normal_pass with U = b, coef = 1: U <- A w - b = -r, G = -A^T r, ss = |r|^2)."""
        if self._g0 is None:
            self._g0 = self._col_norms(self._rmatmat(self._B))
        k = self._B.shape[1]
        fused = None
        if isinstance(self._A, torch.Tensor) and W.ndim == 2:
            if self._ones is None or self._ones.device != self._B.device:
                self._ones = torch.ones(k, dtype=torch.float64, device=self._B.device)
            fused = self._normal_pass(W, self._B.clone(), self._ones)
        if fused is not None:
            G, ss = fused
            G = (-G).to(self._B.dtype)
            misfit = ss.clamp_min(0.0).sqrt()
        else:
            R = self._residual(W)
            G = self._rmatmat(R)
            misfit = self._col_norms(R)
        if self._reg:
            G = G - self._reg * W
        abs_res = self._col_norms(G)
        dt = self._B.dtype
        return {"abs_res": abs_res.to(dt), "rel_res": (abs_res / self._g0).to(dt),
                "misfit": misfit.to(dt)}

    def _check_termination_criteria(self, internal_metrics: dict, atol: float, rtol: float):
        tol = torch.clamp(rtol * self._g0, min=atol)
        self._mask = (internal_metrics["abs_res"].double() > tol).cpu()
        return not bool(self._mask.any())

    def solve(self, solver_config, W_init, callback_fn=None, callback_args=[], callback_kwargs={},
              callback_freq=10, log_in_wandb=False, wandb_init_kwargs=None):
        _is_solver_config(solver_config, "solver_config")
        _is_torch_tensor(W_init, "W_init")
        if log_in_wandb and wandb_init_kwargs is None:
            raise ValueError("wandb_init_kwargs must be specified if log_in_wandb is True")
        _dist.reset_seed_stream()  # (one rank per GPU: this solve's draws follow rank 0's generator as of now)
        atol, rtol = solver_config.atol, solver_config.rtol
        logger = Logger(
            log_freq=callback_freq,
            log_fn=self._get_log_fn(callback_fn, callback_args, callback_kwargs),
            wandb_kwargs=self._get_wandb_kwargs(
                log_in_wandb=log_in_wandb, wandb_init_kwargs=wandb_init_kwargs,
                solver_name=_get_solver_name(solver_config), solver_config=solver_config,
                callback_freq=callback_freq,
            ),
            sync=_device_sync(self._B),
        )
        solver = _get_solver(model=self, W_init=W_init, solver_config=solver_config)
        return self._train(
            logger=logger,
            termination_fn=lambda m: self._check_termination_criteria(m, atol, rtol),
            solver=solver,
            max_iters=solver_config.max_iters,
        )
