"""``LinSys``: the regularised system (A + reg I) W = B (rlaopt/models/linsys.py:14-159).

Public behaviour follows the reference: B is made 2-D, the convergence mask starts all
True, metrics are per-column ||B - (A W + reg W)||_2 and its ratio to ||B||_2, a column is
finished when abs_res <= max(rtol ||b||, atol).  ``_exec_context()`` is this build's
addition: it hides how (A + reg I) X is evaluated - one fused streaming product for dense
tensors and materialised kernel operators, a row-sharded product plus collectives for an
SPMD operator, ``A @ X`` for any other operator - so that the solvers are written once.
"""

from __future__ import annotations

import os
from typing import Any, Callable

import torch

from .. import dist as _dist
from ..linops import _is_linop_or_torch_tensor
from ..ops import get_backend
from ..solvers import _get_solver, _get_solver_name, _is_solver_config
from ..utils import Logger, _is_callable, _is_nonneg_float, _is_torch_tensor
from .model import Model

__all__ = ["LinSys"]


class _ExecContext:
    """Evaluation of (A + reg I) X, residuals and k x k Gram blocks for one system."""

    def __init__(self, system: "LinSys"):
        self.A, self.reg = system.A, float(system.reg)
        self.be = get_backend()
        self.n = system.B.shape[0]
        self.sharded = bool(getattr(self.A, "is_row_sharded", False))
        self.group = getattr(self.A, "_group", None)
        if self.sharded:
            self.lo, self.hi = self.A.row_range
            self.dev = self.A.local_device
        self.B = self.take_rows(system.B)

    def take_rows(self, X: torch.Tensor) -> torch.Tensor:
        return X[self.lo : self.hi].to(self.dev) if self.sharded else X

    def gather_rows(self, X_loc: torch.Tensor) -> torch.Tensor:
        return _dist.all_gather_rows(X_loc, self.n, self.group) if self.sharded else X_loc

    def shifted_matmat(self, X_loc: torch.Tensor) -> torch.Tensor:
        A, reg = self.A, self.reg
        if self.sharded:
            return A.local_matmat(self.gather_rows(X_loc), diag_shift=reg)
        if isinstance(A, torch.Tensor):
            return self.be.matmat(A, X_loc, diag_shift=reg)
        fused = getattr(A, "_rk_shifted_matmat", None)
        if fused is not None:
            Y = fused(X_loc, reg)
            if Y is not None:
                return Y
        Y = A @ X_loc  # opaque user operator
        return Y + reg * X_loc if reg != 0.0 else Y

    def residual(self, W_loc: torch.Tensor) -> torch.Tensor:
        if isinstance(self.A, torch.Tensor) and not self.sharded:
            return self.be.matmat(self.A, W_loc, alpha=-1.0, beta=1.0, C=self.B, diag_shift=-self.reg)
        return self.B - self.shifted_matmat(W_loc)

    def gram(self, X: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
        G = self.be.rmatmat(X, Y, out_f64=True)
        if self.sharded:
            _dist.all_reduce_sum(G, self.group)
        return G

    def shifted_matmat_with_gram(self, X_loc: torch.Tensor):
        """((A + reg I) X)[local rows] and X^T (A + reg I) X.  Row-sharded operators that can
        fold the Gram block into the product's own collective do so (one all-reduce instead of
        two per PCG iteration); everything else computes the two separately."""
        fused = getattr(self.A, "_rk_spmd_matmat_gram", None) if self.sharded else None
        if fused is not None:
            out = fused(self.gather_rows(X_loc), self.reg)
            if out is not None:
                return out
        AX = self.shifted_matmat(X_loc)
        return AX, self.gram(X_loc, AX)

    def precond_apply_with_gram(self, P, R_loc: torch.Tensor):
        """Z = P^-1 R and R^T Z; one collective when the (row-sharded) preconditioner can fuse
        them (Nystrom), otherwise apply + separate Gram."""
        fused = getattr(P, "_inverse_matmul_with_gram", None) if self.sharded else None
        if fused is not None and R_loc.ndim == 2:
            return fused(R_loc)
        Z = P._inv @ R_loc
        return Z, self.gram(R_loc, Z)

    def col_norms(self, X: torch.Tensor) -> torch.Tensor:
        return self.gram(X, X).diagonal().clamp_min(0.0).sqrt()


def _device_sync(t: torch.Tensor):
    """What the logger calls before reading the clock (utils/logger.py)."""
    if not t.is_cuda:
        return None
    dev = t.device
    return lambda: torch.cuda.synchronize(dev)


class LinSys(Model):
    """Positive-definite linear system (A + reg * I) W = B."""

    def __init__(self, A, B: torch.Tensor, reg: float = 0.0,
                 A_row_oracle: Callable | None = None, A_blk_oracle: Callable | None = None):
        self._check_inputs(A, B, reg, A_row_oracle, A_blk_oracle)
        self._A = A
        self._B = B[:, None] if B.ndim == 1 else B
        self._reg = reg
        self._A_row_oracle, self._A_blk_oracle = A_row_oracle, A_blk_oracle
        self._mask = torch.ones(self._B.shape[1], dtype=torch.bool)
        self._ctx = None
        self._B_norms = None
        self._active_solver = None  # lets the metric share one operator pass with the solver

    A = property(lambda self: self._A)
    B = property(lambda self: self._B)
    reg = property(lambda self: self._reg)
    A_row_oracle = property(lambda self: self._A_row_oracle)
    A_blk_oracle = property(lambda self: self._A_blk_oracle)
    mask = property(lambda self: self._mask)

    def _check_inputs(self, A: Any, B: Any, reg: Any, A_row_oracle: Any, A_blk_oracle: Any):
        _is_linop_or_torch_tensor(A, "A")
        _is_torch_tensor(B, "B")
        _is_nonneg_float(reg, "reg")
        for fn, name in ((A_row_oracle, "A_row_oracle"), (A_blk_oracle, "A_blk_oracle")):
            if fn is not None:
                _is_callable(fn, name)
        if (A_row_oracle is None) != (A_blk_oracle is None):  # both or neither
            missing = "A_blk_oracle" if A_blk_oracle is None else "A_row_oracle"
            given = "A_row_oracle" if A_blk_oracle is None else "A_blk_oracle"
            raise ValueError(f"{missing} must be provided if {given} is provided")

    def _exec_context(self) -> _ExecContext:
        if self._ctx is None:
            self._ctx = _ExecContext(self)
        return self._ctx

    def _compute_internal_metrics(self, W: torch.Tensor):
        ctx = self._exec_context()
        if self._B_norms is None:
            self._B_norms = ctx.col_norms(ctx.B)
        W_rows = ctx.take_rows(W)
        fused = getattr(self._active_solver, "_fused_log_product", None)
        AW = fused(W_rows) if fused is not None else None
        R = ctx.residual(W_rows) if AW is None else ctx.B - AW
        abs_res = ctx.col_norms(R)
        dt = self._B.dtype
        return {"abs_res": abs_res.to(dt), "rel_res": (abs_res / self._B_norms).to(dt)}

    def _check_termination_criteria(self, internal_metrics: dict, atol: float, rtol: float):
        abs_res = internal_metrics["abs_res"].double()
        if self._B_norms is None:
            ctx = self._exec_context()
            self._B_norms = ctx.col_norms(ctx.B)
        tol = torch.clamp(rtol * self._B_norms, min=atol)
        self._mask = (abs_res > tol).cpu()  # the one host read per logged iteration
        return not bool(self._mask.any())

    def solve(self, solver_config, W_init, callback_fn=None, callback_args=[], callback_kwargs={},
              callback_freq=10, log_in_wandb=False, wandb_init_kwargs=None):
        _is_solver_config(solver_config, "solver_config")
        _is_torch_tensor(W_init, "W_init")
        if log_in_wandb and wandb_init_kwargs is None:
            raise ValueError("wandb_init_kwargs must be specified if log_in_wandb is True")
        if (getattr(self._A, "_inprocess_multi_device", False) and callback_fn is None and not log_in_wandb
                and os.environ.get("RK_MULTI_DEVICE", "spmd") != "inprocess"):
            # one process naming several GPUs (the reference's own call): run the solve as one fresh worker process
            # per device - the one-rank-per-GPU layout this engine is built for - and hand back rank 0's result
            from .. import spmd

            own_oracles = (self._A_row_oracle is not None
                           and getattr(self._A_row_oracle, "__self__", None) is self._A
                           and getattr(self._A_blk_oracle, "__self__", None) is self._A)
            if self._A_row_oracle is None or own_oracles:
                W, log = spmd.solve_kernel_system(self._A, self._B, self._reg, solver_config, W_init,
                                                  callback_freq=callback_freq, with_oracles=own_oracles)
                return W.to(W_init.device), log
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
        plan = getattr(self._A, "_rk_plan", None)
        if plan is not None and _get_solver_name(solver_config) == "pcg":
            # how many operator products this solve will ask for at most: one per iteration, one per logged residual
            # (fused with an iteration's when narrow), the sketch counted as rank / 16 passes - a storage decision the
            # operator may still take (kernels/base.py::_prefer_matrix_free)
            rank = getattr(getattr(solver_config, "precond_config", None), "rank", 0) or 0
            plan(solver_config.max_iters * (1.0 + 1.0 / max(callback_freq, 1)) + rank / 16.0)
        solver = _get_solver(model=self, W_init=W_init, solver_config=solver_config)
        self._active_solver = solver
        return self._train(
            logger=logger,
            termination_fn=lambda m: self._check_termination_criteria(m, atol, rtol),
            solver=solver,
            max_iters=solver_config.max_iters,
        )
