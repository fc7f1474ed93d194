"""Sketch-and-project ("ASkotch") block-coordinate solver (rlaopt/solvers/sap.py:25-175).

Per iteration: draw a block of coordinates, build a block preconditioner from the
blk x blk kernel block (matrix-free, index-gathered in the kernel), estimate the step size
1 / lambda_max(P_b^-1 (K_bb + reg I)) by power iteration, evaluate the block gradient with
the blk x n row oracle and apply Nesterov-accelerated updates on the rows of the block.

The heavy products (row oracle, block oracle, Nystrom assembly / apply) run through the HIP
library; the n x k Nesterov bookkeeping is element-wise torch on device tensors.
Not reproduced: the reference's ``UnboundLocalError`` for NewtonConfig with rho != reg
(sap.py:89-111) - that case runs the power iteration like every inexact preconditioner.
"""

from __future__ import annotations

import os
from typing import TYPE_CHECKING

import torch

from .. import dist as _dist
from ..linops import LinOp
from ..preconditioners import (
    IdentityConfig, NewtonConfig, NystromConfig, Preconditioner, PreconditionerConfig,
)
from ..preconditioners import _get_precond as _pf_get_precond
from ..spectral_estimators import randomized_powering
from ..utils import quiet_host
from .configs import SAPAccelConfig
from .solver import Solver

if TYPE_CHECKING:
    from ..models import LinSys

VALID_PRECONDS = [IdentityConfig, NewtonConfig, NystromConfig]


class SAP(Solver):
    def __init__(self, system: "LinSys", W_init: torch.Tensor, precond_config: PreconditionerConfig,
                 device: torch.device, blk_sz: int, accel: bool,
                 accel_config: SAPAccelConfig | None, power_iters: int):
        self.system = system
        if type(precond_config) not in VALID_PRECONDS:
            raise TypeError(
                f"Valid preconditioner configs for SAP are {VALID_PRECONDS}, "
                f"but received {type(precond_config)}"
            )
        self.precond_config = precond_config
        self._W = W_init.clone()
        self.device = device
        self.blk_sz, self.accel, self.accel_config = blk_sz, accel, accel_config
        self.power_iters = power_iters
        n = system.A.shape[0]
        self.probs = torch.ones(n) / n
        self._side = None     # lazily created side stream (see _step)
        self._hi = None
        self._overlap = os.environ.get("RK_SAP_OVERLAP", "1") != "0"
        self._cols_cache = None  # (mask, active columns on the device, all active?) of the last step
        self._blocks = None   # optional injected block sequence (parity runs)
        self._starts = None   # optional injected power-iteration start vectors
        if accel:
            mu, nu = accel_config.mu, accel_config.nu
            self.beta = 1 - (mu / nu) ** 0.5
            self.gamma = 1 / (mu * nu) ** 0.5
            self.alpha = 1 / (1 + self.gamma * nu)
            self.V = self._W.clone()
            self.Y = self._W.clone()

    @property
    def W(self):
        return self._W

    def inject_draws(self, blocks=None, starts=None):
        """Feed externally drawn blocks / power-iteration start vectors (iterables)."""
        self._blocks = iter(blocks) if blocks is not None else None
        self._starts = iter(starts) if starts is not None else None

    def _get_precond(self, blk: torch.Tensor, K_bb=None, before_host_wait=None) -> Preconditioner:
        P = _pf_get_precond(self.precond_config)
        K_bb = self.system.A_blk_oracle(blk) if K_bb is None else K_bb
        if isinstance(self.precond_config, NystromConfig):
            # (assembly + damping in one host round trip; the block preconditioner is only ever inverted)
            P._update_block(K_bb, self.device, self.system.reg, before_host_wait=before_host_wait)
            return P
        P._update(K_bb, self.device)
        P._update_damping(baseline_rho=self.system.reg)
        return P

    def _get_blk(self) -> torch.Tensor:
        if self._blocks is not None:
            return next(self._blocks)
        # The reference draws torch.multinomial(uniform probs, blk, replacement=False) on the
        # host (sap.py:76-86; 62 ms at n = 500 000, numpy fallback past 2**24 categories).  With
        # uniform probabilities that is a uniformly random subset: draw it as the head of a
        # random permutation on the solver's device (seeded from torch's default generator).
        # In an SPMD run every rank must pick the SAME block: the seed comes from rank 0.
        n = self.probs.shape[0]
        seed = _dist.shared_randint(1)[0]
        if self._W.is_cuda and os.environ.get("RK_SAP_DRAW", "feistel") != "randperm":
            from ..ops import get_backend

            be = get_backend()
            if hasattr(be, "sample_distinct"):
                # one launch: the head of a keyed pseudo-random permutation of [0, n) (the sort-based
                # randperm below is 27 launches, 0.19 - 0.35 ms of every step at n = 500 000)
                return be.sample_distinct(n, self.blk_sz, self._W.device, seed)
        gen = torch.Generator(device=self._W.device)
        gen.manual_seed(seed)
        return torch.randperm(n, device=self._W.device, generator=gen)[: self.blk_sz]

    def _fused_stepsize(self, K_bb, blk_precond, v0):
        """1 / lambda_max on the device, no host synchronisation (``rk_power_iteration``): needs the
        block oracle as one resident tensor and a preconditioner whose inverse is I or
        (I - U M U^T) / rho.  Returns a 0-dim tensor, or None when this layout does not apply."""
        from ..ops import get_backend
        from ..preconditioners.identity import Identity
        from ..preconditioners.nystrom import Nystrom

        be = get_backend()
        dense = getattr(K_bb, "_rk_dense", None)
        if (dense is None or not hasattr(be, "power_iteration") or not dense.is_cuda
                or os.environ.get("RK_SAP_FUSED_POWER", "1") == "0"):
            return None
        dt = self._W.dtype
        if v0 is None:
            from ..sketches import _device_randn  # the draw randomized_powering would make

            v0 = _device_randn(self.blk_sz, 1, dt, dense.device)[:, 0]
        v0 = v0.to(device=dense.device, dtype=dt).contiguous().clone()
        if isinstance(blk_precond, Nystrom) and not blk_precond._sharded:
            state = be.power_iteration(dense, self.system.reg, v0, U=blk_precond.U,
                                       M=blk_precond._apply_factor(), rho=blk_precond._rho_now(),
                                       max_iters=self.power_iters, rtol=1e-3)
        elif isinstance(blk_precond, Identity):
            state = be.power_iteration(dense, self.system.reg, v0, max_iters=self.power_iters, rtol=1e-3)
        else:
            return None
        return state[0].reciprocal().to(dt)

    def _get_stepsize(self, blk: torch.Tensor, blk_precond: Preconditioner, K_bb=None):
        if isinstance(self.precond_config, NewtonConfig) and self.precond_config.rho == self.system.reg:
            return 1.0  # exact block solve
        if K_bb is None:
            K_bb = self.system.A_blk_oracle(blk)
        reg = self.system.reg
        v0 = next(self._starts).to(device=self.device, dtype=self._W.dtype) if self._starts else None
        if v0 is None and _dist.spmd_active():
            # one rank per GPU: the start vector must be the SAME on every rank (a rank-local
            # torch.randn, as in spectral_norm.py:16, would give each rank its own step size)
            from ..sketches import _device_randn

            v0 = _device_randn(self.blk_sz, 1, self._W.dtype, self.device)[:, 0]
        eta = self._fused_stepsize(K_bb, blk_precond, v0)
        if eta is not None:
            return eta
        S = LinOp(
            device=self.device,
            shape=torch.Size((self.blk_sz, self.blk_sz)),
            matvec=blk_precond._inverse_matmul_compose(lambda v: K_bb @ v + reg * v),
            dtype=self._W.dtype,
        )
        max_eig, _ = randomized_powering(S, max_iters=self.power_iters, v0=v0)
        return float(max_eig) ** -1.0

    def _precond_and_stepsize(self, blk, before_host_wait=None):
        """Block preconditioner and step size from ONE evaluation of the block oracle (the
        reference builds K(X[blk], X[blk]) twice per step, sap.py:64-66 and :95)."""
        K_bb = self.system.A_blk_oracle(blk)
        blk_precond = self._get_precond(blk, K_bb, before_host_wait)
        return blk_precond, self._get_stepsize(blk, blk_precond, K_bb)

    def _side_stream(self):
        """Second HIP stream for the block gradient (single-process GPU runs only: in an SPMD run
        the row-oracle product contains a collective, which stays in program order)."""
        if not self._W.is_cuda or _dist.spmd_active() or not self._overlap:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self._W.device)
        return self._side

    def _chain_stream(self):
        """High-priority stream for the preconditioner / step-size chain while the gradient kernel
        runs (measured 14.6 -> 14.0 ms per ASkotch step; the big kernel's 4 000 workgroups keep the
        dispatcher busy, so the overlap stays partial - reserving CUs with a CU-masked stream was
        slower).  RK_SAP_PRIORITY=0 keeps the chain on the current stream."""
        if os.environ.get("RK_SAP_PRIORITY", "1") == "0":
            return None
        if self._hi is None:
            self._hi = torch.cuda.Stream(device=self._W.device, priority=-1)
        return self._hi

    def _block_gradient(self, W, B, blk):
        return self.system.A_row_oracle(blk) @ W + self.system.reg * W[blk, :] - B[blk, :]

    def _get_block_update(self, W, B, blk, blk_precond):
        return blk_precond._inv @ self._block_gradient(W, B, blk)

    def _fused_update(self, cols, blk_d, d, eta) -> bool:
        """All columns active and device-resident iterates: the whole Nesterov bookkeeping in two
        launches (``rk_sap_update``) instead of ~12 element-wise / index_put kernels."""
        from ..ops import get_backend

        be = get_backend()
        if (not hasattr(be, "sap_update") or not self._W.is_cuda or cols.numel() != self._W.shape[1]
                or os.environ.get("RK_SAP_FUSED_UPDATE", "1") == "0"):
            return False
        if self.accel:
            return be.sap_update(self._W, self.V, self.Y, blk_d, d, eta, True, self.beta, self.gamma, self.alpha)
        return be.sap_update(self._W, None, None, blk_d, d, eta, False)

    def _step(self):
        with quiet_host():
            self._step_impl()

    def _step_impl(self):
        mask = self.system.mask
        cached = self._cols_cache
        if cached is None or cached[0].shape != mask.shape or not torch.equal(cached[0], mask):
            # (the set of active columns changes a handful of times per solve: one host -> device copy then,
            # not one per step)
            cached = (mask.clone(), torch.nonzero(mask)[:, 0].to(self._W.device), bool(mask.all()))
            self._cols_cache = cached
        _, cols, all_active = cached
        if cols.numel() == 0:
            return
        blk = self._get_blk()
        blk_d = blk.to(self._W.device)
        base = self.Y if self.accel else self._W
        # all columns active: the iterate and B themselves, not 20 MB copies of them per step
        loc = base if all_active else base[:, cols]
        Bc = self.system.B if all_active else self.system.B[:, cols]
        side = self._side_stream()
        if side is not None:
            # The block gradient (the blk x n row-oracle product: the largest kernel of the step)
            # depends on neither the block preconditioner nor the step size; it runs on a side stream
            # while the host factorises the r x r blocks of the preconditioner.  It is queued at the
            # assembly's first host wait rather than up front: kernels of the latency-bound chain that
            # share the device with it run ~4x slower, so the chain's device-only head goes first.
            # (A software pipeline over steps - block, preconditioner and step size of step k + 1 prepared
            # beside step k's gradient; they depend on the draws only - gives bitwise the same iterates and
            # was measured twice: round 2, and round 3 with the gradient at 1.5 ms and two workgroups per CU:
            # 2.16 against 2.07 ms per step.  The chain's kernels slow down beside the product by as much as
            # the overlap saves.)
            main = torch.cuda.current_stream(self._W.device)
            box = {}

            def launch_grad():
                if "grad" not in box:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        box["grad"] = self._block_gradient(loc, Bc, blk_d)

            early = os.environ.get("RK_SAP_GRAD_FIRST", "0") == "1"
            if early:
                launch_grad()
            hi = self._chain_stream()
            if hi is not None:  # the latency-bound chain on a high-priority stream
                hi.wait_stream(main)
                with torch.cuda.stream(hi):
                    blk_precond, eta = self._precond_and_stepsize(blk, launch_grad)
                main.wait_stream(hi)
            else:
                blk_precond, eta = self._precond_and_stepsize(blk, launch_grad)
            launch_grad()  # (preconditioners that never wait on the host)
            grad = box["grad"]
            main.wait_stream(side)
            grad.record_stream(main)
            if isinstance(eta, torch.Tensor):
                eta.record_stream(main)
        else:
            blk_precond, eta = self._precond_and_stepsize(blk)
            grad = self._block_gradient(loc, Bc, blk_d)
        d = blk_precond._inv @ grad
        if self._fused_update(cols, blk_d, d, eta):
            return
        if not self.accel:
            Wm = self._W[:, cols]
            Wm[blk_d] -= eta * d
            self._W[:, cols] = Wm
            return
        Wm = loc.clone()                      # W <- Y, then the block step
        Wm[blk_d] -= eta * d
        Vm = self.beta * self.V[:, cols] + (1 - self.beta) * loc
        Vm[blk_d] -= (eta * self.gamma) * d
        self._W[:, cols] = Wm
        self.V[:, cols] = Vm
        self.Y[:, cols] = self.alpha * Vm + (1 - self.alpha) * Wm
