"""Masked block PCG (rlaopt/solvers/pcg.py:13-93), device-resident.

The recurrences are the reference's true block CG (k x k alpha and beta):

    AP = (A + reg I) P            alpha = (P^T AP)^-1 RZ      W += P alpha    R -= AP alpha
    Z  = Pinv R                   RZ' = R^T Z                 beta = RZ^-1 RZ'
    P  = Z + P beta

but nothing n-long is ever touched by the host: AP is one streaming product with the
"+ reg P" term fused in, the k x k Gram blocks come out of tall-skinny products in float64,
the two k x k solves run in a one-workgroup kernel on the device, and the W/R and P updates
are fused element-wise kernels reading alpha / beta from device memory - so an iteration is
a fixed sequence of launches with no host synchronisation (the host only looks at data on
the logging iterations, like the reference's termination test).

Converged columns (system.mask False) are dropped by compacting the n x k state to the
active columns when the mask changes (the reference re-slices every array on every step,
pcg.py:54-90).  With a row-sharded operator the state is row-partitioned across ranks.  Per
iteration: one all-gather of the search directions, one all-reduce carrying the partial product
together with P^T AP, and one carrying U^T R together with R^T R (from which R^T Z follows) - the
Gram blocks ride in the collectives the product and the preconditioner need anyway.
"""

from __future__ import annotations

import os
from typing import TYPE_CHECKING

import torch

from ..ops import get_backend
from ..preconditioners import PreconditionerConfig
from ..preconditioners import _get_precond as _pf_get_precond
from ..utils import quiet_host
from .solver import Solver

if TYPE_CHECKING:
    from ..models import LinSys


class PCG(Solver):
    def __init__(self, system: "LinSys", W_init: torch.Tensor, precond_config: PreconditionerConfig,
                 device: torch.device):
        self.system = system
        self.precond_config = precond_config
        self.device = device
        self._be = get_backend()
        ctx = system._exec_context()
        self._ctx = ctx
        if W_init.ndim != 2 or W_init.shape != system.B.shape:
            # linsys.py:39-40 unsqueezes B but not W_init; a 1-D W_init silently broadcasts in
            # the reference - refuse it instead
            raise ValueError(
                f"W_init must have shape {tuple(system.B.shape)}, got {tuple(W_init.shape)}"
            )
        k = system.B.shape[1]
        self._k = k
        self._W_all = ctx.take_rows(W_init).clone().contiguous()  # n_loc x k, every column
        self._active = list(range(k))
        self.P = self._get_precond()
        self._Wc = self._W_all.clone()
        self._Rc = ctx.residual(self._Wc)  # B - (A W + reg W), local rows
        Z, self.RZ = ctx.precond_apply_with_gram(self.P, self._Rc)  # RZ: k x k float64 on the device
        self._Pc = Z.clone()
        self._AP_next = None  # (A + reg I) P computed ahead of time by _fused_log_product
        # opt-in.  Measured on one MI355X at n = 50 000 (0.94 ms / iteration, tail 0.11 ms): no gain -
        # the host enqueues far ahead of the device, so the dependent launches already run back to
        # back; it only pays when the host is the bottleneck.  And graph capture shares the process
        # with whatever else holds device objects (a garbage collection freeing one mid-capture
        # aborts the process), so it is not on by default.
        self._use_graph = os.environ.get("RK_PCG_GRAPH", "0") == "1"
        self._graph, self._graph_key, self._graph_AP, self._eager_tails = None, None, None, 0

    # reference attribute names, as full arrays (debug / tests)
    @property
    def W(self) -> torch.Tensor:
        self._W_all[:, self._active] = self._Wc
        return self._ctx.gather_rows(self._W_all)

    @property
    def R(self) -> torch.Tensor:
        return self._ctx.gather_rows(self._Rc)

    @property
    def P_(self) -> torch.Tensor:
        return self._ctx.gather_rows(self._Pc)

    def _get_precond(self):
        P = _pf_get_precond(self.precond_config)
        P._update(self.system.A, self.device)
        P._update_damping(baseline_rho=self.system.reg)
        return P

    # widest fused product that is still cheaper than two passes over the operator (see
    # _fused_log_product): float32 has one-pass streaming kernels up to 16 columns (packed symmetric:
    # 17.7 ms against 12.3 + 15.1 ms for two passes at n = 200 000), float64 up to 4
    _FUSE_MAX_COLS = {torch.float32: 16, torch.float64: 4}

    def _fused_log_product(self, W_rows: torch.Tensor):
        """(A + reg I) [P, W] in ONE pass over the operator on a logging iteration: the W part
        feeds the true-residual metric (the reference spends a whole extra product on it,
        models/linsys.py:96-99), the P part is kept for the next step, which then skips its
        own product.  Returns (A + reg I) W for the local rows, or None when the combined
        width would leave the fast streaming kernels."""
        ka = len(self._active)
        widest = self._FUSE_MAX_COLS.get(W_rows.dtype, 4)
        if ka == 0 or ka + W_rows.shape[1] > widest or self._AP_next is not None:
            return None
        Y = self._ctx.shifted_matmat(torch.cat([self._Pc, W_rows], dim=1))
        self._AP_next = Y[:, :ka].contiguous()
        return Y[:, ka:]

    def _compact(self, mask) -> None:
        want = [j for j in range(self._k) if bool(mask[j])]
        want = [j for j in want if j in self._active]  # a finished column never restarts
        if want == self._active:
            return
        self._W_all[:, self._active] = self._Wc
        pos = [self._active.index(j) for j in want]
        idx = torch.tensor(pos, device=self._Wc.device)
        self._Wc = self._Wc[:, idx].contiguous()
        self._Rc = self._Rc[:, idx].contiguous()
        self._Pc = self._Pc[:, idx].contiguous()
        if self._AP_next is not None:
            self._AP_next = self._AP_next[:, idx].contiguous()
        self.RZ = self.RZ[idx][:, idx].contiguous()
        self._active = want

    def _step(self):
        with quiet_host():  # (a no-op inside LinSys.solve, which already holds it)
            self._step_impl()

    def _step_impl(self):
        mask = self.system.mask
        if not bool(mask.any()):
            return
        self._compact(mask)
        if not self._active:
            return
        be, ctx = self._be, self._ctx
        AP, self._AP_next = self._AP_next, None
        if AP is None:
            AP, PAP = ctx.shifted_matmat_with_gram(self._Pc)  # (A + reg I) P, local rows; P^T AP
        else:
            PAP = ctx.gram(self._Pc, AP)
        if not self._tail_replay(AP, PAP):
            self.RZ = self._tail(AP, PAP)

    def _tail(self, AP, PAP):
        """Everything after the operator product: alpha, W/R update, preconditioner apply, beta,
        P update.  ~12 dependent launches on n x k vectors: launch-latency-bound once the product
        is short (small systems, many GPUs).  Returns the new R^T Z."""
        be, ctx = self._be, self._ctx
        if PAP is None:
            PAP = ctx.gram(self._Pc, AP)
        alpha = be.small_solve(PAP, self.RZ)
        be.pcg_update_wr(self._Wc, self._Rc, self._Pc, AP, alpha)
        Z, RZ_new = ctx.precond_apply_with_gram(self.P, self._Rc)
        beta = be.small_solve(self.RZ, RZ_new)
        be.pcg_update_p(self._Pc, Z, beta)
        return RZ_new

    # -- HIP graph of the tail -------------------------------------------------------------------
    def _tail_graph_allowed(self) -> bool:
        return (self._use_graph and self._Wc.is_cuda and not self._ctx.sharded
                and type(self._be).__name__ == "HipBackend")

    def _tail_replay(self, AP, PAP) -> bool:
        """Replay the captured tail (capturing it first when the state buffers changed).  The
        graph reads AP from a fixed buffer and leaves R^T Z in ``self.RZ`` in place, so nothing
        it touches is re-bound between replays.  False -> caller runs the eager tail."""
        if not self._tail_graph_allowed():
            return False
        self._eager_tails += 1
        if self._eager_tails <= 2:  # first iterations eager: one-time allocations and module loads
            return False
        key = (self._Wc.data_ptr(), self._Rc.data_ptr(), self._Pc.data_ptr(), self.RZ.data_ptr(),
               tuple(self._Wc.shape), id(self.P))
        try:
            if self._graph is None or self._graph_key != key:
                import gc

                self._graph_AP = torch.empty_like(AP)
                self._graph_AP.copy_(AP)
                g = torch.cuda.CUDAGraph()
                gc_was_on = gc.isenabled()
                gc.disable()  # no destructor of a device object may run while the stream captures
                try:
                    with torch.cuda.graph(g):
                        rz = self._tail(self._graph_AP, None)
                        self.RZ.copy_(rz)
                finally:
                    if gc_was_on:
                        gc.enable()
                self._graph, self._graph_key = g, key
                # capture only records: the state is unchanged, fall through to the replay
            else:
                self._graph_AP.copy_(AP)
            self._graph.replay()
            return True
        except Exception as e:  # capture not possible here: stay eager for good
            import warnings

            warnings.warn(f"rlaopt_amd: PCG tail graph disabled ({type(e).__name__}: {e})")
            self._use_graph = False
            self._graph = None
            return False
