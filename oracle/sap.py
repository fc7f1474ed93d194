"""Oracle: sketch-and-project ("ASkotch") block-coordinate solver.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Restates rlaopt/solvers/sap.py:25-175 and rlaopt/spectral_estimators/spectral_norm.py:
11-29.  Random draws are made with the same torch calls in the same order as the
reference (block: ``torch.multinomial``; sketch: oracle/sketches.py; power-iteration
start: ``torch.randn(d)`` in the DEFAULT dtype, spectral_norm.py:16), so with one seed
on the CPU the oracle and the imported reference follow the same trajectory.

Deliberately NOT reproduced: the latent ``UnboundLocalError`` for a Newton
preconditioner with rho != reg (sap.py:89-111); here that case runs the power
iteration like every other inexact preconditioner.
"""

from __future__ import annotations

import torch

from . import nystrom as _nys
from . import sketches as _sk


def randomized_powering(apply_S, d: int, device, max_iters: int = 10, rtol: float = 1e-3, v0=None):
    """spectral_norm.py:11-29: power iteration with Rayleigh quotient v^T (S v)."""
    omega = torch.randn(d, device=device) if v0 is None else v0
    v = omega / torch.linalg.norm(omega, 2)
    i, err, sig = 0, torch.inf, 0.0
    sig_new = None
    while i < max_iters and err > rtol * sig:
        v_new = apply_S(v)
        sig_new = torch.dot(v, v_new)
        v = v_new / torch.linalg.norm(v_new, 2)
        err = torch.abs(sig_new - sig)
        sig = sig_new
        i += 1
    return sig_new, v


class SAP:
    """``row_oracle(blk)`` and ``blk_oracle(blk)`` return objects supporting ``@``.

    precond: ("identity",) | ("newton", rho) | ("nystrom", rank, rho, sketch, adaptive)
    """

    def __init__(
        self,
        n,
        row_oracle,
        blk_oracle,
        B,
        reg,
        W_init,
        precond,
        blk_sz,
        accel=True,
        mu=None,
        nu=None,
        power_iters=10,
        device="cpu",
        blk_sequence=None,
        omega_sequence=None,
        start_sequence=None,
    ):
        self.n, self.row_oracle, self.blk_oracle = n, row_oracle, blk_oracle
        self.B, self.reg = B, reg
        self.W = W_init.clone()
        self.precond, self.blk_sz = precond, blk_sz
        self.accel, self.power_iters, self.device = accel, power_iters, device
        self.probs = torch.ones(n) / n  # sap.py:54
        self._blk_iter = iter(blk_sequence) if blk_sequence is not None else None
        # (pre-drawn test matrices / power-iteration starts: parity runs that share every draw with the
        # product; None = drawn here with the reference's own torch calls)
        self._omega_iter = iter(omega_sequence) if omega_sequence is not None else None
        self._start_iter = iter(start_sequence) if start_sequence is not None else None
        if accel:  # sap.py:58-64
            self.beta = 1 - (mu / nu) ** 0.5
            self.gamma = 1 / (mu * nu) ** 0.5
            self.alpha = 1 / (1 + self.gamma * nu)
            self.V = self.W.clone()
            self.Y = self.W.clone()

    # -- pieces ------------------------------------------------------------------------
    def _draw_block(self):
        if self._blk_iter is not None:
            return next(self._blk_iter)
        return torch.multinomial(self.probs, self.blk_sz, replacement=False)  # sap.py:78

    def _block_precond(self, blk):
        """sap.py:70-74 -> (apply_Pinv, exact?)"""
        Kbb = self.blk_oracle(blk)
        kind = self.precond[0]
        if kind == "identity":
            return (lambda x: x), False
        if kind == "newton":
            rho = self.precond[1]
            eye = torch.eye(self.blk_sz, dtype=self.B.dtype, device=self.device)
            M = Kbb @ eye  # newton.py:66-71
            M = M + rho * eye
            L = torch.linalg.cholesky(M, upper=False)

            def pinv(x):
                X = x[:, None] if x.ndim == 1 else x
                w = torch.linalg.solve_triangular(L, X, upper=False)
                w = torch.linalg.solve_triangular(L.T, w, upper=True)
                return w[:, 0] if x.ndim == 1 else w

            return pinv, rho == self.reg
        _, rank, rho, sketch, adaptive = self.precond
        if self._omega_iter is not None:
            Om = next(self._omega_iter)
        else:
            Om = _sk.get_sketch_matrix(sketch, "right", rank, self.blk_sz, self.B.dtype, self.device)
        st = _nys.build(Kbb, Om, rho)
        _nys.update_damping(st, self.reg, adaptive)
        return (lambda x: _nys.inverse_matmul(st, x)), False

    def _stepsize(self, blk, pinv, exact):
        """sap.py:88-111: 1 / lambda_max(P_b^-1 (K_bb + reg I))."""
        if exact:
            return 1.0
        Kbb = self.blk_oracle(blk)
        sig, _ = randomized_powering(
            lambda v: pinv(Kbb @ v + self.reg * v),
            self.blk_sz,
            self.device,
            max_iters=self.power_iters,
            v0=next(self._start_iter) if self._start_iter is not None else None,
        )
        return sig ** (-1.0)

    # -- one iteration -------------------------------------------------------------------
    def step(self, mask):
        """sap.py:129-175."""
        if not bool(mask.any()):
            return
        blk = self._draw_block()
        pinv, exact = self._block_precond(blk)
        eta = self._stepsize(blk, pinv, exact)
        loc = self.Y[:, mask] if self.accel else self.W[:, mask]
        Bm = self.B[:, mask]
        grad = self.row_oracle(blk) @ loc + self.reg * loc[blk, :] - Bm[blk, :]  # sap.py:121-123
        d = pinv(grad)
        if self.accel:
            self.W[:, mask] = self.Y[:, mask].clone()
            upd = torch.zeros_like(self.W[:, mask])
            upd[blk] = eta * d
            self.W[:, mask] -= upd
            self.V[:, mask] = self.beta * self.V[:, mask] + (1 - self.beta) * self.Y[:, mask]
            vupd = torch.zeros_like(self.V[:, mask])
            vupd[blk] = eta * self.gamma * d
            self.V[:, mask] -= vupd
            self.Y[:, mask] = self.alpha * self.V[:, mask] + (1 - self.alpha) * self.W[:, mask]
        else:
            upd = torch.zeros_like(self.W[:, mask])
            upd[blk] = eta * d
            self.W[:, mask] -= upd
