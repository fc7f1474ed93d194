"""Oracle: one kernel-ridge system solved end to end on the host.

TEST / CPU-BASELINE INFRASTRUCTURE - see ``oracle/__init__.py``.

The composite oracle of SURVEY 8(c): the kernel operator (closed forms of rlaopt/kernels/standard.py:31-85,
here ``oracle/rk_oracle.c`` on the host cores) behind the reference's own solver stack as restated in
``oracle/nystrom.py`` (preconditioners/nystrom.py:55-152), ``oracle/pcg.py`` (solvers/pcg.py:13-93) and
``oracle/linsys.py`` (models/linsys.py:96-107, models/model.py:89-114), in the dtype of the data - i.e.
what ``LinSys(A, B, reg).solve(PCGConfig(precond_config=NystromConfig(...)))`` computes on the CPU.
"""

from __future__ import annotations

import time

import torch

from . import cbind
from . import linsys as ol
from . import nystrom as on
from . import pcg as opcg


class KernelOperator:
    """``A @ V`` for A = const_scaling * K(X, X): matrix-free (one pass of the C product per call, what
    the reference's pykeops operator does) or from the dense matrix written once (``dense=True``)."""

    def __init__(self, kind, X, lengthscale=1.0, const_scaling=1.0, dense=False):
        self.kind, self.X, self.ell, self.scale = kind, X, lengthscale, const_scaling
        self.dtype = X.dtype
        self.shape = (X.shape[0], X.shape[0])
        self.products = 0
        self.product_s = 0.0
        self.K = cbind.kernel_matrix(kind, X, X, lengthscale, const_scaling) if dense else None

    def __matmul__(self, V):
        t = time.perf_counter()
        if self.K is not None:
            # (few columns: the C row-streaming product; the sketch product: BLAS gemm)
            out = cbind.dense_matmat(self.K, V) if (V.ndim == 1 or V.shape[1] <= 16) else self.K @ V
        else:
            out = cbind.kernel_matmat(self.kind, self.X, self.X, self.ell, V, self.scale)
        self.product_s += time.perf_counter() - t
        self.products += 1
        return out


def solve(X, B, Omega, reg, *, rtol, atol=0.0, max_iters=200, callback_freq=10, kind="rbf", lengthscale=1.0,
          adaptive=True, dense=False):
    """Nystrom-preconditioned block PCG on (K(X, X) + reg I) W = B with the test matrix ``Omega`` (n x r);
    rho = reg, adaptive damping as in the reference default (preconditioners/configs.py:82).  Returns a
    dict: W, log (iteration -> abs_res / rel_res), iters, rel_res (last logged, per column), build_s,
    loop_s, products, product_s."""
    if B.ndim == 1:
        B = B[:, None]
    A = KernelOperator(kind, X, lengthscale, dense=dense)
    t0 = time.perf_counter()
    st = on.build(A, Omega, reg)
    on.update_damping(st, reg, adaptive=adaptive)
    build_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    solver = opcg.BlockPCG(lambda V: A @ V, lambda R: on.inverse_matmul(st, R), B, reg, torch.zeros_like(B))
    W, log = ol.run(solver, lambda V: A @ V, B, reg, max_iters, atol, rtol, callback_freq=callback_freq)
    loop_s = time.perf_counter() - t0
    last = max(log)
    return {"W": W, "log": log, "iters": last, "rel_res": log[last]["rel_res"], "rho": st.rho,
            "build_s": build_s, "loop_s": loop_s, "products": A.products, "product_s": A.product_s}
