"""The LSQR oracle (parity unpinned upstream) against scipy.sparse.linalg.lsqr and the
closed-form ridge solution.  CPU only."""

import numpy as np
import pytest
import scipy.sparse.linalg as spla
import torch

from oracle import lsqr as olsqr


@pytest.mark.parametrize("reg", [0.0, 0.3])
def test_lsqr_oracle_matches_scipy_iterates(reg):
    g = torch.Generator().manual_seed(0)
    m, d = 400, 30
    A = torch.randn(m, d, dtype=torch.float64, generator=g) * torch.logspace(0, -2, d, dtype=torch.float64)
    b = torch.randn(m, dtype=torch.float64, generator=g)
    for iters in (1, 3, 10):
        x, _ = olsqr.lsqr(A, b, reg=reg, iters=iters)
        ref = spla.lsqr(A.numpy(), b.numpy(), damp=reg**0.5, atol=0, btol=0, conlim=0, iter_lim=iters)[0]
        np.testing.assert_allclose(x.numpy(), ref, rtol=1e-7, atol=1e-9)
    # later iterates of two Golub-Kahan implementations drift apart (loss of orthogonality);
    # what must agree is the quality of the solution they reach
    x, _ = olsqr.lsqr(A, b, reg=reg, iters=60)
    ref = torch.from_numpy(spla.lsqr(A.numpy(), b.numpy(), damp=reg**0.5, atol=0, btol=0, conlim=0, iter_lim=60)[0])
    gx = torch.linalg.norm(A.T @ (b - A @ x) - reg * x)
    gr = torch.linalg.norm(A.T @ (b - A @ ref) - reg * ref)
    g0 = torch.linalg.norm(A.T @ b)
    assert float(gx) < 1e-3 * float(g0) and 0.1 < float(gx / gr) < 10


def test_lsqr_oracle_preconditioned_converges_to_ridge_solution():
    g = torch.Generator().manual_seed(1)
    m, d, k, reg = 600, 40, 3, 0.05
    A = torch.randn(m, d, dtype=torch.float64, generator=g) * torch.logspace(0, -3, d, dtype=torch.float64)
    B = torch.randn(m, k, dtype=torch.float64, generator=g)
    want = torch.linalg.solve(A.T @ A + reg * torch.eye(d, dtype=torch.float64), A.T @ B)
    # sketch-and-precondition factor from a Gaussian sketch (skpre.py:93-98): G = L L^T, R = L^T
    Om = torch.randn(4 * d, m, dtype=torch.float64, generator=g) / (4 * d) ** 0.5
    Y = Om @ A
    L = torch.linalg.cholesky(Y.T @ Y + reg * torch.eye(d, dtype=torch.float64))
    Rinv = torch.linalg.inv(L.T)
    x_pre, h_pre = olsqr.lsqr(A, B, reg=reg, Rinv=Rinv, iters=25)
    x_raw, h_raw = olsqr.lsqr(A, B, reg=reg, iters=25)
    assert float((x_pre - want).norm() / want.norm()) < 1e-8
    assert float(h_pre[-1].max()) < 1e-3 * float(h_raw[-1].max())  # the preconditioner matters
    # warm start: x0 = solution -> stays there
    x_ws, _ = olsqr.lsqr(A, B, reg=reg, Rinv=Rinv, x0=want + 1e-3, iters=15)
    assert float((x_ws - want).norm() / want.norm()) < 1e-8
