"""The Gaussian test matrix is drawn on the float16 grid and the packed operator hands that fact to the sketch
product (rk_symm_gemm_rhs16: one piece of Omega, two products per pair)."""
import pytest
import torch

from _oracle_backend import OracleBackend
from rlaopt_amd.ops import f16_grid, get_backend, on_f16_grid, use_backend
from rlaopt_amd.sketches import get_sketch


@pytest.fixture(autouse=True)
def oracle_backend():
    with use_backend(OracleBackend()):
        yield


def _gauss(dtype):
    return get_sketch("gauss", "right", 64, 200, dtype, torch.device("cpu"))


def test_gauss_right_sketch_is_on_the_float16_grid(monkeypatch):
    for dtype in (torch.float32, torch.float64):
        Om = _gauss(dtype).Omega_mat
        assert on_f16_grid(Om) and Om.dtype == dtype and torch.equal(Om, Om.half().to(dtype))
        # still a Gaussian sketch: mean 0, variance 1 / s
        assert abs(float(Om.mean())) < 0.02 and abs(float(Om.var()) * 64 - 1) < 0.1
    monkeypatch.setenv("RK_SKETCH_F16_GRID", "0")
    Om = _gauss(torch.float32).Omega_mat
    assert not on_f16_grid(Om) and not torch.equal(Om, Om.half().float())


def test_left_sketches_and_injected_matrices_are_left_alone():
    sk = get_sketch("gauss", "left", 64, 200, torch.float32, torch.device("cpu"))
    assert not on_f16_grid(sk.Omega_mat)
    x = torch.randn(5, 7)
    y = f16_grid(x)
    assert on_f16_grid(y) and not on_f16_grid(y[:, :3]) and not on_f16_grid(y.clone())


def test_packed_operator_passes_the_grid_to_the_sketch_product(monkeypatch):
    from rlaopt_amd.kernels import RBFLinOp
    from rlaopt_amd.kernels.configs import KernelConfig

    be = get_backend()
    seen = []
    real = be.symm_gemm
    monkeypatch.setattr(be, "symm_gemm", lambda *a, **k: seen.append(k.get("rhs_f16")) or real(*a, **k))
    monkeypatch.setenv("RK_KERNEL_MODE", "sympack")
    torch.manual_seed(0)
    X = torch.randn(300, 5)
    A = RBFLinOp(X, X, KernelConfig(const_scaling=1.0, lengthscale=1.0))
    assert getattr(A, "_sym", None) is not None
    Om = get_sketch("gauss", "right", 40, 300, torch.float32, torch.device("cpu")).Omega_mat
    Y = A @ Om
    assert seen and all(seen)
    seen.clear()
    Y2 = A @ Om.clone()  # the same values without the tag: the general entry
    assert seen and not any(seen)
    assert torch.allclose(Y, Y2, rtol=1e-5, atol=1e-6)


def test_rounded_sketch_preconditions_like_the_unrounded_one(monkeypatch):
    """Same seed, same draw, entries rounded to the float16 grid or not: the Nystrom spectrum agrees to the rounding level
    of the test matrix and block PCG takes the same number of iterations to a few percent (a random test matrix rounded to
    11 bits is as good a test matrix)."""
    from rlaopt_amd.kernels import RBFLinOp
    from rlaopt_amd.kernels.configs import KernelConfig
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.solvers import PCGConfig

    dev = torch.device("cpu")
    g = torch.Generator().manual_seed(3)
    n, d, r = 1500, 6, 60
    X = torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5
    B = torch.randn(n, 2, dtype=torch.float64, generator=g)
    reg = 1e-6 * n
    out = {}
    for grid in ("1", "0"):
        monkeypatch.setenv("RK_SKETCH_F16_GRID", grid)
        torch.manual_seed(11)  # (the sketch seed comes from torch's default generator)
        A = RBFLinOp(X, X, KernelConfig(const_scaling=1.0, lengthscale=1.0))
        cfg = PCGConfig(device=dev, max_iters=300, atol=0.0, rtol=1e-9,
                        precond_config=NystromConfig(rank=r, rho=reg, sketch="gauss"))
        system = LinSys(A, B, reg=reg)
        W, log = system.solve(cfg, torch.zeros(n, 2, dtype=torch.float64), callback_freq=1)
        out[grid] = (max(log), W, system._active_solver.P.S.clone() if hasattr(system, "_active_solver") and hasattr(
            system._active_solver, "P") and hasattr(system._active_solver.P, "S") else None)
    it1, W1, S1 = out["1"]
    it0, W0, S0 = out["0"]
    assert abs(it1 - it0) <= max(2, 0.03 * it0), (it1, it0)  # (291 and 288 here: two different but equally good test matrices)
    assert torch.allclose(W1, W0, rtol=1e-5, atol=1e-7)
    if S1 is not None and S0 is not None:
        assert float(((S1 - S0).abs() / S0.abs().clamp_min(1e-12)).max()) < 5e-2
