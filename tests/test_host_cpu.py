"""Host-side logic of the package on CPU, with the oracle injected as compute backend
(``tests/_oracle_backend.py``).  The cases live in ``stack_cases.py``."""

import pytest
import torch

from _oracle_backend import OracleBackend
from rlaopt_amd.ops import use_backend
from stack_cases import *  # noqa: F401,F403


@pytest.fixture
def dev():
    return torch.device("cpu")


@pytest.fixture(autouse=True)
def oracle_backend():
    with use_backend(OracleBackend()):
        yield


def test_quiet_host_limits_and_restores_threads():
    """utils.quiet_host: small CPU thread pool inside solver loops, re-entrant, restored on exit."""
    from rlaopt_amd.utils import quiet_host, small_host_lapack

    prev = torch.get_num_threads()
    try:
        torch.set_num_threads(max(prev, 4))
        outer = torch.get_num_threads()
        with quiet_host(max_threads=2):
            assert torch.get_num_threads() == min(outer, 2)
            with quiet_host(max_threads=1):  # nested: the outermost context decides
                assert torch.get_num_threads() == min(outer, 2)
                with small_host_lapack(64):  # (a pool at most twice the wanted size is left alone: no team rebuild)
                    assert torch.get_num_threads() == min(outer, 2)
                assert torch.get_num_threads() == min(outer, 2)
            assert torch.get_num_threads() == min(outer, 2)
        assert torch.get_num_threads() == outer
        if outer > 2:
            with small_host_lapack(64):  # far too wide for a 64 x 64 factorisation: one thread
                assert torch.get_num_threads() == 1
            assert torch.get_num_threads() == outer
        with pytest.raises(RuntimeError):
            with quiet_host(max_threads=2):
                raise RuntimeError("boom")
        assert torch.get_num_threads() == outer  # restored after an exception too
    finally:
        torch.set_num_threads(prev)


@pytest.mark.parametrize("dt", [torch.float64, torch.float32])
def test_nystrom_inverse_with_gram_matches_apply_then_gram(dt):
    """Nystrom._inverse_matmul_with_gram (the one-collective form used by row-sharded PCG):
    Z equals P^-1 X and the k x k block equals X^T Z computed from Z."""
    from rlaopt_amd import sketches
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.preconditioners import _get_precond

    g = torch.Generator().manual_seed(11)
    n, r, k = 300, 40, 3
    Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, generator=g))
    A = ((Q * torch.arange(1, n + 1, dtype=torch.float64) ** -2.0) @ Q.T).to(dt)
    A = 0.5 * (A + A.T)
    X = torch.randn(n, k, dtype=dt, generator=g)
    Om = (torch.randn(r, n, dtype=dt, generator=g) / r**0.5).T.contiguous()
    with use_backend(OracleBackend()):
        P = _get_precond(NystromConfig(rank=r, rho=1e-3, sketch="gauss"))
        with sketches.inject(Om):
            P._update(A, torch.device("cpu"))
        P._update_damping(baseline_rho=1e-3)
        Z, XtZ = P._inverse_matmul_with_gram(X)
        Zref = P._inv @ X
    tol = 1e-10 if dt == torch.float64 else 2e-4
    torch.testing.assert_close(Z, Zref, rtol=tol, atol=tol)
    want = X.double().T @ Zref.double()
    assert XtZ.dtype == torch.float64
    assert float((XtZ - want).abs().max() / want.abs().max()) < (1e-9 if dt == torch.float64 else 1e-4)


def test_several_cuda_devices_in_one_process_warn_and_point_to_torchrun():
    """The reference's in-process multi-GPU call is not silently accepted: it warns (what it costs, how to
    launch one rank per GPU instead) before anything is moved to a device."""
    import warnings

    from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig

    X = torch.randn(50, 3, dtype=torch.float64)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        try:
            DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={torch.device("cuda:0"), torch.device("cuda:1")})
        except Exception:  # noqa: BLE001 - no GPU here: the construction itself cannot finish
            pass
    msgs = [str(w.message) for w in rec if issubclass(w.category, RuntimeWarning)]
    assert any("one rank per GPU" in m and "torch.distributed.run" in m for m in msgs), msgs
