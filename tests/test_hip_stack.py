"""The parity tests proper: the full stack through the C ABI on cuda:0 against the
reference's golden vectors.  The cases live in ``stack_cases.py``."""

import pytest
import torch

from stack_cases import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_pcg_tail_graph_replay_matches_eager(monkeypatch):
    """RK_PCG_GRAPH=1: the captured tail of a PCG iteration (alpha, updates, preconditioner apply,
    beta) replayed as a HIP graph gives bitwise the iterates of the eager launches, across a
    column compaction (which re-captures)."""
    import torch

    from rlaopt_amd import sketches
    from rlaopt_amd.kernels import KernelConfig, RBFLinOp
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.solvers.pcg import PCG

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    n, d, k, r = 3000, 6, 3, 64
    X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).to(dev)
    B = torch.randn(n, k, dtype=torch.float64, generator=g).to(dev)
    Om = (torch.randn(r, n, dtype=torch.float64, generator=g) / r**0.5).T.contiguous().to(dev)
    reg = 1e-3 * n
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RK_PCG_GRAPH", mode)
        A = RBFLinOp(X, X, KernelConfig(lengthscale=1.0))
        system = LinSys(A, B, reg=reg)
        with sketches.inject(Om):
            s = PCG(system, torch.zeros_like(B), NystromConfig(rank=r, rho=reg, sketch="gauss"), dev)
        for i in range(12):
            if i == 7:  # drop a column: the state is compacted and the graph re-captured
                system._mask = torch.tensor([True, False, True])
            s._step()
        torch.cuda.synchronize()
        assert (s._graph is not None) == (mode == "1")
        out[mode] = (s.W.clone(), s.R.clone(), s.RZ.clone())
    for a, b in zip(out["0"], out["1"]):
        assert torch.equal(a, b)



def test_operator_chooses_storage_by_expected_products(monkeypatch):
    """RK_KERNEL_MODE=auto with a solver's hint (LinSys.solve -> A._rk_plan): few expected products and an allocation
    that has to be paid for -> the operator stays storage-free; many products, or free memory already cached by the
    allocator, or RK_ALLOC_MS_PER_GB=0 -> strip-packed storage.  Either way the solve is the same solve."""
    from rlaopt_amd import sketches
    from rlaopt_amd.kernels import KernelConfig, RBFLinOp
    from rlaopt_amd.kernels import base as kb
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.solvers import PCGConfig

    d0 = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    n, d, r = 3000, 16, 64
    X = (torch.randn(n, d, generator=g) / d**0.5).to(d0)
    B = torch.randn(n, 1, generator=g).to(d0)
    Om = (torch.randn(r, n, generator=g) / r**0.5).T.contiguous().to(d0)
    monkeypatch.delenv("RK_KERNEL_MODE", raising=False)
    # nothing cached: the allocation is priced
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *_a, **_k: torch.cuda.memory_allocated(d0))
    monkeypatch.setenv("RK_ALLOC_MS_PER_GB", "50")
    assert kb._prefer_matrix_free("rbf", X, 20.0) and not kb._prefer_matrix_free("rbf", X, 5000.0)
    assert not kb._prefer_matrix_free("laplace", X, 20.0) and not kb._prefer_matrix_free("rbf", X.double(), 20.0)
    reg = 1e-3 * n
    cfg = PCGConfig(device=d0, max_iters=20, atol=0.0, rtol=1e-4,
                    precond_config=NystromConfig(rank=r, rho=reg, sketch="gauss"))
    A = RBFLinOp(X, X, KernelConfig(lengthscale=1.0))
    assert A.storage_mode == "sympack"
    with sketches.inject(Om):
        W_free, log_free = LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=5)
    assert A.storage_mode == "matrix_free"
    # priced at zero -> storage
    monkeypatch.setenv("RK_ALLOC_MS_PER_GB", "0")
    A2 = RBFLinOp(X, X, KernelConfig(lengthscale=1.0))
    with sketches.inject(Om):
        W_pack, log_pack = LinSys(A2, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=5)
    assert A2.storage_mode == "sympack"
    assert sorted(log_free) == sorted(log_pack)
    assert float((W_free - W_pack).norm() / W_pack.norm()) < 1e-3
    # a cached block of that size costs nothing either
    monkeypatch.setenv("RK_ALLOC_MS_PER_GB", "50")
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *_a, **_k: torch.cuda.memory_allocated(d0) + (1 << 30))
    assert not kb._prefer_matrix_free("rbf", X, 20.0)
    # an operator that has already materialised keeps its storage
    A2._rk_plan(1.0)
    assert A2.storage_mode == "sympack"
