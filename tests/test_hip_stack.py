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

