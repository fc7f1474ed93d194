"""``rlaopt_amd.spmd``: a caller in ONE process, one fresh worker process per device (CPU: two gloo ranks; the workers
compute with the oracle stand-in backend).  Covers the generic entry, failure reporting, and the dispatch of a whole
kernel-system solve (what ``LinSys.solve`` does for an operator built with several devices in one process)."""

import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from rlaopt_amd import spmd  # noqa: E402

CPU = torch.device("cpu")


def _sum_of_ranks(ctx, offset, scale=1.0):
    import torch.distributed as td

    t = torch.tensor([float(ctx.rank + offset)]) * scale
    td.all_reduce(t)
    assert td.get_world_size() == ctx.world and td.get_backend() == "gloo"
    return {"sum": t, "rank": ctx.rank, "device": str(ctx.device)}


def _fails_on_rank_one(ctx):
    if ctx.rank == 1:
        raise ValueError("boom on rank one")
    return 0


def test_run_starts_one_rank_per_device_and_returns_rank_zero():
    out = spmd.run(_sum_of_ranks, [CPU, CPU, CPU], args=(10,), kwargs={"scale": 2.0}, timeout=120)
    assert out["rank"] == 0 and out["device"] == "cpu"
    assert float(out["sum"]) == 2.0 * (10 + 11 + 12)


def test_run_reports_a_failing_rank():
    with pytest.raises(RuntimeError, match="boom on rank one"):
        spmd.run(_fails_on_rank_one, [CPU, CPU], timeout=120)
    with pytest.raises(ValueError):
        spmd.run(lambda ctx: 0, [CPU])  # not importable by name


def test_solve_of_a_one_process_multi_device_operator_is_dispatched_to_spmd_workers(monkeypatch):
    """LinSys.solve on an operator flagged as 'several devices in one process' goes through spmd.solve_kernel_system:
    two worker ranks rebuild the SPMD operator (strip-packed, row-partitioned state) and solve; the result equals the
    in-process solve of the same system."""
    from _oracle_backend import OracleBackend
    from rlaopt_amd import sketches
    from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.ops import use_backend
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.solvers import PCGConfig

    g = torch.Generator().manual_seed(8)
    n, d, k = 90, 4, 2
    dt = torch.float64
    X = torch.randn(n, d, dtype=dt, generator=g) / d**0.5
    B = torch.randn(n, k, dtype=dt, generator=g)
    reg = 1e-3 * n
    cfg = PCGConfig(device=CPU, max_iters=100, atol=0.0, rtol=1e-10, precond_config=NystromConfig(rank=16, rho=reg))
    with use_backend(OracleBackend()):
        A = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={CPU})
        torch.manual_seed(5)
        W_ref, log_ref = LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=5)
        # flag it the way a devices={cuda:0, cuda:1} construction does, with two (CPU) shard devices
        A._inprocess_multi_device = True
        A._shard_devices = [CPU, CPU]
        calls = {}
        real = spmd.solve_kernel_system

        def spy(*a, **kw):
            calls["n"] = calls.get("n", 0) + 1
            return real(*a, init="_oracle_backend:install_for_worker", timeout=240, **kw)

        monkeypatch.setattr(spmd, "solve_kernel_system", spy)
        W, log = LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=5)
        assert calls == {"n": 1}
        # a user callback keeps the solve in this process
        W2, _ = LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_fn=lambda w, m: 0.0, callback_freq=5)
        assert calls == {"n": 1}
        monkeypatch.setenv("RK_MULTI_DEVICE", "inprocess")
        LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=5)
        assert calls == {"n": 1}
    res = torch.linalg.norm(B - (A_dense(X) @ W + reg * W), dim=0) / torch.linalg.norm(B, dim=0)
    assert bool((res <= 1e-9).all()), res
    assert max(log) <= max(log_ref) + 5  # (its own sketch draw: the iteration count may differ by a logging window)
    assert float((W - W_ref).norm() / W_ref.norm()) < 1e-7
    assert sorted(log)[0] == 0 and "internal_metrics" in log[0]["metrics"]


def A_dense(X):
    from oracle import kernels as ok

    return ok.kernel_matrix("rbf", X, X, 1.0)
