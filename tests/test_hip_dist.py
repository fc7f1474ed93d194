"""SPMD control flow with the real HIP kernels on the one-GPU test box (see
tests/_dist_gpu_worker.py): 2 ranks sharing cuda:0 over gloo, and a single-rank RCCL group."""

import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("backend,world,n", [("gloo", 2, 3001), ("gloo", 3, 2000), ("nccl", 1, 3001), ("nccl-abi", 1, 3001)])
def test_spmd_on_gpu(backend, world, n):
    """("nccl-abi": the single-rank RCCL group with the collectives routed through the C ABI's own transport,
    rk_comm_* - RK_COMM=abi.)"""
    abi = backend == "nccl-abi"
    backend = "nccl" if abi else backend
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if backend == "nccl":
            env["RK_FORCE_SPMD"] = "1"
        if abi:
            env["RK_COMM"] = "abi"
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(HERE, "_dist_gpu_worker.py"), backend, str(n)],
            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
        ))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
    assert outs[0].count("DIST-GPU-OK") == 2


def test_rk_comm_transport_single_rank():
    """rk_comm_* (include/rk.h): unique id, communicator, the three collectives on device buffers
    (a world of one: every collective is the identity) and the error path, through the raw C ABI."""
    import ctypes

    import torch

    from rlaopt_amd import _lib

    lib = _lib.load()
    ident = ctypes.create_string_buffer(128)
    _lib.check(lib.rk_comm_unique_id(ident), "rk_comm_unique_id")
    assert any(ident.raw)
    comm = ctypes.c_void_p()
    _lib.check(lib.rk_comm_create(ctypes.byref(comm), 1, 0, ident), "rk_comm_create")
    assert comm.value
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream(dev).cuda_stream
    for dt, code in ((torch.float32, 0), (torch.float64, 1)):
        x = torch.randn(1000, 3, dtype=dt, device=dev)
        ref = x.clone()
        _lib.check(lib.rk_comm_allreduce_sum(comm, x.data_ptr(), x.numel(), code, st), "allreduce")
        out = torch.empty_like(x)
        _lib.check(lib.rk_comm_allgather(comm, x.data_ptr(), out.data_ptr(), x.numel(), code, st), "allgather")
        rs = torch.empty_like(x)
        _lib.check(lib.rk_comm_reduce_scatter_sum(comm, x.data_ptr(), rs.data_ptr(), x.numel(), code, st), "reduce_scatter")
        torch.cuda.synchronize()
        assert torch.equal(x, ref) and torch.equal(out, ref) and torch.equal(rs, ref)
    with pytest.raises(ValueError):
        _lib.check(lib.rk_comm_allreduce_sum(comm, None, 4, 0, st), "allreduce")
    with pytest.raises(ValueError):
        _lib.check(lib.rk_comm_allreduce_sum(comm, x.data_ptr(), 4, 7, st), "allreduce")
    _lib.check(lib.rk_comm_destroy(comm), "rk_comm_destroy")


@pytest.mark.parametrize("devices,backend", [(1, "nccl"), (2, "gloo")])
def test_one_process_multi_device_solve_runs_in_worker_processes(devices, backend, monkeypatch):
    """rlaopt_amd/spmd.py on the GPU box: the whole solve of an operator flagged 'several devices in one process' is
    executed by fresh worker processes - one RCCL rank on cuda:0 (RK_FORCE_SPMD: the collective code path with a world
    of one), then two gloo ranks sharing cuda:0 - with the real HIP kernels (strip-packed SPMD storage); rank 0's
    result equals the in-process solve."""
    import torch

    from rlaopt_amd import sketches, spmd
    from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig
    from rlaopt_amd.models import LinSys
    from rlaopt_amd.preconditioners import NystromConfig
    from rlaopt_amd.solvers import PCGConfig

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    n, d, k = 2500, 8, 2
    dt = torch.float64
    X = (torch.randn(n, d, dtype=dt, generator=g) / d**0.5).to(dev)
    B = torch.randn(n, k, dtype=dt, generator=g).to(dev)
    reg = 1e-3 * n
    cfg = PCGConfig(device=dev, max_iters=100, atol=0.0, rtol=1e-9, precond_config=NystromConfig(rank=32, rho=reg))
    A = DistributedRBFLinOp(X, X, KernelConfig(lengthscale=1.0), devices={dev})
    W_ref, log_ref = LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=5)
    A._inprocess_multi_device = True
    A._shard_devices = [dev] * devices
    if devices == 1:
        monkeypatch.setenv("RK_FORCE_SPMD", "1")
    real = spmd.solve_kernel_system
    monkeypatch.setattr(spmd, "solve_kernel_system", lambda *a, **kw: real(*a, backend=backend, timeout=300, **kw))
    W, log = LinSys(A, B, reg=reg).solve(cfg, torch.zeros_like(B), callback_freq=5)
    assert W.device == B.device
    assert max(log) <= max(log_ref) + 5
    K = A @ torch.eye(n, dtype=dt, device=dev)
    res = torch.linalg.norm(B - (K @ W + reg * W), dim=0) / torch.linalg.norm(B, dim=0)
    assert bool((res <= 1e-9).all()), res
    assert float((W - W_ref).norm() / W_ref.norm()) < 1e-6
