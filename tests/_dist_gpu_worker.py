"""Worker for tests/test_hip_dist.py: one rank of an SPMD run whose shards all live on cuda:0.

backend "gloo": W ranks share the one GPU of the test box; collectives are staged through the
host (rlaopt_amd/dist.py) - exercises the real HIP kernels under the SPMD control flow.
backend "nccl": a single-rank RCCL group with RK_FORCE_SPMD=1 - exercises the RCCL calls
themselves (all_gather_into_tensor / all_reduce on device buffers, float32 and float64).
"""

import os
import sys

import torch
import torch.distributed as td

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import cbind  # noqa: E402
from oracle import kernels as ok  # noqa: E402
from rlaopt_amd import sketches  # noqa: E402
from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig  # noqa: E402
from rlaopt_amd.models import LinSys  # noqa: E402
from rlaopt_amd.preconditioners import NystromConfig  # noqa: E402
from rlaopt_amd.solvers import PCGConfig  # noqa: E402


def main():
    backend, n = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if backend == "nccl":
        td.init_process_group("nccl", device_id=dev)
    else:
        td.init_process_group("gloo")
    rank, size = td.get_rank(), td.get_world_size()
    for dt, rtol in ((torch.float64, 1e-9), (torch.float32, 1e-4)):
        gen = torch.Generator().manual_seed(0)
        d, k, r = 8, 2, 64
        X = (torch.randn(n, d, dtype=torch.float64, generator=gen) / d**0.5).to(dt)
        B = torch.randn(n, k, dtype=torch.float64, generator=gen).to(dt)
        Om = (torch.randn(r, n, dtype=torch.float64, generator=gen) / r**0.5).T.contiguous().to(dt)
        M = torch.randn(n, 3, dtype=torch.float64, generator=gen).to(dt)
        lim = 1e-11 if dt == torch.float64 else 2e-5
        A = DistributedRBFLinOp(X.to(dev), X.to(dev), KernelConfig(lengthscale=1.0), devices={dev})
        assert A.is_row_sharded
        want = cbind.kernel_matmat("rbf", X, X, 1.0, M)
        got = (A @ M.to(dev)).cpu()
        assert float((got - want).norm() / want.norm()) < lim
        got = (A.T @ M.to(dev)).cpu()  # symmetric data: K^T M == K M
        assert float((got - want).norm() / want.norm()) < lim
        blk = torch.randperm(n, generator=gen)[: n // 10]
        got = (A.row_oracle(blk) @ M.to(dev)).cpu()
        assert float((got - want[blk]).norm() / want[blk].norm()) < lim
        w = torch.randn(blk.numel(), 2, dtype=torch.float64, generator=gen).to(dt)
        Kbb = ok.kernel_matrix("rbf", X[blk].double(), X[blk].double(), 1.0)
        got = (A.blk_oracle(blk) @ w.to(dev)).cpu()
        assert float((got.double() - Kbb @ w.double()).norm() / (Kbb @ w.double()).norm()) < lim
        reg = (1e-5 if dt == torch.float64 else 1e-4) * n
        cfg = PCGConfig(device=dev, max_iters=300, atol=0.0, rtol=rtol,
                        precond_config=NystromConfig(rank=r, rho=reg, sketch="gauss"))
        system = LinSys(A, B.to(dev), reg=reg)
        with sketches.inject(Om):
            W, log = system.solve(cfg, torch.zeros(n, k, dtype=dt, device=dev), callback_freq=5)
        Wc = W.cpu()
        res = torch.linalg.norm((B - (cbind.kernel_matmat("rbf", X, X, 1.0, Wc) + reg * Wc)).double(), dim=0)
        res = res / torch.linalg.norm(B.double(), dim=0)
        assert bool((res <= 2 * rtol).all()), res
        assert 5 <= max(log) < 300
        if rank == 0:
            print(f"DIST-GPU-OK backend={backend} world={size} dtype={dt} iters={max(log)} res={float(res.max()):.2e}")
        A.shutdown()
    td.barrier()
    td.destroy_process_group()


if __name__ == "__main__":
    main()
