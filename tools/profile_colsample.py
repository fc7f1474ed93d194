#!/usr/bin/env python3
"""Where the Nystrom assembly with the column-sampling sketch spends its time (n = 200 000, r = 512, float32): every
backend call and every host LAPACK call of ``Nystrom._update`` timed with a device synchronisation around it."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_problem  # noqa: E402
from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig  # noqa: E402
from rlaopt_amd.ops import get_backend  # noqa: E402
from rlaopt_amd.preconditioners import NystromConfig  # noqa: E402
from rlaopt_amd.preconditioners.nystrom import Nystrom  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--rank", type=int, default=512)
    ap.add_argument("--sketch", default="colsample")
    ap.add_argument("--cprofile", action="store_true", help="host-side call profile of one more assembly")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    be = get_backend()
    n, d, r = args.n, args.d, args.rank
    X, _B, _Om = make_problem(n, d, 8, torch.float32)
    Xd = X.to(dev)
    A = DistributedRBFLinOp(Xd, Xd, KernelConfig(lengthscale=1.0), devices={dev})
    A.local_matmat(torch.zeros(n, 1, device=dev))
    torch.cuda.synchronize()
    log = []

    def wrap(obj, name, label=None):
        fn = getattr(obj, name)

        def timed(*a, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize()
            log.append((label or name, 1e3 * (time.perf_counter() - t)))
            return out

        setattr(obj, name, timed)

    for name in ("sample_distinct", "kernel_materialize", "gram_tn", "gemm", "symm_gemm", "randn", "rmatmat", "matmat"):
        if hasattr(be, name):
            wrap(be, name)
    for name in ("cholesky", "eigh", "solve_triangular", "matrix_norm"):
        wrap(torch.linalg, name, "host torch.linalg." + name)
    for rep in range(3):
        log.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        P = Nystrom(NystromConfig(rank=r, rho=1e-6 * n, sketch=args.sketch, damping_mode="adaptive"))
        P._update(A, dev)
        P._update_damping(baseline_rho=1e-6 * n)
        torch.cuda.synchronize()
        total = 1e3 * (time.perf_counter() - t0)
        print(f"pass {rep}: {total:.1f} ms total; timed stages {sum(t for _, t in log):.1f} ms")
        for name, t in log:
            print(f"    {name:40s} {t:8.2f} ms")


def cprofile_pass(args):
    import cProfile
    import pstats

    dev = torch.device("cuda:0")
    n, d, r = args.n, args.d, args.rank
    X, _B, _Om = make_problem(n, d, 8, torch.float32)
    Xd = X.to(dev)
    A = DistributedRBFLinOp(Xd, Xd, KernelConfig(lengthscale=1.0), devices={dev})
    A.local_matmat(torch.zeros(n, 1, device=dev))
    for rep in range(2):
        P = Nystrom(NystromConfig(rank=r, rho=1e-6 * n, sketch=args.sketch, damping_mode="adaptive"))
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        pr.enable()
        P._update(A, dev)
        P._update_damping(baseline_rho=1e-6 * n)
        torch.cuda.synchronize()
        pr.disable()
        if rep == 1:
            pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    if "--cprofile" in sys.argv:
        ap = argparse.ArgumentParser()
        ap.add_argument("--n", type=int, default=200000)
        ap.add_argument("--d", type=int, default=64)
        ap.add_argument("--rank", type=int, default=512)
        ap.add_argument("--sketch", default="colsample")
        ap.add_argument("--cprofile", action="store_true")
        cprofile_pass(ap.parse_args())
        sys.exit(0)
    main()
