#!/usr/bin/env python3
"""Where the Nystrom assembly of the headline system spends its time (n = 200 000, r = 512, float32):
each stage of ``Nystrom._update`` timed with a device synchronisation behind it."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import make_problem  # noqa: E402
from rlaopt_amd import sketches  # noqa: E402
from rlaopt_amd.kernels import DistributedRBFLinOp, KernelConfig  # noqa: E402
from rlaopt_amd.ops import get_backend  # noqa: E402
from rlaopt_amd.preconditioners import NystromConfig  # noqa: E402
from rlaopt_amd.preconditioners import nystrom as nys  # noqa: E402
from rlaopt_amd.utils import small_host_lapack  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--rank", type=int, default=512)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    be = get_backend()
    n, d, r = args.n, args.d, args.rank
    X, B, Om = make_problem(n, d, r, torch.float32)
    from bench import on_grid

    Xd, Omd = X.to(dev), on_grid(Om.to(dev))  # (on the float16 grid, as the Gaussian sketch draws it)
    A = DistributedRBFLinOp(Xd, Xd, KernelConfig(lengthscale=1.0), devices={dev})
    A.local_matmat(torch.zeros(n, 1, device=dev))
    torch.cuda.synchronize()

    def timed(label, fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        print(f"  {label:58s} {1e3 * (time.perf_counter() - t):8.2f} ms", flush=True)
        return out

    for rep in range(2):
        print(f"pass {rep} (pass 0 includes one-time allocations / module loads)")
        t_all = time.perf_counter()
        Y = timed("Y = K Omega (rk_symm_gemm incl. split image)", lambda: A.local_matmat(Omd))
        both = timed("[Omega^T Y, Y^T Y] (rk_gram_tn x2, float64 MFMA)",
                     lambda: torch.stack([be.gram_tn(Omd, Y), be.gram_tn(Y, Y)]))
        host = timed("device -> host (2 r^2 float64)", lambda: both.cpu())
        core, gram = host[0], host[1]

        def small():
            shift = torch.finfo(torch.float32).eps * torch.trace(core)
            c = core + shift * torch.eye(r, dtype=torch.float64)
            with small_host_lapack(r):
                L = torch.linalg.cholesky(c)
                H = torch.linalg.solve_triangular(L, gram, upper=False)
                H = torch.linalg.solve_triangular(L, H.T.contiguous(), upper=False)
                H = 0.5 * (H + H.T)
                ev, V = torch.linalg.eigh(H)
                ev, V = ev.flip(0), V.flip(1)
                sig = ev.clamp_min(0).sqrt()
                M = torch.linalg.solve_triangular(L.T.contiguous(), V, upper=True) / sig.clamp_min(1e-300)[None, :]
            return M

        M = timed("host: Cholesky, 2 triangular solves, eigh, back-solve (r x r)", small)
        Md = timed("host -> device (r^2)", lambda: M.to(device=dev, dtype=torch.float32))
        U = timed("U = Y M (rk_gemm)", lambda: be.gemm(Y, Md))
        G2 = timed("U^T U (rk_gram_tn)", lambda: be.gram_tn(U, U))

        def lowdin():
            with small_host_lapack(r):
                return nys._inv_sqrt_near_identity(G2.cpu())

        fix = timed("host: (U^T U)^-1/2", lowdin)
        U2 = timed("U <- U fix (rk_gemm)", lambda: be.gemm(U, fix.to(device=dev, dtype=torch.float32)))
        print(f"  sum of the stages: {1e3 * (time.perf_counter() - t_all):.1f} ms")
        del Y, U, U2
        # and the real thing
        P = nys.Nystrom(NystromConfig(rank=r, rho=1e-6 * n, sketch="gauss"))
        torch.cuda.synchronize()
        t = time.perf_counter()
        with sketches.inject(Omd):
            P._update(A, dev)
        P._update_damping(1e-6 * n)
        torch.cuda.synchronize()
        print(f"  Nystrom._update + damping: {1e3 * (time.perf_counter() - t):.1f} ms")
        del P


if __name__ == "__main__":
    main()
