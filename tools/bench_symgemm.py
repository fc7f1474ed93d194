#!/usr/bin/env python3
"""The sketch product Y = K Omega from the packed strips (rk_symm_gemm) against the panel + float32-MFMA route.

    python tools/bench_symgemm.py [--n 200000] [--d 64] [--r 512] [--reps 3]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rlaopt_amd.ops import get_backend  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--r", type=int, nargs="+", default=[512, 100, 256])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--fold", type=int, nargs="+", default=[16, 64, 8])
    ap.add_argument("--stride", type=int, default=1,
                    help="> 1: time rank 0's partial of a cyclic strip distribution over this many ranks (and stop)")
    ap.add_argument("--check-rows", type=int, default=2048)
    args = ap.parse_args()
    be = get_backend()
    dev = torch.device("cuda:0")
    n, d = args.n, args.d
    g = torch.Generator().manual_seed(0)
    X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).float().to(dev)
    if args.stride > 1:
        for first in (0, args.stride - 1):
            Kp = be.sympack_materialize("rbf", X, 1.0, first=first, stride=args.stride)
            for r in args.r:
                Om = (torch.randn(n, r, generator=g) / r**0.5).to(dev)
                be.symm_gemm(Kp, n, Om, first=first, stride=args.stride)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    be.symm_gemm(Kp, n, Om, first=first, stride=args.stride)
                e1.record()
                torch.cuda.synchronize()
                print(f"rank {first} of {args.stride}: r={r}: partial sketch product {e0.elapsed_time(e1) / args.reps:.2f} ms "
                      f"({Kp.numel() * 4 / 1e9:.1f} GB of strips)", flush=True)
            del Kp
        return
    t0 = time.time()
    Kp = be.sympack_materialize("rbf", X, 1.0)
    torch.cuda.synchronize()
    print(f"n={n} d={d}: packed triangle {Kp.numel() * 4 / 1e9:.1f} GB in {time.time() - t0:.2f}s", flush=True)
    for r in args.r:
        Om = (torch.randn(n, r, generator=g) / r**0.5).to(dev)
        rows = torch.randperm(n, generator=g)[: args.check_rows].to(dev)
        want = None  # (float64 reference rows: computed after the first timing, see below)
        if r == args.r[0]:  # the route this kernel replaces: row-major panels + the float32 matrix cores
            rpp = 8192
            panel = torch.empty((rpp, n), dtype=torch.float32, device=dev)
            Yo = torch.empty((n, r), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            t0 = time.time()
            for a in range(0, n, rpp):
                b = min(n, a + rpp)
                pn = be.kernel_materialize("rbf", X[a:b], X, 1.0, out=panel[: b - a])
                Yo[a:b] = be.matmat(pn, Om)
            torch.cuda.synchronize()
            ms = (time.time() - t0) * 1e3
            want = be.kernel_materialize("rbf", X[rows].double(), X.double(), 1.0) @ Om.double()
            err = float(((Yo[rows].double() - want).norm(dim=0) / want.norm(dim=0)).max())
            print(f"r={r:4d} panels + rk_gemm (float32 MFMA): {ms:8.2f} ms, max column error vs float64 {err:.2e}", flush=True)
            del panel, Yo
        from rlaopt_amd.ops import f16_grid

        Omg = f16_grid(Om)  # the test matrix as the Gaussian sketch draws it: on the float16 grid
        wantg = None
        variants = ([("f16x2", a, "ph", False) for a in args.fold] + [("f16x2", args.fold[0], "lockstep", False),
                    ("f16x2", args.fold[0], "ph", True), ("bf16x3", args.fold[0], "lockstep", False)])
        for split, fold, impl, grid in variants:
            os.environ["RK_SYMGEMM_FOLD"] = str(fold)
            os.environ["RK_SYMGEMM_IMPL"] = impl
            if split == "bf16x3":
                os.environ["RK_SYMGEMM_SPLIT"] = "bf16x3"
            else:
                os.environ.pop("RK_SYMGEMM_SPLIT", None)
            V = Omg if grid else Om
            Y = be.symm_gemm(Kp, n, V, rhs_f16=grid)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                Y = be.symm_gemm(Kp, n, V, rhs_f16=grid)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            terms = 6 if split == "bf16x3" else (2 if grid else 3)
            if want is None:  # (after the timing: float64 GEMMs right before it lower the clocks the kernel gets)
                want = be.kernel_materialize("rbf", X[rows].double(), X.double(), 1.0) @ Om.double()
            if grid and wantg is None:
                wantg = be.kernel_materialize("rbf", X[rows].double(), X.double(), 1.0) @ Omg.double()
            w = wantg if grid else want
            err = float(((Y[rows].double() - w).norm(dim=0) / w.norm(dim=0)).max())
            label = (f"{split} fold {fold:3d} " + ("one piece of V (on the float16 grid), phase kernel" if grid else
                     ("phase kernel" if impl == "ph" and split == "f16x2" else "lockstep kernel")))
            print(f"r={r:4d} {label}: {ms:8.2f} ms  {2 * n * n * r / ms / 1e9:7.1f} TFLOP/s float32-equivalent, "
                  f"{terms * 2 * n * n * max(r, 128) / ms / 1e9:7.1f} TFLOP/s on the 16-bit cores (padded), "
                  f"max column error vs float64 {err:.2e}", flush=True)
        os.environ.pop("RK_SYMGEMM_IMPL", None)
        del Om, Y, want
    os.environ.pop("RK_SYMGEMM_SPLIT", None)
    os.environ.pop("RK_SYMGEMM_FOLD", None)


if __name__ == "__main__":
    main()
