#!/usr/bin/env python3
"""The sketch product of a matrix-free operator, Y = K(X, X) Omega with the entries evaluated inside the product
(rk_kernel_sketch_matmat), against the strip-subset route (temporary packed strips + rk_symm_gemm) and the stored-strip
product (rk_symm_gemm on the resident 80 GB triangle).

    python tools/bench_ksketch.py [--n 200000] [--d 64] [--r 512 256 100] [--reps 3] [--only-fused]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rlaopt_amd.kernels import KernelConfig, RBFLinOp  # noqa: E402
from rlaopt_amd.ops import get_backend  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--r", type=int, nargs="+", default=[512, 256, 100])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kind", default="rbf")
    ap.add_argument("--only-fused", action="store_true")
    ap.add_argument("--check-rows", type=int, default=256)
    args = ap.parse_args()
    be = get_backend()
    dev = torch.device("cuda:0")
    n, d = args.n, args.d
    g = torch.Generator().manual_seed(0)
    X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).float().to(dev)
    gc = be.gram_center(X, X, 1.0)
    assert gc is not None
    for r in args.r:
        Om = (torch.randn(n, r, generator=g) / r**0.5).to(dev)
        rec = {"n": n, "d": d, "r": r, "kind": args.kind}
        ms, Y = timed(lambda: be.kernel_sketch_matmat(args.kind, X, X, 1.0, Om, gram_center=gc), args.reps)
        rec["fused"] = {"ms": round(ms, 2), "TFLOPs_f32_equiv": round(2 * n * n * r / ms / 1e9, 1),
                        "TFLOPs_16bit_cores": round(3 * 2 * n * n * (max(r, 128) + 2 * d * (-(-r // 256))) / ms / 1e9, 1)}
        from rlaopt_amd.ops import f16_grid

        Omg = f16_grid(Om)  # (as the Gaussian sketch draws it: one piece of Omega, two products per pair)
        ms_g, Yg = timed(lambda: be.kernel_sketch_matmat(args.kind, X, X, 1.0, Omg, gram_center=gc, rhs_f16=True), args.reps)
        rec["fused_rhs16"] = {"ms": round(ms_g, 2)}
        if not args.only_fused:
            os.environ["RK_KERNEL_MODE"] = "matrix_free"
            os.environ["RK_FREE_WIDE_FUSED"] = "0"
            try:
                A = RBFLinOp(X, X, KernelConfig(lengthscale=1.0))
                ms_s, Ys = timed(lambda: A @ Om, 1)
                rec["strip_subsets"] = {"ms": round(ms_s, 2)}
            finally:
                del os.environ["RK_KERNEL_MODE"], os.environ["RK_FREE_WIDE_FUSED"]
        if args.check_rows:
            rows = torch.randperm(n, generator=g)[: args.check_rows].to(dev)
            want = be.kernel_materialize(args.kind, X[rows].double(), X.double(), 1.0) @ Om.double()
            rec["fused"]["max_col_err_vs_f64"] = float(((Y[rows].double() - want).norm(dim=0) / want.norm(dim=0)).max())
            if "strip_subsets" in rec:
                rec["strip_subsets"]["max_col_err_vs_f64"] = float(((Ys[rows].double() - want).norm(dim=0) / want.norm(dim=0)).max())
        print(rec, flush=True)


if __name__ == "__main__":
    main()
