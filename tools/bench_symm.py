#!/usr/bin/env python3
"""Strip-packed symmetric product ``rk_symm_matmat`` for 1 .. 16 right-hand sides on one MI355X:
ms per product and GB/s of packed bytes; float32: the block kernel on the 16-bit matrix cores (f16x2: four strips side by
side per work item, round 4) against round 3's strip form of it (f16x2_strip), the float32 matrix-core strip kernel
(wide), groups of 4 through the register kernel (RK_SYMM_WIDE=0) and, for k <= 4, the register kernel."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.ops import get_backend


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--ks", default="1,2,4,5,8,10,16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--check-rows", type=int, default=1024, help="rows compared with a float64 product (0: none)")
    ap.add_argument("--modes", default="", help="comma list restricting the kernels timed (f16x2,wide,groups_of_4,register,"
                                                "f16x2_strip); f16x2 = the block kernel, f16x2_strip = round 3's strip kernel")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    be = get_backend()
    n, d = args.n, args.d
    g = torch.Generator().manual_seed(0)
    dt = torch.float32 if args.dtype == "f32" else torch.float64
    X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).to(dt).to(dev)
    Kp = be.sympack_materialize("rbf", X, 1.0)
    nbytes = Kp.numel() * Kp.element_size()
    for k in [int(v) for v in args.ks.split(",")]:
        V = torch.randn(n, k, dtype=torch.float64, generator=g).to(dt).to(dev)
        rec = {"n": n, "k": k, "packed_GB": round(nbytes / 1e9, 2)}
        outs = {}
        modes = ["f16x2", "wide", "groups_of_4"] if args.dtype == "f32" else ["wide", "groups_of_4"]
        if k <= 4:
            modes = ["register", "f16x2"] if args.dtype == "f32" else ["register"]
        if args.dtype == "f32":
            modes.append("f16x2_strip")
        if args.modes:
            modes = [m for m in modes if m in args.modes.split(",")]
        # (the accuracy check - float64 GEMMs - runs AFTER the timing of this k: right behind it the matrix-core /
        # LDS-staged kernels measured 8 - 15 % slower, the HBM-bound k = 1 kernel the same: clock state)
        for mode in modes:
            os.environ["RK_SYMM_WIDE"] = "0" if mode == "groups_of_4" else "1"
            os.environ["RK_SYMM_F16X2_MIN"] = "1"
            os.environ["RK_SYMM_F16X2_MIN_BYTES"] = "0"
            kmax = 1.0 if mode.startswith("f16x2") else None
            if mode == "f16x2_strip":
                os.environ["RK_SYMM_H2_IMPL"] = "strip"
            else:
                os.environ.pop("RK_SYMM_H2_IMPL", None)
            _sm = be.symm_matmat
            be_symm = lambda Kp_, n_, V_: _sm(Kp_, n_, V_, kmax=kmax)  # noqa: E731
            Y = be_symm(Kp, n, V)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                be_symm(Kp, n, V)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            outs[mode] = Y
            rec[mode] = {"ms": round(ms, 3), "packed_GBps_per_pass": round(nbytes / ms / 1e6, 1)}
        if args.check_rows:
            rows = torch.randperm(n, generator=g)[: args.check_rows].to(dev)
            want = be.kernel_materialize("rbf", X[rows].double(), X.double(), 1.0) @ V.double()
            for mode, Y in outs.items():
                rec[mode]["max_col_err_vs_f64"] = float(((Y[rows].double() - want).norm(dim=0) / want.norm(dim=0)).max())
            del want
        os.environ["RK_SYMM_WIDE"] = "1"
        os.environ.pop("RK_SYMM_H2_IMPL", None)
        os.environ.pop("RK_SYMM_F16X2_MIN", None)
        os.environ.pop("RK_SYMM_F16X2_MIN_BYTES", None)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
