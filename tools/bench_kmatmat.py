#!/usr/bin/env python3
"""Matrix-free kernel product ``rk_kernel_matmat`` on one MI355X: G pair-evaluations/s of the three
formulations side by side in ONE process (the environment switches are read per call):

  bf16x3   three-way bf16 split on the bf16 matrix cores   (default)
  mfma32   float32 matrix cores                            (RK_KMAT_BF16X3=0)
  direct   direct-difference VALU kernels                  (RK_KMAT_MFMA=0)

plus the worst entry-wise deviation of each from the float64 closed form on a sample of columns."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.ops import get_backend

MODES = {"bf16x3": {}, "bf16x3_full": {"RK_KMAT_SYM": "0"}, "mfma32": {"RK_KMAT_BF16X3": "0"},
         "direct": {"RK_KMAT_MFMA": "0"}}


def set_mode(mode):
    for key in ("RK_KMAT_BF16X3", "RK_KMAT_MFMA", "RK_KMAT_SYM"):
        os.environ.pop(key, None)
    os.environ.update(MODES[mode])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cases", default="200000x200000x64x1,200000x200000x32x1,5000x500000x50x10g,"
                                       "25000x200000x64x1,200000x200000x64x4,5000x500000x50x16g,50000x50000x32x1,"
                                       "100000x100000x100x1,100000x100000x128x10")
    ap.add_argument("--modes", default="bf16x3,mfma32,direct")
    ap.add_argument("--kind", default="rbf")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    be = get_backend()
    out = []
    for case in args.cases.split(","):
        gather = case.endswith("g")
        n1, n2, d, k = (int(v) for v in case.rstrip("g").split("x"))
        g = torch.Generator().manual_seed(0)
        X = (torch.randn(n2, d, dtype=torch.float64, generator=g) / d**0.5).float().to(dev)
        V = torch.randn(n2, k, dtype=torch.float64, generator=g).float().to(dev)
        if gather:  # the SAP row oracle: K(X[blk], X) @ V
            idx1 = torch.randperm(n2, generator=g)[:n1].to(dev)
            X1 = X
        else:
            idx1 = None
            X1 = X if n1 == n2 else X[:n1].contiguous()
        gc = be.gram_center(X1, X, 1.0)
        assert gc is not None
        # float64 reference on 16 sampled rows
        rows = torch.arange(0, n1, max(1, n1 // 16))[:16].to(dev)
        A = (X1[idx1[rows]] if gather else X1[rows]).double()
        D2 = (A * A).sum(1)[:, None] + (X.double() ** 2).sum(1)[None, :] - 2 * A @ X.double().T
        if args.kind == "rbf":
            Kref = torch.exp(-0.5 * D2.clamp_min(0))
        elif args.kind == "matern32":
            Dm = D2.clamp_min(0).sqrt() * 3**0.5
            Kref = (1 + Dm) * torch.exp(-Dm)
        elif args.kind == "matern52":
            D0 = D2.clamp_min(0).sqrt()
            Kref = (1 + 5**0.5 * D0 + 5.0 / 3.0 * D0 * D0) * torch.exp(-(5**0.5) * D0)
        else:
            raise SystemExit("--kind: rbf, matern32 or matern52")
        want = Kref @ V.double()
        rec = {"case": case, "kind": args.kind, "pairs": n1 * n2}
        for mode in args.modes.split(","):
            set_mode(mode)
            run = lambda: be.kernel_matmat(args.kind, X1, X, 1.0, V, idx1=idx1, gram_center=gc)  # noqa: E731
            Y = run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            err = float((Y[rows].double() - want).abs().max() / want.abs().max())
            rec[mode] = {"ms": round(ms, 3), "Gpair_s": round(n1 * n2 / ms / 1e6, 1), "rel_err_rows": err}
        set_mode("bf16x3")
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
