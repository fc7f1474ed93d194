#!/usr/bin/env python3
"""Feasibility probe: does the HBM-bound packed-triangle product co-run with the matrix-core-bound
matrix-free product on one MI355X?  K = [[K11, K12], [K21, K22]] with K11 (m0 x m0) streamed from the
packed triangle and the border K21 / K12 / K22 recomputed on the bf16 matrix cores, on two streams."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.ops import get_backend


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--m0", type=str, default="160000,170000,150000")
    ap.add_argument("--d", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    be = get_backend()
    n, d = args.n, args.d
    g = torch.Generator().manual_seed(0)
    X = (torch.randn(n, d, dtype=torch.float64, generator=g) / d**0.5).float().to(dev)
    v = torch.randn(n, 1, dtype=torch.float64, generator=g).float().to(dev)
    gc = be.gram_center(X, X, 1.0)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def timed(fn, reps=6):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    for m0 in [int(x) for x in args.m0.split(",")]:
        X1 = X[:m0].contiguous()
        Kp = be.sympack_materialize("rbf", X1, 1.0)
        v1 = v[:m0].contiguous()
        Xb = X[m0:].contiguous()
        vb = v[m0:].contiguous()

        def stream_part():
            return be.symm_matmat(Kp, m0, v1)

        def free_part():
            a = be.kernel_matmat("rbf", Xb, X, 1.0, v, gram_center=gc)      # rows m0..n, all columns
            b = be.kernel_matmat("rbf", X1, Xb, 1.0, vb, gram_center=gc)    # K12 v2
            return a, b

        def both():
            cur = torch.cuda.current_stream(dev)
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                stream_part()
            with torch.cuda.stream(s2):
                free_part()
            cur.wait_stream(s1)
            cur.wait_stream(s2)

        def both_rev():
            cur = torch.cuda.current_stream(dev)
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                free_part()
            with torch.cuda.stream(s1):
                stream_part()
            cur.wait_stream(s1)
            cur.wait_stream(s2)

        rec = {"n": n, "m0": m0, "stream_ms": round(timed(stream_part), 3), "free_ms": round(timed(free_part), 3),
               "both_ms": round(timed(both), 3), "both_free_first_ms": round(timed(both_rev), 3)}
        # correctness of the split: y = K v assembled from the pieces vs the full matrix-free product
        y1 = stream_part()
        a, b = free_part()
        y = torch.cat([y1 + b, a])
        full = be.kernel_matmat("rbf", X, X, 1.0, v, gram_center=None if os.environ.get("RK_PROBE_DIRECT") else gc)
        rec["rel_err_vs_full"] = float((y - full).norm() / full.norm())
        print(json.dumps(rec), flush=True)
        del Kp


if __name__ == "__main__":
    main()
