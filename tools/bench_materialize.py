#!/usr/bin/env python3
"""Materialisation of kernel-matrix storage: the packed triangle (n = 200 000), a row-major shard and an ASkotch block.

    python tools/bench_materialize.py [--n 200000] [--d 64]     (RK_HIP_LIBRARY selects a tuning build)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rlaopt_amd.ops import get_backend  # noqa: E402


def timed(fn, reps=3):
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
    a = ap.parse_args()
    be = get_backend()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    X = (torch.randn(a.n, a.d, generator=g) / a.d**0.5).to(dev)
    for kind in ("rbf", "matern52", "laplace"):
        buf = torch.empty(be.sympack_elems(a.n, 0, 1), dtype=torch.float32, device=dev)
        ms, Kp = timed(lambda: be.sympack_materialize(kind, X, 1.0, out=buf))
        chk = float(Kp[:: max(1, Kp.numel() // 1000003)].double().sum())
        print(f"packed triangle n={a.n} d={a.d} {kind:9s}: {ms:8.2f} ms  ({Kp.numel() / ms / 1e6:7.1f} Gentry/s, "
              f"{Kp.numel() * 4 / ms / 1e6:7.1f} GB/s written)  checksum {chk:.10e}", flush=True)
        del buf, Kp
    rows = 25000
    gc = be.gram_center(X, X, 1.0)  # (the matrix-core evaluation needs the centre; RK_SYMPACK_MC=0 ignores it)
    ms, K = timed(lambda: be.kernel_materialize("rbf", X[:rows], X, 1.0, gram_center=gc))
    print(f"row-major shard {rows} x {a.n} rbf: {ms:8.2f} ms  ({K.numel() / ms / 1e6:7.1f} Gentry/s)  checksum "
          f"{float(K[::97, ::1013].double().sum()):.10e}", flush=True)
    del K
    Xb = (torch.randn(500000, 50, generator=g) / 50**0.5).to(dev)
    blk = torch.randperm(500000, generator=g)[:5000].to(dev)
    Xs = Xb[blk].contiguous()
    gcb = be.gram_center(Xb, Xb, 1.0)
    ms, Kb = timed(lambda: be.kernel_materialize("rbf", Xs, Xs, 1.0, gram_center=gcb), reps=20)
    print(f"ASkotch block 5000 x 5000 d=50 rbf: {1e3 * ms:8.1f} us  checksum {float(Kb.double().sum()):.10e}", flush=True)


if __name__ == "__main__":
    main()
