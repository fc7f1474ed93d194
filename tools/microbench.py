#!/usr/bin/env python3
"""Per-kernel timings on one MI355X (HIP events on the launch stream).  Development aid:
prints achieved GB/s / TFLOP/s for the main kernels at bench-like shapes."""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.ops import get_backend


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="skinny,skinny_t,gemm,gram,kmat,kmatmat")
    ap.add_argument("--n", type=int, default=100000)
    args = ap.parse_args()
    be = get_backend()
    dev = torch.device("cuda:0")
    what = args.what.split(",")
    n = args.n
    if "skinny" in what:
        for dt in (torch.float32, torch.float64):
            es = 4 if dt == torch.float32 else 8
            for m in (n, n // 8):
                A = torch.randn(m, n, dtype=dt, device=dev)
                for k in (1, 4, 16):
                    V = torch.randn(n, k, dtype=dt, device=dev)
                    med, best = timeit(lambda: be.matmat(A, V, diag_shift=0.1, row0=0))
                    gb = (m * n + n * k + m * k) * es / 1e9
                    print(f"skinny {dt} m={m} n={n} k={k}: {med*1e3:.3f} ms  {gb/med:.0f} GB/s (best {gb/best:.0f})", flush=True)
                del A
    if "symm1" in what:  # the headline kernel only (tuning variants)
        X = torch.randn(200000, 64, device=dev) / 8
        Kp = be.sympack_materialize("rbf", X, 1.0)
        V = torch.randn(200000, 1, device=dev)
        med, best = timeit(lambda: be.symm_matmat(Kp, 200000, V, diag_shift=0.1), reps=9)
        print(f"symm n=200000 k=1: {med*1e3:.3f} ms (best {best*1e3:.3f}) {Kp.numel()*4/1e9/med:.0f} GB/s", flush=True)
        del Kp
    if "foot" in what:  # does the streaming rate depend on the footprint? (20 / 40 / 80 GB)
        V = torch.randn(200000, 1, device=dev)
        for rows in (25000, 50000, 100000):
            A = torch.empty(rows, 200000, device=dev)
            A.normal_()
            med, best = timeit(lambda: be.matmat(A, V, diag_shift=0.1), reps=7)
            print(f"skinny {rows}x200000 k=1: {med*1e3:.3f} ms (best {best*1e3:.3f}) {A.numel()*4/1e9/med:.0f} GB/s", flush=True)
            del A
    if "shape" in what:  # one 40 GB buffer streamed as matrices of different row lengths
        buf = torch.empty(10_000_000_000, device=dev)
        buf.normal_()
        for cols in (256, 1024, 8192, 65536, 200000):
            rows = buf.numel() // cols
            A = buf[: rows * cols].view(rows, cols)
            V = torch.randn(cols, 1, device=dev)
            med, best = timeit(lambda: be.matmat(A, V), reps=5)
            print(f"skinny {rows}x{cols} k=1: {med*1e3:.3f} ms (best {best*1e3:.3f}) {A.numel()*4/1e9/med:.0f} GB/s", flush=True)
        del buf, A
    if "nt" in what:  # the two headline streaming kernels at bench size
        X = torch.randn(200000, 64, device=dev) / 8
        Kp = be.sympack_materialize("rbf", X, 1.0)
        V = torch.randn(200000, 1, device=dev)
        med, best = timeit(lambda: be.symm_matmat(Kp, 200000, V, diag_shift=0.1), reps=7)
        print(f"symm n=200000 k=1: {med*1e3:.3f} ms (best {best*1e3:.3f}) {Kp.numel()*4/1e9/med:.0f} GB/s", flush=True)
        del Kp
        A = be.kernel_materialize("rbf", X[:25000], X, 1.0)
        med, best = timeit(lambda: be.matmat(A, V, diag_shift=0.1), reps=7)
        print(f"skinny 25000x200000 k=1: {med*1e3:.3f} ms (best {best*1e3:.3f}) {A.numel()*4/1e9/med:.0f} GB/s", flush=True)
        del A
    if "tall" in what:
        for dt in (torch.float64, torch.float32):
            es = 8 if dt == torch.float64 else 4
            m, d = 2_000_000, 512
            A = torch.randn(m, d, dtype=dt, device=dev)
            x = torch.randn(d, 1, dtype=dt, device=dev)
            u = torch.randn(m, 1, dtype=dt, device=dev)
            med, _ = timeit(lambda: be.matmat(A, x))
            print(f"tall A x {dt} {m}x{d}: {med*1e3:.3f} ms {m*d*es/med/1e9:.0f} GB/s", flush=True)
            med, _ = timeit(lambda: be.rmatmat(A, u))
            print(f"tall A^T u {dt} {m}x{d}: {med*1e3:.3f} ms {m*d*es/med/1e9:.0f} GB/s", flush=True)
            med, _ = timeit(lambda: be.rmatmat(u, u, out_f64=True))
            print(f"tall u^T u {dt} m={m}: {med*1e3:.3f} ms", flush=True)
            med, _ = timeit(lambda: be.axpby_cols(u, None, u))
            print(f"tall axpby {dt} m={m}: {med*1e3:.3f} ms", flush=True)
            del A
    if "symm" in what:
        for dt in ((torch.float32,) if n > 150000 else (torch.float32, torch.float64)):
            es = 4 if dt == torch.float32 else 8
            X = torch.randn(n, 64, dtype=dt, device=dev) / 8
            Kp = be.sympack_materialize("rbf", X, 1.0)
            torch.cuda.synchronize()
            med, best = (0.0, 0.0)
            print(f"sympack materialize {dt} n={n}: {med*1e3:.2f} ms  ({Kp.numel()*es/1e9:.2f} GB packed)", flush=True)
            for k in (1, 2, 4):
                V = torch.randn(n, k, dtype=dt, device=dev)
                med, best = timeit(lambda: be.symm_matmat(Kp, n, V, diag_shift=0.1))
                gb = Kp.numel() * es / 1e9
                print(f"symm {dt} n={n} k={k}: {med*1e3:.3f} ms  {gb/med:.0f} GB/s of packed bytes = {n*n*es/1e9/med:.0f} GB/s dense-equivalent", flush=True)
            del Kp
    if "skinny_t" in what:
        for r, k in ((512, 1), (512, 4), (100, 10)):
            U = torch.randn(2 * n, r, dtype=torch.float32, device=dev)
            R = torch.randn(2 * n, k, dtype=torch.float32, device=dev)
            med, best = timeit(lambda: be.rmatmat(U, R, out_f64=True))
            gb = (2 * n * r + 2 * n * k) * 4 / 1e9
            print(f"skinny_t U^T R rows={2*n} r={r} k={k}: {med*1e3:.3f} ms {gb/med:.0f} GB/s", flush=True)
            T = torch.randn(r, k, dtype=torch.float32, device=dev)
            med, best = timeit(lambda: be.matmat(U, T, alpha=-1.0, beta=1.0, C=R))
            print(f"skinny U T rows={2*n} r={r} k={k}: {med*1e3:.3f} ms {gb/med:.0f} GB/s", flush=True)
    if "gemm" in what:
        for dt in (torch.float32, torch.float64):
            m = n // 2
            A = torch.randn(m, n, dtype=dt, device=dev)
            B = torch.randn(n, 512, dtype=dt, device=dev)
            med, best = timeit(lambda: be.gemm(A, B), reps=3, warm=1)
            fl = 2.0 * m * n * 512
            print(f"gemm {dt} {m}x{n}x512: {med*1e3:.2f} ms {fl/med/1e12:.1f} TF/s", flush=True)
            del A
        Y = torch.randn(2 * n, 512, dtype=torch.float32, device=dev)
        M = torch.randn(512, 512, dtype=torch.float32, device=dev)
        med, best = timeit(lambda: be.gemm(Y, M))
        print(f"gemm Y M {2*n}x512x512: {med*1e3:.2f} ms {2.0*2*n*512*512/med/1e12:.1f} TF/s", flush=True)
    if "gram" in what:
        Y = torch.randn(2 * n, 512, dtype=torch.float32, device=dev)
        med, best = timeit(lambda: be.gram_tn(Y, Y))
        print(f"gram Y^T Y rows={2*n} r=512: {med*1e3:.2f} ms {2.0*2*n*512*512/med/1e12:.1f} TF/s(f64)", flush=True)
    if "kmat" in what:
        for d in (32, 64):
            X = torch.randn(n, d, dtype=torch.float32, device=dev) / d**0.5
            m = n // 2
            med, best = timeit(lambda: be.kernel_materialize("rbf", X[:m], X, 1.0), reps=3, warm=1)
            print(f"materialize rbf f32 {m}x{n} d={d}: {med*1e3:.2f} ms {m*n/med/1e9:.1f} Gpair/s  write {m*n*4/med/1e9:.0f} GB/s", flush=True)
    if "kmatmat" in what:
        for d, k in ((64, 1), (50, 10)):
            X = torch.randn(n, d, dtype=torch.float32, device=dev) / d**0.5
            V = torch.randn(n, k, dtype=torch.float32, device=dev)
            m = n // 4
            med, best = timeit(lambda: be.kernel_matmat("rbf", X[:m], X, 1.0, V), reps=3, warm=1)
            print(f"fused matmat rbf f32 {m}x{n} d={d} k={k}: {med*1e3:.2f} ms {m*n/med/1e9:.1f} Gpair/s", flush=True)
        blk = torch.randperm(n, device=dev)[:5000]
        med, best = timeit(lambda: be.kernel_matmat("rbf", X, X, 1.0, V, idx1=blk), reps=3, warm=1)
        print(f"row-oracle 5000x{n} d=50 k=10: {med*1e3:.2f} ms {5000*n/med/1e9:.1f} Gpair/s", flush=True)


if __name__ == "__main__":
    main()
