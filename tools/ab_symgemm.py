#!/usr/bin/env python3
"""rk_symm_gemm: the two kernels side by side in one process (RK_SYMGEMM_IMPL = lockstep | ph): equality of the results
bit for bit on small and ragged sizes, then alternating timings on the bench size.

    python tools/ab_symgemm.py [--n 200000] [--r 512] [--rounds 3] [--stride 1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from rlaopt_amd.ops import get_backend  # noqa: E402


def run(be, impl, *a, **k):
    os.environ["RK_SYMGEMM_IMPL"] = "ph" if impl == "rhs16" else impl
    try:
        return be.symm_gemm(*a, rhs_f16=(impl == "rhs16"), **k)  # ("rhs16": the one-piece kernel, V rounded to its grid)
    finally:
        del os.environ["RK_SYMGEMM_IMPL"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--r", type=int, nargs="+", default=[512])
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--impls", nargs="+", default=["lockstep", "ph"])
    ap.add_argument("--skip-small", action="store_true")
    ap.add_argument("--trace", action="store_true",
                    help="a -DRK_SG_TRACE=1 build (RK_HIP_LIBRARY): print the cycle stamps after the barriers of 16 steps")
    args = ap.parse_args()
    be = get_backend()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    if not args.skip_small:
        for n, k, first, stride in [(300, 40, 0, 1), (1111, 96, 0, 1), (2500, 130, 0, 1), (5003, 512, 0, 1), (4100, 200, 1, 3),
                                    (4100, 200, 0, 3), (4100, 200, 2, 3), (9000, 300, 4, 8), (777, 17, 0, 2), (12345, 1024, 0, 1)]:
            X = (torch.randn(n, 16, generator=g) / 4).to(dev)
            Kp = be.sympack_materialize("rbf", X, 1.0, first=first, stride=stride)
            V = torch.randn(n, k, generator=g).to(dev)
            a = run(be, "lockstep", Kp, n, V, first=first, stride=stride)
            b = run(be, "ph", Kp, n, V, first=first, stride=stride)
            b2 = run(be, "ph", Kp, n, V, first=first, stride=stride)
            torch.cuda.synchronize()
            same = torch.equal(a, b) and torch.equal(b, b2)
            print(f"n={n} k={k} strips {first}::{stride}: equal {same}  max|diff| {(a - b).abs().max().item():.3e} "
                  f"(|Y| max {a.abs().max().item():.3e})", flush=True)
            if not same:
                bad = (a != b).nonzero()
                print("   first differing entries (row, col):", bad[:8].tolist(), " count", bad.shape[0], flush=True)
    n = args.n
    X = (torch.randn(n, 64, dtype=torch.float64, generator=g) / 8).float().to(dev)
    Kp = be.sympack_materialize("rbf", X, 1.0, first=0, stride=args.stride)
    for r in args.r:
        Om = (torch.randn(n, r, generator=g) / r**0.5).half().float().to(dev)  # (on the float16 grid: every kernel agrees)
        ref = None
        for rd in range(args.rounds):
            for impl in args.impls:
                out = run(be, impl, Kp, n, Om, first=0, stride=args.stride)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    out = run(be, impl, Kp, n, Om, first=0, stride=args.stride)
                e1.record()
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                print(f"r={r} stride={args.stride} round {rd} {impl:9s}: {e0.elapsed_time(e1) / 3:.2f} ms  equal to first: "
                      f"{torch.equal(out, ref)}", flush=True)
        if args.trace:
            dump_trace(be)


def dump_trace(be):
    """(a -DRK_SG_TRACE=1 build; the one-piece kernel of the LAST launch: run with --impls rhs16)"""
    import ctypes

    P = 6
    buf = (ctypes.c_ulonglong * (16 * P * 2))()
    be._lib.rk_debug_symgemm_trace.restype = ctypes.c_int
    be._lib.rk_debug_symgemm_trace.argtypes = [ctypes.c_void_p]
    assert be._lib.rk_debug_symgemm_trace(buf) == 0
    names = ["L start", "M start (barrier passed)", "operands in", "matrix-core work issued", "vmcnt wait done", "lgkmcnt wait done"]
    t = [[[buf[(q * P + p) * 2 + g] for g in range(2)] for p in range(P)] for q in range(16)]
    t0 = t[0][0][0]
    print("cycle stamps of one workgroup, relative (group 0 | group 1): " + ", ".join(names))
    for q in range(16):
        print(f"step {q:2d}: " + " ".join(f"{t[q][p][0] - t0:6d}|{t[q][p][1] - t0:6d}" for p in range(P)))
    for g in range(2):
        def mean(f):
            v = [f(q) for q in range(15)]
            return sum(v) / len(v)
        print(f"group {g}: L issue + barrier wait {mean(lambda q: t[q][1][g] - t[q][0][g]):.0f}, operand wait {mean(lambda q: t[q][2][g] - t[q][1][g]):.0f}, "
              f"M work {mean(lambda q: t[q][3][g] - t[q][2][g]):.0f}, vmcnt wait {mean(lambda q: t[q][4][g] - t[q][3][g]):.0f}, "
              f"lgkmcnt wait {mean(lambda q: t[q][5][g] - t[q][4][g]):.0f}, barrier {mean(lambda q: t[q + 1][0][g] - t[q][5][g]):.0f}; "
              f"step {mean(lambda q: t[q + 1][0][g] - t[q][0][g]):.0f}")


if __name__ == "__main__":
    main()
