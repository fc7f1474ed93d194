"""In-process A/B timing of tuning variants of librk_hip.so (built by tools/variants.sh).

Process-to-process (physical page placement) and box-to-box differences are ~3 % on this
kernel - more than most tuning steps - so the variants are loaded side by side with ctypes and
timed alternately on the SAME buffers:

    python tools/ab.py --kernel symm  joint quarters j_rb16
    python tools/ab.py --kernel skinny base nopipe
"""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rlaopt_amd import _lib  # noqa: E402
from rlaopt_amd.ops import get_backend  # noqa: E402


def load_variant(name):
    name = name.split("@")[0]  # "main@strip": the main build with RK_SYMM_H2_IMPL=strip around its calls
    path = os.path.join(ROOT, "build_var", name, "librk_hip.so") if name != "main" else _lib.LIB_PATH
    lib = ctypes.CDLL(path)
    for sym, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(lib, sym)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def time_call(fn, reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    fn()
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--kernel", default="symm", choices=["symm", "symmh2", "skinny", "kmatmat"])
    ap.add_argument("--n", type=int, default=200000)
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--store", default="same", choices=["same", "f32"], help="symm: packed storage dtype")
    ap.add_argument("--stride", type=int, default=1, help="symm: strips first, first+stride, ... (rank 0 of p)")
    ap.add_argument("--d", type=int, default=50, help="kmatmat: features")
    ap.add_argument("--blk", type=int, default=5000, help="kmatmat: gathered rows (row oracle)")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    dti = _lib.RK_F32 if a.dtype == "f32" else _lib.RK_F64
    be = get_backend()
    libs = {v: load_variant(v) for v in a.variants}
    n, k = a.n, a.k
    V = torch.randn(n, k, dtype=dt, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    if a.kernel == "symm":
        X = torch.randn(n, 64, dtype=dt, device=dev) / 8
        Kp = be.sympack_materialize("rbf", X, 1.0, store_dtype=torch.float32 if a.store == "f32" else None,
                                    first=0, stride=a.stride)
        sti = _lib.RK_F32 if Kp.dtype == torch.float32 else _lib.RK_F64
        Y = torch.empty(n, k, dtype=dt, device=dev)
        nbytes = Kp.numel() * Kp.element_size()

        def call(lib):
            wsn = lib.rk_symm_matmat_ws_bytes(dti, sti, n, k, 0, a.stride)
            ws = torch.empty(wsn, dtype=torch.uint8, device=dev)

            def fn():
                rc = lib.rk_symm_matmat(dti, sti, Kp.data_ptr(), n, 0, a.stride, V.data_ptr(), k, k, 1.0, 0.1,
                                        Y.data_ptr(), k, ws.data_ptr(), wsn, stream)
                assert rc == 0, lib.rk_last_error()
            return fn
    elif a.kernel == "symmh2":  # the float16-split packed symmetric product (rk_symm_matmat_f16x2)
        X = torch.randn(n, 64, dtype=dt, device=dev) / 8
        Kp = be.sympack_materialize("rbf", X, 1.0, first=0, stride=a.stride)
        Y = torch.empty(n, k, dtype=dt, device=dev)
        nbytes = Kp.numel() * Kp.element_size()

        def call(lib):
            wsn = lib.rk_symm_matmat_f16x2_ws_bytes(n, k, 0, a.stride)
            ws = torch.empty(wsn, dtype=torch.uint8, device=dev)

            def fn():
                rc = lib.rk_symm_matmat_f16x2(Kp.data_ptr(), n, 0, a.stride, 1.0, V.data_ptr(), k, k, 1.0, 0.1,
                                              Y.data_ptr(), k, ws.data_ptr(), wsn, stream)
                assert rc == 0, lib.rk_last_error()
            return fn
    elif a.kernel == "kmatmat":
        X = torch.randn(n, a.d, dtype=dt, device=dev) / a.d**0.5
        blk = torch.randperm(n, device=dev)[: a.blk].contiguous()
        Y = torch.empty(a.blk, k, dtype=dt, device=dev)
        nbytes = a.blk * n  # "bytes" column = pair evaluations here
        gcen = be.gram_center(X, X, 1.0)  # (None: the direct-difference kernels are timed)

        def call(lib):
            wsn = lib.rk_kernel_matmat_ws_bytes(dti, a.blk, n, k)
            ws = torch.empty(max(wsn, 16), dtype=torch.uint8, device=dev)

            def fn():
                rc = lib.rk_kernel_matmat(dti, 0, X.data_ptr(), a.blk, a.d, blk.data_ptr(), X.data_ptr(), n,
                                          a.d, None, a.d, 1.0, None, 1.0, V.data_ptr(), k, k, Y.data_ptr(),
                                          k, 0, None if gcen is None else gcen.data_ptr(), ws.data_ptr(), wsn,
                                          stream)
                assert rc == 0, lib.rk_last_error()
            return fn
    else:
        A = torch.empty(a.rows, n, dtype=dt, device=dev)
        A.normal_()
        Y = torch.empty(a.rows, k, dtype=dt, device=dev)
        nbytes = A.numel() * A.element_size()

        def call(lib):
            wsn = lib.rk_matmat_skinny_ws_bytes(dti, a.rows, n, k)
            ws = torch.empty(max(wsn, 16), dtype=torch.uint8, device=dev)

            def fn():
                rc = lib.rk_matmat_skinny(dti, A.data_ptr(), a.rows, n, n, V.data_ptr(), k, k, 1.0, 0.0,
                                          None, 0, 0.0, 0, Y.data_ptr(), k, ws.data_ptr(), wsn, stream)
                assert rc == 0, lib.rk_last_error()
            return fn
    def with_env(v, fn):
        if "@" not in v:
            def plain():
                os.environ.pop("RK_SYMM_H2_IMPL", None)
                fn()
            return plain
        impl = v.split("@")[1]

        def wrapped():
            os.environ["RK_SYMM_H2_IMPL"] = impl
            fn()
        return wrapped

    fns = {v: with_env(v, call(lib)) for v, lib in libs.items()}
    ref = None
    for v, fn in fns.items():  # same answers (fp reassociation only)
        fn()
        torch.cuda.synchronize()
        if ref is None:
            ref = Y.clone()
        else:
            err = float((Y - ref).abs().max() / ref.abs().max())
            print(f"# {v}: max rel diff vs {a.variants[0]} = {err:.2e}", flush=True)
    res = {v: [] for v in fns}
    for _ in range(a.rounds):
        for v, fn in fns.items():
            res[v].append(time_call(fn, a.reps))
    for v, ts in res.items():
        med = statistics.median(ts)
        unit = "Gpair/s" if a.kernel == "kmatmat" else "GB/s"
        print(f"{a.kernel} {v:12s} median {med:.3f} ms  ({nbytes / med / 1e6:.0f} {unit})  rounds: "
              + " ".join(f"{t:.3f}" for t in ts), flush=True)


if __name__ == "__main__":
    main()
