#!/usr/bin/env python3
"""Row-major streaming product ``rk_matmat_skinny`` (the p = 8 shard shape 25 000 x 200 000, float32)
for 1 .. 16 right-hand sides: ms and GB/s; 9 <= k <= 16 with the matrix-core kernel against the
LDS-shared VALU kernel (RK_SKINNY_MFMA=0).  ``--f64``: float64 (a 12 500 x 200 000 shard)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from rlaopt_amd.ops import get_backend

dev = torch.device("cuda:0")
be = get_backend()
dt = torch.float64 if "--f64" in sys.argv else torch.float32
m, n = (12_500 if dt == torch.float64 else 25_000), 200_000
A = torch.randn(m, n, device=dev, dtype=dt)
# Warm the card before the first measurement: right after the 20 GB allocation and its fill the first
# launches run at about half rate (the k = 1 rows of an earlier version of this tool read 4.1 TB/s for a
# kernel that streams 6.7) - half a second of the product itself, then every figure below is the MEDIAN of
# five groups of five launches.
_w = torch.randn(n, 1, device=dev, dtype=dt)
import time as _time

_t = _time.time()
while _time.time() - _t < 0.5:
    for _ in range(10):
        be.matmat(A, _w)
    torch.cuda.synchronize()
for k in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16):
    V = torch.randn(n, k, device=dev, dtype=dt)
    rec = {"m": m, "n": n, "k": k, "dtype": str(dt)}
    kmin = 7 if dt == torch.float64 else 9
    for mode in (["mfma", "valu_lds"] if k >= kmin else (["register", "valu_lds"] if k in (3, 4) else ["default"])):
        os.environ["RK_SKINNY_MFMA"] = "0" if mode == "valu_lds" else "2"
        os.environ["RK_SKINNY_LDS_MIN"] = "5" if mode == "register" else "3"
        for _ in range(5):
            be.matmat(A, V)
        torch.cuda.synchronize()
        groups = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                be.matmat(A, V)
            e1.record()
            torch.cuda.synchronize()
            groups.append(e0.elapsed_time(e1) / 5)
        ms = sorted(groups)[2]
        rec[mode] = {"ms": round(ms, 3), "GBps": round((m * n + n * k + m * k) * A.element_size() / ms / 1e6, 1)}
    os.environ["RK_SKINNY_MFMA"] = "1"
    os.environ["RK_SKINNY_LDS_MIN"] = "3"
    print(json.dumps(rec), flush=True)
