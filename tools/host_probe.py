#!/usr/bin/env python3
"""What the GPU box's host gives this process: CPU affinity, cgroup quota, and how the two host paths
of the CPU baseline (MKL sgemv over a dense K; the C/OpenMP matrix-free product) scale with threads."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import cbind  # noqa: E402

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "-", e)
try:
    print("loadavg", open("/proc/loadavg").read().strip())
    mem = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/meminfo") if l.startswith(("MemTotal", "MemAvailable"))}
    print("mem", mem)
except OSError:
    pass
print("torch threads", torch.get_num_threads(), "omp", cbind.num_threads())
lib = cbind.load()
n, d = 12000, 64
g = torch.Generator().manual_seed(0)
X = (torch.randn(n, d, generator=g) / d**0.5)
v = torch.randn(n, 1, generator=g)
for t in (8, 16, 32, 64, 128):
    torch.set_num_threads(t)
    lib.omp_set_num_threads(t)
    t0 = time.time(); K = cbind.kernel_matrix("rbf", X, X, 1.0); tk = time.time() - t0
    K @ v
    t0 = time.time()
    for _ in range(20):
        K @ v
    tg = (time.time() - t0) / 20
    t0 = time.time(); cbind.kernel_matmat("rbf", X[:4000], X, 1.0, v); tm = time.time() - t0
    print(f"threads {t:4d}: dense K build {tk:.3f}s  sgemv {tg * 1e3:.2f} ms ({n * n * 4 / tg / 1e9:.0f} GB/s)  "
          f"matrix-free 4000 rows {tm:.3f}s ({4000 * n / tm / 1e9:.3f} Gpair/s)", flush=True)
