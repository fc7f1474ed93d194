#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace CSV of tools/bench_sap.py into a per-step timeline: for every
row-oracle product launch (the big kernel) list how much of the latency-bound chain (every other
kernel of the step) ran while it was executing.  Usage: sap_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
big = [e for e in ev if "kmat_matmat_bf16x3_kernel" in e[2] or "kmat_matmat_mfma_kernel" in e[2]]
big = [e for e in big if e[1] - e[0] > 1_000_000]  # > 1 ms
print(f"{len(ev)} kernels, {len(big)} row-oracle products")
for i in range(1, min(len(big), 9)):
    s0, s1 = big[i - 1][0], big[i][0]  # one step: from this product's start to the next one's
    b0, b1 = big[i - 1][0], big[i - 1][1]
    step = [e for e in ev if s0 <= e[0] < s1 and e is not big[i - 1]]
    inside = [e for e in step if e[0] >= b0 and e[1] <= b1]
    busy_in = sum(e[1] - e[0] for e in inside)
    busy_out = sum(e[1] - e[0] for e in step if e not in inside)
    queues = sorted({e[3] for e in step} | {big[i - 1][3]})
    print(f"step {i}: period {(s1 - s0) / 1e6:.2f} ms, product {(b1 - b0) / 1e6:.2f} ms on queue {big[i-1][3]}; "
          f"{len(step)} other kernels ({len(inside)} inside the product, {busy_in / 1e6:.2f} ms busy; "
          f"outside {busy_out / 1e6:.2f} ms busy), queues {queues}")
    if i == 2:
        last = b0
        for e in step[:400]:
            where = "IN " if (e[0] >= b0 and e[1] <= b1) else "out"
            print(f"   {where} +{(e[0] - b0) / 1e3:8.1f} us  dur {(e[1] - e[0]) / 1e3:7.1f} us  gap {(e[0] - last) / 1e3:7.1f}  q{e[3]}  {e[2][:70]}")
            last = e[1]
