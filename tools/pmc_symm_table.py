#!/usr/bin/env python3
"""Table of the counters collected by tools/pmc_symm.sh: per k, HBM bytes of the block kernel and of the finishing kernel
(FETCH_SIZE x 2 + WRITE_SIZE, in KB, corrected as MI355X_MICROARCH.md 'HBM' prescribes) against the algorithmic bytes,
and the matrix cores' busy fraction.   pmc_symm_table.py gpurun_out/<tag>_symm_pmc [n]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
strips = -(-n // 256)
packed = sum(min((s + 1) * 256, n) for s in range(strips)) * 256 * 4
print(f"packed triangle n = {n}: {packed / 1e9:.2f} GB;  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024\n")
print("| k | kernel | launches | FETCH_SIZE x 2 (GB) | WRITE_SIZE (GB) | HBM GB per launch | algorithmic GB | ratio | MFMA busy |")
print("|---|---|---|---|---|---|---|---|---|")
for k in (1, 5, 10, 16):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{root}/k{k}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            for key in ("symm_blk_h2_kernel", "symm_blk_finish_kernel"):
                if key in name:
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tot = 0.0
    for key in ("symm_blk_h2_kernel", "symm_blk_finish_kernel"):
        c = {m: sum(v) / len(v) for m, v in acc[key].items() if v}
        if "FETCH_SIZE" not in c:
            continue
        fetch, write = 2.0 * c["FETCH_SIZE"] * 1024 / 1e9, c.get("WRITE_SIZE", 0.0) * 1024 / 1e9
        tot += fetch + write
        busy = ""
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            # MFMA_BUSY is summed over the SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: busy / (gui / 8 x 1024 SIMDs)
            busy = f"{100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0):.1f} %"
        algo = packed + 2 * n * k * 4 if key == "symm_blk_h2_kernel" else 0
        print(f"| {k} | `{key}` | {len(acc[key]['FETCH_SIZE'])} | {fetch:.2f} | {write:.3f} | {fetch + write:.2f} | "
              f"{algo / 1e9:.2f} | {'' if not algo else f'{(fetch + write) / (algo / 1e9):.3f}'} | {busy} |")
    if tot:
        algo = packed + 2 * n * k * 4
        print(f"| {k} | **product (both launches)** | | | | **{tot:.2f}** | {algo / 1e9:.2f} | **{tot / (algo / 1e9):.3f}** | |")
