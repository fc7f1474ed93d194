#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_kmatmat.sh: pmc_table.py <dir> [name filter]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        lt = name.find("<")
        key = name[name.rfind("::", 0, lt) + 2 : name.find(">") + 1] + f" grid={r.get('Grid_Size', '?')}"
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for key in sorted(acc):
    print(key)
    c = {k: v[0] / v[1] for k, v in acc[key].items()}
    for k in sorted(c):
        print(f"   {k:32s} {c[k]:.4g}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
            if k in c:
                print(f"   {k + ' / SQ_WAVE_CYCLES':44s} {c[k] / wc:.3f}")
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs (own pass each)
        print(f"   {'matrix-core busy = MFMA_BUSY / (GUI_ACTIVE / 8 x 1024 SIMDs)':44s} "
              f"{100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.1f} %   "
              f"(shader clock over the launch: GUI_ACTIVE / 8 cycles)")
    if "SQ_BUSY_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print(f"   {'MFMA_BUSY / (BUSY_CYCLES x 4 SIMDs.. raw ratio)':44s} {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
