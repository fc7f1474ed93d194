#!/usr/bin/env python3
"""Condense rocprofv3 CSV output of `tools/gpu_round.sh <tag>` into committed summaries:

  profiles/<tag>_kernel_stats.csv      per-kernel totals of the bench run (--kernel-trace --stats)
  profiles/<tag>_summary.md            readable table + HBM traffic of the dominant kernel
  profiles/<tag>_traffic.json          machine-readable HBM bytes per launch (bench.py reads it)

HBM traffic follows MI355X_MICROARCH.md "HBM": bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 with the
gfx950 correction FETCH_SIZE x 2 for wide coalesced streaming reads; the two counters are
collected in separate passes."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
dst = "profiles"
os.makedirs(dst, exist_ok=True)
stats = f"{src}/{tag}_prof/{tag}_kernel_stats.csv"
shutil.copy(stats, f"{dst}/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/{tag}_bench_under_rocprof.log", f"{dst}/{tag}_bench_under_rocprof.log")
shutil.copy(f"{src}/{tag}_bench_n1.log", f"{dst}/{tag}_bench_n1.log")
rows = list(csv.DictReader(open(stats)))


def medians(trace_path):
    """Median / min duration (us) per kernel name from a --kernel-trace CSV: a process's first launch of a
    kernel includes its code-object load (one 199 ms first launch made a 2.2 ms kernel average 12 ms in the
    round-2 ASkotch table) - the median is what a steady-state launch costs."""
    per = {}
    if os.path.exists(trace_path):
        for r in csv.DictReader(open(trace_path)):
            per.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for name, v in per.items():
        v.sort()
        out[name] = (v[len(v) // 2], v[0])
    return out


bench = json.loads([l for l in open(f"{src}/{tag}_bench_n1.log") if l.startswith("{")][-1])
storage = bench["roofline"].get("storage", "materialize")
kdesc = bench["roofline"].get("kernel", "")
if storage.startswith("sympack"):
    KNAME = ("symm_blk_reg_kernel" if "symm_blk_reg_kernel" in kdesc else
             "symm_blk_h2_kernel" if "symm_blk_h2_kernel" in kdesc else "symm_strip_kernel<float, float, 1>")
else:
    KNAME = "skinny_nn_kernel<float, 1, 8"
# the product's second launch (it reads the partial tables the first one wrote): its traffic belongs to the product
FINISH = {"symm_blk_reg_kernel": "symm_blk_finish_t_kernel", "symm_blk_h2_kernel": "symm_blk_finish_kernel", "symm_strip_kernel<float, float, 1>": "symm_finish_kernel<float>"}.get(KNAME)


def pmc(kind):
    path = f"{src}/{tag}_pmc_{kind}/{tag}_counter_collection.csv"
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if KNAME in r["Kernel_Name"] and int(r["Grid_Size"]) > 100_000]
    big = [v for v in vals if v > 0.5 * max(vals)]  # the shard products, not the small U @ t ones
    fin = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
           if FINISH is not None and FINISH in r["Kernel_Name"]]
    fin = [v for v in fin if v > 0.5 * max(fin)] if fin else []
    return sum(big) / len(big) + (sum(fin) / len(fin) if fin else 0.0), len(big)


trace = list(csv.DictReader(open(f"{src}/{tag}_prof/{tag}_kernel_trace.csv")))
shard = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in trace
         if KNAME in r["Kernel_Name"]]
shard = [v for v in shard if v > 0.5 * max(shard)]
shard_avg_ms = sum(shard) / len(shard)
fetch, nf = pmc("fetch")
write, nw = pmc("write")
traffic = (2.0 * fetch + write) * 1024.0
algo = bench["roofline"]["bytes_per_launch"]
out = {
    "tag": tag, "kernel": KNAME, "storage": storage, "launches_sampled": [nf, nw],
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "fetch_correction": 2.0,
    "hbm_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": algo,
    "traffic_over_algorithmic": traffic / algo,
    "workload": bench["config"]["workload"], "n_gpus": bench["n_gpus"],
    "rocprof_shard_product_avg_ms": shard_avg_ms, "rocprof_shard_product_launches": len(shard),
}
json.dump(out, open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
with open(f"{dst}/{tag}_summary.md", "w") as f:
    f.write(f"# rocprofv3 summary `{tag}` - `bench.py` (defaults) on one MI355X\n\n")
    f.write(f"bench line: value {bench['value']:.2f} iter/s, {bench['ms_per_step']:.2f} ms/step; "
            f"roofline achieved {bench['roofline']['achieved']:.0f} GB/s "
            f"({100 * bench['roofline']['frac']:.1f} % of 8000), avg launch "
            f"{bench['roofline']['avg_launch_ms']:.3f} ms (HIP events in bench.py)\n\n")
    f.write("## kernel stats (rocprofv3 --kernel-trace --stats, whole process incl. setup)\n\n")
    f.write("| kernel | calls | total ms | median us | min us | avg us | % |\n|---|---|---|---|---|---|---|\n")
    med = medians(f"{src}/{tag}_prof/{tag}_kernel_trace.csv")
    for r in rows[:16]:
        md, mn = med.get(r["Name"], (float("nan"), float("nan")))
        f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {md:.1f} | {mn:.1f} | "
                f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    f.write(f"\nDominant kernel: `{KNAME}` (storage mode {storage}). Its shard-product launches alone "
            f"(kernel-trace rows of that kernel, long ones only): **{len(shard)} launches, "
            f"average {shard_avg_ms:.3f} ms** (rocprofv3) vs {bench['roofline']['avg_launch_ms']:.3f} ms "
            f"(HIP events inside bench.py, un-profiled run).\n\n")
    # matrix-core evidence: the Nystrom sketch product Y = K Omega (rk_symm_gemm: float16 split on the 16-bit cores)
    import re as _re
    mwl = _re.search(r"n=(\d+) d=\d+ k=\d+ nystrom_rank=(\d+)", bench["config"]["workload"])
    gem = [r for r in rows if "symgemm_" in r["Name"]]
    if mwl and gem:
        nn, rr = int(mwl.group(1)), int(mwl.group(2))
        flops = 2.0 * nn * nn * rr / bench["n_gpus"]  # float32-equivalent, per launch
        md = med.get(gem[0]["Name"], (float(gem[0]["AverageNs"]) / 1e3, 0.0))[0] * 1e-6  # s
        one_piece = "symgemm_ph_kernel<8, 1>" in gem[0]["Name"] or "symgemm_ph_kernel<4, 1>" in gem[0]["Name"]
        terms = 6 if "SgBf16" in gem[0]["Name"] else (2 if one_piece else 3)
        f.write("## Matrix cores: the Nystrom sketch product Y = K Omega from the packed strips (`rk_symm_gemm`)\n\n")
        f.write(f"`{gem[0]['Name'][:gem[0]['Name'].find('(')].replace('void rk::', '')}`"
                f"{' (Omega on the float16 grid: one piece, two products per pair)' if one_piece else ''}: {sum(int(r['Calls']) for r in gem)} launches (assembly + its warm repeat), median {md * 1e3:.1f} ms for "
                f"2 n^2 r = {flops:.3e} flop float32-equivalent -> **{flops / md / 1e12:.0f} TFLOP/s float32-equivalent = "
                f"{terms * flops / md / 1e15:.2f} PFLOP/s on the 16-bit matrix cores** ({terms} exact products per pair; "
                f"{100 * terms * flops / md / 2.5e15:.0f} % of their 2.5 PFLOP/s dense peak).\n\n")
    mf = f"{src}/{tag}_pmc_mfma/{tag}_counter_collection.csv"
    if os.path.exists(mf) and gem:
        acc = {}
        for r in csv.DictReader(open(mf)):
            if "symgemm_" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if acc.get("SQ_VALU_MFMA_BUSY_CYCLES") and acc.get("GRBM_GUI_ACTIVE"):
            busy = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"])
            gui = sum(acc["GRBM_GUI_ACTIVE"])
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs (#MFMA x 16 cycles for v_mfma_f32_16x16x32_f16);
            # GRBM_GUI_ACTIVE is reported once per XCD (8 of them) -> utilisation = busy / (gui / 8 x 1024 SIMDs)
            util = busy / (gui / 8.0 * 1024.0)
            f.write(f"PMC over the same launches (own pass): SQ_VALU_MFMA_BUSY_CYCLES = {busy:.4g} (= {busy / 16:.4g} MFMA x 16 "
                    f"cycles), GRBM_GUI_ACTIVE = {gui:.4g} summed over 8 XCDs -> **matrix cores busy {100 * util:.1f} %** of the "
                    f"1024 SIMDs' cycles.\n\n")
    # the "next" rows: ASkotch and LSQR benches
    for name, title in (("sap", "ASkotch / SAP (BASELINE config 5, tools/bench_sap.py)"),
                        ("lsqr", "LSQR + SkPre (BASELINE config 4, tools/bench_lsqr.py)")):
        blog = f"{src}/{tag}_bench_{name}.log"
        st = f"{src}/{tag}_prof_{name}/{tag}_{name}_kernel_stats.csv"
        if not (os.path.exists(blog) and os.path.exists(st)):
            continue
        shutil.copy(blog, f"{dst}/{tag}_bench_{name}.log")
        shutil.copy(st, f"{dst}/{tag}_{name}_kernel_stats.csv")
        line = [l for l in open(blog) if l.startswith("{")][-1].strip()
        f.write(f"## {title}\n\nun-profiled run: `{line[:700]}`\n\n")
        f.write("| kernel (rocprofv3 --kernel-trace --stats of the same script) | calls | total ms | median us | min us | avg us | % |\n|---|---|---|---|---|---|---|\n")
        med2 = medians(f"{src}/{tag}_prof_{name}/{tag}_{name}_kernel_trace.csv")
        for r in list(csv.DictReader(open(st)))[:8]:
            md, mn = med2.get(r["Name"], (float("nan"), float("nan")))
            f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {md:.1f} | {mn:.1f} | "
                    f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
        f.write("\n")
    # matrix-free kernel product (rk_kernel_matmat): formulations side by side + matrix-core counter
    klog = f"{src}/{tag}_bench_kmatmat.log"
    if os.path.exists(klog):
        shutil.copy(klog, f"{dst}/{tag}_bench_kmatmat.log")
        f.write("## Matrix-free kernel product `rk_kernel_matmat` (tools/bench_kmatmat.py)\n\n")
        f.write("| case n1 x n2 x d x k (g: gathered rows) | bf16x3 Gpair/s | fp32 MFMA Gpair/s | direct VALU Gpair/s | "
                "worst row error vs float64 (bf16x3 / fp32 MFMA / direct) |\n|---|---|---|---|---|\n")
        for l in open(klog):
            if not l.startswith("{"):
                continue
            r = json.loads(l)
            g = lambda m, key: r.get(m, {}).get(key, float("nan"))  # noqa: E731
            f.write(f"| {r['case']}{'' if r.get('kind', 'rbf') == 'rbf' else ' ' + r['kind']} | {g('bf16x3', 'Gpair_s'):.0f} | {g('mfma32', 'Gpair_s'):.0f} | "
                    f"{g('direct', 'Gpair_s'):.0f} | {g('bf16x3', 'rel_err_rows'):.1e} / "
                    f"{g('mfma32', 'rel_err_rows'):.1e} / {g('direct', 'rel_err_rows'):.1e} |\n")
        f.write("\n")
        kf = f"{src}/{tag}_pmc_kmatmat/{tag}_counter_collection.csv"
        if os.path.exists(kf):
            per = {}
            for r in csv.DictReader(open(kf)):
                if "kmat_matmat_bf16x3_kernel" in r["Kernel_Name"]:
                    key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])
                    per.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            f.write("PMC (own pass, per launch): matrix-core busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)\n\n")
            f.write("| kernel instantiation | grid | MFMA busy cycles | matrix-core busy |\n|---|---|---|---|\n")
            for (kn, grid), c in per.items():
                if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
                    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
                    gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
                    f.write(f"| `{kn}` | {grid} | {busy:.4g} | {100 * busy / (gui / 8.0 * 1024.0):.1f} % |\n")
            f.write("\n")
    for name, title in (("symm", "Packed symmetric product, 1 .. 16 right-hand sides (tools/bench_symm.py)"),
                        ("skinny", "Row-major shard product 25 000 x 200 000, 1 .. 16 right-hand sides (tools/bench_skinny.py)"),
                        ("matrix_free", "Storage-free PCG: RK_KERNEL_MODE=matrix_free bench.py --steps 20")):
        blog = f"{src}/{tag}_bench_{name}.log"
        if not os.path.exists(blog):
            continue
        shutil.copy(blog, f"{dst}/{tag}_bench_{name}.log")
        f.write(f"## {title}\n\n```\n")
        for l in open(blog):
            if l.startswith("{"):
                f.write(l[:420].rstrip() + "\n")
        f.write("```\n\n")
    for name, title in (("bench_c2", "BASELINE config 2: bench.py --n 50000 --d 32 --steps 200"),
                        ("bench_symgemm", "Sketch product Y = K Omega from the packed strips (tools/bench_symgemm.py)"),
                        ("nystrom_setup", "Nystrom assembly by stage (tools/profile_nystrom_setup.py)"),
                        ("symgemm_sq_counters", "SQ counters of the sketch kernel (tools/pmc_symgemm.sh)"),
                        ("sap_solve", "ASkotch end to end: tools/bench_sap.py --solve (300 iterations, callback_freq 100)"),
                        ("bench_gloo_world2", "Two gloo ranks on one GPU: the multi-rank path of bench.py (n = 100 000)"),
                        ("bench_gloo_world2_selflaunch", "`RK_BENCH_BACKEND=gloo RK_BENCH_ONE_GPU=1 python3 bench.py --gpus 2`: bench.py starts its own ranks"),
                        ("symm_pmc_table", "Packed symmetric product (block kernel + finishing kernel): HBM traffic and matrix-core busy per k (tools/pmc_symm.sh)"),
                        ("wr_probe", "What stores cost beside a streaming read (tools/wr_probe.hip)"),
                        ("bench_ksketch", "Sketch product of a matrix-free operator: entries evaluated inside the product (tools/bench_ksketch.py)"),
                        ("bench_f64", "Secondary line: bench.py --dtype f64 (float64 storage, 160 GB packed)"),
                        ("bench_f64_f32storage", "Secondary line: bench.py --dtype f64 --storage f32 (float64 iteration over the float32 triangle)"),
                        ("profile_colsample", "Nystrom assembly with the column-sampling sketch by stage (tools/profile_colsample.py)"),
                        ("profile_gauss", "Nystrom assembly with the Gaussian sketch by stage (tools/profile_colsample.py --sketch gauss)"),
                        ("host_probe", "Host of the box (tools/host_probe.py)")):
        ext = "txt" if name.endswith("counters") else ("md" if name.endswith("table") else "log")
        blog = f"{src}/{tag}_{name}.{ext}"
        if not os.path.exists(blog):
            continue
        shutil.copy(blog, f"{dst}/{tag}_{name}.{ext}")
        f.write(f"## {title}\n\n```\n")
        for l in open(blog):
            if l.startswith(("[W", "[Gloo", "/opt/amdgpu", "W1", "*****")) or not l.strip():
                continue
            f.write(l[:600].rstrip() + "\n")
        f.write("```\n\n")
    kf = f"{src}/{tag}_pmc_ksketch/{tag}_counter_collection.csv"
    if os.path.exists(kf):
        c = {}
        for r in csv.DictReader(open(kf)):
            if "kgemm_kernel" in r["Kernel_Name"]:
                c.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
            busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
            gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
            f.write(f"## Matrix cores of `kgemm_kernel` (rk_kernel_sketch_matmat, n = 200 000, r = 512; own PMC pass)\n\n"
                    f"SQ_VALU_MFMA_BUSY_CYCLES = {busy:.4g} per launch, GRBM_GUI_ACTIVE = {gui:.4g} (8 XCDs) -> **matrix cores busy "
                    f"{100 * busy / (gui / 8.0 * 1024.0):.1f} %** over {len(c['GRBM_GUI_ACTIVE'])} launch(es).\n\n")
    for name in ("smoke", "pytest_gpu"):
        blog = f"{src}/{tag}_{name}.log"
        if os.path.exists(blog):
            tail = [l.rstrip() for l in open(blog) if l.strip()][-2:]
            f.write(f"## {name}\n\n```\n" + "\n".join(t[:300] for t in tail) + "\n```\n\n")
    f.write("## HBM traffic of the shard product (separate --pmc passes)\n\n")
    f.write(f"FETCH_SIZE = {fetch:.0f} KB, WRITE_SIZE = {write:.0f} KB per launch -> "
            f"(2 x FETCH + WRITE) x 1024 = {traffic / 1e9:.2f} GB vs algorithmic {algo / 1e9:.2f} GB "
            f"(ratio {traffic / algo:.3f}).\n")
print(json.dumps(out))
