"""Turn gpurun_out/<tag>_* (written by collect_profiles.sh on the GPU box) into the committed profiles/<tag>_* summaries."""
import collections, csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out/", "profiles/"
os.makedirs(dst, exist_ok=True)
rows = list(csv.reader(open(f"{src}{tag}_stats/bench_kernel_stats.csv")))
with open(f"{dst}{tag}_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(rows[0])
    for r in rows[1:]:
        if "gptqhip" in r[0] or float(r[4]) >= 1.0: w.writerow([r[0][:160]] + r[1:])
kt = list(csv.DictReader(open(f"{src}{tag}_stats/bench_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in kt:
    if "skinny" in r["Kernel_Name"]:
        agg[(int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
per = [{"grid": list(k), "launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3} for k, v in sorted(agg.items())]
allv = [d for v in agg.values() for d in v]
def pmc(d, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f"{src}{tag}_{d}/bench_counter_collection.csv"))
            if "skinny" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals) if vals else None
bench = json.loads(open(f"{src}{tag}_bench.json").read())
fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
summ = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline ; separate --pmc passes with --no-graph (tests/dev/collect_profiles.sh)",
        "kernel": "gptqhip::skinny_kernel (decode)", "launches": len(allv), "avg_kernel_us_rocprof": sum(allv) / len(allv) / 1e3,
        "avg_launch_us_bench_events": bench["roofline"]["avg_launch_us"], "per_grid": per,
        "FETCH_SIZE_KB_per_launch_raw": fetch, "WRITE_SIZE_KB_per_launch_raw": write,
        "hbm_read_bytes_per_launch_corrected": 2 * 1024 * fetch,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE uncalibrated (negligible here)",
        "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
        "traffic_over_algorithmic": 2 * 1024 * fetch / bench["roofline"]["bytes_per_launch"],
        "sq_per_launch": {n: pmc("pmc_sq", n) for n in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU")}}
# prefill kernel counters
try:
    t = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{src}{tag}_pmc_tiled/p_counter_collection.csv")):
        if "tiled_kernel" in r["Kernel_Name"]: t[r["Counter_Name"]].append(float(r["Counter_Value"]))
    tk = {k: sum(v) / len(v) for k, v in t.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tk and "GRBM_GUI_ACTIVE" in tk:
        tk["mfma_busy_fraction"] = (tk["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (tk["GRBM_GUI_ACTIVE"] / 8)
    summ["tiled_kernel_M8192_4096x4096"] = tk
except Exception as e:
    summ["tiled_kernel_M8192_4096x4096"] = str(e)
json.dump(summ, open(f"{dst}{tag}_pmc_summary.json", "w"), indent=1)
for f in ("bench.json", "bench_70b_tp1.json", "bench_bf16.json", "bench_bf16_exact_optin.json", "torch_gpu_baseline.txt", "gemm_tflops.txt", "gemm_tflops_bf16.txt", "m_sweep.txt", "configs.txt", "stream_probe.txt"):
    if os.path.exists(f"{src}{tag}_{f}"): shutil.copy(f"{src}{tag}_{f}", f"{dst}{tag}_{f}")
print(json.dumps({k: summ[k] for k in ("avg_kernel_us_rocprof", "avg_launch_us_bench_events", "traffic_over_algorithmic")}, indent=1))
print(open(f"{dst}{tag}_bench.json").read()[:400])
