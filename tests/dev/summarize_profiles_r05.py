"""Turn gpurun_out/r04_* (written by collect_profiles_r05.sh on the GPU box) into the committed profiles/r04_* summaries."""
import collections, csv, json, os, re, shutil, sys
tag = "r05"
src, dst = "gpurun_out/", "profiles/"
os.makedirs(dst, exist_ok=True)
rows = list(csv.reader(open(f"{src}{tag}_stats/bench_kernel_stats.csv")))
with open(f"{dst}{tag}_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(rows[0])
    for r in rows[1:]:
        if "gptqhip" in r[0] or float(r[4]) >= 1.0: w.writerow([r[0][:200]] + r[1:])
# the bench line printed by the SAME profiled run
m = re.search(r'^\{"metric".*$', open(f"{src}{tag}_stats_bench.log").read(), re.M)
bench_prof = json.loads(m.group(0))
json.dump(bench_prof, open(f"{dst}{tag}_bench_under_rocprof.json", "w"))
kt = list(csv.DictReader(open(f"{src}{tag}_stats/bench_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in kt:
    if "skinny_kernel" in r["Kernel_Name"]:
        glue = re.search(r"skinny_kernel<([^>]*)>", r["Kernel_Name"]).group(1).split(",")[-2].strip()
        agg[(int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]), glue)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = {(6144 // 16, "1"): "qkv (RMSNorm in)", (4096 // 16, "0"): "o / down (residual + stats out)", (28672 // 16, "1"): "gate_up (RMSNorm in, SiLU*mul out)"}
per = []
for (gx, wg, glue), v in sorted(agg.items()):
    per.append({"grid_threads": gx, "workgroup": wg, "blocks": gx // wg, "glue_template": glue, "launches": len(v),
                "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3})
allv = [d for v in agg.values() for d in v]
def pmc(d, name):
    p = f"{src}{tag}_{d}/bench_counter_collection.csv"
    if not os.path.exists(p):
        ps = f"{src}{tag}_{d}/pmc_summary.json"      # aggregated on the GPU box (tests/dev/pmc_agg.py) when the CSV was too big
        if os.path.exists(ps):
            v = json.load(open(ps)).get("skinny_kernel", {}).get(name)
            return None if v is None else v["avg_per_dispatch_row"]
        return None
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(p)) if "skinny_kernel" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(vals) / len(vals) if vals else None
fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
summ = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-configs "
                   "(kernel table and bench line from the SAME run); separate --pmc passes with --no-graph (tests/dev/collect_profiles_r05.sh)",
        "kernel": "gptqhip::skinny_kernel<...,GLUE> (batch-1 decode op)", "launches": len(allv),
        "avg_kernel_us_rocprof": sum(allv) / len(allv) / 1e3,
        "bench_line_same_run": {k: bench_prof[k] for k in ("value", "ms_per_step")} | {"avg_launch_us": bench_prof["roofline"]["avg_launch_us"],
                                                                                       "frac": bench_prof["roofline"]["frac"]},
        "per_grid": per,
        "FETCH_SIZE_KB_per_launch_raw": fetch, "WRITE_SIZE_KB_per_launch_raw": write,
        "hbm_read_bytes_per_launch_corrected": None if fetch is None else 2 * 1024 * fetch,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE uncalibrated (negligible here)",
        "algorithmic_bytes_per_launch": bench_prof["roofline"]["bytes_per_launch"],
        "traffic_over_algorithmic": None if fetch is None else 2 * 1024 * fetch / bench_prof["roofline"]["bytes_per_launch"],
        "sq_per_launch": {n: pmc("pmc_sq", n) for n in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU")}
                         | {n: pmc("pmc_sq2", n) for n in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES")}}
sq = summ["sq_per_launch"]
if sq.get("SQ_INSTS_VALU") and summ["avg_kernel_us_rocprof"]:
    # 1024 SIMDs, 4 cycles per wave64 VALU instruction, 2.4 GHz: share of the average launch the VALU pipes are issuing
    summ["derived"] = {"valu_instr_per_simd": sq["SQ_INSTS_VALU"] / 1024,
                       "valu_issue_us_per_simd_at_2p4GHz": sq["SQ_INSTS_VALU"] / 1024 * 4 / 2400,
                       "valu_share_of_avg_launch": sq["SQ_INSTS_VALU"] / 1024 * 4 / 2400 / summ["avg_kernel_us_rocprof"],
                       "note": "averaged over the four launch shapes of a layer; on gate_up alone the loop issues 66 VALU per 1 KiB chunk = 6.3 us of its 14.7"}
json.dump(summ, open(f"{dst}{tag}_pmc_summary.json", "w"), indent=1)
for f in ("bench_line.json", "bench_detail.json", "bench_line_200steps.json", "bench.json", "bench_bf16.json", "bench_modules.json", "decode_ops.txt", "eager_overhead.txt", "e2e_llama8b.txt", "configs.txt", "torch_gpu_baseline.txt",
          "gemm_tflops.txt", "gemm_tflops_bf16.txt", "gemm_tflops_bf16_bf16scales.txt", "e2e_llama8b_actorder.txt", "e2e_llama8b_actorder_hfprefill.txt",
          "e2e_llama8b_hfprefill.txt", "mid_m_sweep.txt"):
    if os.path.exists(f"{src}{tag}_{f}"): shutil.copy(f"{src}{tag}_{f}", f"{dst}{tag}_{f}")
print(json.dumps({k: summ[k] for k in ("avg_kernel_us_rocprof", "bench_line_same_run", "traffic_over_algorithmic")}, indent=1))
for p in per: print(p)
