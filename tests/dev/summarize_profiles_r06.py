"""Turn gpurun_out/r06_* (written by tests/dev/collect_profiles_r06.sh on the GPU box) into the committed profiles/r06_* summaries."""
import collections, csv, json, os, re, shutil

tag, src, dst = "r06", "gpurun_out/", "profiles/"
rows = list(csv.reader(open(f"{src}{tag}_stats/bench_kernel_stats.csv")))
with open(f"{dst}{tag}_bench_kernel_stats.csv", "w", newline="") as f:
    w = csv.writer(f); w.writerow(rows[0])
    for r in rows[1:]:
        if "gptqhip" in r[0] or float(r[4]) >= 1.0: w.writerow([r[0][:200]] + r[1:])
m = re.search(r'^\{"metric".*$', open(f"{src}{tag}_stats_bench.log").read(), re.M)
bench_prof = json.loads(m.group(0))
json.dump(bench_prof, open(f"{dst}{tag}_bench_under_rocprof.json", "w"))
kt = list(csv.DictReader(open(f"{src}{tag}_stats/bench_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in kt:
    mm = re.search(r"gptqhip::(skinny1p?_kernel|skinny_kernel|decode_stream_kernel)", r["Kernel_Name"])
    if mm:
        agg[(mm.group(1), int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
names = {196608: "qkv (RMSNorm in)", 131072: "o / down", 458752: "gate_up"}
per = [{"kernel": k, "grid_threads": gx, "workgroup": wg, "blocks": gx // wg, "launches": len(v), "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3,
        "max_us": max(v) / 1e3} for (k, gx, wg), v in sorted(agg.items())]
allv = [d for v in agg.values() for d in v]
pmc = json.load(open(f"{src}{tag}_pmc.json"))
dec = {k: v for k, v in pmc.get("decode", {}).items() if re.match(r"(skinny1p?_kernel|skinny_kernel|decode_stream_kernel)", k)}
til = {wl: {k: v for k, v in pmc.get(wl, {}).items() if k.startswith("tiled_kernel")} for wl in ("tiled8192", "tiled128")}


def derive(e, chunks=None):
    d = {}
    if "SQ_INSTS_VALU" in e and "SQ_ACTIVE_INST_VALU" in e:
        d["valu_busy_quad_cycles_per_simd"] = e["SQ_ACTIVE_INST_VALU"] / 1024
    if "SQ_WAVE_CYCLES" in e:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if n in e: d[n.lower() + "_share_of_wave_cycles"] = e[n] / e["SQ_WAVE_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
        # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs here (one instance each)
        d["mfma_busy_fraction"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (e["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_ACTIVE_INST_VALU" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
        d["valu_busy_fraction"] = (4 * e["SQ_ACTIVE_INST_VALU"] / 1024) / (e["GRBM_GUI_ACTIVE"] / 8)
    if chunks:
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_INSTS_MFMA"):
            if n in e: d[n.lower() + "_per_1KiB_chunk"] = e[n] / chunks
    if "FETCH_SIZE" in e:
        d["hbm_read_bytes_corrected"] = 2 * 1024 * e["FETCH_SIZE"]
    return d


chunks_of = {"grid=98304": 384 * 32, "grid=196608": 384 * 32, "grid=131072 wg=512": 256 * 32, "grid=131072": 256 * 32, "grid=229376": 256 * 112, "grid=458752": 1792 * 32, "grid=262144": 256 * 32}
for k, e in dec.items():
    ch = next((v for kk, v in chunks_of.items() if kk in k), None)
    if "wg=896" in k or ("skinny1" in k and "grid=229376" in k): ch = 256 * 112
    if k.startswith("skinny1p_kernel"): ch = 1792 * 32          # the persistent form walks the fused gate_up's 1792 tiles with 256 blocks
    e["derived"] = derive(e, ch)
for wl in til:
    for k, e in til[wl].items():
        e["derived"] = derive(e)
fetch = [e["FETCH_SIZE"] * e.get("dispatch_rows", 1) for e in dec.values() if "FETCH_SIZE" in e]
rows_n = [e.get("dispatch_rows", 1) for e in dec.values() if "FETCH_SIZE" in e]
write = [e["WRITE_SIZE"] * e.get("dispatch_rows", 1) for e in dec.values() if "WRITE_SIZE" in e]
f_avg = sum(fetch) / sum(rows_n) if rows_n else None
w_avg = sum(write) / sum(rows_n) if rows_n else None
summ = {"command": "tests/dev/collect_profiles_r06.sh: rocprofv3 --kernel-trace --stats -- python bench.py --steps 200 --warmup 20 (kernel table and bench line from the SAME run); "
                   "tests/dev/pmc_passes.sh: one counter group per rocprofv3 --kernel-trace --pmc pass over tests/dev/pmc_decode.py (4-layer eager chain) and tests/dev/pmc_tiled.py",
        "decode_launches_in_stats_run": len(allv), "avg_decode_kernel_us_rocprof": sum(allv) / len(allv) / 1e3,
        "bench_line_same_run": {k: bench_prof[k] for k in ("value", "ms_per_step")} | {"avg_launch_us": bench_prof["roofline"]["avg_launch_us"], "frac": bench_prof["roofline"]["frac"]},
        "per_grid": per,
        "FETCH_SIZE_KB_per_launch_raw": f_avg, "WRITE_SIZE_KB_per_launch_raw": w_avg,
        "hbm_read_bytes_per_launch_corrected": None if f_avg is None else 2 * 1024 * f_avg,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read -> doubled; WRITE_SIZE uncalibrated (negligible here)",
        "algorithmic_bytes_per_launch": bench_prof["roofline"]["bytes_per_launch"],
        "traffic_over_algorithmic": None if f_avg is None else 2 * 1024 * f_avg / bench_prof["roofline"]["bytes_per_launch"],
        "decode_counters_per_launch": dec, "prefill_counters_per_launch": til,
        "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; "
                 "GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md per-instruction constants / PMC slots)"}
json.dump(summ, open(f"{dst}{tag}_pmc_summary.json", "w"), indent=1)
for f in ("bench_line.json", "bench_detail.json", "bench_line_200steps.json", "bench_bf16.json", "bench_bitfaithful.json", "e2e_llama8b.txt"):
    if os.path.exists(f"{src}{tag}_{f}"): shutil.copy(f"{src}{tag}_{f}", f"{dst}{tag}_{f}")
print(json.dumps({k: summ[k] for k in ("avg_decode_kernel_us_rocprof", "bench_line_same_run", "traffic_over_algorithmic")}, indent=1))
for p in per: print(p)
for k, e in dec.items(): print(k, e.get("derived"))
for wl in til:
    for k, e in til[wl].items(): print(wl, k, e.get("derived"))
