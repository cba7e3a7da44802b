#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout 900 -rf -p no:cacheprovider > gpurun_out/r3_pytest_final.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r3_pytest_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
bash tests/dev/collect_profiles_r03.sh > gpurun_out/r3_collect.log 2>&1
python - <<'PY'
import json
b = json.load(open("gpurun_out/r03_bench.json"))
print("bench", b["value"], b["roofline"]["frac"], b["roofline"]["traffic_source"][:40])
for c in b["configs"]:
    print(c.get("config"), c.get("workload", "")[:90], round(c.get("value", 0), 1), c.get("unit"), c.get("error", ""))
print("prefill", b["prefill"]["value"])
print("e2e", {k: b["e2e"].get(k) for k in ("eager_tokens_per_s", "graph_tokens_per_s", "prefill_tokens_per_s")})
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["c1_ms"], b["cpu_baseline"]["c4_awq_ms"])
PY
GPTQHIP_BENCH_SHARE_GPU=1 timeout 900 python3 bench.py --gpus 2 --steps 50 --warmup 5 > gpurun_out/r3_bench_share2.out 2> gpurun_out/r3_bench_share2.err; echo "share rc=$?"; grep -o '"n_gpus": [0-9]*, "ranks_seen": [0-9]*' gpurun_out/r3_bench_share2.out; grep -o '"config": "C5", "tp": [0-9]*' gpurun_out/r3_bench_share2.out
