#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -k "tp_chain" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -30 > gpurun_out/r4c12_pytest.txt
cat gpurun_out/r4c12_pytest.txt
( time timeout 900 python bench.py > gpurun_out/r4c12_bench.json 2> gpurun_out/r4c12_bench.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4c12_bench.json"))
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["frac"])
cb = d.get("cpu_baseline", {})
print("cpu_baseline", {k: cb.get(k) for k in ("kind", "value", "cores", "c1_ms", "c4_awq_ms", "leg_s", "c1_compiled_note")})
for c in d.get("configs", []):
    if c.get("config") == "T1":
        if "error" in c: print(c); continue
        for x in c["cases"]:
            print(x["case_id"], x["kernel"], "eager %.1f TF (%.1f us)" % (x["tflops"], x["mean_ms"] * 1e3), "graph %.1f TF (%.1f us)" % (x["tflops_graph"], x["us_graph"]), "mfma %.3f hbm %.3f" % (x["roofline"]["mfma_frac"], x["roofline"]["hbm_frac"]))
    else:
        print(c.get("config"), str(c.get("workload"))[:70], c.get("value"), c.get("unit"), c.get("error"))
PY
tail -3 gpurun_out/r4c12_bench.err
