#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "8192 4096 4096" "65536 4096 4096" "128 4096 28672" "2048 4096 4096"; do
  tag=$(echo $cfg | tr ' ' 'x')
  rm -rf /tmp/pmc_$tag
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tests/dev/pmc_tiled.py $cfg > /tmp/pmc_$tag.log 2>&1
  rm -rf /tmp/pmc2_$tag
  timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d /tmp/pmc2_$tag -o p -- python $R/tests/dev/pmc_tiled.py $cfg > /tmp/pmc2_$tag.log 2>&1
  python - "$tag" <<'PY'
import csv, glob, sys, collections, json
tag = sys.argv[1]
out = {}
for d in ("/tmp/pmc_" + tag, "/tmp/pmc2_" + tag):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(tag, d, "no counter file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "tiled_kernel<" in r["Kernel_Name"] and "repack" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "tiled_kernel<" in r["Kernel_Name"] and "repack" not in r["Kernel_Name"]]
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
    out["avg_kernel_us_" + d.split("/")[-1].split("_")[0]] = sum(dur) / len(dur)
print(json.dumps({tag: out}))
PY
done > $R/gpurun_out/pmc_tiled_r03.jsonl 2>&1
cat $R/gpurun_out/pmc_tiled_r03.jsonl
