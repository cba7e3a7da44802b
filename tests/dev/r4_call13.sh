#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "## persistent-layer probe (tests/dev/persist_probe.hip)"
timeout 300 python tests/dev/persist_probe.py
echo "## prefill kernel: shipped | no group-constant loads | no weight + constant loads (timing builds, wrong results), eager us per call"
for lib in "" nometa nobload; do
  if [ -z "$lib" ]; then timeout 300 python tests/dev/tiled_ablate.py; else GPTQHIP_LIB=$R/tests/dev/ablate/libgptqhip_$lib.so timeout 300 python tests/dev/tiled_ablate.py; fi
done
} > gpurun_out/r4c13_probe.txt 2>&1
cat gpurun_out/r4c13_probe.txt
cd /tmp
rocprofv3 -L > $R/gpurun_out/r4c13_counters.txt 2>&1
grep -o "Name:[ \t]*[A-Za-z0-9_]*" $R/gpurun_out/r4c13_counters.txt | sed 's/Name:[ \t]*//' | sort -u | grep -E "^(SQ_INSTS_VMEM|SQ_INST_CYCLES|SQ_WAIT|SQ_ACTIVE_INST|SQ_INSTS_SMEM|TA_|TCP_|TD_)" | tr '\n' ' ' | fold -w 200 > $R/gpurun_out/r4c13_counter_names.txt
head -c 3000 $R/gpurun_out/r4c13_counter_names.txt; echo
for lib in shipped nometa nobload; do
  for grp in "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
    tag=${lib}_$(echo $grp | cut -d' ' -f1)
    rm -rf /tmp/pmc_$tag
    if [ "$lib" = shipped ]; then L=""; else L=$R/tests/dev/ablate/libgptqhip_$lib.so; fi
    GPTQHIP_LIB=$L timeout 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_$tag -o p -- python $R/tests/dev/pmc_tiled.py 8192 4096 4096 > /tmp/pmc_$tag.log 2>&1
    python - "$lib" "$tag" <<'PY'
import csv, glob, sys, collections, json
lib, tag = sys.argv[1], sys.argv[2]
d = "/tmp/pmc_" + tag
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
out = {}
if f:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "tiled_kernel<" in r["Kernel_Name"] and "repack" not in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if kt:
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(kt[0])) if "tiled_kernel<" in r["Kernel_Name"] and "repack" not in r["Kernel_Name"]]
        if dur: out["avg_kernel_us"] = sum(dur) / len(dur)
else:
    out["error"] = open("/tmp/pmc_%s.log" % tag).read()[-300:]
print(json.dumps({lib: out}))
PY
  done
done > $R/gpurun_out/r4c13_pmc.jsonl 2>&1
cat $R/gpurun_out/r4c13_pmc.jsonl
