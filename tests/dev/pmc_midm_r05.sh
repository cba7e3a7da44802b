#!/bin/bash
# round 5: PMC passes of the prefill kernel at M = 128 on the reference benchmark's 4096x11008 layer: round-4 plan (128 rows x 256 columns, 4 K-slices)
# vs round-5 plan (64 rows x 128 columns, no split-K).  Counters in their own passes (no trace domains besides --kernel-trace).
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "r4plan 2 4" "r5plan 0 0"; do
  set -- $cfg
  for ctr in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    timeout 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$1_$tag -o t -- python $R/tests/dev/pmc_tiled.py 128 4096 11008 $2 $3 > /tmp/pmc_$1_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections, os
for plan in ("r4plan", "r5plan"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in glob.glob(f"/tmp/pmc_{plan}_*"):
        if not os.path.isdir(d): continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = "tiled_kernel" if "tiled_kernel" in r["Kernel_Name"] else "splitk_reduce" if "splitk_reduce" in r["Kernel_Name"] else None
                if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"/tmp/pmc_{plan}_*.log"):
        for ln in open(f):
            if ln.startswith("PLAN"): plan_txt = ln.strip()
    print(plan, plan_txt)
    for k in agg:
        print("  ", k, {c: round(sum(v) / len(v), 1) for c, v in sorted(agg[k].items())}, "dispatch rows per counter:", len(next(iter(agg[k].values()))))
PY
