#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "64 4096 4096" "512 4096 4096"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 90 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/split_$tag -o p -- python $R/tests/dev/midm_split.py $cfg > /dev/null 2>&1
  f=$(find $R/gpurun_out/split_$tag -name "*kernel_trace.csv" | head -1)
  echo "== $cfg"
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "tiled_kernel" in n or "splitk_reduce" in n:
        key = ("main" if "tiled_kernel" in n else "reduce", r.get("Grid_Size_Z", r.get("Grid_Size", "")), r.get("Grid_Size_X",""), r.get("Grid_Size_Y",""))
        agg.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    v = sorted(v)[: max(1, len(v) - 2)]
    print(k, "n=%d" % len(v), "avg %.1f us" % (sum(v) / len(v) / 1e3))
PY
done
