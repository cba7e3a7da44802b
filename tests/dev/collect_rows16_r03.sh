#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 evidence for the small-batch kernels (decode ops at 16 rows; decode kernel family at 32 / 64 rows)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_rows16_stats -o rows -- python $R/tests/dev/rows_bench.py 16 > $O/r03_rows16_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r03_rows16_fetch -o rows -- python $R/tests/dev/rows_bench.py 16 > $O/r03_rows16_fetch.log 2>&1
MIDM_KERNELS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_midm_stats -o midm -- python $R/tests/dev/midm.py 32,64 > $O/r03_midm_stats.log 2>&1
python3 - <<'PY'
import csv, collections, os, glob
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for tag, name in (("r03_rows16_stats", "rows"), ("r03_midm_stats", "midm")):
    f = glob.glob(f"{O}/{tag}/**/*kernel_stats.csv", recursive=True)
    if f:
        rows = list(csv.reader(open(f[0])))
        with open(f"{O}/{tag}_kernels.csv", "w", newline="") as g:
            w = csv.writer(g); w.writerow(rows[0])
            for r in rows[1:]:
                if "gptqhip" in r[0]: w.writerow([r[0][:160]] + r[1:])
f = glob.glob(f"{O}/r03_rows16_fetch/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[0])):
        if "gptqhip" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            k = (r["Kernel_Name"][:110], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    with open(f"{O}/r03_rows16_fetch_summary.txt", "w") as g:
        g.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE -- rows_bench.py 16: average FETCH_SIZE (KB, raw; x2 = bytes on gfx950 for wide coalesced reads) per dispatch\n")
        for k, v in sorted(acc.items()):
            g.write(f"{k[0]} grid {k[1]}: {v[0] / v[1]:.1f} KB raw x2 = {2 * v[0] / v[1] / 1024:.2f} MB ({v[1]} dispatches)\n")
PY
find $O/r03_rows16_stats $O/r03_rows16_fetch $O/r03_midm_stats -type f -size +4M -delete 2>/dev/null
cat $O/r03_rows16_stats_kernels.csv | cut -c1-200; cat $O/r03_rows16_fetch_summary.txt | cut -c1-220; cat $O/r03_midm_stats_kernels.csv | cut -c1-200
