"""dev (runs on the GPU box after tests/dev/pmc_passes.sh): per (workload, kernel family, grid) averages of every counter of every pass.
argv: tag, directory holding pmc_<tag>_<workload>_<i>/, output json."""
import collections
import csv
import glob
import json
import os
import re
import sys

tag, base, out = sys.argv[1], sys.argv[2], sys.argv[3]
res = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
dur = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(f"{base}/pmc_{tag}_*")):
    if not os.path.isdir(d):
        continue
    wl = re.match(rf"pmc_{tag}_(.*)_\d+$", os.path.basename(d)).group(1)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = re.search(r"gptqhip::(\w+)", k)
            if not m:
                continue
            fam = m.group(1)
            key = f"{fam} grid={r.get('Grid_Size', '?')} wg={r.get('Workgroup_Size', '?')}"
            res[wl][key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"gptqhip::(\w+)", r["Kernel_Name"])
            if not m:
                continue
            gx = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            wg = int(r["Workgroup_Size_X"])
            dur[wl][f"{m.group(1)} grid={gx} wg={wg}"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
summ = {}
for wl in res:
    summ[wl] = {}
    for key, cs in res[wl].items():
        e = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
        e["dispatch_rows"] = len(next(iter(cs.values())))
        dk = dur[wl].get(key)
        if dk:
            e["avg_us_under_pmc"] = sum(dk) / len(dk) / 1e3
        summ[wl][key] = e
json.dump(summ, open(out, "w"), indent=1)
for wl in summ:
    for key, e in summ[wl].items():
        print(wl, key, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()})
