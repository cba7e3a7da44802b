"""dev (runs on the GPU box): average every counter of a rocprofv3 --pmc pass per kernel family into <dir>/pmc_summary.json
(the raw counter_collection.csv of a whole bench run is tens of MB)."""
import collections, csv, json, sys
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
with open(f"{d}/bench_counter_collection.csv") as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        fam = "skinny_kernel" if "skinny_kernel" in k else ("tiled_kernel" if "tiled_kernel" in k else None)
        if fam is None:
            continue
        a = acc[fam][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = {fam: {c: {"avg_per_dispatch_row": v[0] / v[1], "rows": v[1]} for c, v in cs.items()} for fam, cs in acc.items()}
json.dump(out, open(f"{d}/pmc_summary.json", "w"), indent=1)
print(json.dumps(out)[:400])
