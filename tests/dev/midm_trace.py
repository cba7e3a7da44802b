"""dev (round 5): kernel-level phase timeline of the mid-M launches the round-4 verdict set gates on -- main kernel, the gap to the
split-K reduce kernel, the reduce kernel, the gap to the next launch -- from a rocprofv3 --kernel-trace of back-to-back graph replays.

    python tests/dev/midm_trace.py run        # the workload (run it under rocprofv3 --kernel-trace --output-format csv -d DIR -o t)
    python tests/dev/midm_trace.py sum DIR    # summary of DIR/**/t_kernel_trace.csv
"""
import collections
import csv
import os
import sys

CASES = [(128, 4096, 4096), (72, 4096, 11008), (128, 4096, 11008), (136, 4096, 11008), (128, 11008, 4096), (192, 11008, 4096)]
REPS = 40

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from gptqmodel_amd import ops
    dev, gs = "cuda", 128
    for (M, K, N) in CASES:
        sets = []
        for _ in range(8):      # rotating weight copies: cold weights like a model's layers
            qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
            qz = torch.zeros((K // gs, N // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
            sets.append(ops.repack_tiled(qw, qz, sc, None, gs, 4))
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        print(f"CASE {M} {K} {N} {ops.plan_describe(M, K, N, gs)}", flush=True)
        torch.cuda.synchronize()
        marker = torch.zeros(M * 1000 + (K // 128), device=dev)      # a memset whose SIZE tags the case in the trace
        marker.zero_()
        for _ in range(REPS // 8 + 1):
            for qw_t, meta in sets:
                ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
        torch.cuda.synchronize()
else:
    rows = []
    for root, _, files in os.walk(sys.argv[2]):
        for fn in files:
            if fn.endswith("kernel_trace.csv"):
                rows += list(csv.DictReader(open(os.path.join(root, fn))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # a case = a run of tiled_kernel (+ splitk_reduce_kernel) launches between two non-gptqhip kernels
    runs, cur = [], []
    for r in rows:
        n = r["Kernel_Name"]
        if "tiled_kernel" in n or "splitk_reduce" in n:
            cur.append(r)
        elif cur:
            if len(cur) >= 20:
                runs.append(cur)
            cur = []
    if len(cur) >= 20:
        runs.append(cur)
    for case, run in zip(CASES, runs):
        main = [r for r in run if "tiled_kernel" in r["Kernel_Name"]]
        red = [r for r in run if "splitk_reduce" in r["Kernel_Name"]]
        d = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        seq = run[8:]       # skip the first (cold instruction cache, clock ramp)
        t = collections.defaultdict(list)
        for a, b in zip(seq, seq[1:]):
            gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
            t["gap main->reduce" if "tiled_kernel" in a["Kernel_Name"] and "splitk" in b["Kernel_Name"] else
              "gap reduce->main" if "splitk" in a["Kernel_Name"] else "gap main->main"].append(gap)
        for r in seq:
            t["main kernel" if "tiled_kernel" in r["Kernel_Name"] else "reduce kernel"].append(d(r))
        per = {k: sum(v) / len(v) for k, v in t.items()}
        total = per.get("main kernel", 0) + per.get("reduce kernel", 0) + per.get("gap main->reduce", 0) + per.get("gap reduce->main", per.get("gap main->main", 0))
        grid = main[0]["Grid_Size_X"], main[0].get("Grid_Size_Z", "?"), main[0]["Workgroup_Size_X"]
        print(f"M={case[0]} K={case[1]} N={case[2]}: launches {len(main)} main + {len(red)} reduce, grid threads x/z {grid[0]}/{grid[1]}: "
              + ", ".join(f"{k} {v:.2f} us" for k, v in sorted(per.items())) + f" | sum per call {total:.2f} us")
