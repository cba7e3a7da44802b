"""dev (run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o t`): the prefill kernel at M = 64..1024 with forced tile heights and
split-K factors -- per configuration the main-kernel and reduce-kernel durations and the gap between them.  Run it twice: under
rocprofv3 it launches (mode "run", prints the launch order to stdout as CFG lines); `python tiled_midm_trace.py summarize DIR LOG`
joins the CFG lines with the kernel trace."""
import csv
import glob
import os
import re
import sys

ITERS = 6


def run():
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from gptqmodel_amd import ops
    dev = "cuda"
    for (K, N) in [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)]:
        sets = []
        for _ in range(4):
            qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
            qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
            sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
            sets.append(ops.repack_tiled(qw, qz, sc, None, 128, 4))
        for M in (128, 256, 512, 1024):
            x = (torch.randn(M, K, device=dev) * 0.5).half()
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            for variant in (0, 3, 2, 1):          # auto, 64-, 128-, 256-row tiles
                for split in ((0,) if variant == 0 else (1, 2, 4, 8)):
                    if variant == 1 and M < 256:
                        continue
                    ops.set_tuning(split, 2, variant)
                    try:
                        for it in range(ITERS):
                            qw_t, meta = sets[it % 4]
                            ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, out=out)
                        torch.cuda.synchronize()
                        print("CFG", M, K, N, variant, split, ops.plan_describe(M, K, N, 128).replace(" ", ","), flush=True)
                    except RuntimeError as e:
                        print("SKIP", M, K, N, variant, split, str(e)[:80], flush=True)
    ops.set_tuning(0, 0, 0)


def summarize(d, log):
    path = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if ("tiled_kernel<" in r["Kernel_Name"] and "repack" not in r["Kernel_Name"]) or "splitk_reduce" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # every gemm call = one tiled launch (two when the plan has a 128-row tail launch) + one reduce launch when split;
    # CFG lines come in launch order, ITERS calls each
    cfgs = [l.split() for l in open(log) if l.startswith("CFG ")]
    calls, it = [], iter(rows)
    r = next(it, None)
    for c in cfgs:
        n_main = 2 if "tail_cols=0" not in c[6] else 1
        for _ in range(ITERS):
            cur = {"main": 0.0, "reduce": 0.0, "gap": 0.0, "grid": None}
            for _ in range(n_main):
                assert r is not None and "tiled_kernel" in r["Kernel_Name"], (c, r and r["Kernel_Name"][:40])
                cur["main"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                cur["end"] = int(r["End_Timestamp"])
                cur["grid"] = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Z"]))
                r = next(it, None)
            if r is not None and "reduce" in r["Kernel_Name"]:
                cur["reduce"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                cur["gap"] = (int(r["Start_Timestamp"]) - cur["end"]) / 1e3
                r = next(it, None)
            calls.append(cur)
    assert r is None, "unconsumed kernel rows"
    print("# M K N variant(0 auto,3=64,2=128,1=256 rows) forced_split | plan | main us | gap us | reduce us | total us   (avg of the last 4 of 6 calls)")
    for i, c in enumerate(cfgs):
        cs = calls[i * ITERS + 2:(i + 1) * ITERS]
        avg = lambda k: sum(x[k] for x in cs) / len(cs)
        print(f"M={c[1]:>5} K={c[2]:>5} N={c[3]:>5} v={c[4]} s={c[5]} {c[6]:60s} grid={cs[0]['grid']} main {avg('main'):7.1f} gap {avg('gap'):5.1f} reduce {avg('reduce'):6.1f} "
              f"total {avg('main') + avg('gap') + avg('reduce'):7.1f}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        run()
