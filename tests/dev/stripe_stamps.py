"""Dev tool: per-block phase breakdown of the stripe kernel (a -DGPTQHIP_STRIPE_STAMPS build, tests/dev/stripe_stamps.sh).
Stamps are s_memrealtime ticks (100 MHz); printed in microseconds relative to the earliest block start."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import _lib
_lib.LIB_PATH = os.environ.get("GPTQHIP_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", "libgptqhip_stamps.so"))
from gptqmodel_amd import ops
dev = "cuda"; gs = 128
NAMES = ["start", "dequeued", "first data", "main loop", "exchange", "published", "ticket", "summed", "stored", "2nd dequeue", "exit ticket"]
stamps = torch.zeros((256, 16), dtype=torch.int64, device=dev)
os.environ["GPTQHIP_STRIPE_STAMP_PTR"] = hex(stamps.data_ptr())
shapes = [tuple(int(v) for v in t.split("x")) for t in (sys.argv[1] if len(sys.argv) > 1 else "4096x4096,4096x11008,11008x4096").split(",")]
Ms = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128").split(",")]
wt = int(os.environ.get("STRIPE_WT", "0"))
for (K, N) in shapes:
    sets = []
    for _ in range(6):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
        sets.append(ops.repack_tiled(qw, qz, sc, None, gs, 4))
    for M in Ms:
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        ops.set_tuning(wt, 3, 0)
        for qw_t, meta in sets:   # the last launch (cold weights, warm code) is the one whose stamps survive
            stamps.zero_()
            ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = stamps.cpu().numpy().astype(np.float64)
        t0 = st[:, 0].min()
        us = np.where(st[:, :11] > 0, (st[:, :11] - t0) / 100.0, np.nan)
        xcc = st[:, 15].astype(int)
        print(f"== K={K} N={N} M={M} wt={wt}  {ops.plan_describe(M, K, N, gs)}  blocks per XCD: {np.bincount(xcc, minlength=8).tolist()}")
        print("   phase          median-end   max-end   median-duration  max-duration  (us; n blocks)")
        prev = None
        for i, name in enumerate(NAMES):
            col = us[:, i]
            ok = ~np.isnan(col)
            if not ok.any():
                continue
            dur = ""
            if i > 0:
                # duration against the latest earlier stamp this block has
                pr = np.full(256, np.nan)
                for j in range(i - 1, -1, -1):
                    pr = np.where(np.isnan(pr), us[:, j], pr)
                d = (col - pr)[ok]
                dur = f"{np.median(d):10.2f} {d.max():12.2f}"
            print(f"   {name:13s} {np.nanmedian(col):9.2f} {np.nanmax(col):9.2f}   {dur}   ({int(ok.sum())})")
        last = int(np.nanargmax(np.nanmax(us, axis=1)))
        print(f"   slowest block {last} (xcc {xcc[last]}): " + " ".join(f"{n}={us[last, i]:.2f}" for i, n in enumerate(NAMES) if not np.isnan(us[last, i])))
        ops.set_tuning(0, 0, 0)
