"""dev: the prefill planner's choice vs forced 64- / 128- / 256-row tiles (split-K left to the planner) over a sweep of M, eager
back-to-back launches on one box; flags every point where a forced tile height beats the planner by more than 4 %."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops  # noqa: E402

dev = "cuda"
MS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [160, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536, 2048, 3072, 4096]


def t_us(f, it):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for (K, N) in [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (8192, 10240), (8192, 57344)]:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, 128, 4)
    for M in MS:
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        o = torch.empty((M, N), dtype=torch.float16, device=dev)
        f = lambda: ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, out=o)
        res = {}
        for v in (0, 3, 2, 1):
            if (v == 3 and M > 512) or (v == 1 and M < 256):
                continue
            ops.set_tuning(0, 2, v)
            res[v] = t_us(f, 10 if M * N <= (1 << 26) else 4)
        ops.set_tuning(0, 2, 0)
        plan = ops.plan_describe(M, K, N, 128).replace("tiled ", "").replace(" gather=0", "")
        best = min(res, key=res.get)
        flag = "  <-- planner loses %.0f %%" % (100 * (res[0] / res[best] - 1)) if res[0] > 1.04 * res[best] else ""
        print(f"K={K:5d} N={N:5d} M={M:5d}: auto {res[0]:8.1f} ({plan}) | " + " | ".join(f"bm{ {3: 64, 2: 128, 1: 256}[v] } {res[v]:8.1f}" for v in (3, 2, 1) if v in res) + flag, flush=True)
ops.set_tuning(0, 0, 0)
