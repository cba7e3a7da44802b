"""Batch decode (M = 2..32): planner vs forced waves per block, rotating cold weight copies, graph replay."""
import sys, os, itertools
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops
from skinny_sweep import make, time_cfg
dev = "cuda"
shapes = [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)]
for k, n in shapes:
    copies = max(4, min(24, (400 << 20) // (k * n // 2)))
    sets = [make(k, n, 128, dev) for _ in range(copies)]
    for M in (2, 4, 8, 16, 32):
        x = torch.randn(M, k, device=dev, dtype=torch.float16)
        res = []
        for W in (0, 4, 7, 8, 14, 16):
            ops.set_tuning(0, 1, W)
            try:
                res.append((time_cfg(x, sets, n, 128, reps=3), W))
            except RuntimeError:
                pass
        ops.set_tuning(0, 0, 0)
        auto = [r for r in res if r[1] == 0][0][0]
        print(f"K={k} N={n} M={M:2d}: auto {auto:6.2f} us | " + " ".join(f"W{w}:{us:6.2f}" for us, w in res if w), flush=True)
    del sets
    torch.cuda.empty_cache()
