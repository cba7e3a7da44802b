"""dev (GPU box): where the act-order prefill penalty on o_proj-shaped layers goes -- 4096x4096 at M = 65536, the same layer with and without
desc_act, event-timed per call; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gptqmodel_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(3)
dtype = torch.float16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = (torch.randn((M, 4096), device=dev, generator=gen) * 0.5).to(dtype)
act = bench.make_gptq(4096, 4096, 128, dev, gen, dtype, desc_act=True)
plain = bench.make_gptq(4096, 4096, 128, dev, gen, dtype, desc_act=False)


def t(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


fl = 2.0 * M * 4096 * 4096
for rnd in range(2):
    ta = t(lambda: act(x))
    tp = t(lambda: plain(x))
    xg = ops.gather_cols(x, act.perm)
    tg = t(lambda: ops.gather_cols(x, act.perm))
    tpg = t(lambda: plain(xg))
    tpre = t(lambda: act.forward_pregathered(xg)) if hasattr(act, "forward_pregathered") else float("nan")
    print(f"M={M}: act-order {ta:8.1f} us ({fl / ta / 1e6:6.0f} TF) | plain {tp:8.1f} us ({fl / tp / 1e6:6.0f} TF) | gather alone {tg:7.1f} us | "
          f"plain on the gathered buffer {tpg:8.1f} us | act-order layer, pre-gathered input {tpre:8.1f} us", flush=True)
