"""dev (GPU box): how long the chip needs to reach its steady clock from idle -- the Llama-3-8B decode chain replayed from a HIP graph, us per token in
windows of 10 replays after the GPU idled for `idle_s`; then the same for a 4096^2 prefill GEMM at M = 8192."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gptqmodel_amd.utils.decode_chain import DecodeStep  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
dtype = torch.float16
cfg = bench.LLAMA3_8B
layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype, n_layers=32)
step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
step.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    step.run()
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        step.run()
    for idle_s in (2.0, 0.2):
        stream.synchronize()
        time.sleep(idle_s)
        n_win, per = 40, 10
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
        ev[0].record(stream)
        for w in range(n_win):
            for _ in range(per):
                g.replay()
            ev[w + 1].record(stream)
        stream.synchronize()
        us = [ev[i].elapsed_time(ev[i + 1]) / per * 1e3 for i in range(n_win)]
        print(f"decode chain after {idle_s} s idle, us per token in windows of {per} replays:", " ".join(f"{u:.0f}" for u in us), flush=True)
    lin = bench.make_gptq(4096, 4096, 128, dev, gen, dtype)
    x = (torch.randn((8192, 4096), device=dev, generator=gen) * 0.5).to(dtype)
    lin(x)
    stream.synchronize()
    time.sleep(2.0)
    n_win = 40
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_win + 1)]
    ev[0].record(stream)
    for w in range(n_win):
        lin(x)
        ev[w + 1].record(stream)
    stream.synchronize()
    us = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n_win)]
    print("4096^2 M=8192 prefill after 2 s idle, us per call:", " ".join(f"{u:.0f}" for u in us), flush=True)
