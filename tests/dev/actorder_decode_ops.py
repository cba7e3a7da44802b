"""dev (GPU box): us per launch of the four decode ops of a desc_act=True Llama-3-8B layer (in-kernel permutation) next to the plain layer's, graph replay."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gptqmodel_amd import ops  # noqa: E402
from gptqmodel_amd.utils.decode_chain import DecodeStep  # noqa: E402

dtype, dev = torch.float16, torch.device("cuda:0")
cfg = bench.LLAMA3_8B
stream = torch.cuda.Stream()
for name, desc in (("plain", False), ("act-order", True)):
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype, desc_act=desc), dev, gen, dtype, n_layers=16)
    step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
    per = []
    for j in range(4):
        sel = step.ops[j::4]
        def fn():
            for op in sel:
                ops.launch_decode_op(op, dev)
        ms, g = bench.time_graph(fn, stream, 40, 8)
        del g
        per.append(ms * 1e3 / len(sel))
    print(f"{name}: " + " ".join(f"{n} {u:.2f}" for n, u in zip(["qkv", "o", "gate_up", "down"], per)) + f" | sum {sum(per):.2f} us", flush=True)
    del step, layers
