"""dev (GPU box): us per launch of the four Llama-3-8B decode ops (with their glue, 32 distinct-weight launches per graph) by forced waves
per block, for a decode form.  argv: form [fp16|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gptqmodel_amd import ops  # noqa: E402
from gptqmodel_amd.utils.decode_chain import DecodeStep  # noqa: E402

form = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dtype = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float16
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
cfg = bench.LLAMA3_8B
layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype, n_layers=32)
step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
stream = torch.cuda.Stream()
ops.set_decode_form(form)
names = ["qkv", "o", "gate_up", "down"]
for j in range(4):
    sel = step.ops[j::4]
    res = []
    for waves in (0, 4, 5, 6, 7, 8, 10, 12, 14, 16):
        ops.set_tuning(0, 0, waves)
        def fn():
            for op in sel:
                ops.launch_decode_op(op, dev)
        try:
            ms, g = bench.time_graph(fn, stream, 40, 8)
            del g
            res.append(f"W{waves or 'auto'} {ms * 1e3 / len(sel):.2f}")
        except RuntimeError as e:
            res.append(f"W{waves} --")
    K, N = sel[0].K, sel[0].N
    ops.set_tuning(0, 0, 0)
    print(f"form {form} {names[j]} K={K} N={N} plan[{ops.plan_describe(1, K, N, 128)}]: " + " | ".join(res), flush=True)
