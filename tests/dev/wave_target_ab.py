"""dev (GPU box): the preload form's wave target (GPTQHIP_SK1_WAVE_TARGET, read once per process) on the 8B and 70B decode chains: tokens/s under graph replay.
argv: model (8b | 70b)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gptqmodel_amd import ops  # noqa: E402
from gptqmodel_amd.utils.decode_chain import DecodeStep  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "8b"
cfg = bench.LLAMA3_8B if which == "8b" else bench.LLAMA3_70B
n_layers = 32 if which == "8b" else 20
dtype, dev = torch.float16, torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype, n_layers=n_layers)
step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
step.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
stream = torch.cuda.Stream()
ms, g = bench.time_graph(step.run, stream, 100, 20)
per = []
for j in range(4):
    sel = step.ops[j::4]
    def fn():
        for op in sel:
            ops.launch_decode_op(op, dev)
    m2, g2 = bench.time_graph(fn, stream, 40, 8)
    del g2
    per.append(m2 * 1e3 / len(sel))
print(f"{which} target={os.environ.get('GPTQHIP_SK1_WAVE_TARGET', '4096')}: chain {ms * 1e3 / n_layers:.2f} us/layer | " +
      " ".join(f"{n} {u:.2f}" for n, u in zip(["qkv", "o", "gate_up", "down"], per)), flush=True)
