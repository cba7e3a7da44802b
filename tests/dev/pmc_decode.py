"""dev (run under rocprofv3 --pmc ...): a few EAGER decode steps of a 4-layer Llama-3-8B-shaped chain (distinct weights per layer),
so that every dispatch of the four decode-op shapes is counted.  argv[1]: fp16 | bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gptqmodel_amd.utils.decode_chain import DecodeStep  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
cfg = bench.LLAMA3_8B
layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype, n_layers=4)
step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
step.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
for _ in range(4):
    step.run()
torch.cuda.synchronize()
print("ok", float(step.out.float().abs().max()))
