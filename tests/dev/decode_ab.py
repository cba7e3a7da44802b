"""dev (GPU box): A/B of the batch-1 decode forms on the Llama-3-8B chain -- tokens/s of the whole dependent chain under graph replay
and us per launch of each of the four op shapes (32 distinct-weight launches per graph), for form 0 (bit-faithful skinny kernel) and
form 1 (stream kernel) at several waves-per-block settings.  argv: [fp16|bf16] [layers]"""
import os
import sys

import torch

if os.environ.get("GPTQHIP_LIB_VARIANT"):
    os.environ["GPTQHIP_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ablate", os.environ["GPTQHIP_LIB_VARIANT"])
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402
from gptqmodel_amd import ops  # noqa: E402
from gptqmodel_amd.utils.decode_chain import DecodeStep  # noqa: E402

dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float16
n_layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
cfg = bench.LLAMA3_8B
layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype, n_layers=n_layers)
step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
step.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
stream = torch.cuda.Stream()
names = ["qkv", "o", "gate_up", "down"]


def per_op(j):
    sel = step.ops[j::4]
    def fn():
        for op in sel:
            ops.launch_decode_op(op, dev)
    ms, g = bench.time_graph(fn, stream, 50, 10)
    del g
    return ms * 1e3 / len(sel)


def chain():
    ms, g = bench.time_graph(step.run, stream, 100, 20)
    del g
    return ms


ref_out = None
for form, waves in [(int(f), 0) for f in (os.environ.get('AB_FORMS', '0,3,4,-1,0,3').split(','))]:
    ops.set_decode_form(form)
    ops.set_tuning(0, 0, waves)
    ms = chain()
    torch.cuda.synchronize()
    out = step.out.float().clone()
    if ref_out is None:
        ref_out = out
    err = float((out - ref_out).abs().max() / ref_out.abs().max())
    per = [per_op(j) for j in range(4)]
    print(f"form={form} waves={waves or 'auto'}: chain {ms * 1e3 / n_layers:.2f} us/layer = {1e3 / (ms * 32 / n_layers):.1f} tokens/s (32-layer equivalent) | "
          + " ".join(f"{n} {u:.2f}" for n, u in zip(names, per)) + f" | sum {sum(per):.2f} us | chain-out rel diff vs first {err:.2e}", flush=True)
