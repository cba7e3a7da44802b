"""dev (GPU box): per-wave phase clocks of decode_stream_kernel (tests/dev/stream_stamps_build.sh builds the instrumented library).
Prints, per op shape, medians over all waves of: start -> DMAs issued -> small operands landed -> prologue barrier -> end, and the time a
wave spends per main-loop phase (vm wait | LDS reads | DMA issue | compute | reduce barrier | epilogue)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
os.environ["GPTQHIP_LIB"] = os.path.join(HERE, "ablate", "libgptqhip_stamps.so")
stamps = torch.zeros(4096 * 16 * 16, dtype=torch.int64, device="cuda:0")
os.environ["GPTQHIP_STREAM_STAMPS_PTR"] = hex(stamps.data_ptr())
import bench  # noqa: E402
from gptqmodel_amd import ops  # noqa: E402
from gptqmodel_amd.utils.decode_chain import DecodeStep  # noqa: E402

dtype = torch.float16
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
cfg = bench.LLAMA3_8B
layers = bench.build_stack(cfg, lambda k, n: bench.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype, n_layers=4)
step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
step.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
if len(sys.argv) > 1:
    ops.set_tuning(0, 0, int(sys.argv[1]))
for _ in range(3):
    step.run()
torch.cuda.synchronize()
names = ["qkv", "o", "gate_up", "down"]
for j in range(4):
    # a warm-up of the other ops in front, then the op under test alone on a cold weight set (layer 3's)
    for op in step.ops[:8]:
        ops.launch_decode_op(op, dev)
    torch.cuda.synchronize()
    stamps.zero_()
    ops.launch_decode_op(step.ops[12 + j], dev)
    torch.cuda.synchronize()
    s = stamps.view(-1, 16, 16).cpu()
    live = s[:, :, 0] > 0
    nb = int(live.any(dim=1).sum())
    s = s[:nb].double()
    lv = live[:nb]
    # clocks may differ per XCD (block b runs on XCD b % 8): everything relative to the earliest wave start of the block's XCD
    rel = torch.zeros_like(s)
    for x in range(8):
        sel = s[x::8]
        t0 = sel[:, :, 0][lv[x::8]].min()
        rel[x::8] = sel - t0
    t = rel[lv]
    q = lambda v: "%6.0f %6.0f %6.0f" % (float(v.quantile(0.05)), float(v.median()), float(v.quantile(0.95)))
    print(f"{names[j]}: {nb} blocks, {t.shape[0]} waves; ticks since the XCD's first wave start (p5 median p95):")
    for i, nm in enumerate(["wave start", "DMAs issued", "small operands landed", "prologue barrier passed", "loop + epilogue done"]):
        print(f"    {nm:26s} {q(t[:, i])}")
