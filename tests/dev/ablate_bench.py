"""dev: per-shape time of the plain decode op with an alternate (ablated) library.  usage: ablate_bench.py [libpath] [tag]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import gptqmodel_amd._lib as L
if len(sys.argv) > 1 and sys.argv[1] != "-":
    L.LIB_PATH = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else "product"
import bench as B
from gptqmodel_amd import ops
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
dtype = torch.float16
NL = 24
stream = torch.cuda.Stream()
line = [f"{tag:10s}"]
for name, K, N in [("o", 4096, 4096), ("qkv", 4096, 6144), ("down", 14336, 4096), ("gate_up", 4096, 28672)]:
    lins = [B.make_gptq(K, N, 128, dev, gen, dtype) for _ in range(NL)]
    xin = (torch.randn(K, device=dev, generator=gen) * 0.5).to(dtype)
    outs = [torch.empty(N, dtype=dtype, device=dev) for _ in range(NL)]
    plain = [ops.make_decode_op(xin, l.qweight, l.meta, None, o, K, N, 128, 4, l._scale_dtype) for l, o in zip(lins, outs)]
    def run():
        for d in plain:
            ops.launch_decode_op(d, dev)
    ms, g = B.time_graph(run, stream, 30, 5)
    line.append(f"{name} {ms*1e3/NL:6.2f}")
    del g, lins, plain, outs
    torch.cuda.empty_cache()
print(" | ".join(line), flush=True)
