"""Do L2 / Infinity-Cache contents survive a dependent kernel boundary?  Same layer launched back to back (hot) vs distinct
layers (cold), with the product's non-temporal weight loads and with a plain-load build (GPTQHIP_LIB=...plainloads.so)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import _lib
if os.environ.get("GPTQHIP_LIB"):
    _lib.LIB_PATH = os.environ["GPTQHIP_LIB"]
import torch
import bench as B
from gptqmodel_amd import ops
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
stream = torch.cuda.Stream()
NL = 16
for name, K, N in (("o", 4096, 4096), ("qkv", 4096, 6144), ("down", 14336, 4096), ("gate_up", 4096, 28672)):
    lins = [B.make_gptq(K, N, 128, dev, gen, torch.float16) for _ in range(NL)]
    x = (torch.randn(K, device=dev, generator=gen) * 0.5).half()
    outs = [torch.empty(N, dtype=torch.float16, device=dev) for _ in range(NL)]
    cold = [ops.make_decode_op(x, l.qweight, l.meta, None, o, K, N, 128, 4, l._scale_dtype) for l, o in zip(lins, outs)]
    hot = [cold[0]] * NL
    pair = [cold[i // 2 * 2] for i in range(NL)]   # every layer twice in a row: 2nd launch may hit what the 1st left behind
    res = []
    for tag, seq in (("cold", cold), ("hot(same layer)", hot), ("pairs(AABB..)", pair)):
        def run():
            for d in seq:
                ops.launch_decode_op(d, dev)
        ms, g = B.time_graph(run, stream, 30, 5)
        res.append(f"{tag} {ms * 1e3 / NL:6.2f} us")
        del g
    print(f"{name:8s} {B.algorithmic_bytes(1, K, N) / 1e6:6.2f} MB | " + " | ".join(res), flush=True)
    del lins, outs, cold, hot, pair
    torch.cuda.empty_cache()
