"""dev: M=1 / M=8 launch time on layer shapes whose chunk count has awkward factors (Llama-2-7B, Qwen2-7B, ...)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from gptqmodel_amd import ops
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
dtype = torch.float16
stream = torch.cuda.Stream()
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for name, K, N in [("llama2-7b down", 11008, 4096), ("llama2-7b gate_up", 4096, 22016), ("qwen2-7b down", 18944, 3584), ("qwen2-7b qkv", 3584, 4608),
                   ("qwen2-7b gate_up", 3584, 37888), ("llama3-8b down", 14336, 4096)]:
    NL = max(4, min(24, (700 << 20) // (K * N // 2)))
    lins = [B.make_gptq(K, N, 128, dev, gen, dtype) for _ in range(NL)]
    line = [f"{tag:6s} {name:18s} K={K:5d} N={N:5d} supported={ops.decode_supported(K, N, 128)}"]
    for M in (1, 8):
        x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).to(dtype)
        def run():
            for l in lins:
                ops.gemm(x, l.qweight, l.meta, None, None, N, 128, 4, l._scale_dtype)
        ms, g = B.time_graph(run, stream, 20, 3)
        us = ms * 1e3 / NL
        line.append(f"M={M}: {us:6.2f} us {B.algorithmic_bytes(M, K, N)/us/1e6:5.2f} TB/s")
        del g
    print(" | ".join(line), flush=True)
    del lins
    torch.cuda.empty_cache()
