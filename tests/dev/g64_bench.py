"""dev: group_size 32 / 64 decode launch time (regular pipeline vs GPTQHIP_NO_PAD... the generic path can be forced with an odd wave count)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from gptqmodel_amd import ops
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
dtype = torch.float16
stream = torch.cuda.Stream()
for gs in (128, 64, 32):
    for K, N in [(4096, 4096), (4096, 28672), (14336, 4096)]:
        NL = max(4, min(24, (700 << 20) // (K * N // 2)))
        lins = [B.make_gptq(K, N, gs, dev, gen, dtype) for _ in range(NL)]
        line = [f"g={gs:3d} K={K:5d} N={N:5d}"]
        for M in (1, 8):
            x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).to(dtype)
            def run():
                for l in lins:
                    ops.gemm(x, l.qweight, l.meta, None, None, N, gs, 4, l._scale_dtype)
            ms, g = B.time_graph(run, stream, 20, 3)
            line.append(f"M={M}: {ms*1e3/NL:6.2f} us")
            del g
        print(" | ".join(line), flush=True)
        del lins
        torch.cuda.empty_cache()
