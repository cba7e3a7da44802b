"""dev: decode op with full glue at M = 1..16 rows (argv[1]: comma-separated row counts), Llama-3-8B layer shapes (us per launch, 24 distinct-weight layers per graph)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench as B
from gptqmodel_amd import ops
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
dtype = torch.float16
NL = 24
stream = torch.cuda.Stream()
SHAPES = [("qkv", 4096, 6144, True, False, False), ("o", 4096, 4096, False, True, False),
          ("gate_up", 4096, 28672, True, False, True), ("down", 14336, 4096, False, True, False)]
for M in ([int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (1, 2, 4, 6, 8)):
    tot = 0.0
    line = [f"M={M}"]
    for name, K, N, rms, res, paired in SHAPES:
        lins = [B.make_gptq(K, N, 128, dev, gen, dtype) for _ in range(NL)]
        x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).to(dtype)
        nw = torch.ones(K, dtype=dtype, device=dev)
        resid = torch.zeros((M, N), dtype=dtype, device=dev)
        outs = [torch.empty((M, N // 2 if paired else N), dtype=dtype, device=dev) for _ in range(NL)]
        st_in = torch.ones((M, K // 16), dtype=torch.float32, device=dev)
        st_out = torch.zeros((M, N // 16), dtype=torch.float32, device=dev)
        dops = [ops.make_decode_op(x, l.qweight, l.meta, None, o, K, N, 128, 4, l._scale_dtype, in_glue=ops.GLUE_RMSNORM if rms else ops.GLUE_NONE,
                                   norm_weight=nw if rms else None, residual=resid if res else None, stats_in=st_in if rms else None,
                                   stats_out=st_out if res else None, out_glue=ops.OUT_SILU_MUL_PAIRED if paired else ops.OUT_NONE, M=M)
                for l, o in zip(lins, outs)]
        def run():
            for d in dops:
                ops.launch_decode_op(d, dev)
        ms, g = B.time_graph(run, stream, 30, 5)
        us = ms * 1e3 / NL
        tot += us
        line.append(f"{name} {us:6.2f}")
        del g, lins, dops, outs
        torch.cuda.empty_cache()
    line.append(f"layer {tot:6.2f} us -> {M * 1e6 / (32 * tot):7.1f} tokens/s (linear stack, {M} rows per step)")
    print(" | ".join(line), flush=True)
