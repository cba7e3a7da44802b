"""Per-shape launch cost of the two decode kernels on cold weights: 24 distinct-weight layers of one shape captured as one
graph (stream order), gemv1 decode op (with its glue) vs skinny kernel (ops.gemm, M=1, constant x)."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench as B
from gptqmodel_amd import ops

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
dtype = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] != "bf16") else torch.bfloat16
NL = 24
SHAPES = [("qkv", 4096, 6144, ops.GLUE_RMSNORM, False), ("o", 4096, 4096, ops.GLUE_NONE, True),
          ("gate_up", 4096, 28672, ops.GLUE_RMSNORM, False), ("down", 14336, 4096, ops.GLUE_NONE, True)]
stream = torch.cuda.Stream()
for name, K, N, glue, res in SHAPES:
    lins = [B.make_gptq(K, N, 128, dev, gen, dtype) for _ in range(NL)]
    xin = (torch.randn(2 * K if glue == ops.GLUE_SILU_MUL else K, device=dev, generator=gen) * 0.5).to(dtype)
    nw = torch.ones(K, dtype=dtype, device=dev)
    resid = torch.zeros(N, dtype=dtype, device=dev)
    outs = [torch.empty(N, dtype=dtype, device=dev) for _ in range(NL)]
    st_in = torch.ones(K // 16, dtype=torch.float32, device=dev)
    st_out = torch.zeros(N // 16, dtype=torch.float32, device=dev)
    dops = [ops.make_decode_op(xin, l.qweight, l.meta, None, o, K, N, 128, 4, l._scale_dtype, in_glue=glue,
                               norm_weight=nw if glue == ops.GLUE_RMSNORM else None, residual=resid if res else None,
                               stats_in=st_in if glue == ops.GLUE_RMSNORM else None, stats_out=st_out if res else None,
                               out_glue=ops.OUT_SILU_MUL_PAIRED if name == "gate_up" else ops.OUT_NONE)
            for l, o in zip(lins, outs)]
    plain = [ops.make_decode_op(xin, l.qweight, l.meta, None, o, K, N, 128, 4, l._scale_dtype) for l, o in zip(lins, outs)]
    x2 = xin[:K].reshape(1, K).contiguous()

    def run_gemv1():
        for d in dops:
            ops.launch_decode_op(d, dev)

    def run_gemv1_plain():
        for d in plain:
            ops.launch_decode_op(d, dev)

    def run_skinny():
        for l in lins:
            ops.gemm(x2, l.qweight, l.meta, None, None, N, 128, 4, l._scale_dtype)
    by = B.algorithmic_bytes(1, K, N)
    res_line = [f"{name:8s} K={K:5d} N={N:5d} {by/1e6:6.2f} MB"]
    for tag, fn in (("decode+glue", run_gemv1), ("decode", run_gemv1_plain), ("gemm(M=1)", run_skinny)):
        ms, g = B.time_graph(fn, stream, 30, 5)
        us = ms * 1e3 / NL
        res_line.append(f"{tag} {us:6.2f} us {by/us/1e6:5.2f} TB/s")
        del g
    print(" | ".join(res_line), flush=True)
    del lins, dops, plain, outs
    torch.cuda.empty_cache()
