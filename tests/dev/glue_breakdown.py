"""Which piece of the decode op's glue costs what: one shape, 24 cold-weight layers per graph, variants toggled per launch."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench as B
from gptqmodel_amd import ops

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
dtype = torch.float16
NL = 24
stream = torch.cuda.Stream()
import traceback
SH = [("gate_up", 4096, 28672), ("qkv", 4096, 6144), ("down", 14336, 4096)]
if len(sys.argv) > 1 and sys.argv[1] == "70b":
    NL = 8
    SH = [("qkv70", 8192, 10240), ("o70", 8192, 8192), ("gate_up70", 8192, 57344), ("down70", 28672, 8192)]
for name, K, N in SH:
    lins = [B.make_gptq(K, N, 128, dev, gen, dtype) for _ in range(NL)]
    xin = (torch.randn(2 * K, device=dev, generator=gen) * 0.5).to(dtype)
    nw = torch.ones(K, dtype=dtype, device=dev)
    resid = torch.zeros(N, dtype=dtype, device=dev)
    outs = [torch.empty(N, dtype=dtype, device=dev) for _ in range(NL)]
    st_in = torch.ones(K // 16, dtype=torch.float32, device=dev)
    st_out = torch.zeros(N // 16, dtype=torch.float32, device=dev)
    if os.environ.get("ONLY"):
        keep = os.environ["ONLY"].split(",")
    else:
        keep = None
    variants = {
        "plain": dict(),
        "rms(stats)": dict(in_glue=ops.GLUE_RMSNORM, norm_weight=nw, stats_in=st_in),
        "rms(block)": dict(in_glue=ops.GLUE_RMSNORM, norm_weight=nw),
        "paired-out": dict(out_glue=ops.OUT_SILU_MUL_PAIRED),
        "rms(stats)+paired": dict(in_glue=ops.GLUE_RMSNORM, norm_weight=nw, stats_in=st_in, out_glue=ops.OUT_SILU_MUL_PAIRED),
        "residual": dict(residual=resid),
        "residual+stats_out": dict(residual=resid, stats_out=st_out),
        "silu-in": dict(in_glue=ops.GLUE_SILU_MUL) ,
    }
    line = [f"{name:8s}"]
    for tag, kw in variants.items():
      if keep and tag not in keep:
          continue
      try:
        dops = [ops.make_decode_op(xin, l.qweight, l.meta, None, o, K, N, 128, 4, l._scale_dtype, **kw) for l, o in zip(lins, outs)]
        def run():
            for d in dops:
                ops.launch_decode_op(d, dev)
        ms, g = B.time_graph(run, stream, 30, 5)
        line.append(f"{tag} {ms * 1e3 / NL:6.2f}")
        del g
      except Exception as e:
        line.append(f"{tag} ERR {str(e)[:60]}")
    print(" | ".join(line), flush=True)
    del lins, outs
    torch.cuda.empty_cache()
