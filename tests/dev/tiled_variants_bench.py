"""Prefill kernel on the variants that are confined to 128-row tiles (8-bit weights, group 32/64): 128- vs 256-row tiles."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops
dev = "cuda"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
gen = torch.Generator(device=dev); gen.manual_seed(0)
for bits, gs in ((4, 128), (8, 128), (4, 64), (4, 32)):
    for M, K, N in ((2048, 4096, 4096), (8192, 4096, 4096), (8192, 4096, 14336)):
        pf = 32 // bits
        qw = torch.randint(-2**31, 2**31 - 1, (K // pf, N), dtype=torch.int32, device=dev, generator=gen)
        qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // pf), dtype=torch.int32, device=dev, generator=gen)
        sc = (torch.rand((K // gs, N), device=dev, generator=gen) * 0.01 + 0.005).half()
        qw_t, meta = ops.repack_tiled(qw, qz, sc, None, gs, bits)
        x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half()
        res = []
        outs = {}
        for tag, variant in (("128-row", 2), ("256-row", 1)):
            try:
                ops.set_tuning(0, 2, variant)
                ms = timeit(lambda: ops.gemm(x, qw_t, meta, None, None, N, gs, bits, torch.float16))
                outs[tag] = ops.gemm(x, qw_t, meta, None, None, N, gs, bits, torch.float16)
                res.append(f"{tag} {2.0 * M * K * N / ms / 1e9:7.1f} TF")
            except Exception as e:
                res.append(f"{tag} ERR {str(e)[:50]}")
            finally:
                ops.set_tuning(0, 0, 0)
        same = "same-bits" if len(outs) == 2 and torch.equal(outs["128-row"], outs["256-row"]) else "DIFFER"
        print(f"bits={bits} g={gs:3d} M={M} K={K} N={N}: " + " | ".join(res) + f" | {same}", flush=True)
