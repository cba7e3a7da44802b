"""dev: us per call of gptqhip_gemm's own plan on a fixed list of serving-batch cases (graph replay, rotating cold weights, best of 3); for A/B
builds via GPTQHIP_LIB."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops  # noqa: E402

dev, gs = "cuda", 128
CASES = [(4096, 11008, 72), (4096, 11008, 128), (4096, 11008, 136), (4096, 11008, 192), (11008, 4096, 128), (11008, 4096, 192), (4096, 4096, 128),
         (4096, 4096, 192), (4096, 28672, 96), (4096, 6144, 256), (14336, 4096, 64), (4096, 4096, 640)]


def gtime(fn, n_launch, reps=6):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_launch)


res = []
for (K, N, M) in CASES:
    copies = max(4, min(16, (400 << 20) // (K * N // 2)))
    sets = []
    for _ in range(copies):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.zeros((K // gs, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
        sets.append(ops.repack_tiled(qw, qz, sc, None, gs, 4))
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)

    def fn():
        for qw_t, meta in sets:
            ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
    res.append(f"{K}x{N}@{M}: {min(gtime(fn, len(sets)) for _ in range(3)):.2f}")
    del sets
print(" | ".join(res))
