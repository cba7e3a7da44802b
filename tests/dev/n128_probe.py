"""dev (round 5): the prefill kernel with 128-column blocks (one column tile per wave) against the planner's 256-column choice:
parity (against the default path's output) and time per launch (graph replay, rotating cold weights)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops  # noqa: E402

dev, gs = "cuda", 128
MS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 72, 128, 136, 192, 256, 384]
SHAPES = ([tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES")
          else [(4096, 11008), (11008, 4096), (4096, 4096), (4096, 28672), (4096, 6144), (14336, 4096)])


def gtime(fn, n_launch, reps=4):
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


for (K, N) in SHAPES:
    copies = max(4, min(16, (400 << 20) // (K * N // 2)))
    sets = []
    for _ in range(copies):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
        sets.append(ops.repack_tiled(qw, qz, sc, None, gs, 4))
    for M in MS:
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        bias = (torch.randn(N, device=dev) * 0.1).half()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)

        def fn():
            for qw_t, meta in sets:
                ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
        ops.set_tuning(0, 0, 0)
        ref = ops.gemm(x, sets[0][0], sets[0][1], bias, None, N, gs, 4, torch.float16).float()
        res = [f"auto {gtime(fn, len(sets)):.1f} ({ops.plan_describe(M, K, N, gs).replace('tiled ', '').replace(' tail_cols=0 gather=0', '')})"]
        chunks = K // 128
        for bm in (64, 128):
            tiles = -(-N // 128) * -(-M // bm)
            for s in (1, 2, 3, 4, 6, 8):
                if s > 1 and (tiles * s > 320 or s > chunks // 4):
                    continue
                ops.set_tuning(s, 2, 1000 + bm)
                got = ops.gemm(x, sets[0][0], sets[0][1], bias, None, N, gs, 4, torch.float16).float()
                err = float((got - ref).abs().max() / ref.abs().max())
                res.append(f"n128 bm{bm} s{s}: {gtime(fn, len(sets)):.1f}" + ("" if err < 1e-3 else f" ERR {err:.2e}"))
        ops.set_tuning(0, 0, 0)
        print(f"K={K} N={N} M={M}: " + " | ".join(res), flush=True)
    del sets
    torch.cuda.empty_cache()
