"""dev (round 5): calibration sweep for the prefill planner with tile heights in steps of 16 rows.  For every (shape, M) times the planner's
own choice (`auto`) and forced (tile height, split-K) points around the launch model's optimum per height; graph replay over rotating
weight copies (cold weights, the dependent-launch gap included -- what bench.py's T1 `us_graph` measures).  One line per point:
    P K N M bm s us            (bm = 0: the planner's choice, with its plan text)
tests/dev/midm_fit.py refits the model's coefficients from these lines."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops  # noqa: E402

dev, gs = "cuda", 128
HEIGHTS = [int(v) for v in os.environ.get("HEIGHTS", "32,48,64,80,96,112,128,256").split(",")]
MS = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [72, 96, 128, 136, 160, 192, 256, 320, 512]
SHAPES = ([tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES")
          else [(4096, 11008), (11008, 4096), (4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)])
FORCE = {256: 1, 128: 2, 64: 3}
BN = int(os.environ.get("BN", "256"))        # 128: the 128-column-block form (force code 1000 + rows); lines then read `P K N M bm s us` with bm = 1000 + rows


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


def splits_to_try(M, K, N, bm):
    chunks = -(-K // 128)
    tiles = -(-N // BN) * -(-M // bm)
    smax = max(1, min(chunks // 4, 256 // max(1, tiles), 16))
    cand = {1, smax, max(1, smax - 1), max(1, smax // 2), max(1, (smax * 3) // 4)}
    if bm <= 64 and os.environ.get("OVERSUB"):      # two blocks per CU fit (<= 128 VGPRs, 48 KiB of LDS): up to 512 blocks
        s2 = max(1, min(chunks // 4, 512 // max(1, tiles), 16))
        cand |= {s2, max(1, (smax + s2) // 2)}
    return sorted(s for s in cand if s * M * N <= (16 << 20) or s == 1)


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
        out = torch.empty((M, N), dtype=torch.float16, device=dev)

        def fn():
            for qw_t, meta in sets:
                ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
        ops.set_tuning(0, 0, 0)
        print(f"P {K} {N} {M} 0 0 {gtime(fn, len(sets)):.2f}  # {ops.plan_describe(M, K, N, gs)}", flush=True)
        for bm in HEIGHTS:
            if (bm < 64 and M > 4 * bm) or (bm <= 64 and M > 1024):
                continue
            if bm == 256 and (M < 192 or BN == 128):
                continue
            for s in splits_to_try(M, K, N, bm):
                ops.set_tuning(s, 2, 1000 + bm if BN == 128 else FORCE.get(bm, bm))
                d = ops.plan_describe(M, K, N, gs)
                if f"bm={bm} " not in d or (BN == 128) != ("bn=128" in d):
                    continue
                s_eff = int(d.split("splits=")[1].split(" ")[0])
                print(f"P {K} {N} {M} {1000 + bm if BN == 128 else bm} {s_eff} {gtime(fn, len(sets)):.2f}", flush=True)
        ops.set_tuning(0, 0, 0)
    del sets
    torch.cuda.empty_cache()
