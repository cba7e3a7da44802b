"""Dev tool: small/mid-M regime -- skinny (force_kernel=1) vs tiled (force_kernel=2).  Graph of many launches over
rotating weight copies (a single-kernel graph replay has a ~10 us floor that would swamp the kernels)."""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import _lib
if os.environ.get("GPTQHIP_LIB"):      # dev A/B builds (tests/dev/ablate/*.so)
    _lib.LIB_PATH = os.environ["GPTQHIP_LIB"]
from gptqmodel_amd import ops
dev = "cuda"; gs = 128
KERNS = tuple(int(v) for v in os.environ.get("MIDM_KERNELS", "1,2").split(","))   # 1 decode kernel, 2 prefill kernel, 0 gptqhip_gemm's own choice
def gtime(fn, n_launch, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n_launch)
Ms = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024]
SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["MIDM_SHAPES"].split(",")] if os.environ.get("MIDM_SHAPES") else [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)]
for (K, N) in SHAPES:
    copies = max(4, min(32, (600 << 20) // (K * N // 2)))
    sets = []
    for _ in range(copies):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
        sets.append(ops.repack_tiled(qw, qz, sc, None, gs, 4))
    for M in Ms:
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        res = []
        for kern in KERNS:
            if kern == 2 and M < 5: continue
            if kern == 1 and M > 256: continue
            ops.set_tuning(0, kern, 0)
            def fn():
                for qw_t, meta in sets: ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
            us = gtime(fn, len(sets))
            res.append(f"{ {0: 'auto', 1: 'skinny', 2: 'tiled'}[kern] } {us:.1f}us {2*M*K*N/us/1e6:.0f}TF" + (f" [{ops.plan_describe(M, K, N, gs).split(' ')[0]}]" if kern == 0 else ""))
        ops.set_tuning(0, 0, 0)
        print(f"K={K} N={N} M={M}: " + " | ".join(res), flush=True)
    del sets
