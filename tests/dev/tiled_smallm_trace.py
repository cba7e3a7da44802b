"""dev (run under rocprofv3 --kernel-trace): the prefill kernel at small M -- main kernel vs split-K reduce kernel durations."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops
dev = "cuda"
for (K, N) in [(4096, 4096), (4096, 28672), (14336, 4096)]:
    sets = []
    for _ in range(6):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
        sets.append(ops.repack_tiled(qw, qz, sc, None, 128, 4))
    for M in (48, 64, 128, 256):
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        ops.set_tuning(0, 2, 0)
        for it in range(12):
            qw_t, meta = sets[it % 6]
            ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, out=out)
        torch.cuda.synchronize()
        print("PLAN", M, K, N, ops.plan_describe(M, K, N, 128), flush=True)
ops.set_tuning(0, 0, 0)
