"""Dev tool (run under rocprofv3 --kernel-trace): main-kernel duration vs chunks per block, to separate the fixed
prologue/epilogue cost of a split-K block from its per-chunk cost."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
M, K, N = (int(v) for v in sys.argv[1:4])
dev = "cuda"
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
sets = [ops.repack_tiled(qw, qz, sc, None, 128, 4) for _ in range(8)]
x = (torch.randn(M, K, device=dev) * 0.5).half()
out = torch.empty((M, N), dtype=torch.float16, device=dev)
for s in (1, 2, 4, 8, 16, 32):
    ops.set_tuning(s, 2, 0)
    for it in range(12):
        qw_t, meta = sets[it % 8]
        ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, out=out)
    torch.cuda.synchronize()
ops.set_tuning(0, 0, 0)
