"""Dev tool (run under rocprofv3 --kernel-trace --stats): act-order prefill, to read the gather pre-pass kernel's own duration."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
M, K, N = (int(v) for v in sys.argv[1:4])
dev, gs = "cuda", 128
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
g_idx = (torch.randperm(K, device=dev) // gs).int()
perm = torch.argsort(g_idx.long(), stable=True).int()
qw_t, meta = ops.repack_tiled(qw, qz, sc, perm, gs, 4)
x = (torch.randn(M, K, device=dev) * 0.5).half()
out = torch.empty((M, N), dtype=torch.float16, device=dev)
for _ in range(20):
    ops.gemm(x, qw_t, meta, None, perm, N, gs, 4, torch.float16, out=out)
torch.cuda.synchronize()
