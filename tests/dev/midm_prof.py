"""Dev tool: run one mid-M shape repeatedly (for rocprofv3 --kernel-trace --stats: per-kernel durations of the
main + split-K reduce launches)."""
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
for it in range(40):
    qw_t, meta = sets[it % 8]
    ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, out=out)  # default dispatch
torch.cuda.synchronize()
