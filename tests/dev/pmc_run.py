"""Dev tool: a handful of skinny-kernel launches over rotating weights for rocprofv3 --pmc / --kernel-trace."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
k, n = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "4096x28672").split("x"))
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda"
sets = []
for _ in range(copies):
    qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev)
    qz = torch.full((k // 128, n // 8), -2004318072, dtype=torch.int32, device=dev)
    sc = (torch.rand((k // 128, n), device=dev) * 0.01 + 0.005).half()
    sets.append(ops.repack_tiled(qw, qz, sc, None, 128, 4))
x = torch.randn(1, k, device=dev, dtype=torch.float16)
for _ in range(3):
    for qw_t, meta in sets:
        ops.gemm(x, qw_t, meta, None, None, n, 128, 4, torch.float16)
torch.cuda.synchronize()
