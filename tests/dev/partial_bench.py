import sys, torch
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
dev = "cuda"
def run(M, K, N, variant, partial):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, 128, 4)
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    ops.set_tuning(0, 0, variant)
    for _ in range(10): ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, partial_f32=partial)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, partial_f32=partial)
    e1.record(); torch.cuda.synchronize()
    ops.set_tuning(0, 0, 0)
    ms = e0.elapsed_time(e1) / 10
    return 2.0 * M * K * N / ms / 1e9
for (M, K, N) in [(8192, 4096, 4096), (8192, 3584, 8192), (4096, 14336, 4096)]:
    print(M, K, N, "16-bit bm256 %.0f | f32 bm256 %.0f | f32 bm128 %.0f | 16-bit bm128 %.0f" % (run(M,K,N,1,False), run(M,K,N,1,True), run(M,K,N,2,True), run(M,K,N,2,False)), flush=True)
