"""Does plan_tiled pick the fastest tile height?  auto vs forced 256 / 128 / 64-row tiles over M x (K, N), 4-bit g128 fp16."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops
dev = "cuda"
def timeit(fn, iters=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
gen = torch.Generator(device=dev); gen.manual_seed(0)
for K, N in ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev, generator=gen)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev, generator=gen)
    sc = (torch.rand((K // 128, N), device=dev, generator=gen) * 0.01 + 0.005).half()
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, 128, 4)
    for M in (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384):
        x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half()
        res = []
        for tag, variant in (("auto", 0), ("256", 1), ("128", 2), ("64", 3)):
            try:
                ops.set_tuning(0, 2 if variant else 0, variant)
                ms = timeit(lambda: ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16))
                res.append(f"{tag} {ms*1e3:8.1f} us")
            except Exception as e:
                res.append(f"{tag} ERR")
            finally:
                ops.set_tuning(0, 0, 0)
        print(f"K={K} N={N} M={M:6d}: " + " | ".join(res), flush=True)
