"""Dev microbenchmark: the act-order gather pass (gptqhip_gather_cols) and the fused RMSNorm + gather (gptqhip_rmsnorm_gather)
as HBM streams -- us per launch and GB/s (read + write of [M, K] 16-bit), checked against torch indexing.
GPTQHIP_LIB=<path> times another build (A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gptqmodel_amd import _lib  # noqa: E402

if os.environ.get("GPTQHIP_LIB"):
    _lib.LIB_PATH = os.environ["GPTQHIP_LIB"]
from gptqmodel_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    torch.manual_seed(0)
    print(f"lib: {_lib.LIB_PATH if hasattr(_lib, 'LIB_PATH') else 'default'}")
    for dtype in (torch.float16, torch.bfloat16):
        for M, K in ((65536, 4096), (8192, 4096), (2048, 4096), (16384, 8192), (2048, 14336)):
            x = torch.randn(M, K, device="cuda", dtype=dtype)
            w = (torch.randn(K, device="cuda") * 0.1 + 1).to(dtype)
            perm = torch.randperm(K, device="cuda").to(torch.int32)
            out = ops.gather_cols(x, perm)
            assert torch.equal(out, x[:, perm.long()])
            t_g = timeit(lambda: ops.gather_cols(x, perm))
            o2 = ops.rmsnorm_gather(x, w, 1e-5, perm)
            var = x.float().pow(2).mean(-1, keepdim=True)
            ref = (w * (x.float() * torch.rsqrt(var + 1e-5)).to(dtype))[:, perm.long()]
            bad = (o2.float() - ref.float()).abs().max().item()
            t_r = timeit(lambda: ops.rmsnorm_gather(x, w, 1e-5, perm))
            t_n = timeit(lambda: ops.rmsnorm_gather(x, w, 1e-5, None))
            t_c = timeit(lambda: out.copy_(x))
            gb = 2 * M * K * 2 / 1e3
            print(f"{str(dtype)[6:]:9s} M={M:6d} K={K:6d}: gather {t_g:8.1f} us {gb / t_g:7.0f} GB/s | rmsnorm+gather {t_r:8.1f} us "
                  f"{gb / t_r:7.0f} GB/s (max abs diff {bad:.3g}) | rmsnorm {t_n:8.1f} us {gb / t_n:7.0f} GB/s | torch copy {t_c:8.1f} us "
                  f"{gb / t_c:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
