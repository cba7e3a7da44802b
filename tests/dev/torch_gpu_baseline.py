"""Dev tool: the reference's BACKEND.TORCH arithmetic on THIS GPU, next to the fused kernel.

BACKEND.TORCH on a ROCm device (gptqmodel/nn_modules/qlinear/torch.py:326-347) = dequantise the whole [K,N] weight to
fp16, then aten matmul (hipBLASLt).  Three columns per shape:
  fused   : gptqhip_gemm (this repo)
  dense   : torch.matmul(x, W_fp16) alone -- the hipBLASLt ceiling with the weight already dequantised
  deq+mm  : our standalone dequant kernel + torch.matmul -- a generous stand-in for the reference's torch path
            (its own dequant is several elementwise aten kernels and therefore slower than ours)
"""
import sys, torch
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
dev = "cuda"
DT = torch.bfloat16 if "bf16" in sys.argv else torch.float16

def timeit(fn, iters):
    """One HIP graph of `iters` calls (no per-call launch overhead), best of 3 replays after a warm-up replay."""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(3): fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(iters): fn()
        g.replay(); st.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); g.replay(); e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters)
    return best

def run(M, K, N, gs=128):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, gs, 4)
    x = (torch.randn(M, K, device=dev) * 0.5).to(DT)
    out = torch.empty((M, N), dtype=DT, device=dev)
    W = ops.dequant_tiled(qw_t, meta, None, K, N, gs, 4, torch.float16, DT)
    iters = 10 if M >= 512 else 50
    f_f = lambda: ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
    f_d = lambda: torch.matmul(x, W, out=out)
    f_r = lambda: torch.matmul(x, ops.dequant_tiled(qw_t, meta, None, K, N, gs, 4, torch.float16, DT), out=out)
    t_f = t_d = t_r = 1e30
    for _ in range(2):  # interleaved: the clocks ramp and throttle, no leg should always go first
        t_d = min(t_d, timeit(f_d, iters)); t_f = min(t_f, timeit(f_f, iters)); t_r = min(t_r, timeit(f_r, iters))
    fl = 2.0 * M * K * N / 1e9
    print(f"M={M} K={K} N={N}: fused {t_f*1e3:.1f} us {fl/t_f:.0f} TF | dense {t_d*1e3:.1f} us {fl/t_d:.0f} TF | "
          f"deq+mm {t_r*1e3:.1f} us {fl/t_r:.0f} TF | fused/dense {t_d/t_f:.2f} fused/deq+mm {t_r/t_f:.2f}", flush=True)

for (M, K, N) in [(1,4096,6144),(1,4096,4096),(1,4096,28672),(1,14336,4096),(16,4096,4096),(128,4096,4096),(512,4096,4096),
                  (2048,4096,4096),(8192,4096,4096),(2048,4096,14336),(2048,14336,4096),(8192,4096,28672),(65536,4096,4096)]:
    run(M, K, N)
