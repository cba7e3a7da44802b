"""Dev tool: dequant-GEMM TFLOPS of the prefill kernel (methodology of the reference's
scripts/benchmark_marlin_a100.py: tflops = 2*M*K*N / t, random int32 qweight, warmup then timed iters)."""
import sys, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops
dev = "cuda"
DT = torch.bfloat16 if "bf16" in sys.argv else torch.float16
def run(M, K, N, gs=128, iters=20):
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).to(torch.bfloat16 if "bf16s" in sys.argv else torch.float16)
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, gs, 4)
    x = (torch.randn(M, K, device=dev) * 0.5).to(DT)
    out = torch.empty((M, N), dtype=DT, device=dev)
    for _ in range(25): ops.gemm(x, qw_t, meta, None, None, N, gs, 4, sc.dtype, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.gemm(x, qw_t, meta, None, None, N, gs, 4, sc.dtype, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * K * N / ms / 1e9
variants = [0] if len(sys.argv) < 2 or not sys.argv[1][0].isdigit() else [int(v) for v in sys.argv[1].split(",")]  # 1 / 2 force 256- / 128-row tiles
for (M, K, N) in [(128,4096,4096),(512,4096,4096),(2048,4096,4096),(8192,4096,4096),(2048,4096,14336),(2048,14336,4096),(8192,4096,28672),(65536,4096,4096),(2048,4096,6144),(4096,4096,6144),(2048,4096,28672),(4096,4096,28672),(2560,4096,4096),(3072,4096,4096),(1024,4096,14336),(1024,4096,28672),(1536,4096,6144)]:
    res = []
    for v in variants:
        ops.set_tuning(0, 0, v)
        ms, tf = run(M, K, N)
        res.append(f"v{v}: {ms:.3f} ms {tf:.1f} TFLOPS")
    ops.set_tuning(0, 0, 0)
    print(f"M={M} K={K} N={N}: " + " | ".join(res), flush=True)
