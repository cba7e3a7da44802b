"""Dev tool: 8-bit GPTQ and AWQ (asymmetric) through the same kernels -- decode us/launch and prefill TFLOPS."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
dev = "cuda"; gs = 128
def graph_time(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s): fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / reps
def evtime(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
K, N = 4096, 14336
for name, bits in (("gptq-w4", 4), ("gptq-w8", 8), ("awq-w4", 4)):
    pf = 32 // bits
    sets = []
    for _ in range(12):
        sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
        if name.startswith("awq"):
            qw_a = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device=dev)
            qz_a = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
            qw, qz = ops.repack_awq(qw_a, qz_a)
        else:
            qw = torch.randint(-2**31, 2**31 - 1, (K // pf, N), dtype=torch.int32, device=dev)
            qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // pf), dtype=torch.int32, device=dev)
        sets.append(ops.repack_tiled(qw, qz, sc, None, gs, bits))
    x1 = torch.randn(1, K, device=dev, dtype=torch.float16)
    def dec():
        for qw_t, meta in sets: ops.gemm(x1, qw_t, meta, None, None, N, gs, bits, torch.float16)
    us = graph_time(dec) * 1e3 / len(sets)
    wbytes = K * N * bits // 8
    xM = (torch.randn(4096, K, device=dev) * 0.5).half()
    out = torch.empty((4096, N), dtype=torch.float16, device=dev)
    qw_t, meta = sets[0]
    ms = evtime(lambda: ops.gemm(xM, qw_t, meta, None, None, N, gs, bits, torch.float16, out=out))
    print(f"{name}: decode M=1 {K}x{N}: {us:.2f} us/launch {wbytes/us/1e6:.2f} TB/s | prefill M=4096: {2*4096*K*N/ms/1e9:.0f} TFLOPS", flush=True)
    del sets
