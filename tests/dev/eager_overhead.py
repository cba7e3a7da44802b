"""Host overhead of the eager product path: HipGptqLinear.forward called from Python per launch vs the
same launches replayed from a HIP graph, 4096x4096 and the fused gate_up shape, M=1."""
import sys, os, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench as B

dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
for K, N in ((4096, 4096), (4096, 28672)):
    lins = [B.make_gptq(K, N, 128, dev, gen, torch.float16) for _ in range(16)]
    x = (torch.randn(1, K, device=dev) * 0.5).half()
    for l in lins:
        l(x)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        for l in lins:
            l(x)
    t_issue = time.perf_counter() - t0       # host time to ISSUE (may run ahead of the GPU)
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t0
    stream = torch.cuda.Stream()
    ms, g = B.time_graph(lambda: [l(x) for l in lins], stream, 100, 10)
    print(f"K={K} N={N}: eager forward() host issue {t_issue/n/16*1e6:6.2f} us/call, eager end-to-end {t_total/n/16*1e6:6.2f} us/call, "
          f"graph replay {ms*1e3/16:6.2f} us/call", flush=True)
    del lins, g
    torch.cuda.empty_cache()
