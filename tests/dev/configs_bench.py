"""Dev tool: the other BASELINE configs as kernel-level numbers: bf16 decode, AWQ decode, act-order prefill."""
import sys, torch
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
dev = "cuda"
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def graph_time(fn, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) / reps
K, N, gs = 4096, 28672, 128
# decode, rotating copies
for act, sdt in ((torch.float16, torch.float16), (torch.bfloat16, torch.float16), (torch.bfloat16, torch.bfloat16)):
    sets = []
    for _ in range(10):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).to(sdt)
        sets.append(ops.repack_tiled(qw, qz, sc, None, gs, 4))
    x = torch.randn(1, K, device=dev).to(act)
    def step():
        for qw_t, meta in sets: ops.gemm(x, qw_t, meta, None, None, N, gs, 4, sdt)
    ms = graph_time(step)
    print(f"decode M=1 {K}x{N} act={act} scales={sdt}: {ms*1e3/len(sets):.2f} us/launch  {K*N/2/(ms*1e-3/len(sets))/1e12:.2f} TB/s", flush=True)
    del sets
# decode with act-order (desc_act=True): the permutation is applied inside the kernel at M == 1
for (K2, N2) in ((4096, 28672), (14336, 4096), (4096, 6144)):
    sets, perms = [], []
    for _ in range(10):
        qw = torch.randint(-2**31, 2**31 - 1, (K2 // 8, N2), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (K2 // gs, N2 // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K2 // gs, N2), device=dev) * 0.01 + 0.005).half()
        g_idx = (torch.randperm(K2, device=dev) // gs).int()
        pm = torch.argsort(g_idx.long(), stable=True).int()
        sets.append(ops.repack_tiled(qw, qz, sc, pm, gs, 4) + (pm,))
    x = torch.randn(1, K2, device=dev).half()
    def step_p():
        for qw_t, meta, pm in sets: ops.gemm(x, qw_t, meta, None, pm, N2, gs, 4, torch.float16)
    def step_n():
        for qw_t, meta, pm in sets: ops.gemm(x, qw_t, meta, None, None, N2, gs, 4, torch.float16)
    tp, tn = graph_time(step_p), graph_time(step_n)
    print(f"decode M=1 {K2}x{N2} fp16: desc_act=True {tp*1e3/len(sets):.2f} us/launch | desc_act=False {tn*1e3/len(sets):.2f} us/launch", flush=True)
    del sets
# prefill with act-order gather (C3): M = 2048 and 8192, 4096x4096
K, N = 4096, 4096
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
qz = torch.randint(-2**31, 2**31 - 1, (K // gs, N // 8), dtype=torch.int32, device=dev)
sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
g_idx = (torch.randperm(K, device=dev) // gs).int()
perm = torch.argsort(g_idx.long(), stable=True).int()
qw_t, meta = ops.repack_tiled(qw, qz, sc, perm, gs, 4)
qw_t0, meta0 = ops.repack_tiled(qw, qz, sc, None, gs, 4)
for M in (2048, 8192, 65536):
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    t0 = timeit(lambda: ops.gemm(x, qw_t0, meta0, None, None, N, gs, 4, torch.float16, out=out))
    t1 = timeit(lambda: ops.gemm(x, qw_t, meta, None, perm, N, gs, 4, torch.float16, out=out))
    print(f"prefill M={M} 4096x4096: desc_act=False {2*M*K*N/t0/1e9:.0f} TF ({t0:.3f} ms) | desc_act=True (x gather pre-pass) {2*M*K*N/t1/1e9:.0f} TF ({t1:.3f} ms)", flush=True)
