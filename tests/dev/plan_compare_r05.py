import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gptqmodel_amd import ops
dev, gs = "cuda", 128
def gtime(fn, n_launch, reps=4):
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
    return e0.elapsed_time(e1) * 1e3 / (reps * n_launch)
CASES = [(14336,4096,896,(128,2),(256,4)),(28672,8192,513,(128,1),(256,2)),(28672,8192,640,(128,1),(256,2)),(28672,8192,896,(128,1),(256,2)),
         (11008,4096,2048,(128,1),(256,2)),(5120,5120,513,(64,1),(128,2)),(5120,5120,640,(64,1),(128,2)),(5120,5120,1024,(128,1),(256,3))]
FORCE = {256: 1, 128: 2, 64: 3}
for (K,N,M,old,new) in CASES:
    copies = max(4, min(16, (400 << 20) // (K * N // 2)))
    sets = []
    for _ in range(copies):
        qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
        qz = torch.zeros((K // gs, N // 8), dtype=torch.int32, device=dev)
        sc = (torch.rand((K // gs, N), device=dev) * 0.01 + 0.005).half()
        sets.append(ops.repack_tiled(qw, qz, sc, None, gs, 4))
    x = (torch.randn(M, K, device=dev) * 0.5).half(); out = torch.empty((M, N), dtype=torch.float16, device=dev)
    def fn():
        for qw_t, meta in sets: ops.gemm(x, qw_t, meta, None, None, N, gs, 4, torch.float16, out=out)
    res = []
    for rep in range(2):
        for (bm, s) in (old, new):
            ops.set_tuning(s, 2, FORCE[bm]); d = ops.plan_describe(M, K, N, gs)
            res.append(f"bm{bm}s{s}: {gtime(fn, len(sets)):.1f}")
    ops.set_tuning(0, 0, 0)
    print(K, N, M, "old/new/old/new:", " | ".join(res), "| auto:", ops.plan_describe(M, K, N, gs), flush=True)
    del sets
