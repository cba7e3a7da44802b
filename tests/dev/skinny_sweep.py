"""Dev tool (not a test): sweep skinny-kernel launch geometry per shape on the GPU.  Times graph replays of N
back-to-back launches over ROTATING weight copies (> 512 MB total, defeats the 256 MB Infinity Cache)."""
import sys, itertools, json
import numpy as np, torch
sys.path.insert(0, "/root/repo"); 
from gptqmodel_amd import ops

def bytes_alg(m, k, n, gs=128):
    g = k // gs
    return k * n // 2 + g * n * 2 + g * n // 2 + m * (k + n) * 2

def make(k, n, gs, dev):
    qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev)
    qz = torch.full((k // gs, n // 8), -2004318072, dtype=torch.int32, device=dev)
    sc = (torch.rand((k // gs, n), device=dev) * 0.01 + 0.005).half()
    return ops.repack_tiled(qw, qz, sc, None, gs, 4)

def time_cfg(x, sets, n, gs, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for qw_t, meta in sets[:2]:
            ops.gemm(x, qw_t, meta, None, None, n, gs, 4, torch.float16)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for qw_t, meta in sets:
                ops.gemm(x, qw_t, meta, None, None, n, gs, 4, torch.float16)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets))

def main():
    dev = "cuda"
    shapes = [(4096, 4096), (4096, 1024), (4096, 14336), (14336, 4096), (4096, 6144), (4096, 28672)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    M = 1
    for k, n in shapes:
        per = k * n // 2
        copies = max(4, min(64, (600 << 20) // per))
        sets = [make(k, n, 128, dev) for _ in range(copies)]
        x = torch.randn(M, k, device=dev, dtype=torch.float16)
        res = []
        for W, S in itertools.product([0, 4, 8, 16], [0, 1, 2, 4, 8]):
            if (W == 0) != (S == 0): continue
            ops.set_tuning(S, 0, W)
            try:
                us = time_cfg(x, sets, n, 128)
            except RuntimeError as e:
                continue
            res.append((us, W, S))
        ops.set_tuning(0, 0, 0)
        b = bytes_alg(M, k, n)
        best = min(res)
        print(f"K={k} N={n} copies={copies}: " + " ".join(f"[W{w}S{s} {us:.1f}us]" for us, w, s in sorted(res)[:6]),
              f"| heuristic {[r for r in res if r[1]==0][0][0]:.1f}us | best {b/best[0]/1e6:.2f} TB/s", flush=True)

main()
