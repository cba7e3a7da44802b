"""Dev experiment: does prefetching the NEXT launch's weights into the Infinity Cache on a side stream (overlapping the
current GEMV) shorten the dependent chain of decode GEMVs?"""
import sys, torch, ctypes
sys.path.insert(0, "/root/repo")
from gptqmodel_amd import ops
lib = ctypes.CDLL("/root/repo/tests/dev/libprobe.so")
lib.probe_launch.argtypes = [ctypes.c_void_p]*2 + [ctypes.c_int]*4 + [ctypes.c_void_p]
dev = "cuda"
gs = 128
shapes = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
L = 32
mods = []
for _ in range(L):
    for k, n in shapes:
        qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev)
        qz = torch.full((k // gs, n // 8), -2004318072, dtype=torch.int32, device=dev)
        sc = (torch.rand((k // gs, n), device=dev) * 0.01 + 0.005).half()
        qw_t, meta = ops.repack_tiled(qw, qz, sc, None, gs, 4)
        mods.append((k, n, qw_t, meta))
xs = {k: torch.randn(1, k, device=dev, dtype=torch.float16) for k in (4096, 14336)}
sink = torch.zeros(4096, dtype=torch.int32, device=dev)

def run(prefetch_dist, nt):
    main = torch.cuda.Stream(); side = torch.cuda.Stream()
    def step():
        evs = []
        for i, (k, n, qw_t, meta) in enumerate(mods):
            if prefetch_dist > 0 and i + prefetch_dist < len(mods):
                kk, nn, qq, _ = mods[i + prefetch_dist]
                # the prefetch of launch i+d may start once launch i-1 is done (bounded run-ahead)
                if evs: side.wait_event(evs[-1])
                lib.probe_launch(qq.data_ptr(), sink.data_ptr(), nn // 16, kk // 128, 8, nt, side.cuda_stream)
            ops.gemm(xs[k], qw_t, meta, None, None, n, gs, 4, torch.float16)
            ev = torch.cuda.Event(); ev.record(main); evs.append(ev)
        main.wait_stream(side)
    with torch.cuda.stream(main):
        side.wait_stream(main)
        step(); main.synchronize(); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            side.wait_stream(main)
            step()
        g.replay(); main.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(20): g.replay()
        e1.record(main); main.synchronize()
    return e0.elapsed_time(e1) / 20
for d, nt in [(0, 0), (1, 0), (1, 1), (2, 0), (3, 0)]:
    ms = run(d, nt)
    print(f"prefetch distance {d} nt={nt}: {ms:.3f} ms/token  {1e3/ms:.0f} tok/s", flush=True)
