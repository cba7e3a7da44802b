import ctypes, sys, torch
lib = ctypes.CDLL("/root/repo/tests/dev/libprobe.so")
lib.probe_launch.argtypes = [ctypes.c_void_p]*2 + [ctypes.c_int]*4 + [ctypes.c_void_p]
def run(k, n, waves, nt, copies):
    tiles, chunks = n // 16, k // 128
    bufs = [torch.randint(0, 2**31-1, (tiles*chunks*256,), dtype=torch.int32, device="cuda") for _ in range(copies)]
    out = torch.zeros(tiles, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for b in bufs[:2]: lib.probe_launch(b.data_ptr(), out.data_ptr(), tiles, chunks, waves, nt, s.cuda_stream)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for b in bufs: lib.probe_launch(b.data_ptr(), out.data_ptr(), tiles, chunks, waves, nt, s.cuda_stream)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5): g.replay()
        e1.record(s); s.synchronize()
    us = e0.elapsed_time(e1)*1e3/(5*copies)
    return us
for (k, n) in [(4096,1024),(4096,4096),(4096,14336),(14336,4096),(4096,28672),(8192,28672)]:
    per = k*n//2; copies = max(4, min(64, (600<<20)//per))
    res = []
    for waves in (4, 8, 16):
        for nt in (0, 1):
            us = run(k, n, waves, nt, copies)
            if us: res.append(f"W{waves}{'nt' if nt else '  '} {us:.1f}us {per/us/1e6:.2f}TB/s")
    print(f"K={k} N={n}: " + " | ".join(res), flush=True)
