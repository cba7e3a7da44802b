"""Dev probe: is a re-read of a just-read buffer (Infinity-Cache warm) faster than a cold HBM read?"""
import ctypes, sys, torch
lib = ctypes.CDLL("/root/repo/tests/dev/libprobe.so")
lib.probe_launch.argtypes = [ctypes.c_void_p]*2 + [ctypes.c_int]*4 + [ctypes.c_void_p]
def run(k, n, waves, nt, copies, reps=20):
    tiles, chunks = n // 16, k // 128
    bufs = [torch.randint(0, 2**31-1, (tiles*chunks*256,), dtype=torch.int32, device="cuda") for _ in range(copies)]
    out = torch.zeros(tiles, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for b in bufs[:2]: lib.probe_launch(b.data_ptr(), out.data_ptr(), tiles, chunks, waves, nt, s.cuda_stream)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(max(1, 16 // copies)):
                for b in bufs: lib.probe_launch(b.data_ptr(), out.data_ptr(), tiles, chunks, waves, nt, s.cuda_stream)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): g.replay()
        e1.record(s); s.synchronize()
    n_launch = reps * max(1, 16 // copies) * copies
    return e0.elapsed_time(e1) * 1e3 / n_launch
for (k, n) in [(4096, 4096), (4096, 28672), (8192, 28672)]:
    per = k * n // 2
    for copies in (1, 2, 4, 16):
        if per * copies > (2 << 30): continue
        for nt in (0, 1):
            us = run(k, n, 8, nt, copies)
            print(f"K={k} N={n} {per/1e6:.0f}MB x{copies} copies ({per*copies/1e6:.0f} MB working set) nt={nt}: {us:.2f} us  {per/us/1e6:.2f} TB/s", flush=True)
