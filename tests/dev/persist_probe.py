"""Dev probe driver (tests/dev/persist_probe.hip): per-layer time of four launches vs one persistent launch with grid barriers
(+ weight prefetch across the barrier), Llama-3-8B op sizes, distinct weights per layer (cold), graph replay."""
import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libpersist_probe.so"))
lib.layer_probe_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_size_t, ctypes.c_int] + [ctypes.c_void_p] * 3
dev = "cuda"
shapes = {"8b": [(4096, 4096), (4096, 28672), (14336, 4096), (4096, 6144)], "70b": [(8192, 8192), (8192, 57344), (28672, 8192), (8192, 10240)]}
for name, ops in shapes.items():
    kib = [k * n // 2 // 1024 for k, n in ops]
    stride = max(kib) * 1024
    layers = 6 if name == "8b" else 2
    bufs = [torch.randint(0, 2**31 - 1, (layers * stride // 4,), dtype=torch.int32, device=dev) for _ in range(4)]
    bar = torch.zeros(4096, dtype=torch.int32, device=dev)
    out = torch.zeros(1024, dtype=torch.int32, device=dev)
    total_mb = sum(kib) / 1024
    for mode in (0, 1, 2):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            def fn():
                rc = lib.layer_probe_launch(mode, *[b.data_ptr() for b in bufs], *kib, stride, layers, bar.data_ptr(), out.data_ptr(), s.cuda_stream)
                assert rc == 0, rc
            fn(); s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s): fn()
            g.replay(); s.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10): g.replay()
            e1.record(s); s.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * layers)
        st = int(bar.view(-1)[8 * 64 + 64 + 8 * 64 + 8 * 64].item())
        print(f"{name}: mode {mode} ({['4 launches per layer', 'persistent + grid barrier', 'persistent + barrier + weight prefetch across it'][mode]}): "
              f"{us:.1f} us per layer, {total_mb / us * 1e6 / 1e6:.2f} TB/s of {total_mb:.0f} MB, barrier status {st}", flush=True)
