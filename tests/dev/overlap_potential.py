"""Dev experiment (NOT a valid benchmark: it drops the data dependencies a real decoder has): how much of the decode
launch ramp/tail could a dependency-flag scheme recover?  The same 128 launches per token are captured (a) on one stream and
(b) alternating over two / four streams so that consecutive launches may overlap."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
cfg = bench.LLAMA3_8B
shapes = bench.launch_shapes(cfg, True)
layers = [[bench.make_linear(k, n, 128, dev, gen) for k, n, _ in shapes] for _ in range(cfg["layers"])]
xs = {k: (torch.randn((1, k), device=dev, generator=gen) * 0.5).half() for k, _, _ in shapes}
mods = [m for layer in layers for m in layer]

def capture(nstreams):
    main = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in range(nstreams)]
    with torch.cuda.stream(main):
        for s in side:
            with torch.cuda.stream(s):
                for m in mods[:4]: m(xs[m.in_features])   # per-stream workspaces outside capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            if nstreams == 1:
                for m in mods: m(xs[m.in_features])
            else:
                for s in side: s.wait_stream(main)
                for i, m in enumerate(mods):
                    with torch.cuda.stream(side[i % nstreams]):
                        m(xs[m.in_features])
                for s in side: main.wait_stream(s)
    return g, main

for ns in (1, 2, 4):
    g, main = capture(ns)
    with torch.cuda.stream(main):
        for _ in range(10): g.replay()
        main.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(100): g.replay()
        e1.record(main); main.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print(f"{ns} stream(s): {ms:.3f} ms/token -> {1e3/ms:.0f} tokens/s", flush=True)
