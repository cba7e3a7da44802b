"""dev: prototype of a cost-model planner for the prefill kernel at mid M (coefficients fitted to profiles/r03_midm_trace.txt) -- measures the
model's (tile height, split-K) choice against the shipped planner's on one box."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import ops  # noqa: E402

COEF = {64: (-1.071, 7.555, 0.998, 0.114), 128: (0.766, 8.78, 1.318, 0.402), 256: (9.314, 4.226, 1.833, 1.257)}
P0 = 0.75


def model(M, K, N, bm, s):
    chunks = -(-K // 128)
    cps = -(-chunks // s)
    s_eff = -(-chunks // cps)
    tiles = -(-N // 256) * -(-M // bm)
    B = tiles * s_eff
    if s_eff == 1:
        full, rem = divmod(tiles, 256)
        r = full + ((P0 + (1 - P0) * rem / 256) if rem else 0.0)
        f = 1.0 if full >= 1 else rem / 256
    else:
        r = 1.0 if B <= 256 else B / 256
        f = min(B, 256) / 256
    a, b, c, d = COEF[bm]
    main = a + b * f + r * cps * (c + d * f)
    red = max(4.6, 1.5 + s_eff * M * N * 4 / 1e6 / 6.2) if s_eff > 1 else 0.0
    return main + red


def choose(M, K, N):
    chunks = -(-K // 128)
    best = None
    for bm in (64, 128, 256):
        if bm == 64 and M > 1024:
            continue
        tiles = -(-N // 256) * -(-M // bm)
        for s in range(1, 17):
            if s > 1 and (s > chunks // 4 or tiles * s > 256 or s * M * N > (16 << 20)):
                break
            t = model(M, K, N, bm, s)
            if best is None or t < best[0]:
                best = (t, bm, s)
    return best


dev = "cuda"
MS = [96, 128, 160, 192, 256, 320, 384, 448, 512, 640, 768, 896, 1024, 1280, 1536, 2048]


def t_us(f, it):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


tot_auto = tot_model = 0.0
for (K, N) in [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (8192, 10240), (8192, 8192), (28672, 8192), (8192, 57344), (5120, 5120), (4096, 11008)]:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, 128, 4)
    for M in MS:
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        o = torch.empty((M, N), dtype=torch.float16, device=dev)
        f = lambda: ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, out=o)
        it = 10 if M * N <= (1 << 26) else 4
        ops.set_tuning(0, 2, 0)
        plan = ops.plan_describe(M, K, N, 128).replace("tiled ", "").replace(" gather=0", "").replace(" tail_cols=0", "")
        ta = t_us(f, it)
        pred, bm, s = choose(M, K, N)
        ops.set_tuning(s, 2, {64: 3, 128: 2, 256: 1}[bm])
        tm = t_us(f, it)
        ta2 = 0.0
        ops.set_tuning(0, 2, 0)
        ta2 = t_us(f, it)
        ta = min(ta, ta2)
        tot_auto += ta
        tot_model += tm
        flag = "  model +%.0f %%" % (100 * (ta / tm - 1)) if tm < 0.96 * ta else ("  MODEL WORSE %.0f %%" % (100 * (tm / ta - 1)) if tm > 1.04 * ta else "")
        print(f"K={K:5d} N={N:5d} M={M:5d}: auto {ta:8.1f} ({plan}) | model bm={bm} s={s} pred {pred:7.1f} meas {tm:8.1f}{flag}", flush=True)
print(f"sum auto {tot_auto:.0f} us, sum model {tot_model:.0f} us")
ops.set_tuning(0, 0, 0)
