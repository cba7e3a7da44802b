"""dev: time the prefill kernel of the library named by GPTQHIP_LIB (timing ablation builds: WRONG results by construction) on a
few shapes; eager back-to-back launches, us per call."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import _lib  # noqa: E402

if os.environ.get("GPTQHIP_LIB"):
    _lib.LIB_PATH = os.environ["GPTQHIP_LIB"]
from gptqmodel_amd import ops  # noqa: E402

dev = "cuda"
CASES = [(8192, 4096, 4096, 0, 0), (2048, 4096, 4096, 0, 0), (128, 4096, 28672, 0, 0), (128, 4096, 4096, 3, 1), (128, 4096, 4096, 0, 0),
         (512, 4096, 28672, 0, 0), (65536, 4096, 4096, 0, 0)]
if os.environ.get("ABLATE_CASES") == "midm2":     # 128-row tiles with and without split-K
    CASES = [(M, K, N, 0, 0) for (K, N) in ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)) for M in (384, 512, 1024, 2048)]
if os.environ.get("ABLATE_CASES") == "midm":      # planner's choice at serving-batch sizes (same-box A/B of two builds: r3_call_ab.sh)
    CASES = [(M, K, N, 0, 0) for (K, N) in ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)) for M in (96, 128, 192, 256)]
out = [os.path.basename(os.environ.get("GPTQHIP_LIB", "shipped"))[:28].ljust(28)]
for (M, K, N, variant, split) in CASES:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, 128, 4)
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    o = torch.empty((M, N), dtype=torch.float16, device=dev)
    ops.set_tuning(split, 2, variant)
    f = lambda: ops.gemm(x, qw_t, meta, None, None, N, 128, 4, torch.float16, out=o)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    it = 20 if M <= 8192 else 5
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        f()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / it * 1e3
    out.append(f"M{M}x{N}{'/bm64s1' if variant else ''}: {us:8.1f} us {2.0 * M * K * N / us / 1e6:6.0f} TF")
    del qw, qz, sc, qw_t, meta, x, o
print(" | ".join(out), flush=True)
