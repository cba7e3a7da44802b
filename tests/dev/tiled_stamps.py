"""dev: per-block phase stamps of the prefill kernel's split-K path (library built from tests/dev/tiled_stamps.patch: the bias pointer is
abused as the stamp buffer; outputs are garbage).  Prints, per configuration, the distribution over blocks of: start offset (ramp), prologue
issue, first-chunk wait, main loop, final barrier, epilogue issue, store drain -- microseconds (100 MHz wall clock: 10 ns steps)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.environ["GPTQHIP_LIB"]
from gptqmodel_amd import ops  # noqa: E402

dev = "cuda"
for (M, K, N, variant, split) in [(128, 4096, 4096, 0, 0), (128, 4096, 4096, 3, 4), (256, 4096, 4096, 0, 0), (128, 14336, 4096, 0, 0), (128, 4096, 6144, 0, 0)]:
    qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
    qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
    sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).half()
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, 128, 4)
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    o = torch.empty((M, N), dtype=torch.float16, device=dev)
    dbg = torch.zeros(1 << 16, dtype=torch.float16, device=dev)     # 128 KiB: up to 2048 blocks x 8 stamps
    ops.set_tuning(split, 2, variant)
    plan = ops.plan_describe(M, K, N, 128)
    for _ in range(4):
        dbg.zero_()
        torch.cuda.synchronize()
        ops.gemm(x, qw_t, meta, dbg, None, N, 128, 4, torch.float16, out=o)
        torch.cuda.synchronize()
    st = dbg.view(torch.int64).cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    t0 = st[:, 0].min()
    us = (st[:, :7] - t0) / 100.0
    names = ["start", "prologue issued", "chunk 0 landed", "main loop done", "barrier", "stores issued", "stores drained"]
    print(f"== M={M} K={K} N={N} {plan}: {len(st)} blocks, kernel span {us[:, 6].max():.1f} us")
    prev = np.zeros(len(st))
    for i, n in enumerate(names):
        col = us[:, i]
        d = col - prev if i else col
        print(f"   {n:16s} at {np.median(col):6.2f} (min {col.min():6.2f} max {col.max():6.2f}) | phase median {np.median(d):5.2f} max {d.max():5.2f}")
        prev = col
    xcd = (st[:, 7] >> 20) & 0xf if False else None
ops.set_tuning(0, 0, 0)
