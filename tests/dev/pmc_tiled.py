"""dev (run under rocprofv3 --pmc ...): a few launches of the prefill kernel.  argv: M K N (default 8192 4096 4096) [variant] [split]; env PMC_DTYPE=bf16: bf16 activations + bf16 scales."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from gptqmodel_amd import _lib  # noqa: E402
if os.environ.get("GPTQHIP_LIB"):      # dev A/B builds (tests/dev/ablate/*.so)
    _lib.LIB_PATH = os.environ["GPTQHIP_LIB"]
from gptqmodel_amd import ops  # noqa: E402

M, K, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 4096, 4096)
dev = "cuda"
qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device=dev)
qz = torch.randint(-2**31, 2**31 - 1, (K // 128, N // 8), dtype=torch.int32, device=dev)
DT = torch.bfloat16 if os.environ.get("PMC_DTYPE") == "bf16" else torch.float16
sc = (torch.rand((K // 128, N), device=dev) * 0.01 + 0.005).to(DT)
qw_t, meta = ops.repack_tiled(qw, qz, sc, None, 128, 4)
x = (torch.randn(M, K, device=dev) * 0.5).to(DT)
out = torch.empty((M, N), dtype=DT, device=dev)
# optional: argv[4] = forced variant (1 / 2 / 3 = 256 / 128 / 64 rows, 32..112, 1000 + rows = 128-column blocks), argv[5] = forced split-K
ops.set_tuning(int(sys.argv[5]) if len(sys.argv) > 5 else 0, 2, int(sys.argv[4]) if len(sys.argv) > 4 else 0)
print("PLAN", ops.plan_describe(M, K, N, 128), flush=True)
for _ in range(6):
    ops.gemm(x, qw_t, meta, None, None, N, 128, 4, DT, out=out)
torch.cuda.synchronize()
