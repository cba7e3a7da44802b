"""Parity over the reference's own TFLOPS benchmark grid (scripts/benchmark_marlin_a100.py:35-44: M = 64..192 on 4096x11008, 11008x4096,
4096x4096, int4 g128) through gptqhip_gemm's OWN plan -- since round 5 the prefill kernel's tile height moves in steps of 16 rows
(32 .. 128), so every batch size of the grid lands on a different (tile height, split-K) pair.  M in {65, 72, 80, 96, 128, 136, 160, 192,
256} x the three shapes x {fp16, bf16} x {plain, act-order}, bias on every second case, against the oracle's dequantised weights
(bit-pinned to the reference, tests/test_oracle_golden.py) and its matmul + rounding chain."""
import numpy as np
import pytest
import torch

from helpers import assert_forward_close, f32_to_torch, synth_gptq, torch_to_f32
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MS = (65, 72, 80, 96, 128, 136, 160, 192, 256)
SHAPES = [(4096, 11008), (11008, 4096), (4096, 4096)]


@pytest.fixture(scope="module")
def ops():
    from gptqmodel_amd import ops as _ops
    assert _ops.device_info(0)["arch"].startswith("gfx950")
    return _ops


@pytest.mark.parametrize("desc_act", [False, True])
@pytest.mark.parametrize("K,N", SHAPES)
def test_reference_benchmark_grid_parity(ops, K, N, desc_act):
    gs = 128
    qweight, qzeros, scales, g_idx = synth_gptq(5000 + K // 128 + N // 256 + int(desc_act), 4, K, N, gs, desc_act=desc_act)
    w16 = O.dequant_gptq(qweight, qzeros, scales, g_idx, 4, "fp16")           # [K, N] fp32 values, each exactly an fp16
    w_for = {"fp16": w16, "bf16": O.round_to(w16, "bf16")}                     # torch.py:331-335: weights.to(x.dtype)
    sc = f32_to_torch(scales, "fp16", DEV)
    perm = torch.from_numpy(O.act_order_perm(g_idx)).to(DEV) if desc_act else None
    qw_t, meta = ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, perm, gs, 4)
    rng = np.random.RandomState(K + N)
    heights = set()
    for i, M in enumerate(MS):
        plan = ops.plan_describe(M, K, N, gs, 4, desc_act)
        assert plan.startswith("tiled bm="), (M, plan)
        heights.add(int(plan.split("bm=")[1].split(" ")[0]))
        for act in ("fp16", "bf16"):
            x = O.round_to(rng.randn(M, K).astype(np.float32) * 0.5, act)
            bias = O.round_to(rng.randn(N).astype(np.float32) * 0.1, act) if (i + (act == "bf16")) % 2 else None
            out = ops.gemm(f32_to_torch(x, act, DEV), qw_t, meta, None if bias is None else f32_to_torch(bias, act, DEV), perm, N, gs, 4,
                           sc.dtype)
            torch.cuda.synchronize()
            ref = O.matmul_round(x, w_for[act], bias, act)
            assert_forward_close(torch_to_f32(out), ref, act, tag=(K, N, M, act, desc_act, plan))
    assert len(heights) >= 3, heights       # the grid really exercises several tile heights
