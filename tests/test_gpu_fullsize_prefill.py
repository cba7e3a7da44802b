"""Parity AT the size bench.py quotes C3 on: M = 65536 (+ a ragged tail) -- batch 32 x 2048 tokens -- through the
prefill kernel's persistent tile loop, for the 4096x4096 act-order layer and the fused gate|up layer (4096 -> 28672), fp16 and bf16,
and for the rmsnorm_gather -> forward_pregathered pair the C3 layer entry times.

The oracle cannot form a 65576 x 28672 product in seconds; every output row depends only on its own input row, so rows are SAMPLED:
rows of the first tiles, of every tile of the LAST round of the persistent loop (those run the steady-state code copy with the
previous tile's stores still in flight), and of the ragged tail (the last row tile holds 40 rows).  The rows that are NOT sampled
are covered by two size-independent properties: the output buffer starts out as NaNs and none may survive (nothing unwritten),
and the column sums of the whole output equal (sum of all input rows) @ W (linearity) within the rounding bound."""
import numpy as np
import pytest
import torch

from helpers import assert_forward_close, f32_to_torch, synth_gptq, torch_to_f32
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
M_FULL = 65536 + 40


@pytest.fixture(scope="module")
def ops():
    from gptqmodel_amd import ops as _ops
    assert _ops.device_info(0)["arch"].startswith("gfx950")
    return _ops


def _rows_to_check(ops, M, K, N, gs, rng):
    """Sampled rows: first / boundary rows, one row inside every tile of the persistent loop's last round, the ragged tail."""
    desc = ops.plan_describe(M, K, N, gs)
    assert desc.startswith("tiled bm=256"), desc
    bm, bn = 256, 256
    nbx, nby = -(-N // bn), -(-M // bm)
    ntiles = nbx * nby
    rows = {0, 1, 255, 256, M // 2, M - 41, M - 40, M - 1}
    if ntiles > 256:   # the tile order of gptqhip_tiled_kernel.h (virtual block v -> XCD-contiguous linear index -> (bm, bn))
        q, r = ntiles >> 3, ntiles & 7
        last_round_first = (ntiles - 1) // 256 * 256
        for v in range(last_round_first, ntiles):
            xcd, idx = v & 7, v >> 3
            lin = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
            tile_row = lin // nbx
            lo, hi = tile_row * bm, min(M, tile_row * bm + bm)
            rows.add(int(rng.randint(lo, hi)))
    rows |= {int(v) for v in rng.randint(0, M, size=6)}
    return np.array(sorted(rows))[:48]


def _device_rows(M, K, act, seed):
    """x [M, K] made ON the device (a host array of this size costs more than the test), values exactly representable in `act`."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn((M, K), device=DEV, generator=g) * 0.5).to(torch.float16 if act == "fp16" else torch.bfloat16)


@pytest.mark.parametrize("act", ["fp16", "bf16"])
@pytest.mark.parametrize("K,N,desc_act", [(4096, 4096, True), (4096, 28672, False)])
def test_prefill_at_benchmark_size_sampled_rows(ops, K, N, desc_act, act):
    gs = 128
    qweight, qzeros, scales, g_idx = synth_gptq(2026, 4, K, N, gs, desc_act=desc_act)
    sc = f32_to_torch(scales, "fp16", DEV)
    perm = torch.from_numpy(O.act_order_perm(g_idx)).to(DEV) if desc_act else None
    qw_t, meta = ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, perm, gs, 4)
    x = _device_rows(M_FULL, K, act, 11)
    # the kernel writes into a buffer POISONED with NaNs: any element it leaves unwritten -- sampled row or not -- stays a NaN
    out = torch.full((M_FULL, N), float("nan"), dtype=x.dtype, device=DEV)
    guard = torch.full((4096,), float("nan"), dtype=x.dtype, device=DEV)       # (allocated right behind: an overrun would land here)
    got = ops.gemm(x, qw_t, meta, None, perm, N, gs, 4, sc.dtype, out=out)
    torch.cuda.synchronize()
    assert got.data_ptr() == out.data_ptr()
    assert not bool(torch.isnan(out).any()), "the prefill kernel left output elements unwritten"
    assert bool(torch.isnan(guard).all())
    rows = _rows_to_check(ops, M_FULL, K, N, gs, np.random.RandomState(5))
    idx = torch.from_numpy(rows).to(DEV)
    ref = O.forward_gptq(torch_to_f32(x[idx]), qweight, qzeros, scales, g_idx, 4, None, act, "fp16")
    assert_forward_close(torch_to_f32(out[idx]), ref, act, tag=(K, N, desc_act, act))
    # ... and the UNSAMPLED rows: the GEMM is linear in x, so the column sums of the whole output must equal (sum of all x rows) @ W.
    # Expected side in fp64 from the dequantised weights the reference's chain multiplies with (torch.py:716-717, then .to(x.dtype),
    # torch.py:330); the only difference left is each output element's final rounding, independent and zero-mean: 8 sigma of their sum
    # (ulp <= |out| * 2^-10 fp16 / 2^-7 bf16, variance ulp^2 / 12) + the fp32 accumulation slack.
    w64 = ops.dequant_tiled(qw_t, meta, perm, K, N, gs, 4, sc.dtype).to(x.dtype).double()
    xs = torch.zeros(K, dtype=torch.float64, device=DEV)
    col = torch.zeros(N, dtype=torch.float64, device=DEV)
    sq = torch.zeros(N, dtype=torch.float64, device=DEV)
    ab = torch.zeros(N, dtype=torch.float64, device=DEV)
    for r0 in range(0, M_FULL, 8192):
        o = out[r0:r0 + 8192].double()
        xs += x[r0:r0 + 8192].double().sum(0)
        col += o.sum(0)
        sq += (o * o).sum(0)
        ab += o.abs().sum(0)
        del o
    want = xs @ w64
    ulp = 2.0 ** (-10 if act == "fp16" else -7)
    tol = 8.0 * torch.sqrt(sq / 12.0) * ulp + 2e-6 * ab
    worst = ((col - want).abs() / tol).max().item()
    assert worst <= 1.0, f"column sums over all {M_FULL} rows off by {worst:.2f} x the rounding bound"


@pytest.mark.parametrize("act", ["fp16", "bf16"])
def test_prefill_layer_path_at_benchmark_size(ops, act):
    """rmsnorm_gather (RMSNorm + the act-order gather in one pass) -> forward_pregathered of the fused q|k|v module at M = 65576:
    the pair bench.py's C3 layer entry times, against the oracle composed with HF's LlamaRMSNorm on sampled rows."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    K, N, gs = 4096, 6144, 128
    qweight, qzeros, scales, g_idx = synth_gptq(77, 4, K, N, gs, desc_act=True)
    lin = HipGptqLinear(bits=4, group_size=gs, sym=False, desc_act=True, in_features=K, out_features=N, bias=False, register_buffers=False)
    lin.qweight, lin.qzeros = torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV)
    lin.scales, lin.g_idx, lin.bias = f32_to_torch(scales, "fp16", DEV), torch.from_numpy(g_idx).to(DEV), None
    lin.qzero_format(format=2)
    lin.eval()
    lin.post_init()
    assert lin.perm is not None
    h = _device_rows(M_FULL, K, act, 12) * 2.0
    w = f32_to_torch(1.0 + 0.1 * np.random.RandomState(3).randn(K).astype(np.float32), act, DEV)
    out = lin.forward_pregathered(ops.rmsnorm_gather(h, w, 1e-5, lin.perm))
    torch.cuda.synchronize()
    rows = _rows_to_check(ops, M_FULL, K, N, gs, np.random.RandomState(6))
    idx = torch.from_numpy(rows).to(DEV)
    hr, wr = torch_to_f32(h[idx]), torch_to_f32(w)
    xn = np.stack([O.rmsnorm_ref(hr[i], wr, 1e-5, act) for i in range(len(rows))])
    ref = O.forward_gptq(xn, qweight, qzeros, scales, g_idx, 4, None, act, "fp16")
    assert_forward_close(torch_to_f32(out[idx]), ref, act, tag=("layer path", act))
