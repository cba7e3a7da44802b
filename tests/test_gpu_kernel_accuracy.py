"""The reference's CPU kernel unit parity test (tests/test_torch_kernel_accuracy.py) restated for HipGptqLinear: same cases
(bits 2..8 x {gptq_p, gptq_v2}, batched shapes, planar == continuous, desc_act with a shuffled g_idx, symmetric zero-point, larger
shapes), same input recipe (:46-58), same logical-code reference (:77-87) and the same tolerances (:104-125, :229).  Like the
reference's test the module is packed by its own pack_block (the device packer, every bit width and layout); the oracle's packers
(pinned to the reference by tests/golden/ref_pack*.npz) must produce the same words."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ALL_BITS = (2, 3, 4, 5, 6, 7, 8)
DUAL_LAYOUT_BITS = (2, 3, 4, 8)


def _format_cases():
    return [(b, "gptq_p") for b in ALL_BITS] + [(b, "gptq_v2") for b in DUAL_LAYOUT_BITS]


def _make_inputs(bits, in_features, out_features, group_size, desc_act=False, seed=0):
    torch.manual_seed(seed + bits)
    maxq = (1 << bits) - 1
    groups = in_features // group_size
    linear = nn.Linear(in_features, out_features, bias=True)
    scales = torch.rand(out_features, groups) * 0.01 + 0.005
    zeros = torch.randint(0, maxq + 1, (out_features, groups)).float()
    if desc_act:
        g_idx = (torch.randperm(in_features) // group_size).to(torch.int32)
    else:
        g_idx = torch.arange(in_features, dtype=torch.int32) // group_size
    return linear, scales, zeros, g_idx


def _codes(linear, scales, zeros, g_idx, bits):
    maxq = (1 << bits) - 1
    scale_full, zero_full = scales[:, g_idx.long()], zeros[:, g_idx.long()]
    return torch.round((linear.weight.data + zero_full * scale_full) / scale_full).clamp(0, maxq), scale_full, zero_full


def _reference_weight(linear, scales, zeros, g_idx, bits):
    codes, scale_full, zero_full = _codes(linear, scales, zeros, g_idx, bits)
    return ((codes - zero_full) * scale_full.to(torch.float16).float()).T.contiguous()      # [in, out]


def _module(bits, fmt, linear, scales, zeros, g_idx, group_size, desc_act=False, sym=False):
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.const import FORMAT
    k, n = linear.in_features, linear.out_features
    m = HipGptqLinear(bits=bits, group_size=group_size, sym=sym, desc_act=desc_act, in_features=k, out_features=n, bias=True,
                      format=FORMAT(fmt))
    planar = True if (fmt == "gptq_p" and bits == 3) else None
    assert bool(m.planar) == (bits in (5, 6, 7) or bool(planar))
    m.pack_block(linear, scales.clone(), zeros.clone(), g_idx.clone())
    codes, _, _ = _codes(linear, scales, zeros, g_idx, bits)
    assert np.array_equal(m.qweight.cpu().numpy(), O.pack_rows_any(codes.T.contiguous().to(torch.uint8).numpy(), bits, planar))
    assert np.array_equal(m.qzeros.cpu().numpy(), O.pack_cols_any(zeros.T.contiguous().to(torch.uint8).numpy(), bits, planar))
    assert tuple(m.qweight.shape) == (k * bits // 32, n) and tuple(m.qzeros.shape) == (k // group_size, n * bits // 32)
    return m.to(DEV).eval()


def _packed_module_and_reference(bits, fmt, in_features=64, out_features=32, group_size=32, desc_act=False, seed=0):
    linear, scales, zeros, g_idx = _make_inputs(bits, in_features, out_features, group_size, desc_act=desc_act, seed=seed)
    return _module(bits, fmt, linear, scales, zeros, g_idx, group_size, desc_act=desc_act), _reference_weight(linear, scales, zeros, g_idx, bits), linear


@pytest.mark.parametrize("bits,fmt", _format_cases())
def test_dequantize_weight_matches_reference(bits, fmt):
    module, ref, _ = _packed_module_and_reference(bits, fmt)
    for _ in range(2):                       # checkpoint layout, then the kernel layout
        dequant = module.dequantize_weight().float().cpu()
        assert dequant.shape == ref.shape
        assert torch.allclose(dequant, ref, atol=1e-4, rtol=0)
        module.post_init()


@pytest.mark.parametrize("bits,fmt", _format_cases())
def test_forward_batched_shapes(bits, fmt):
    module, ref, linear = _packed_module_and_reference(bits, fmt)
    module.post_init()
    torch.manual_seed(bits)
    x = torch.randn(2, 3, 64, dtype=torch.float16) * 0.5
    out = module(x.to(DEV)).cpu()
    ref_out = x.float().reshape(-1, 64) @ ref + linear.bias.data.float()
    assert out.shape == (2, 3, 32)
    assert torch.allclose(out.float().reshape(-1, 32), ref_out, atol=5e-3, rtol=1e-2)


@pytest.mark.parametrize("bits", DUAL_LAYOUT_BITS)
def test_planar_and_continuous_forward_identical(bits):
    linear, scales, zeros, g_idx = _make_inputs(bits, 64, 32, 32)
    m_planar = _module(bits, "gptq_p", linear, scales, zeros, g_idx, 32)
    m_continuous = _module(bits, "gptq_v2", linear, scales, zeros, g_idx, 32)
    assert torch.equal(m_planar.dequantize_weight(), m_continuous.dequantize_weight())
    if bits == 3:
        assert not torch.equal(m_planar.qweight, m_continuous.qweight)      # different words, same values
    m_planar.post_init()
    m_continuous.post_init()
    torch.manual_seed(bits)
    x = (torch.randn(4, 64, dtype=torch.float16) * 0.5).to(DEV)
    assert torch.equal(m_planar(x), m_continuous(x))


@pytest.mark.parametrize("bits", ALL_BITS)
def test_forward_desc_act_shuffled_g_idx(bits):
    fmt = "gptq_p" if bits in (3, 5, 6, 7) else "gptq_v2"
    module, ref, linear = _packed_module_and_reference(bits, fmt, desc_act=True, seed=7)
    module.post_init()
    assert module.perm is not None
    torch.manual_seed(bits)
    x = torch.randn(4, 64, dtype=torch.float16) * 0.5
    out = module(x.to(DEV)).cpu()
    ref_out = x.float() @ ref + linear.bias.data.float()
    assert torch.allclose(out.float(), ref_out, atol=5e-3, rtol=1e-2)


@pytest.mark.parametrize("bits", ALL_BITS)
def test_forward_sym_zero_point(bits):
    fmt = "gptq_p" if bits in (3, 5, 6, 7) else "gptq_v2"
    in_features, out_features, group_size = 64, 32, 32
    maxq = (1 << bits) - 1
    torch.manual_seed(100 + bits)
    linear = nn.Linear(in_features, out_features, bias=True)
    groups = in_features // group_size
    scales = torch.rand(out_features, groups) * 0.01 + 0.005
    zeros = torch.full((out_features, groups), float((maxq + 1) // 2))
    g_idx = torch.arange(in_features, dtype=torch.int32) // group_size
    module = _module(bits, fmt, linear, scales, zeros, g_idx, group_size, sym=True)
    ref = _reference_weight(linear, scales, zeros, g_idx, bits)
    assert torch.allclose(module.dequantize_weight().float().cpu(), ref, atol=1e-4, rtol=0)
    module.post_init()
    x = torch.randn(4, in_features, dtype=torch.float16) * 0.5
    out = module(x.to(DEV)).cpu()
    ref_out = x.float() @ ref + linear.bias.data.float()
    assert torch.allclose(out.float(), ref_out, atol=5e-3, rtol=1e-2)


@pytest.mark.parametrize("bits", ALL_BITS)
def test_forward_larger_shapes(bits):
    fmt = "gptq_p" if bits in (3, 5, 6, 7) else "gptq_v2"
    in_features, out_features, group_size = 256, 128, 64
    module, ref, linear = _packed_module_and_reference(bits, fmt, in_features=in_features, out_features=out_features,
                                                       group_size=group_size, seed=42)
    module.post_init()
    torch.manual_seed(bits)
    x = torch.randn(8, in_features, dtype=torch.float16) * 0.5
    out = module(x.to(DEV)).cpu()
    ref_out = x.float() @ ref + linear.bias.data.float()
    assert out.shape == (8, out_features)
    assert torch.allclose(out.float(), ref_out, atol=2e-2, rtol=1e-2)
