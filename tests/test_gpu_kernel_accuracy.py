"""Unit parity of HipGptqLinear at EVERY bit width and word layout of the reference's torch kernel, in the manner of the reference's
CPU kernel test (tests/test_torch_kernel_accuracy.py: bits 2..8 x {gptq_p, gptq_v2}; dequantize_weight against the weights rebuilt
from the logical codes, atol 1e-4; forward against x @ W + b, atol 5e-3 / rtol 1e-2, 2e-2 on the larger layer; act-order with a
shuffled g_idx; symmetric zero-point; split-plane and continuous words giving identical results).  The module is packed by its own
pack_block (the device packer); the words must equal what the oracle's packers -- pinned to the reference's pack_block by
tests/golden/ref_pack*.npz -- produce from the same codes."""
import itertools

import numpy as np
import pytest
import torch

from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WIDTHS = (2, 3, 4, 5, 6, 7, 8)
BOTH_LAYOUTS = (2, 3, 4, 8)          # widths that exist as continuous AND as split-plane words
LAYOUT_CASES = [(b, "gptq_p") for b in WIDTHS] + [(b, "gptq_v2") for b in BOTH_LAYOUTS]


def native_format(bits):
    return "gptq_p" if bits in (3, 5, 6, 7) else "gptq_v2"


class Case:
    """One quantised layer: a float Linear, per-(column, group) scales and zero-points, the group of every input row; the logical
    codes the packer must store, and the dequantised [K, N] matrix they stand for (scales at the module's fp16 precision)."""

    def __init__(self, bits, fmt, k=64, n=32, gs=32, shuffled=False, centred_zero=False, seed=0):
        gen = torch.Generator().manual_seed(1009 * seed + 17 * bits + k)
        self.bits, self.fmt, self.k, self.n, self.gs, self.shuffled, self.sym = bits, fmt, k, n, gs, shuffled, centred_zero
        top = (1 << bits) - 1
        self.linear = torch.nn.Linear(k, n, bias=True)
        with torch.no_grad():
            self.linear.weight.copy_(torch.empty(n, k).uniform_(-0.12, 0.12, generator=gen))
            self.linear.bias.copy_(torch.empty(n).uniform_(-0.1, 0.1, generator=gen))
        self.scales = torch.empty(n, k // gs).uniform_(0.005, 0.015, generator=gen)
        self.zeros = (torch.full((n, k // gs), float((top + 1) // 2)) if centred_zero
                      else torch.randint(0, top + 1, (n, k // gs), generator=gen).float())
        rows = torch.randperm(k, generator=gen) if shuffled else torch.arange(k)
        self.g_idx = (rows // gs).to(torch.int32)
        s_row, z_row = self.scales[:, self.g_idx.long()], self.zeros[:, self.g_idx.long()]           # [N, K]
        self.codes = torch.clamp(torch.round((self.linear.weight.data + z_row * s_row) / s_row), 0, top)
        self.weight = ((self.codes - z_row) * s_row.half().float()).T.contiguous()                    # [K, N]
        self.planar = True if (fmt == "gptq_p" and bits == 3) else None

    def module(self):
        from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
        from gptqmodel_amd.utils.const import FORMAT
        m = HipGptqLinear(bits=self.bits, group_size=self.gs, sym=self.sym, desc_act=self.shuffled, in_features=self.k,
                          out_features=self.n, bias=True, format=FORMAT(self.fmt))
        assert bool(m.planar) == (self.bits in (5, 6, 7) or bool(self.planar))
        m.pack_block(self.linear, self.scales.clone(), self.zeros.clone(), self.g_idx.clone())
        assert tuple(m.qweight.shape) == (self.k * self.bits // 32, self.n)
        assert tuple(m.qzeros.shape) == (self.k // self.gs, self.n * self.bits // 32)
        want_w = O.pack_rows_any(self.codes.T.contiguous().to(torch.uint8).numpy(), self.bits, self.planar)
        want_z = O.pack_cols_any(self.zeros.T.contiguous().to(torch.uint8).numpy(), self.bits, self.planar)
        assert np.array_equal(m.qweight.cpu().numpy(), want_w) and np.array_equal(m.qzeros.cpu().numpy(), want_z)
        return m.to(DEV).eval()

    def expect(self, x):
        return x.float().reshape(-1, self.k) @ self.weight + self.linear.bias.data.float()


def close(got, want, atol, rtol):
    return torch.allclose(got.float().cpu(), want, atol=atol, rtol=rtol)


@pytest.mark.parametrize("bits,fmt", LAYOUT_CASES)
def test_dequantize_weight_is_the_logical_codes(bits, fmt):
    case = Case(bits, fmt)
    m = case.module()
    assert close(m.dequantize_weight(), case.weight, 1e-4, 0)          # from the checkpoint words ...
    m.post_init()
    assert close(m.dequantize_weight(), case.weight, 1e-4, 0)          # ... and from the kernel layout


@pytest.mark.parametrize("bits,fmt", LAYOUT_CASES)
def test_forward_keeps_leading_dimensions(bits, fmt):
    case = Case(bits, fmt)
    m = case.module()
    m.post_init()
    x = torch.randn(2, 3, case.k, generator=torch.Generator().manual_seed(bits)).half() * 0.5
    out = m(x.to(DEV))
    assert out.shape == (2, 3, case.n) and out.dtype == torch.float16
    assert close(out.reshape(-1, case.n), case.expect(x), 5e-3, 1e-2)


@pytest.mark.parametrize("bits", BOTH_LAYOUTS)
def test_split_plane_and_continuous_words_are_the_same_layer(bits):
    a, b = Case(bits, "gptq_p"), Case(bits, "gptq_v2")
    ma, mb = a.module(), b.module()
    assert torch.equal(a.codes, b.codes)
    assert torch.equal(ma.qweight, mb.qweight) == (bits != 3)          # only 3-bit words differ between the layouts
    assert torch.equal(ma.dequantize_weight(), mb.dequantize_weight())
    ma.post_init()
    mb.post_init()
    x = (torch.randn(4, a.k, generator=torch.Generator().manual_seed(bits)).half() * 0.5).to(DEV)
    assert torch.equal(ma(x), mb(x))


VARIANTS = {"act_order": dict(shuffled=True, seed=7), "centred_zero": dict(centred_zero=True, seed=100),
            "larger": dict(k=256, n=128, gs=64, seed=42)}


@pytest.mark.parametrize("bits,variant", list(itertools.product(WIDTHS, VARIANTS)))
def test_forward_variants(bits, variant):
    case = Case(bits, native_format(bits), **VARIANTS[variant])
    m = case.module()
    if variant == "centred_zero":
        assert close(m.dequantize_weight(), case.weight, 1e-4, 0)
    m.post_init()
    assert (m.perm is not None) == case.shuffled
    rows = 8 if variant == "larger" else 4
    x = torch.randn(rows, case.k, generator=torch.Generator().manual_seed(bits)).half() * 0.5
    out = m(x.to(DEV))
    assert out.shape == (rows, case.n)
    assert close(out, case.expect(x), 2e-2 if variant == "larger" else 5e-3, 1e-2)
