"""HipAwqLinear -- BACKEND.AWQ_HIP: MI355X fused dequant-matmul QuantLinear for AWQ (FORMAT.GEMM) checkpoints.

Drop-in for AwqTorchLinear (gptqmodel/nn_modules/qlinear/torch_awq.py:20) on DEVICE.ROCM.  post_init() relayouts
the N-packed interleaved AWQ tensors once on the device into the K-packed layout the MFMA kernel consumes
(semantics of unpack_reorder_pack, packing_utils.py:90-103 -- what the reference's own ExllamaV2-AWQ path does);
zero-points are used as-is (asymmetric, no +-1: REQUIRES_FORMAT_V2 = False like torch_awq.py:41).
"""
from __future__ import annotations

from typing import Optional

import torch

from ...utils.adapter import Adapter, Lora
from ...utils.backend import BACKEND
from ...utils.const import DEVICE, FORMAT, METHOD, PLATFORM
from . import AWQuantLinear
from .hip_common import flatten_input, hip_validate_once


class HipAwqLinear(AWQuantLinear):
    SUPPORTS_BACKENDS = [BACKEND.AWQ_HIP]
    SUPPORTS_METHODS = [METHOD.AWQ]
    SUPPORTS_FORMATS = {FORMAT.GEMM: 120}
    SUPPORTS_BITS = [4]
    SUPPORTS_GROUP_SIZE = [-1, 32, 64, 128]
    SUPPORTS_DESC_ACT = [True, False]
    SUPPORTS_SYM = [True, False]
    SUPPORTS_SHARDS = True
    SUPPORTS_TRAINING = False
    SUPPORTS_AUTO_PADDING = False
    SUPPORTS_IN_FEATURES_DIVISIBLE_BY = [32]
    SUPPORTS_OUT_FEATURES_DIVISIBLE_BY = [8]
    SUPPORTS_DEVICES = [DEVICE.ROCM]
    SUPPORTS_PLATFORM = [PLATFORM.LINUX]
    SUPPORTS_PACK_DTYPES = [torch.int32]
    SUPPORTS_ADAPTERS = [Lora]
    SUPPORTS_DTYPES = [torch.float16, torch.bfloat16]

    REQUIRES_FORMAT_V2 = False
    QUANT_TYPE = "hip_awq"

    def __init__(self, bits: int, group_size: int, sym: bool, desc_act: bool, in_features: int, out_features: int,
                 bias: bool = False, pack_dtype: torch.dtype = torch.int32, adapter: Adapter = None,
                 register_buffers: bool = True, **kwargs):
        super().__init__(bits=bits, group_size=group_size, sym=sym, desc_act=desc_act, in_features=in_features,
                         out_features=out_features, bias=bias, pack_dtype=pack_dtype,
                         backend=kwargs.pop("backend", BACKEND.AWQ_HIP), adapter=adapter,
                         register_buffers=register_buffers, **kwargs)
        self._ready = False
        self._rt = {}  # dtype -> (scales, bias) cast to the runtime compute dtype

    @classmethod
    def validate_once(cls):
        return hip_validate_once()

    def post_init(self):
        if self.scales is not None and self.scales.dtype not in (torch.float16, torch.bfloat16):
            self.scales = self.scales.to(torch.float16)  # torch_awq.py:81-86
        if self.bias is not None and self.bias.dtype not in (torch.float16, torch.bfloat16):
            self.bias = self.bias.to(torch.float16)
        super().post_init()
        from gptqmodel_amd import ops
        if not self.qweight.is_cuda:
            raise RuntimeError("HipAwqLinear.post_init: buffers must be on the ROCm device (no CPU fallback)")
        if self.qweight.shape != (self.in_features, self.out_features // 8):
            raise RuntimeError(f"unexpected AWQ qweight shape {tuple(self.qweight.shape)}")
        qw, qz = ops.repack_awq(self.qweight.data, self.qzeros.data)
        self.qweight.data = qw   # now [K/8, N], K-packed sequential
        self.qzeros.data = qz    # now sequential nibble order
        self.scales.data = self.scales.data.contiguous()
        self._ready = True

    def _runtime(self, dtype: torch.dtype):
        """AwqTorchLinear._ensure_runtime_dtype (torch_awq.py:149-155): scales and bias are cast to the compute
        dtype BEFORE the dequant multiply."""
        hit = self._rt.get(dtype)
        if hit is None or hit[0].device != self.scales.device:
            sc = self.scales if self.scales.dtype == dtype else self.scales.to(dtype).contiguous()
            b = None
            if self.bias is not None:
                b = self.bias if self.bias.dtype == dtype else self.bias.to(dtype).contiguous()
            hit = (sc, b)
            self._rt = {dtype: hit}
        return hit

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self._ready:
            raise RuntimeError("HipAwqLinear.forward called before post_init()")
        from gptqmodel_amd import ops
        out_shape = x.shape[:-1] + (self.out_features,)
        x2, in_dtype = flatten_input(x, self.in_features)
        scales, bias = self._runtime(x2.dtype)
        out = ops.gemm(x2, self.qweight, self.qzeros, scales, bias, None, self.group_size, self.bits)
        if self.adapter:
            out = self.adapter.apply(x=x2, out=out)
        if out.dtype != in_dtype:
            out = out.to(in_dtype)
        return out.reshape(out_shape)

    def dequantize_weight(self) -> torch.Tensor:
        from gptqmodel_amd import ops
        return ops.dequant(self.qweight, self.qzeros, self.scales, None, self.group_size, self.bits)


__all__ = ["HipAwqLinear"]
