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
    EXACT_BF16_DECODE = False  # opt-in, see HipGptqLinear.EXACT_BF16_DECODE

    def __init__(self, bits: int, group_size: int, sym: bool, desc_act: bool, in_features: int, out_features: int,
                 bias: bool = False, pack_dtype: torch.dtype = torch.int32, adapter: Adapter = None,
                 register_buffers: bool = True, **kwargs):
        super().__init__(bits=bits, group_size=group_size, sym=sym, desc_act=desc_act, in_features=in_features,
                         out_features=out_features, bias=bias, pack_dtype=pack_dtype,
                         backend=kwargs.pop("backend", BACKEND.AWQ_HIP), adapter=adapter,
                         register_buffers=register_buffers, **kwargs)
        self._ready = False
        self._rt = {}  # compute dtype -> (meta, bias): dequant constants with scales cast to that dtype

    @classmethod
    def validate_once(cls):
        return hip_validate_once()

    def post_init(self):
        if self.scales is not None and self.scales.dtype not in (torch.float16, torch.bfloat16):
            self.scales = self.scales.to(torch.float16)  # torch_awq.py:81-86
        if self.bias is not None and self.bias.dtype not in (torch.float16, torch.bfloat16):
            self.bias = self.bias.to(torch.float16)
        super().post_init()
        if self._ready:
            return
        from gptqmodel_amd import ops
        if not self.qweight.is_cuda:
            raise RuntimeError("HipAwqLinear.post_init: buffers must be on the ROCm device (no CPU fallback)")
        if self.qweight.shape != (self.in_features, self.out_features // 8):
            raise RuntimeError(f"unexpected AWQ qweight shape {tuple(self.qweight.shape)}")
        qw, qz = ops.repack_awq(self.qweight.data, self.qzeros.data)  # AWQ -> K-packed sequential
        self.scales.data = self.scales.data.contiguous()
        qw_t, meta = ops.repack_tiled(qw, qz, self.scales.data, None, self.group_size, self.bits)
        self.qweight.data = qw_t  # tiled words; the AWQ-layout copy is released
        self.qzeros.data = qz     # sequential nibble order (kept: meta is rebuilt from it on a dtype change)
        self._rt = {self.scales.dtype: (meta, self.bias)}
        self._ready = True

    def _runtime(self, dtype: torch.dtype):
        """AwqTorchLinear._ensure_runtime_dtype (torch_awq.py:149-155): scales and bias are cast to the compute
        dtype BEFORE the dequant multiply."""
        hit = self._rt.get(dtype)
        if hit is not None and hit[0].device != self.qweight.device:
            hit = None  # the module was moved (.to(device)) after post_init: rebuild the constants next to the weights
        if hit is None:
            from gptqmodel_amd import ops
            sc = self.scales if self.scales.dtype == dtype else self.scales.to(dtype).contiguous()
            b = None
            if self.bias is not None:
                b = self.bias if self.bias.dtype == dtype else self.bias.to(dtype).contiguous()
            _, meta = ops.repack_tiled(None, self.qzeros, sc, None, self.group_size, self.bits)  # meta only
            hit = (meta, b)
            self._rt = {dtype: hit}  # one compute dtype at a time (a model runs in one dtype)
        return hit

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self._ready:
            raise RuntimeError("HipAwqLinear.forward called before post_init()")
        from gptqmodel_amd import ops
        out_shape = x.shape[:-1] + (self.out_features,)
        x2, in_dtype = flatten_input(self._apply_rotation_to_input(x), self.in_features)
        meta, bias = self._runtime(x2.dtype)
        out = ops.gemm(x2, self.qweight, meta, bias, None, self.out_features, self.group_size, self.bits, x2.dtype,
                       exact_bf16=self.EXACT_BF16_DECODE)
        if self.adapter:
            out = self.adapter.apply(x=x2, out=out)
        if out.dtype != in_dtype:
            out = out.to(in_dtype)
        return out.reshape(out_shape)

    def forward_partial(self, x: torch.Tensor) -> torch.Tensor:
        """float32 [.., N] unrounded accumulators without bias (row-parallel tensor-parallel shards all-reduce these
        before the single final rounding, gptqmodel_amd/utils/tp.py) -- same contract as HipGptqLinear.forward_partial."""
        if not self._ready:
            raise RuntimeError("HipAwqLinear.forward_partial called before post_init()")
        from gptqmodel_amd import ops
        x2, _ = flatten_input(x, self.in_features)
        meta, _ = self._runtime(x2.dtype)
        out = ops.gemm(x2, self.qweight, meta, None, None, self.out_features, self.group_size, self.bits, x2.dtype,
                       partial_f32=True)
        return out.reshape(x.shape[:-1] + (self.out_features,))

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        if self._ready:
            raise RuntimeError(f"{self.__class__.__name__} `{self.name}`: state_dict() after post_init() would save the "
                               "kernel (tile-major) layout; save from the original checkpoint instead")
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def dequantize_weight(self) -> torch.Tensor:
        from gptqmodel_amd import ops
        meta, _ = self._runtime(self.scales.dtype)
        return ops.dequant_tiled(self.qweight, meta, None, self.in_features, self.out_features, self.group_size,
                                 self.bits, self.scales.dtype)


__all__ = ["HipAwqLinear"]
