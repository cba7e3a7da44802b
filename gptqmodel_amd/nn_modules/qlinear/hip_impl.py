"""ONE implementation of the two HIP QuantLinear plugin classes, parameterised by the base classes / enums they are built
on.  `make_hip_classes(ns)` is called twice:

  * by gptqmodel_amd/nn_modules/qlinear/hip_gptq.py / hip_awq.py with THIS package's mirror of the plugin contract
    (gptqmodel_amd/nn_modules/qlinear/__init__.py) -- what the repo's tests, bench and examples run, and
  * by integration/gptqmodel_overlay/nn_modules/qlinear/hip.py with the REFERENCE's own GPTQQuantLinear / AWQuantLinear /
    BACKEND / DEVICE / FORMAT / METHOD (gptqmodel/nn_modules/qlinear/__init__.py:727,1634; utils/backend.py:10;
    models/_const.py:41) -- the file a GPTQModel maintainer drops into the upstream tree, where the reference's unmodified
    discovery (utils/importer.py:110-127,169-179) finds the classes and `GPTQModel.load()` becomes a drop-in.

Everything class-specific is here once: SUPPORTS_* declarations, post_init() relayout, forward(), dequantize_weight(),
pack(), forward_partial().  `ns` must provide: GPTQQuantLinear, AWQuantLinear, BACKEND (with GPTQ_HIP / AWQ_HIP members),
DEVICE, FORMAT, METHOD, PLATFORM, Lora.
"""
from __future__ import annotations

from typing import Optional

import torch

from .hip_common import act_order_permutation, flatten_input, hip_validate_once


def _fmt_value(fmt) -> Optional[str]:
    return None if fmt is None else str(getattr(fmt, "value", fmt)).lower()


def _convert_v1_to_v2_inplace(module) -> None:
    """The loader's v1 -> v2 qzeros conversion for one module (reference utils/model.py:814-831): checkpoints in
    FORMAT.GPTQ store zero-1; add 1 to every packed field with int32 wraparound."""
    from gptqmodel_amd.utils.model import shift_v1_qzeros
    module.qzeros.data = shift_v1_qzeros(module.qzeros.data, module.bits, planar=bool(getattr(module, "planar", False)))
    module.qzero_format(format=2)


def make_hip_classes(ns, module_name: str):
    GPTQQuantLinear, AWQuantLinear = ns.GPTQQuantLinear, ns.AWQuantLinear
    BACKEND, DEVICE, FORMAT, METHOD, PLATFORM, Lora = ns.BACKEND, ns.DEVICE, ns.FORMAT, ns.METHOD, ns.PLATFORM, ns.Lora

    class HipGptqLinear(GPTQQuantLinear):
        """BACKEND.GPTQ_HIP: the MI355X (gfx950) fused dequant-matmul QuantLinear for GPTQ checkpoints.  Drop-in for the
        reference's TorchLinear (gptqmodel/nn_modules/qlinear/torch.py:114) on DEVICE.ROCM: same constructor, buffers,
        post_init()/forward()/dequantize_weight() semantics and rounding, but forward() is ONE HIP kernel
        (libgptqhip.so: gptqhip_gemm) instead of ~10 elementwise torch kernels + a dense GEMM.  Selection priority 120
        beats every kernel upstream lists for ROCm."""
        SUPPORTS_BACKENDS = [BACKEND.GPTQ_HIP]
        SUPPORTS_METHODS = [METHOD.GPTQ]
        SUPPORTS_FORMATS = ({FORMAT.GPTQ: 120, FORMAT.GPTQ_V2: 120, FORMAT.GPTQ_P: 120} if hasattr(FORMAT, "GPTQ_P")
                            else {FORMAT.GPTQ: 120, FORMAT.GPTQ_V2: 120})
        # 4 and 8 bits are the kernels' native field widths; 2 / 3 bits are widened to 4-bit fields and 5 / 6 / 7 (planar) to 8-bit fields
        # at post_init (gptqhip_widen_codes: same codes, same zero-points, same results -- the generic dequantize_weight of the
        # reference, qlinear/__init__.py:947-999), at the price of the wider copy's HBM bytes
        SUPPORTS_BITS = [2, 3, 4, 5, 6, 7, 8]
        SUPPORTS_GROUP_SIZE = [-1, 32, 64, 128, 256, 512, 1024]
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

        REQUIRES_FORMAT_V2 = True  # the loader converts v1 qzeros (+0x11111111) first: utils/model.py:750-844
        QUANT_TYPE = "hip_gptq"
        # opt-in: bf16 batch<=4 decode accumulates exact products instead of rounding every weight to bf16 first
        # (GPTQHIP_GEMM_EXACT_BF16, include/gptqhip.h): ~20 % faster, up to 2 output ulps from the reference's chain
        EXACT_BF16_DECODE = False

        def __init__(self, bits: int, group_size: int, sym: bool, desc_act: bool, in_features: int, out_features: int,
                     bias: bool = False, pack_dtype: torch.dtype = torch.int32, adapter=None,
                     register_buffers: bool = True, format=None, **kwargs):
            super().__init__(bits=bits, group_size=group_size, sym=sym, desc_act=desc_act, in_features=in_features,
                             out_features=out_features, bias=bias, pack_dtype=pack_dtype,
                             backend=kwargs.pop("backend", BACKEND.GPTQ_HIP), adapter=adapter,
                             register_buffers=register_buffers, format=format, **kwargs)
            # derived device tensors, filled by post_init() and promoted to non-persistent BUFFERS there (module.to(device)
            # moves them with the weights, state_dict() never carries them).  Plain None attributes until then: None-valued
            # buffers break accelerate's offload hooks (same reason as set_had_K, qlinear/__init__.py:485-505)
            self.perm = None  # act-order row permutation (int32 [K])
            self.meta = None  # pre-baked per-(group, column) constants
            self._scale_dtype = torch.float16
            self._ready = False
            self._bias_cache = None
            # field width of the layout the kernels read (== bits for 4 / 8); 5 / 6 / 7 bits only exist in the planar layout, 3 bits
            # are planar under FORMAT.GPTQ_P only (qlinear/__init__.py:766-773; the reference's base class sets `planar` itself)
            self.kernel_bits = 4 if bits <= 4 else 8
            self.source_bits = bits        # differs from `bits` after widen_in_place()
            if not hasattr(self, "planar"):
                self.planar = bits in (5, 6, 7) or (_fmt_value(format) == "gptq_p" and bits == 3)
            if (bits not in (4, 8) or self.planar) and (in_features % 32 != 0 or out_features % 32 != 0):
                # 3 / 5 / 6 / 7 bits and the planar layouts store whole 32-code blocks (qlinear/__init__.py:780-786); 2-bit rows of
                # qzeros hold 16 columns per word and gptqhip_widen_codes works on 32-column groups.  NotImplementedError keeps the
                # selection loop going (it means "this kernel cannot serve the shape")
                raise NotImplementedError(f"{bits}-bit (planar: {self.planar}) packing needs in / out features divisible by 32: "
                                          f"got {in_features} x {out_features}")

        @classmethod
        def validate_once(cls):
            return hip_validate_once()

        def widen_in_place(self):
            """Turn a 2 / 3 / 5 / 6 / 7-bit module (checkpoint layout, device tensors) into the 4- / 8-bit module that holds the SAME
            code values: qweight / qzeros re-encoded by gptqhip_widen_codes, `bits` becomes the kernel's field width, the original
            width stays in `source_bits`.  What post_init() does on the fly; done up front by the layer fusion (utils/hf_llama.py),
            whose sibling concatenation / gate-up interleaving / act-order folding work on whole 4- / 8-bit words.  Such a module is no
            longer a checkpoint of its original format: save_quantized_checkpoint refuses it."""
            from gptqmodel_amd import ops
            if self.bits in (4, 8) and not self.planar:
                return self
            if self._ready:
                raise RuntimeError("widen_in_place must be called before post_init()")
            if not self.qweight.is_cuda:    # (NotImplementedError: the layer fusion skips the layer, post_init widens on the device later)
                raise NotImplementedError("HipGptqLinear.widen_in_place: buffers must be on the ROCm device (no CPU fallback)")
            if _fmt_value(getattr(self, "format", None)) == "gptq" and self.qzero_format() == 1:
                _convert_v1_to_v2_inplace(self)      # zero - 1 wraps modulo 2^bits: it must be undone at the ORIGINAL width
            qw, qz, wide = ops.widen_codes(self.qweight.data, self.qzeros.data, self.bits, planar=bool(self.planar))
            self.qweight.data, self.qzeros.data = qw, qz
            self.source_bits, self.bits, self.kernel_bits, self.planar = self.bits, wide, wide, False
            self.pack_factor, self.maxq = 32 // wide, 2 ** wide - 1
            return self

        def post_init(self):
            """One-time device-side relayout into the MFMA-tile-major kernel layout (+ act-order row sort).  The
            reference's fast kernels repack here too (marlin.py:246-293, exllamav2.py:114-140); saving a loaded model
            re-reads the checkpoint from disk (models/writer.py:681-685), so replacing `qweight` in place is safe."""
            super().post_init()
            from gptqmodel_amd import ops
            if self._ready:
                return  # already in the kernel layout (idempotent: HF/optimum and the loader may both call it)
            if not self.qweight.is_cuda:
                raise RuntimeError("HipGptqLinear.post_init: buffers must be on the ROCm device (no CPU fallback)")
            if self.scales.dtype not in (torch.float16, torch.bfloat16):
                self.scales.data = self.scales.data.to(torch.float16)
            if _fmt_value(getattr(self, "format", None)) == "gptq" and self.qzero_format() == 1:
                # a v1 checkpoint (zero-1 on disk) that reached post_init unconverted: the reference loader converts
                # before post_init whenever a loaded module has REQUIRES_FORMAT_V2 (models/loader.py:1658-1675); a caller
                # that goes make_quant -> load_state_dict -> gptqmodel_post_init directly gets the same result here
                # instead of zeros that are silently off by one
                _convert_v1_to_v2_inplace(self)
            groups = self.scales.shape[0]
            perm = None
            if self.g_idx is not None and self.g_idx.numel() == self.in_features:
                perm = act_order_permutation(self.g_idx, self.group_size, groups)
            elif self.g_idx is not None and self.g_idx.numel() not in (0, self.in_features):
                raise NotImplementedError("stacked g_idx (num_itr > 1, torch.py:327) is not supported by the HIP kernel")
            qw, qz = self.qweight.data, self.qzeros.data
            if self.bits not in (4, 8):
                # 2 / 3 / 5 / 6 / 7-bit checkpoints run on the 4- / 8-bit kernels: the codes are widened once, here.  Exact, but the
                # RESIDENT weight bytes (and the decode HBM traffic) become those of a 4- / 8-bit model: 2-bit x2, 3-bit x1.33,
                # 5 / 6 / 7-bit x1.6 / x1.33 / x1.14.  The narrow words are released before the tile-major copy is made.
                qw, qz, wide = ops.widen_codes(qw, qz, self.bits, planar=bool(self.planar))
                assert wide == self.kernel_bits
                self.qweight.data = qw      # (qzeros keeps its checkpoint width)
            qw_t, meta = ops.repack_tiled(qw, qz, self.scales.data, perm, self.group_size, self.kernel_bits)
            del qw
            self.qweight.data = qw_t  # tiled words; the checkpoint-layout copy is released
            self._set_derived("meta", meta)
            self._set_derived("perm", perm)
            self._scale_dtype = self.scales.dtype
            self._ready = True

        def _set_derived(self, name: str, tensor: Optional[torch.Tensor]) -> None:
            if name in self._buffers:
                if tensor is None:
                    del self._buffers[name]
                    setattr(self, name, None)
                else:
                    self._buffers[name] = tensor
                return
            if tensor is None:
                setattr(self, name, None)
                return
            if hasattr(self, name):
                delattr(self, name)
            self.register_buffer(name, tensor, persistent=False)

        def _save_to_state_dict(self, destination, prefix, keep_vars):
            if self._ready:
                # after post_init `qweight` holds tile-major words under the checkpoint name: writing them out would
                # produce a checkpoint no loader can read.  The reference saves a loaded quantised model by re-reading
                # the checkpoint from disk (models/writer.py:681-685), never from the live modules.
                raise RuntimeError(f"{self.__class__.__name__} `{self.name}`: state_dict() after post_init() would save "
                                   "the kernel (tile-major) layout; save from the original checkpoint instead")
            super()._save_to_state_dict(destination, prefix, keep_vars)

        def _bias_for(self, dtype: torch.dtype, device: torch.device):
            if self.bias is None:
                return None
            c = self._bias_cache
            if c is None or c.dtype != dtype or c.device != device or c.data_ptr() == 0:
                c = self.bias.to(device=device, dtype=dtype).contiguous()  # torch.py:338-342 casts bias to out dtype
                self._bias_cache = c
            return c

        def forward(self, x: torch.Tensor) -> torch.Tensor:
            if not self._ready:
                raise RuntimeError("HipGptqLinear.forward called before post_init()")
            from gptqmodel_amd import ops
            out_shape = x.shape[:-1] + (self.out_features,)
            x2, in_dtype = flatten_input(self._apply_rotation_to_input(x), self.in_features)
            out = ops.gemm(x2, self.qweight, self.meta, self._bias_for(x2.dtype, x2.device), self.perm,
                           self.out_features, self.group_size, self.kernel_bits, self._scale_dtype,
                           exact_bf16=self.EXACT_BF16_DECODE)
            if self.adapter:
                out = self.adapter.apply(x=x2, out=out)  # torch.py:344-345
            if out.dtype != in_dtype:
                out = out.to(in_dtype)
            return out.reshape(out_shape)

        def forward_pregathered(self, x: torch.Tensor) -> torch.Tensor:
            """forward() for an input whose features are ALREADY in this module's kernel row order (x_in[..., self.perm]; any
            x when the module has no act-order permutation): the act-order gather pass of forward() is skipped.  Used by the
            prefill path of utils.hf_llama, where ops.rmsnorm_gather emits the normalised activations in that order."""
            if not self._ready:
                raise RuntimeError("HipGptqLinear.forward_pregathered called before post_init()")
            if self.adapter:
                raise NotImplementedError("forward_pregathered: adapters see the un-permuted input; use forward()")
            from gptqmodel_amd import ops
            out_shape = x.shape[:-1] + (self.out_features,)
            x2, in_dtype = flatten_input(x, self.in_features)
            out = ops.gemm(x2, self.qweight, self.meta, self._bias_for(x2.dtype, x2.device), None, self.out_features,
                           self.group_size, self.kernel_bits, self._scale_dtype, exact_bf16=self.EXACT_BF16_DECODE)
            if out.dtype != in_dtype:
                out = out.to(in_dtype)
            return out.reshape(out_shape)

        def pack_block(self, linear: torch.nn.Module, scales: torch.Tensor, zeros: torch.Tensor, g_idx: torch.Tensor,
                       block_in: int = 8192, workers: int = 1):
            """Quantise-and-pack a float Linear into this module's checkpoint-layout buffers ON THE DEVICE; same
            signature and bit-exact output as PackableQuantLinear.pack_block (qlinear/__init__.py:1036-1323) at
            every bit width and layout of the class (continuous 2 / 3 / 4 / 8, split-plane 3 under gptq_p, planar 5 / 6 / 7):
            scales / zeros arrive as [out, G]."""
            from gptqmodel_amd import ops
            dev = linear.weight.device if linear.weight.is_cuda else torch.device("cuda", torch.cuda.current_device())
            w = linear.weight.detach().to(dev)
            sc = scales.T.contiguous().to(dev)
            zr = zeros.T.contiguous().to(dev)
            qweight, qzeros = ops.pack_gptq(w, sc, zr, g_idx.to(dev), self.bits, planar=bool(self.planar))
            self.register_buffer("qweight", qweight)
            self.register_buffer("qzeros", qzeros)
            self.register_buffer("scales", sc.to(torch.float16))
            self.register_buffer("g_idx", g_idx.to(device=dev, dtype=torch.int32))
            if linear.bias is not None:
                self.register_buffer("bias", linear.bias.detach().to(device=dev, dtype=torch.float16))
            else:
                self.bias = None
            self.qzero_format(format=2)
            self._ready = False

        pack = pack_block

        def forward_partial(self, x: torch.Tensor) -> torch.Tensor:
            """float32 [.., N] unrounded accumulators without bias: what a row-parallel (K-sharded) tensor-parallel
            layer all-reduces before the single final rounding (gptqmodel_amd/utils/tp.py)."""
            if not self._ready:
                raise RuntimeError("HipGptqLinear.forward_partial called before post_init()")
            from gptqmodel_amd import ops
            x2, _ = flatten_input(x, self.in_features)
            out = ops.gemm(x2, self.qweight, self.meta, None, self.perm, self.out_features, self.group_size, self.kernel_bits,
                           self._scale_dtype, partial_f32=True)
            return out.reshape(x.shape[:-1] + (self.out_features,))

        def dequantize_weight(self, num_itr: int = 1) -> torch.Tensor:
            """[K,N] weights in scales.dtype, bit-exact with TorchLinear.dequantize_weight (torch.py:225)."""
            if num_itr != 1:
                raise NotImplementedError("num_itr > 1 is not supported")
            from gptqmodel_amd import ops
            if not self._ready:  # still in checkpoint layout
                qw, qz = self.qweight, self.qzeros
                if self.bits not in (4, 8):
                    qw, qz, _ = ops.widen_codes(qw, qz, self.bits, planar=bool(self.planar))
                return ops.dequant(qw, qz, self.scales, self.g_idx, self.group_size, self.kernel_bits)
            return ops.dequant_tiled(self.qweight, self.meta, self.perm, self.in_features, self.out_features,
                                     self.group_size, self.kernel_bits, self._scale_dtype)

    class HipQuantEmbeddings(HipGptqLinear):
        """Quantised embedding table on the HIP backend: the mirror of TorchQuantEmbeddings
        (gptqmodel/nn_modules/qlinear/torch.py:764-797).  in_features = num_embeddings, out_features = embedding dim;
        forward takes integer token ids.  Selected by module role, never by backend discovery
        (SUPPORTS_BACKEND_SELECTION False).  Unlike the reference, which dequantises the whole table per call, only the
        requested rows are decoded."""

        # every SUPPORTS_* must be declared on the class itself (verify_supports_params, like upstream's
        # TorchQuantEmbeddings which restates them all)
        SUPPORTS_BACKENDS = [BACKEND.GPTQ_HIP]
        SUPPORTS_BACKEND_SELECTION = False
        SUPPORTS_METHODS = [METHOD.GPTQ]
        SUPPORTS_FORMATS = ({FORMAT.GPTQ: 120, FORMAT.GPTQ_V2: 120, FORMAT.GPTQ_P: 120} if hasattr(FORMAT, "GPTQ_P")
                            else {FORMAT.GPTQ: 120, FORMAT.GPTQ_V2: 120})
        SUPPORTS_BITS = [2, 3, 4, 5, 6, 7, 8]      # like TorchQuantEmbeddings (torch.py:774); widened at post_init like the linear
        SUPPORTS_GROUP_SIZE = [-1, 32, 64, 128, 256, 512, 1024]
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
        SUPPORTS_ADAPTERS = []
        SUPPORTS_DTYPES = [torch.float16, torch.bfloat16]
        QUANT_TYPE = "hip_gptq_embedding"

        def post_init(self):
            super().post_init()
            self._inv_perm = None
            if self.perm is not None:
                inv = torch.empty_like(self.perm)
                inv[self.perm.long()] = torch.arange(self.perm.numel(), dtype=torch.int32, device=self.perm.device)
                self._inv_perm = inv

        def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
            if not self._ready:
                raise RuntimeError("HipQuantEmbeddings.forward called before post_init()")
            from gptqmodel_amd import ops
            return ops.embedding(input_ids, self.qweight, self.meta, self._inv_perm, self.in_features,
                                 self.out_features, self.group_size, self.kernel_bits, self._scale_dtype)

    class HipAwqLinear(AWQuantLinear):
        """BACKEND.AWQ_HIP: MI355X fused dequant-matmul QuantLinear for AWQ (FORMAT.GEMM) checkpoints.  Drop-in for
        AwqTorchLinear (gptqmodel/nn_modules/qlinear/torch_awq.py:20) on DEVICE.ROCM.  post_init() relayouts the N-packed
        interleaved AWQ tensors once on the device into the K-packed layout the MFMA kernel consumes (semantics of
        unpack_reorder_pack, packing_utils.py:90-103 -- what the reference's own ExllamaV2-AWQ path does); zero-points
        are used as-is (asymmetric, no +-1: REQUIRES_FORMAT_V2 = False like torch_awq.py:41)."""
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
                     bias: bool = False, pack_dtype: torch.dtype = torch.int32, adapter=None,
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

        def forward_pregathered(self, x: torch.Tensor) -> torch.Tensor:
            """Same contract as HipGptqLinear.forward_pregathered.  AWQ has no act-order permutation, so this is forward()
            minus the adapter hook; it exists so that the prefill path of utils.hf_llama (ops.rmsnorm_gather feeding the
            GEMM directly) serves AWQ models as well."""
            if not self._ready:
                raise RuntimeError("HipAwqLinear.forward_pregathered called before post_init()")
            if self.adapter:
                raise NotImplementedError("forward_pregathered: adapters are applied by forward()")
            from gptqmodel_amd import ops
            out_shape = x.shape[:-1] + (self.out_features,)
            x2, in_dtype = flatten_input(x, self.in_features)
            meta, bias = self._runtime(x2.dtype)
            out = ops.gemm(x2, self.qweight, meta, bias, None, self.out_features, self.group_size, self.bits, x2.dtype,
                           exact_bf16=self.EXACT_BF16_DECODE)
            if out.dtype != in_dtype:
                out = out.to(in_dtype)
            return out.reshape(out_shape)

        def forward_partial(self, x: torch.Tensor) -> torch.Tensor:
            """float32 [.., N] unrounded accumulators without bias (row-parallel tensor-parallel shards all-reduce these
            before the single final rounding, gptqmodel_amd/utils/tp.py) -- same contract as HipGptqLinear."""
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
                raise RuntimeError(f"{self.__class__.__name__} `{self.name}`: state_dict() after post_init() would save "
                                   "the kernel (tile-major) layout; save from the original checkpoint instead")
            super()._save_to_state_dict(destination, prefix, keep_vars)

        def dequantize_weight(self) -> torch.Tensor:
            from gptqmodel_amd import ops
            meta, _ = self._runtime(self.scales.dtype)
            return ops.dequant_tiled(self.qweight, meta, None, self.in_features, self.out_features, self.group_size,
                                     self.bits, self.scales.dtype)

    for cls in (HipGptqLinear, HipQuantEmbeddings, HipAwqLinear):
        cls._GPTQHIP_KERNEL_CLASS = True   # whichever contract base it was built on (this package's mirror or the reference's own)
        cls.__module__ = module_name       # importable / picklable under the module that exposes them
        cls.__qualname__ = cls.__name__
    return HipGptqLinear, HipQuantEmbeddings, HipAwqLinear
