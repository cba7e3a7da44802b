"""QuantLinear base hierarchy: the host-side mirror of the reference plugin contract
(gptqmodel/nn_modules/qlinear/__init__.py: BaseQuantLinear :73, GroupedQuantLinear :520,
PackedGroupedQuantLinear :664, GPTQQuantLinear :727, AWQuantLinear :1634).

Same class attributes, constructor keywords, buffer names/shapes, validate()/validate_once() protocol and
error conventions (NotImplementedError = "unsupported here, try next candidate"; ValueError = hard config
error) so that kernel selection, checkpoint loading (buffers are registered by name) and the reference's
tests read the same.  Only what the GPTQ/AWQ int4/int8 hot path needs is carried over.
"""
from __future__ import annotations

import copy
import math
import sys
from functools import lru_cache
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ...utils.adapter import Adapter
from ...utils.backend import BACKEND
from ...utils.const import DEVICE, FORMAT, METHOD, PLATFORM


class BaseQuantLinear(nn.Module):
    SUPPORTS_BACKENDS: List[BACKEND] = None
    SUPPORTS_BACKEND_SELECTION: bool = True
    SUPPORTS_METHODS: List[METHOD] = None
    SUPPORTS_FORMATS: Dict[FORMAT, int] = None
    SUPPORTS_BITS: List[int] = None
    SUPPORTS_GROUP_SIZE: List[int] = None
    SUPPORTS_DESC_ACT: List[bool] = None
    SUPPORTS_SYM: List[bool] = None
    SUPPORTS_SHARDS: bool = None
    SUPPORTS_SHARDED_LOAD: bool = True
    SUPPORTS_TRAINING: bool = None
    SUPPORTS_AUTO_PADDING: bool = None
    SUPPORTS_IN_FEATURES_DIVISIBLE_BY: List[int] = None
    SUPPORTS_OUT_FEATURES_DIVISIBLE_BY: List[int] = None
    SUPPORTS_PACK_DTYPES: List[torch.dtype] = None
    SUPPORTS_ADAPTERS: List[type] = None
    SUPPORTS_DEVICES: List[DEVICE] = None
    SUPPORTS_PLATFORM: List[PLATFORM] = None
    SUPPORTS_DTYPES: List[torch.dtype] = None
    REQUIRES_FORMAT_V2: bool = False
    AUTOTUNE: bool = False

    def __init__(self, bits: int, group_size: int, desc_act: bool, sym: bool, in_features: int, out_features: int,
                 bias: bool, pack_dtype: torch.dtype, backend: BACKEND, adapter: Optional[Adapter],
                 name: str = None, dtype: Optional[torch.dtype] = None, **kwargs):
        super().__init__()
        if name is None:
            name = f"{self.__class__.__module__}.{self.__class__.__qualname__}"
        self.name = name
        self.in_features = in_features
        self.out_features = out_features
        self.bits = bits
        self.backend = backend
        self.adapter = copy.deepcopy(adapter)  # adapters hold per-module tensors (qlinear/__init__.py:125)
        self.optimized = False
        # rotation / online-Hadamard state of SpinQuant / QuaRot checkpoints (qlinear/__init__.py:133-141): off by default
        self.online_full_had = False
        self.online_partial_had = False
        self.had_dim = -1
        self.K = 1
        self.had_K = None

        _, err = self.validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym,
                               in_features=in_features, out_features=out_features, pack_dtype=pack_dtype,
                               dtype=dtype, adapter=adapter)
        if err:
            raise err

        self.group_size = group_size if group_size != -1 else in_features
        self.requested_group_size = group_size
        self.desc_act = desc_act
        self.sym = sym
        self.pack_dtype = pack_dtype
        self.pack_dtype_bits = 32
        self.pack_factor = self.pack_dtype_bits // self.bits
        self.maxq = 2 ** self.bits - 1

    # ---- buffers ---------------------------------------------------------------------------------
    def list_buffers(self) -> List[torch.Tensor]:
        out, seen = [], set()
        for state in (self._parameters, self._buffers):
            for t in state.values():
                if isinstance(t, torch.Tensor) and id(t) not in seen:
                    seen.add(id(t))
                    out.append(t)
        return out

    def runtime_device(self) -> Optional[torch.device]:
        for n in ("qweight", "qzeros", "scales", "g_idx", "bias"):
            t = getattr(self, n, None)
            if isinstance(t, torch.Tensor):
                return t.device
        return None

    # ---- lifecycle ---------------------------------------------------------------------------------
    def post_init(self):
        """Called after weights are on their device (gptqmodel/utils/model.py:1335-1340)."""
        self.clear_autotune()
        if self.adapter is not None:
            self.adapter.post_init(weight_key=self.name, device=self.runtime_device(),
                                   lora_A=getattr(self, "lora_A", None), lora_B=getattr(self, "lora_B", None))

    # ---- rotation hook (qlinear/__init__.py:485-518; torch.py:312 applies it to x before the matmul) -------------------
    def set_had_K(self, had_K: Optional[torch.Tensor]) -> None:
        if "had_K" in self._buffers:
            if had_K is None:
                del self._buffers["had_K"]
                self.had_K = None
            else:
                self._buffers["had_K"] = had_K
            return
        if had_K is None:
            self.had_K = None
            return
        if hasattr(self, "had_K"):
            del self.had_K
        self.register_buffer("had_K", had_K, persistent=False)

    def _apply_rotation_to_input(self, x: torch.Tensor) -> torch.Tensor:
        """Identity unless the checkpoint asks for an online Hadamard transform.  That transform lives in the reference's
        quantization/rotation package (out of this path's scope): fail loudly instead of returning unrotated results."""
        if self.online_full_had or self.online_partial_had:
            raise NotImplementedError(f"{self.__class__.__name__}: online Hadamard rotation (SpinQuant/QuaRot) is not "
                                      "implemented by the HIP backend")
        return x

    # ---- optional per-module autotune hook (qlinear/__init__.py:236-255): one-shot, skipped in training; the HIP kernels
    # plan their launch geometry per call (plan_skinny / plan_tiled), so the classes here leave `_autotune` unimplemented
    # and keep autotune disabled
    autotune_enabled: bool = False

    def clear_autotune(self):
        self._autotune_complete = False
        self._autotune_result = None

    def get_autotune_result(self):
        return getattr(self, "_autotune_result", None)

    def _autotune(self, *args, **kwargs):
        raise NotImplementedError(f"{self.__class__.__name__} does not implement `_autotune()`.")

    def maybe_autotune(self, *args, **kwargs):
        if not self.autotune_enabled or self.training:
            return self.get_autotune_result()
        if getattr(self, "_autotune_complete", False):
            return self._autotune_result
        self._autotune_result = self._autotune(*args, **kwargs)
        self._autotune_complete = True
        return self._autotune_result

    def optimize(self, backend: str = "inductor", mode: str = None, fullgraph: bool = False):
        self.optimized = True  # nothing to torch.compile: the kernel is native

    def train(self, mode: bool = True):
        if mode and not self.SUPPORTS_TRAINING and mode != self.training:
            raise NotImplementedError(f"{self.__class__.__name__}: `{self.name}` switching to training mode.")
        return super().train(mode)

    # ---- validation protocol (qlinear/__init__.py:257-332, 340-445, 526-660) -----------------------
    @classmethod
    @lru_cache(maxsize=1024)
    def cached_validate_once(cls) -> Tuple[bool, Optional[Exception]]:
        ok, exp = cls.validate_once()
        if not ok and exp is not None:
            exp.with_traceback(None)
            return False, exp
        return True, None

    @classmethod
    def validate_once(cls) -> Tuple[bool, Optional[Exception]]:
        return True, None

    @classmethod
    def validate(cls, bits: int, group_size: int = -1, desc_act: bool = False, sym: bool = True,
                 in_features: int = None, out_features: int = None, pack_dtype: torch.dtype = None,
                 dtype: Optional[torch.dtype] = None, dynamic: Optional[dict] = None,
                 device: Optional[DEVICE] = None, trainable: Optional[bool] = None,
                 adapter: Optional[Adapter] = None) -> Tuple[bool, Optional[Exception]]:
        ok_once, exp_once = cls.cached_validate_once()
        if not ok_once:
            return False, exp_once
        return cls._validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, in_features=in_features,
                             out_features=out_features, pack_dtype=pack_dtype, dtype=dtype, dynamic=dynamic,
                             device=device, trainable=trainable, adapter=adapter)

    @classmethod
    def verify_supports_params(cls):
        for name, value in BaseQuantLinear.__dict__.items():
            if not name.startswith("SUPPORTS") or callable(value) or value is not None:
                continue
            if name not in cls.__dict__ or cls.__dict__[name] is None:
                raise ValueError(f"{cls.__name__}.{name} is not overridden / cannot be None.")

    @classmethod
    def validate_device(cls, device: DEVICE):
        if device not in cls.SUPPORTS_DEVICES:
            raise NotImplementedError(f"{cls} only supports `{cls.SUPPORTS_DEVICES}`: actual device = `{device}`")

    @classmethod
    def _validate(cls, bits=4, group_size=128, desc_act=False, sym=False, pack_dtype=None, dtype=None, dynamic=None,
                  in_features=None, out_features=None, device=None, trainable=None, adapter=None):
        cls.verify_supports_params()
        NI = NotImplementedError
        if adapter is not None and adapter.__class__ not in cls.SUPPORTS_ADAPTERS:
            return False, NI(f"{cls} does not support adapter: {adapter}")
        if pack_dtype not in cls.SUPPORTS_PACK_DTYPES:
            return False, NI(f"{cls} does not support `pack_dtype`: {pack_dtype}")
        if dtype is not None and dtype not in cls.SUPPORTS_DTYPES:
            return False, NI(f"{cls} only supports `{cls.SUPPORTS_DTYPES}` dtype: actual dtype = `{dtype}`")
        if PLATFORM.ALL not in cls.SUPPORTS_PLATFORM and sys.platform not in [p.value for p in cls.SUPPORTS_PLATFORM]:
            return False, NI(f"{cls} does not support platform: {sys.platform}")
        if DEVICE.ALL not in cls.SUPPORTS_DEVICES and device is not None:
            try:
                cls.validate_device(device)
            except NotImplementedError:
                return False, NI(f"{cls} does not support device: {device}")
        if trainable and not cls.SUPPORTS_TRAINING:
            return False, NI(f"{cls} does not support training.")
        if bits not in cls.SUPPORTS_BITS:
            return False, NI(f"{cls} only supports `{cls.SUPPORTS_BITS}` bits: actual bits = `{bits}`")
        if group_size not in cls.SUPPORTS_GROUP_SIZE and group_size != in_features:
            return False, NI(f"{cls} only supports `{cls.SUPPORTS_GROUP_SIZE}` group_size: actual group_size = `{group_size}`")
        if sym not in cls.SUPPORTS_SYM:
            return False, NI(f"{cls} only supports symmetric `{cls.SUPPORTS_SYM}` quantization: actual sym = `{sym}`")
        if desc_act not in cls.SUPPORTS_DESC_ACT:
            return False, NI(f"{cls} only supports `{cls.SUPPORTS_DESC_ACT}` desc_act: actual desc_act = `{desc_act}`")
        if dynamic is not None:
            for layer, ov in dynamic.items():
                if ov.get("bits", bits) not in cls.SUPPORTS_BITS:
                    return False, NI(f"{cls} only supports `{cls.SUPPORTS_BITS}` bits: dynamic bits for `{layer}`")
                if ov.get("group_size", group_size) not in cls.SUPPORTS_GROUP_SIZE:
                    return False, NI(f"{cls} only supports `{cls.SUPPORTS_GROUP_SIZE}` group_size: dynamic for `{layer}`")
                if ov.get("sym", sym) not in cls.SUPPORTS_SYM or ov.get("desc_act", desc_act) not in cls.SUPPORTS_DESC_ACT:
                    return False, NI(f"{cls}: unsupported dynamic sym/desc_act for `{layer}`")
        if in_features is not None:
            if not all(in_features % d == 0 for d in cls.SUPPORTS_IN_FEATURES_DIVISIBLE_BY):
                return False, NI(f"{cls}: `in_features`: {in_features} must be divisible by {cls.SUPPORTS_IN_FEATURES_DIVISIBLE_BY}.")
            gs = in_features if group_size == -1 else group_size
            if gs <= 0 or (in_features % gs != 0 and not cls.SUPPORTS_AUTO_PADDING):
                return False, NI(f"{cls}: `in_features`: {in_features} must be divisible by `group_size: {group_size}`.")
        if out_features is not None:
            if not all(out_features % d == 0 for d in cls.SUPPORTS_OUT_FEATURES_DIVISIBLE_BY):
                return False, NI(f"{cls}: `out_features`: {out_features} must be divisible by {cls.SUPPORTS_OUT_FEATURES_DIVISIBLE_BY}.")
        return True, None


class GPTQQuantLinear(BaseQuantLinear):
    """GPTQ tensor contract (qlinear/__init__.py:827-865): qweight int32 [K*bits/32, N] K-packed sequential,
    qzeros int32 [G, N*bits/32] N-packed sequential, scales fp16 [G,N], g_idx int32 [K], bias fp16 [N]."""

    def __init__(self, *args, bias: bool = False, register_buffers: bool = False, format: Optional[FORMAT] = None,
                 **kwargs):
        super().__init__(*args, bias=bias, **kwargs)
        self.format = format
        self._qzeros_format = 1
        if register_buffers:
            k, n, g = self.in_features, self.out_features, math.ceil(self.in_features / self.group_size)
            self.register_buffer("qweight", torch.zeros((math.ceil(k * self.bits / 32), n), dtype=torch.int32))
            self.register_buffer("qzeros", torch.zeros((g, math.ceil(n * self.bits / 32)), dtype=torch.int32))
            self.register_buffer("scales", torch.zeros((g, n), dtype=torch.float16))
            self.register_buffer("g_idx", torch.tensor([i // self.group_size for i in range(k)], dtype=torch.int32))
            if bias:
                self.register_buffer("bias", torch.zeros(n, dtype=torch.float16))
            else:
                self.bias = None

    def qzero_format(self, format: int = None) -> int:
        if format is None:
            return self._qzeros_format
        if format not in (1, 2):
            raise ValueError("Unsupported qzero format. Only 1 and 2 are supported.")
        self._qzeros_format = format
        return self._qzeros_format


class AWQuantLinear(BaseQuantLinear):
    """AWQ GEMM tensor contract (qlinear/__init__.py:1634-1668): qweight int32 [K, N/8] N-packed interleaved,
    qzeros int32 [G, N/8], scales fp16 [G,N]; no g_idx."""

    def __init__(self, *args, bias: bool = False, register_buffers: bool = False, **kwargs):
        kwargs.pop("format", None)
        super().__init__(*args, bias=bias, **kwargs)
        if register_buffers:
            k, n, g = self.in_features, self.out_features, self.in_features // self.group_size
            self.register_buffer("qweight", torch.zeros((k, n // self.pack_factor), dtype=torch.int32))
            self.register_buffer("qzeros", torch.zeros((g, n // self.pack_factor), dtype=torch.int32))
            self.register_buffer("scales", torch.zeros((g, n), dtype=torch.float16))
            if bias:
                self.register_buffer("bias", torch.zeros(n, dtype=torch.float16))
            else:
                self.bias = None


__all__ = ["BaseQuantLinear", "GPTQQuantLinear", "AWQuantLinear"]
