"""HipGptqLinear / HipQuantEmbeddings -- BACKEND.GPTQ_HIP on this package's mirror of the reference plugin contract.
The implementation is shared with the upstream-tree overlay: see hip_impl.make_hip_classes."""
from __future__ import annotations

from types import SimpleNamespace

from ...utils.adapter import Lora
from ...utils.backend import BACKEND
from ...utils.const import DEVICE, FORMAT, METHOD, PLATFORM
from . import AWQuantLinear, GPTQQuantLinear
from .hip_impl import make_hip_classes

_NS = SimpleNamespace(GPTQQuantLinear=GPTQQuantLinear, AWQuantLinear=AWQuantLinear, BACKEND=BACKEND, DEVICE=DEVICE,
                      FORMAT=FORMAT, METHOD=METHOD, PLATFORM=PLATFORM, Lora=Lora)
HipGptqLinear, HipQuantEmbeddings, _HipAwqLinear = make_hip_classes(_NS, __name__)
_HipAwqLinear.__module__ = __name__.rsplit(".", 1)[0] + ".hip_awq"

__all__ = ["HipGptqLinear", "HipQuantEmbeddings"]
