"""Enums of the reference's plugin contract that reach the QuantLinear classes, mirrored by value so
checkpoints' quantize_config.json strings parse identically.

Reference definitions: DEVICE / PLATFORM  gptqmodel/models/_const.py:41-84,
FORMAT / METHOD  gptqmodel/quantization/config.py:106-141.
Only the members relevant to the GPTQ/AWQ int4/int8 hot path are listed; the HIP backend is
`DEVICE.ROCM`-only by construction.
"""
from __future__ import annotations

from enum import Enum

import torch


class DEVICE(str, Enum):
    ALL = "all"
    CPU = "cpu"
    CUDA = "cuda"   # a ROCm torch build exposes the GPU as "cuda"; it is normalised to ROCM below
    ROCM = "rocm"

    @property
    def type(self) -> str:
        return "cuda" if self == DEVICE.ROCM else self.value

    def to_torch_device(self) -> torch.device:
        return torch.device("cuda:0") if self in (DEVICE.CUDA, DEVICE.ROCM) else torch.device(self.type)


class PLATFORM(str, Enum):
    ALL = "all"
    LINUX = "linux"


class FORMAT(str, Enum):
    GPTQ = "gptq"        # v1 on disk: qzeros hold zero-1
    GPTQ_V2 = "gptq_v2"  # qzeros hold zero
    GPTQ_P = "gptq_p"    # planar (split-plane) words, qzeros hold zero: what 5 / 6 / 7-bit checkpoints declare, optional for 3 bits
    GEMM = "gemm"        # AWQ GEMM layout


class METHOD(str, Enum):
    GPTQ = "gptq"
    AWQ = "awq"


IS_ROCM = bool(getattr(torch.version, "hip", None))


def normalize_device(value) -> DEVICE:
    """str | int | torch.device | DEVICE -> DEVICE, mapping cuda -> ROCM on a ROCm build
    (reference: models/_const.py:103 normalize_device + utils/importer.py:349-351,519-521)."""
    if isinstance(value, DEVICE):
        dev = value
    elif isinstance(value, int):
        dev = DEVICE.CUDA
    elif isinstance(value, torch.device):
        dev = DEVICE(value.type)
    elif isinstance(value, str):
        dev = DEVICE(value.split(":")[0].lower())
    else:
        raise ValueError(f"Unsupported device value: {value!r}")
    if dev == DEVICE.CUDA and IS_ROCM:
        dev = DEVICE.ROCM
    return dev
