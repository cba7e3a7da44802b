"""BACKEND selector values.  Mirrors gptqmodel/utils/backend.py:10-87 for the members that matter here and
adds the two members this backend introduces (INTEGRATION.md shows the two-line upstream patch):

    GPTQ_HIP = "gptq_hip"     AWQ_HIP = "awq_hip"      (+ method-generic legacy-style alias HIP = "hip")
"""
from __future__ import annotations

from enum import Enum
from typing import Any, Optional, Union


class BACKEND(str, Enum):
    AUTO = "auto"
    AUTO_TRAINABLE = "auto_trainable"
    # new: MI355X-native kernels
    GPTQ_HIP = "gptq_hip"
    AWQ_HIP = "awq_hip"
    HIP = "hip"  # generic alias resolved per quant method, like upstream's TORCH/MARLIN legacy names
    # upstream names kept so that saved configs naming them fail with a clear message instead of a KeyError
    GPTQ_TORCH = "gptq_torch"
    AWQ_TORCH = "awq_torch"
    TORCH = "torch"
    TORCH_AWQ = "torch_awq"


_LEGACY_BACKEND_BY_METHOD = {
    "gptq": {BACKEND.HIP: BACKEND.GPTQ_HIP, BACKEND.TORCH: BACKEND.GPTQ_TORCH},
    "awq": {BACKEND.HIP: BACKEND.AWQ_HIP, BACKEND.TORCH: BACKEND.AWQ_TORCH, BACKEND.TORCH_AWQ: BACKEND.AWQ_TORCH},
}


def normalize_backend(backend: Optional[Union[str, BACKEND]], *, quant_method: Optional[Union[str, Any]] = None):
    """Same resolution rules as gptqmodel/utils/backend.py:153-177: member name or value, then the
    method-specific legacy alias map."""
    if backend is None:
        return None
    if isinstance(backend, BACKEND):
        resolved = backend
    elif isinstance(backend, str):
        s = backend.strip()
        if not s:
            return None
        resolved = BACKEND.__members__.get(s.upper())
        if resolved is None:
            resolved = BACKEND(s.lower())
    else:
        raise TypeError(f"backend must be a string or BACKEND, got `{type(backend)}`")
    method = None if quant_method is None else str(getattr(quant_method, "value", quant_method)).lower()
    if method is None:
        return resolved
    return _LEGACY_BACKEND_BY_METHOD.get(method, {}).get(resolved, resolved)
