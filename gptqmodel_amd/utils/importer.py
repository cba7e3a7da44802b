"""Kernel selection: the mirror of gptqmodel/utils/importer.py (select_quant_linear :495, get_kernel_for_backend
:147, iter_quant_linear_kernels :110, build_kernel_support_maps :182) restricted to the kernels this package
ships.  Discovery is by subclass walk with the same rules (a class participates iff its own __dict__ defines
SUPPORTS_FORMATS and SUPPORTS_BACKEND_SELECTION is truthy); AUTO orders candidates by the priority stored in
SUPPORTS_FORMATS (higher wins) and filters by device exactly like upstream (importer.py:553-556)."""
from __future__ import annotations

import logging
from typing import Dict, List, Optional, Type, Union

import torch

from ..nn_modules.qlinear import BaseQuantLinear
from .adapter import Adapter
from .backend import BACKEND, normalize_backend
from .const import DEVICE, FORMAT, METHOD, normalize_device

log = logging.getLogger("gptqmodel_amd")
_logged_selection = set()


def _log_selected(cls, backend, method, fmt):
    """One line per distinct selection (upstream logs the chosen kernel too, importer.py:602-611)."""
    key = (cls.__name__, str(backend), str(method), str(fmt))
    if key not in _logged_selection:
        _logged_selection.add(key)
        log.info("Kernel: selected `%s` for backend=%s method=%s format=%s", cls.__name__, backend, method, fmt)


def _supports_pack_api(cls: Type[BaseQuantLinear]) -> bool:
    """upstream importer.py:37-42: a kernel can be chosen with pack=True iff it exposes pack()/pack_block()."""
    return callable(getattr(cls, "pack", None)) or callable(getattr(cls, "pack_block", None))


def _import_all_qlinear_kernels() -> None:
    from ..nn_modules.qlinear import hip_awq, hip_gptq  # noqa: F401


def iter_quant_linear_kernels() -> List[Type[BaseQuantLinear]]:
    kernels, seen = [], set()

    def _walk(cls):
        for sub in cls.__subclasses__():
            if sub in seen:
                continue
            seen.add(sub)
            _walk(sub)
            if "SUPPORTS_FORMATS" in sub.__dict__ and getattr(sub, "SUPPORTS_BACKEND_SELECTION", True):
                kernels.append(sub)

    _walk(BaseQuantLinear)
    return kernels


def build_kernel_support_maps():
    _import_all_qlinear_kernels()
    auto: Dict[METHOD, Dict[FORMAT, Dict[BACKEND, Type[BaseQuantLinear]]]] = {}
    support: Dict[METHOD, Dict[FORMAT, List[BACKEND]]] = {}
    entries = []
    for cls in iter_quant_linear_kernels():
        if not isinstance(cls.SUPPORTS_FORMATS, dict):
            raise ValueError(f"{cls.__name__}.SUPPORTS_FORMATS must be a dict of FORMAT -> priority.")
        for backend in cls.SUPPORTS_BACKENDS:
            for method in cls.SUPPORTS_METHODS:
                for fmt, prio in cls.SUPPORTS_FORMATS.items():
                    entries.append((prio, METHOD(method), FORMAT(fmt), BACKEND(backend), cls))
    for prio, method, fmt, backend, cls in sorted(entries, key=lambda e: -e[0]):
        auto.setdefault(method, {}).setdefault(fmt, {})[backend] = cls
        support.setdefault(method, {}).setdefault(fmt, []).append(backend)
    return auto, support


AUTO_BACKEND_KERNEL_MAPPING, BACKEND_TO_METHOD_FORMAT_MAPPING = build_kernel_support_maps()


def get_kernel_for_backend(backend: BACKEND, quant_method: METHOD, fmt: FORMAT) -> Type[BaseQuantLinear]:
    backend = normalize_backend(backend, quant_method=quant_method)
    matches = [c for c in iter_quant_linear_kernels()
               if backend in c.SUPPORTS_BACKENDS and quant_method in c.SUPPORTS_METHODS and fmt in c.SUPPORTS_FORMATS]
    if not matches:
        raise ValueError(f"Unsupported backend: `{backend}` for `{quant_method}` with format `{fmt}` "
                         f"(gptqmodel_amd ships only the HIP kernels; use upstream gptqmodel for the others)")
    if len(matches) > 1:
        raise ValueError(f"Multiple kernels matched backend `{backend}`: {', '.join(c.__name__ for c in matches)}")
    return matches[0]


def select_quant_linear(bits: int, group_size: int, desc_act: bool, sym: bool,
                        device: Optional[Union[DEVICE, str, int, torch.device]] = None,
                        backend: BACKEND = BACKEND.AUTO, format: FORMAT = FORMAT.GPTQ,
                        quant_method: METHOD = METHOD.GPTQ, pack: bool = False, dynamic=None,
                        pack_dtype: torch.dtype = torch.int32, dtype: Optional[torch.dtype] = None,
                        multi_select: bool = False, adapter: Optional[Adapter] = None, is_sharded: bool = False,
                        **_ignored):
    """Same signature / return / error behaviour as upstream select_quant_linear (importer.py:495-654)."""
    if isinstance(format, str):
        format = FORMAT(format.lower())
    if isinstance(quant_method, str):
        quant_method = METHOD(quant_method.lower())
    backend = normalize_backend(backend, quant_method=quant_method) or BACKEND.AUTO
    if device is not None:
        device = normalize_device(device)
    supported = BACKEND_TO_METHOD_FORMAT_MAPPING.get(quant_method)
    if supported is None:
        raise ValueError(f"Unsupported quantization method: `{quant_method}`")
    if format not in supported:
        raise ValueError(f"Unsupported format: `{format}` for quantization method `{quant_method}`")
    trainable = backend == BACKEND.AUTO_TRAINABLE

    if backend in (BACKEND.AUTO, BACKEND.AUTO_TRAINABLE):
        validated, last_err = [], None
        for _, cls in AUTO_BACKEND_KERNEL_MAPPING[quant_method].get(format, {}).items():
            if DEVICE.ALL not in cls.SUPPORTS_DEVICES and device is not None and device not in cls.SUPPORTS_DEVICES:
                continue
            if is_sharded and not getattr(cls, "SUPPORTS_SHARDED_LOAD", cls.SUPPORTS_SHARDS):
                continue
            ok, err = cls.validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym,
                                   pack_dtype=pack_dtype, dtype=dtype, dynamic=dynamic, device=device,
                                   trainable=trainable, adapter=adapter)
            if not ok:
                last_err = err
                continue
            if pack and not _supports_pack_api(cls):
                continue
            if not multi_select:
                _log_selected(cls, backend, quant_method, format)
                return cls
            validated.append(cls)
        if not validated:
            if last_err:
                raise last_err
            raise ValueError("No valid quant linear")
        _log_selected(validated[0], backend, quant_method, format)
        return validated

    qlinear = get_kernel_for_backend(backend, quant_method, format)
    ok, err = qlinear.validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, pack_dtype=pack_dtype,
                               dtype=dtype, dynamic=dynamic, device=device, trainable=trainable)
    if not ok:
        raise ValueError(err)
    if pack and not _supports_pack_api(qlinear):
        raise ValueError(f"Selected backend `{backend}` with kernel `{qlinear.__name__}` cannot pack quantized weights "
                         f"for format `{format}`.")
    _log_selected(qlinear, backend, quant_method, format)
    return [qlinear] if multi_select else qlinear


def hf_select_quant_linear(bits: int, group_size: int, desc_act: bool, sym: bool, checkpoint_format: str,
                           meta=None, pack: bool = False, device_map=None, backend=None, **kw):
    """HF/optimum-stable entry (upstream importer.py:377): picks a kernel from config strings."""
    fmt = FORMAT(str(checkpoint_format).lower())
    method = METHOD.AWQ if fmt == FORMAT.GEMM else METHOD.GPTQ
    device = DEVICE.ROCM
    if isinstance(device_map, dict) and device_map:
        device = normalize_device(next(iter(device_map.values())))
    return select_quant_linear(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, device=device,
                               backend=backend or BACKEND.AUTO, format=fmt, quant_method=method, pack=pack)


def hf_select_quant_linear_v2(bits: int, group_size: int, desc_act: bool, sym: bool, format, quant_method,
                              zero_point=None, dtype=None, meta=None, pack: bool = False, device_map=None, backend=None,
                              **kw):
    """HF/optimum entry with explicit method / format strings (upstream importer.py:413-470): enum and dtype strings are
    normalised here, AWQ's `zero_point` is the negation of `sym`."""
    def _enum(value, cls, field):
        if isinstance(value, cls):
            return value
        try:
            return cls(str(value).lower())
        except ValueError as exc:
            raise ValueError(f"Unsupported {field}: `{value}`") from exc

    method = _enum(quant_method, METHOD, "quant_method")
    fmt = _enum(format, FORMAT, "format")
    if isinstance(dtype, str):
        cand = getattr(torch, dtype.replace("torch.", "").lower(), None)
        if not isinstance(cand, torch.dtype):
            raise ValueError(f"Unsupported dtype: `{dtype}`")
        dtype = cand
    if method == METHOD.AWQ and zero_point is not None:
        sym = not bool(zero_point)
    device = DEVICE.ROCM
    if isinstance(device_map, dict) and device_map:
        device = normalize_device(next(iter(device_map.values())))
    return select_quant_linear(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, device=device,
                               backend=backend or BACKEND.AUTO, format=fmt, quant_method=method, pack=pack, dtype=dtype)
