"""Module-swap / post-load helpers: the subset of gptqmodel/utils/model.py the hot path touches
(make_quant :398, create_quant_module :475, convert_gptq_v1_to_v2_format_module :750, gptqmodel_post_init :1281)."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Type

import torch
import torch.nn as nn

from ..nn_modules.qlinear import BaseQuantLinear
from .backend import BACKEND
from .const import DEVICE, FORMAT, METHOD
from .importer import select_quant_linear

_V1_TO_V2_ADD = {4: 0x11111111, 8: 0x01010101}


def convert_gptq_v1_to_v2_format_module(module: BaseQuantLinear, bits: int, pack_dtype: torch.dtype = torch.int32):
    """v1 checkpoints store zero-1; add 1 to every packed field with int32 wraparound (utils/model.py:814-831)."""
    if pack_dtype != torch.int32 or bits not in _V1_TO_V2_ADD:
        raise NotImplementedError(f"v1->v2 conversion supports int32 words with 4/8 bits, got {pack_dtype}/{bits}")
    add = _V1_TO_V2_ADD[bits]
    if add >= 2 ** 31:
        add -= 2 ** 32
    module.qzeros.data += add  # int32 tensor add wraps like the reference's `+= 0b0001...`
    module.qzero_format(format=2)
    return module


def convert_gptq_v1_to_v2_format(model: nn.Module, bits: int, pack_dtype: torch.dtype = torch.int32):
    for m in model.modules():
        if isinstance(m, BaseQuantLinear) and getattr(m, "REQUIRES_FORMAT_V2", False) and hasattr(m, "qzero_format") \
                and m.qzero_format() == 1:
            convert_gptq_v1_to_v2_format_module(m, bits=m.bits, pack_dtype=pack_dtype)
    return model


def create_quant_module(parent: nn.Module, child_name: str, linear_cls: Type[BaseQuantLinear], bits: int,
                        group_size: int, desc_act: bool, sym: bool, in_features: int, out_features: int, bias: bool,
                        full_name: str, backend: BACKEND, fmt: FORMAT, dtype: Optional[torch.dtype] = None):
    ok, err = linear_cls.validate(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym,
                                  in_features=in_features, out_features=out_features, pack_dtype=torch.int32,
                                  dtype=dtype)
    if err:
        raise err
    new = linear_cls(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, in_features=in_features,
                     out_features=out_features, pack_dtype=torch.int32, bias=bias, name=full_name, backend=backend,
                     register_buffers=True, format=fmt)
    setattr(parent, child_name, new)
    return new


def make_quant(model: nn.Module, names: Iterable[str], bits: int, group_size: int, desc_act: bool, sym: bool,
               backend: BACKEND = BACKEND.AUTO, format: FORMAT = FORMAT.GPTQ, quant_method: METHOD = METHOD.GPTQ,
               device=DEVICE.ROCM, dtype: Optional[torch.dtype] = None) -> List[Type[BaseQuantLinear]]:
    """Replace every nn.Linear whose qualified name is in `names` by the selected QuantLinear class
    (utils/model.py:398-472, 651-727).  NotImplementedError from a candidate => try the next one (AUTO)."""
    wanted = set(names)
    candidates = select_quant_linear(bits=bits, group_size=group_size, desc_act=desc_act, sym=sym, device=device,
                                     backend=backend, format=format, quant_method=quant_method, dtype=dtype,
                                     multi_select=True)
    modules: Dict[str, nn.Module] = dict(model.named_modules())
    for full_name in sorted(wanted):
        sub = modules.get(full_name)
        if sub is None or isinstance(sub, BaseQuantLinear):
            continue
        if not isinstance(sub, nn.Linear):
            raise ValueError(f"make_quant: `{full_name}` is {type(sub).__name__}, expected nn.Linear")
        parent_name, _, child = full_name.rpartition(".")
        parent = modules[parent_name] if parent_name else model
        last = None
        for cls in candidates:
            try:
                create_quant_module(parent, child, cls, bits, group_size, desc_act, sym, sub.in_features,
                                    sub.out_features, sub.bias is not None, full_name, backend, format, dtype)
                last = None
                break
            except NotImplementedError as e:
                last = e
        if last is not None:
            raise ValueError(f"No compatible quant module for `{full_name}`: {last}")
    return candidates


def gptqmodel_post_init(model: nn.Module, use_act_order: bool = False, **_kw) -> nn.Module:
    """Call post_init() on every QuantLinear once its tensors are on the device (utils/model.py:1281-1344)."""
    if isinstance(model, BaseQuantLinear):
        model.post_init()
        return model
    for m in model.modules():
        if isinstance(m, BaseQuantLinear):
            m.post_init()
    return model
