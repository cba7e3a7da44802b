"""Logic shared by the two HIP QuantLinear classes, written against plain module attributes
(qweight/qzeros/scales/g_idx/bias/bits/group_size/adapter) so the same functions serve the classes in this
package and the thin upstream-tree classes shown in INTEGRATION.md."""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def hip_validate_once() -> Tuple[bool, Optional[Exception]]:
    """validate_once() contract (reference qlinear/__init__.py:257-270): (False, ImportError) when the native
    library or a gfx950 device is unusable, so that BACKEND.AUTO falls through to the next candidate."""
    import os
    if os.environ.get("GPTQHIP_DISABLE", "").strip() not in ("", "0"):
        return False, ImportError("gptqmodel_amd HIP kernels disabled by GPTQHIP_DISABLE")
    try:
        from gptqmodel_amd import ops
        if not torch.cuda.is_available():
            return False, ImportError("gptqmodel_amd HIP kernels need a ROCm device (torch.cuda.is_available() is False)")
        ops.device_info(torch.cuda.current_device())
    except Exception as e:  # noqa: BLE001 - any load/probe failure means "kernel unavailable"
        return False, ImportError(f"gptqmodel_amd HIP kernels unavailable: {e}")
    return True, None


def check_g_idx(g_idx: torch.Tensor, groups: int) -> torch.Tensor:
    """Range guard before any kernel trusts g_idx (same concern as the reference's CWE-125 guard
    gptqmodel/nn_modules/qlinear/tritonv2.py:25-50).  Negative entries wrap by +G like torch indexing."""
    g = g_idx.to(torch.int64)
    g = torch.where(g < 0, g + groups, g)
    if g.numel() and (int(g.min()) < 0 or int(g.max()) >= groups):
        raise ValueError(f"g_idx out of range: min={int(g.min())} max={int(g.max())} groups={groups}")
    return g


def act_order_permutation(g_idx: torch.Tensor, group_size: int, groups: int) -> Optional[torch.Tensor]:
    """Returns perm (int32 [K]) such that sorted row k' is original row perm[k'], or None when rows are already in
    group order.  Stable argsort == torch_fused.py:121-151 / utils/marlin.py:368-372 semantics."""
    g = check_g_idx(g_idx, groups)
    k = g.numel()
    seq = torch.arange(k, device=g.device, dtype=torch.int64) // group_size
    if torch.equal(g, seq):
        return None
    perm = torch.argsort(g, stable=True)
    if not torch.equal(g[perm], seq):
        # real GPTQ act-order checkpoints always have exactly group_size rows per group
        # (quantization/gptq.py:1291-1302); anything else cannot be expressed as row-sorted groups
        raise NotImplementedError("HIP kernel requires every group in g_idx to own exactly group_size rows")
    return perm.to(torch.int32)


def flatten_input(x: torch.Tensor, in_features: int):
    if x.shape[-1] != in_features:
        raise RuntimeError(f"input last dim {x.shape[-1]} != in_features {in_features}")
    x2 = x.reshape(-1, in_features)
    in_dtype = x2.dtype
    if in_dtype not in (torch.float16, torch.bfloat16):
        x2 = x2.to(torch.float16)  # ExllamaV2Linear does the same cast (exllamav2.py:147-170)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    return x2, in_dtype
