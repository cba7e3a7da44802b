"""Tensor-parallel sharding of GPTQ/AWQ quantised linears for the Llama-3-70B config (BASELINE.json configs[4]).

The reference has NO tensor parallelism and no collectives (multi-GPU inference is accelerate's
layer round-robin); it only carries predicates inherited from vLLM (marlin_is_k_full /
marlin_repeat_scales_on_all_ranks, gptqmodel/utils/marlin.py:296-305) and a quantise-time padder
(quantization/config.py:1185-1189).  This is therefore new MI355X-first design:

  one process per GPU (torchrun), torch.distributed backend "nccl" (= RCCL over xGMI), Megatron pattern
    q,k,v,gate,up : COLUMN parallel   qweight[:, n0:n1]  qzeros[:, n0/pf:n1/pf]  scales[:, n0:n1]  g_idx full   no comm
    o, down       : ROW parallel      qweight[k0/pf:k1/pf, :]  qzeros[g0:g1]  scales[g0:g1]  g_idx[k0:k1]-g0
                    each rank's kernel returns UNROUNDED fp32 partial sums (GPTQHIP_GEMM_PARTIAL_F32), ONE
                    all-reduce(sum) per row-parallel layer (2 per decoder layer, 160 per 70B token), then the single
                    rounding + bias of the reference -- so TP keeps the single-GPU rounding chain.
  xGMI is point-to-point: the decode message is M*hidden*4 B (32 KB at M=1, 70B) -- latency-bound, any RCCL
  algorithm works; prefill messages (M*hidden*4 B up to GBs) are bandwidth-bound and RCCL spreads them over the 7 links.

Sharding happens on the CHECKPOINT-layout tensors, before post_init() relayouts each shard for the kernel.
All functions are pure tensor ops (CPU-testable); the parallel modules take any local module with
forward()/forward_partial(), so the world_size-2 gloo tests drive them with the oracle as local compute.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def _bounds(total: int, rank: int, world: int, multiple: int, what: str):
    if total % world != 0 or (total // world) % multiple != 0:
        raise ValueError(f"cannot shard {what}={total} over {world} ranks in multiples of {multiple}")
    step = total // world
    return rank * step, (rank + 1) * step


def shard_gptq_column(t: Dict[str, torch.Tensor], rank: int, world: int, bits: int) -> Dict[str, torch.Tensor]:
    """Split along N (out_features).  t: qweight [K/pf,N], qzeros [G,N/pf], scales [G,N], g_idx [K], bias [N]|None."""
    pf = 32 // bits
    n = t["scales"].shape[1]
    n0, n1 = _bounds(n, rank, world, 8, "out_features")
    out = {"qweight": t["qweight"][:, n0:n1].contiguous(), "qzeros": t["qzeros"][:, n0 // pf:n1 // pf].contiguous(),
           "scales": t["scales"][:, n0:n1].contiguous(), "g_idx": t["g_idx"].clone() if t.get("g_idx") is not None else None,
           "bias": t["bias"][n0:n1].contiguous() if t.get("bias") is not None else None}
    return out


def shard_gptq_row(t: Dict[str, torch.Tensor], rank: int, world: int, bits: int, group_size: int) -> Dict[str, torch.Tensor]:
    """Split along K (in_features) in whole groups; g_idx is rebased.  Act-order checkpoints (g_idx not sequential)
    reference arbitrary groups from any K-slice and are rejected here (they need a replicated input + global sort)."""
    pf = 32 // bits
    k = t["qweight"].shape[0] * pf
    k0, k1 = _bounds(k, rank, world, max(group_size, 32), "in_features")
    g0, g1 = k0 // group_size, k1 // group_size
    g_idx = t.get("g_idx")
    if g_idx is not None:
        seq = torch.arange(k, dtype=torch.int64, device=g_idx.device) // group_size
        if not torch.equal(g_idx.to(torch.int64), seq):
            raise NotImplementedError("row-parallel sharding of act-order (desc_act) checkpoints is not supported")
        g_idx = (g_idx[k0:k1] - g0).contiguous()
    return {"qweight": t["qweight"][k0 // pf:k1 // pf].contiguous(), "qzeros": t["qzeros"][g0:g1].contiguous(),
            "scales": t["scales"][g0:g1].contiguous(), "g_idx": g_idx,
            "bias": None}  # the bias is added ONCE after the all-reduce by RowParallelQuantLinear


def shard_awq_column(t, rank, world):
    n = t["scales"].shape[1]
    n0, n1 = _bounds(n, rank, world, 8, "out_features")
    return {"qweight": t["qweight"][:, n0 // 8:n1 // 8].contiguous(), "qzeros": t["qzeros"][:, n0 // 8:n1 // 8].contiguous(),
            "scales": t["scales"][:, n0:n1].contiguous(),
            "bias": t["bias"][n0:n1].contiguous() if t.get("bias") is not None else None}


def shard_awq_row(t, rank, world, group_size):
    k = t["qweight"].shape[0]
    k0, k1 = _bounds(k, rank, world, max(group_size, 32), "in_features")
    g0, g1 = k0 // group_size, k1 // group_size
    return {"qweight": t["qweight"][k0:k1].contiguous(), "qzeros": t["qzeros"][g0:g1].contiguous(),
            "scales": t["scales"][g0:g1].contiguous(), "bias": None}


class ColumnParallelQuantLinear(nn.Module):
    """y_local = local(x): rank r owns output columns [r*N/w, (r+1)*N/w).  No communication unless gather_output."""

    def __init__(self, local: nn.Module, group: Optional[dist.ProcessGroup] = None, gather_output: bool = False):
        super().__init__()
        self.local = local
        self.group = group
        self.gather_output = gather_output

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.local(x)
        if not self.gather_output:
            return y
        world = dist.get_world_size(self.group)
        parts = [torch.empty_like(y) for _ in range(world)]
        dist.all_gather(parts, y.contiguous(), group=self.group)
        return torch.cat(parts, dim=-1)


class RowParallelQuantLinear(nn.Module):
    """x arrives sharded along K (the output of a column-parallel layer).  Each rank computes fp32 partial sums over
    its K-slice, ONE all-reduce(sum) over xGMI combines them, then the reference's rounding chain runs once:
    y = round(sum) ; y = round(y + bias)   (torch.py:337-342)."""

    def __init__(self, local: nn.Module, bias: Optional[torch.Tensor] = None, group: Optional[dist.ProcessGroup] = None):
        super().__init__()
        self.local = local
        self.group = group
        self.bias = bias

    def forward(self, x_shard: torch.Tensor) -> torch.Tensor:
        partial = self.local.forward_partial(x_shard)  # float32, unrounded
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=self.group)
        out = partial.to(x_shard.dtype if x_shard.dtype in (torch.float16, torch.bfloat16) else torch.float16)
        if self.bias is not None:
            out = out + self.bias.to(device=out.device, dtype=out.dtype)
        return out
