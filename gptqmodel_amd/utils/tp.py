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


def _whole_word_bits(bits: int) -> None:
    """The shard helpers slice / permute whole 4- or 8-bit fields of the packed words.  2 / 3 / 5 / 6 / 7-bit (and split-plane)
    checkpoints are sharded AFTER their codes are widened to that layout (ops.widen_codes / HipGptqLinear.widen_in_place: same code
    values) -- slicing their words here would silently cut fields in two."""
    if bits not in (4, 8):
        raise NotImplementedError(f"tensor-parallel sharding works on 4- / 8-bit words (got bits={bits}): widen the codes first "
                                  f"(gptqmodel_amd.ops.widen_codes)")


def shard_gptq_column(t: Dict[str, torch.Tensor], rank: int, world: int, bits: int) -> Dict[str, torch.Tensor]:
    """Split along N (out_features).  t: qweight [K/pf,N], qzeros [G,N/pf], scales [G,N], g_idx [K], bias [N]|None."""
    _whole_word_bits(bits)
    pf = 32 // bits
    n = t["scales"].shape[1]
    n0, n1 = _bounds(n, rank, world, 8, "out_features")
    out = {"qweight": t["qweight"][:, n0:n1].contiguous(), "qzeros": t["qzeros"][:, n0 // pf:n1 // pf].contiguous(),
           "scales": t["scales"][:, n0:n1].contiguous(), "g_idx": t["g_idx"].clone() if t.get("g_idx") is not None else None,
           "bias": t["bias"][n0:n1].contiguous() if t.get("bias") is not None else None}
    return out


def _unpack_rows(qweight: torch.Tensor, bits: int) -> torch.Tensor:
    """[K/pf, N] int32 -> codes uint8 [K, N] (code(pf*r+j, n) = (word >> bits*j) & maxq; qlinear/__init__.py:827-865)."""
    pf = 32 // bits
    sh = (torch.arange(pf, dtype=torch.int32, device=qweight.device) * bits).view(1, pf, 1)
    codes = torch.bitwise_and(torch.bitwise_right_shift(qweight.unsqueeze(1), sh), (1 << bits) - 1)
    return codes.reshape(qweight.shape[0] * pf, qweight.shape[1]).to(torch.uint8)


def _pack_rows(codes: torch.Tensor, bits: int) -> torch.Tensor:
    pf = 32 // bits
    k, n = codes.shape
    c = codes.to(torch.int64).reshape(k // pf, pf, n)
    sh = (torch.arange(pf, dtype=torch.int64, device=codes.device) * bits).view(1, pf, 1)
    w = torch.bitwise_left_shift(c, sh).sum(dim=1)                  # disjoint bit fields: sum == or
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


def shard_gptq_row(t: Dict[str, torch.Tensor], rank: int, world: int, bits: int, group_size: int,
                   act_order: str = "reject") -> Dict[str, torch.Tensor]:
    """Split along K (in_features) in whole groups; g_idx is rebased.

    Act-order checkpoints (desc_act=True: g_idx is a permutation of the group ids) reference arbitrary groups from any
    contiguous K-slice.  act_order="global_sort" follows the rule the reference inherited from vLLM/Marlin for this case
    (`marlin_is_k_full` / `marlin_repeat_scales_on_all_ranks`, gptqmodel/utils/marlin.py:296-305; row sorting
    `marlin_sort_g_idx` :368-372): the rows are sorted by group GLOBALLY first (stable argsort of g_idx), then sliced, so
    every rank again owns whole groups [g0, g1) with its own scales / zeros rows and a sequential local g_idx -- no
    replicated scales, the kernel sees an ordinary shard.  The price is on the activation side: rank r needs the input
    features perm[k0:k1], which are scattered over all ranks' column shards, so the returned dict carries
    `input_index` (int64 [K/world]) and RowParallelQuantLinear all-gathers the sharded activation before selecting them
    (the alternative -- folding the permutation into the producing column-parallel layer -- only exists for MLPs, not for
    attention outputs).  act_order="reject" (default) raises instead."""
    _whole_word_bits(bits)
    pf = 32 // bits
    k = t["qweight"].shape[0] * pf
    k0, k1 = _bounds(k, rank, world, max(group_size, 32), "in_features")
    g0, g1 = k0 // group_size, k1 // group_size
    g_idx = t.get("g_idx")
    input_index = None
    qweight = None
    if g_idx is not None:
        groups = t["scales"].shape[0]
        g64 = g_idx.to(torch.int64)
        g64 = torch.where(g64 < 0, g64 + groups, g64)
        seq = torch.arange(k, dtype=torch.int64, device=g_idx.device) // group_size
        if not torch.equal(g64, seq):
            if act_order != "global_sort":
                raise NotImplementedError("row-parallel sharding of an act-order (desc_act) checkpoint needs "
                                          "act_order='global_sort' (gathered input)")
            perm = torch.argsort(g64, stable=True)
            if not torch.equal(g64[perm], seq):
                raise NotImplementedError("act-order row sharding requires every group to own exactly group_size rows")
            input_index = perm[k0:k1].contiguous()
            qweight = _pack_rows(_unpack_rows(t["qweight"], bits)[input_index], bits)
        g_idx = (seq[k0:k1] - g0).to(torch.int32).contiguous()
    if qweight is None:
        qweight = t["qweight"][k0 // pf:k1 // pf].contiguous()
    out = {"qweight": qweight, "qzeros": t["qzeros"][g0:g1].contiguous(),
           "scales": t["scales"][g0:g1].contiguous(), "g_idx": g_idx,
           "bias": None}  # the bias is added ONCE after the all-reduce by RowParallelQuantLinear
    if input_index is not None:
        out["input_index"] = input_index
    return out


def select_gptq_columns(t: Dict[str, torch.Tensor], cols: torch.Tensor, bits: int) -> Dict[str, torch.Tensor]:
    """The GPTQ checkpoint tensors of the output columns `cols` (int64, any order, len % (32 / bits) == 0): qweight / scales / bias
    columns gathered, the N-packed zero-points unpacked, gathered and re-packed.  Exact: only integer codes move."""
    _whole_word_bits(bits)
    pf = 32 // bits
    if cols.numel() % pf != 0:
        raise ValueError(f"column selection must keep whole packed zero-point words ({pf} columns)")
    cols = cols.to(t["qweight"].device)
    sh = torch.arange(0, 32, bits, dtype=torch.int32, device=cols.device).view(1, 1, pf)
    z = ((t["qzeros"].unsqueeze(2) >> sh) & ((1 << bits) - 1)).reshape(t["qzeros"].shape[0], -1)[:, cols]
    zw = (z.reshape(z.shape[0], -1, pf).to(torch.int64) << sh.to(torch.int64)).sum(dim=2) & 0xFFFFFFFF
    return {"qweight": t["qweight"][:, cols].contiguous(), "qzeros": torch.where(zw >= 2 ** 31, zw - 2 ** 32, zw).to(torch.int32),
            "scales": t["scales"][:, cols].contiguous(), "g_idx": t["g_idx"].clone() if t.get("g_idx") is not None else None,
            "bias": t["bias"][cols].contiguous() if t.get("bias") is not None else None}


def shard_mlp_act_order(gate: Dict[str, torch.Tensor], up: Dict[str, torch.Tensor], down: Dict[str, torch.Tensor], rank: int,
                        world: int, bits: int, group_size: int):
    """Tensor-parallel shards of a Llama MLP whose down_proj is an act-order (desc_act) checkpoint, WITHOUT an activation
    exchange: down_proj's rows are sorted by group globally and sliced (shard_gptq_row(act_order="global_sort"), the Marlin rule
    gptqmodel/utils/marlin.py:296-305,368-372); rank r's rows then need the intermediate features perm[k0:k1] -- and since
    gate_proj / up_proj are column-parallel and connected to down_proj by elementwise ops only, rank r simply OWNS exactly those
    columns of gate / up, in that order (the column-parallel assignment is free).  Returns (gate_r, up_r, down_r); down_r carries
    no input_index, its local g_idx is sequential.  gate / up keep their own (input-side) g_idx."""
    down_r = shard_gptq_row(down, rank, world, bits, group_size, act_order="global_sort")
    idx = down_r.pop("input_index", None)
    if idx is None:   # down_proj has no act-order permutation: the plain contiguous split
        k0, k1 = _bounds(down["qweight"].shape[0] * (32 // bits), rank, world, max(group_size, 32), "in_features")
        idx = torch.arange(k0, k1, dtype=torch.int64)
    return select_gptq_columns(gate, idx, bits), select_gptq_columns(up, idx, bits), down_r


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


# messages above this many bytes of fp32 partial sums take the two-step exchange below (prefill); smaller ones are latency-bound
TWO_STEP_MIN_BYTES = 1 << 20


def reduce_round_gather(partial: torch.Tensor, out_dtype: torch.dtype, bias: Optional[torch.Tensor] = None,
                        group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """sum over ranks of the fp32 partials -> ONE rounding to out_dtype (+ bias) -> the full result on every rank, for the
    bandwidth-bound prefill messages (M x hidden fp32: 1 GiB at M = 32768, hidden 8192) -- SURVEY.md 5 / 8e: a single ring is capped
    by one xGMI link, so the exchange is spread over ALL links:

        1. all-to-all: the partial is cut into `world` row slabs; rank r receives slab r of every rank -- on the fully connected xGMI
           fabric every pair of GPUs has its own link, so the (world - 1) / world of the message that has to move travels over all
           7 links of a GPU at once (a reduce-scatter without a ring);
        2. rank r adds its `world` slabs in RANK ORDER (bit-identical whoever computes it), rounds ONCE like the reference (torch.py:
           337-342), adds the bias;
        3. all-gather of the 16-bit slabs: the second half of the exchange moves 2 bytes per element, not 4 -- a fp32 all-reduce moves
           8 bytes per element over the links, this 6.

    Same rounding chain as RowParallelQuantLinear's all-reduce path (sum in fp32, round once); the fp32 association differs (rank
    order here, RCCL's choice there).  Works on any backend with all_to_all_single / all_gather_into_tensor (RCCL; gloo in the CPU
    tests)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    shape = partial.shape
    n = shape[-1]
    p2 = partial.reshape(-1, n)
    m = p2.shape[0]
    if world == 1:
        out = p2.to(out_dtype)
        if bias is not None:
            out = out + bias.to(device=out.device, dtype=out_dtype)
        return out.reshape(shape)
    rows = -(-m // world)                      # rows per slab (the last slabs are zero-padded)
    if rows * world != m:
        pad = torch.zeros((rows * world - m, n), dtype=p2.dtype, device=p2.device)
        p2 = torch.cat([p2, pad], dim=0)
    recv = torch.empty_like(p2)                # [world, rows, n]: slab `rank` of every rank, in rank order
    dist.all_to_all_single(recv, p2.contiguous(), group=group)
    slabs = recv.view(world, rows, n)
    acc = slabs[0].clone()
    for r in range(1, world):
        acc += slabs[r]                        # rank order: every rank would compute the same bits for this slab
    mine = acc.to(out_dtype)
    if bias is not None:
        mine = mine + bias.to(device=mine.device, dtype=out_dtype)
    full = torch.empty((world * rows, n), dtype=out_dtype, device=mine.device)
    dist.all_gather_into_tensor(full, mine.contiguous(), group=group)
    return full[:m].reshape(shape)


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
    its K-slice, ONE exchange over xGMI combines them, then the reference's rounding chain runs once:
    y = round(sum) ; y = round(y + bias)   (torch.py:337-342).  Three message regimes: decode vectors (<= comm.n_max elements) through
    the one-shot peer-to-peer kernel when a communicator is given; mid-size messages through dist.all_reduce; prefill messages
    (>= 1 MiB of partials) through reduce_round_gather (all-to-all + rank-ordered sum + 16-bit all-gather: every link busy, 6 instead
    of 8 bytes per element on the wire)."""

    def __init__(self, local: nn.Module, bias: Optional[torch.Tensor] = None, group: Optional[dist.ProcessGroup] = None,
                 input_index: Optional[torch.Tensor] = None, comm=None, two_step: bool = True):
        super().__init__()
        self.local = local
        self.two_step = two_step    # large (prefill) messages: reduce_round_gather instead of a fp32 all-reduce
        self.group = group
        self.bias = bias
        # utils.xgmi_allreduce.OneShotAllReduce: small (decode) messages go through ONE peer-to-peer kernel that also applies
        # the rounding + bias; larger ones keep the RCCL all-reduce
        self.comm = comm
        # act-order shards (shard_gptq_row(..., act_order="global_sort")): the input features this rank's rows need, as
        # indices into the FULL (gathered) activation
        self.input_index = input_index

    def _gathered_input(self, x_shard: torch.Tensor) -> torch.Tensor:
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            parts = [torch.empty_like(x_shard) for _ in range(world)]
            dist.all_gather(parts, x_shard.contiguous(), group=self.group)
            x_full = torch.cat(parts, dim=-1)
        else:
            x_full = x_shard
        return x_full.index_select(-1, self.input_index.to(x_full.device)).contiguous()

    def forward(self, x_shard: torch.Tensor) -> torch.Tensor:
        if self.input_index is not None:
            x_shard = self._gathered_input(x_shard)
        partial = self.local.forward_partial(x_shard)  # float32, unrounded
        odt = x_shard.dtype if x_shard.dtype in (torch.float16, torch.bfloat16) else torch.float16
        if self.comm is not None and partial.numel() <= self.comm.n_max and partial.numel() % 4 == 0:
            bias = None if self.bias is None else self.bias.to(device=partial.device, dtype=odt)
            if bias is not None and partial.dim() > 1 and partial.numel() != bias.numel():
                bias = bias.expand(partial.shape).contiguous()
            return self.comm(partial.contiguous(), out_dtype=odt, bias=bias)
        if (self.two_step and dist.is_initialized() and dist.get_world_size(self.group) > 1
                and partial.numel() * 4 >= TWO_STEP_MIN_BYTES):
            # bandwidth-bound (prefill) message: all-to-all + rank-ordered sum + one rounding + 16-bit all-gather over all links
            return reduce_round_gather(partial, odt, self.bias, self.group)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=self.group)
        out = partial.to(x_shard.dtype if x_shard.dtype in (torch.float16, torch.bfloat16) else torch.float16)
        if self.bias is not None:
            out = out + self.bias.to(device=out.device, dtype=out.dtype)
        return out
