"""Batch-1 decode step over a stack of Llama-style decoder layers as a CHAIN of decode ops (gptqhip_decode_linear).

The reference runs a decode step as HF's module graph: per layer 7 QuantLinear.forward calls (torch.py:302-347) with
RMSNorm / SiLU*mul / residual-add torch kernels between them -- ~20 dependent launches per layer at batch 1, each a
kernel boundary of 2-3 us (measured: 437 tokens/s for the Llama-3-8B linear stack, profiles/r02_*).  This helper builds
the same computation from the modules' already-relayouted tensors as 4 launches per layer: sibling projections fused
(utils.model.fuse_siblings) and the glue fused into the GEMV (input RMSNorm / SiLU*mul, residual-add epilogue).
All ops are bound once (raw-pointer structs); a step is ONE host call (gptqhip_decode_linear_seq: 870 tokens/s for the 8B stack
without any graph), or one HIP graph replay after capture.

Per layer (h = residual stream [hidden], activation dtype):
    qkv = rmsnorm(h; w_in) @ Wqkv                       in_glue RMSNORM
    a   = attention(qkv)                                 NOT a quantised linear: the stand-in "a = q" here
    h1  = h + a @ Wo                                     residual epilogue
    a   = silu(g) * u,  g|u = rmsnorm(h1; w_post) @ Wgate_up     in_glue RMSNORM + OUT_SILU_MUL_PAIRED epilogue when the fused
                                                          gate_up columns are interleaved (utils.model.fuse_gate_up_interleaved)
    h2  = h1 + a @ Wdown                                 residual epilogue
  (gate_up fused by plain concatenation instead: gu = ... @ Wgate_up, then in_glue SILU_MUL on the down op.)
RMSNorm statistics travel with the data: the op that writes a residual-stream vector also writes the per-tile sums of its
squares (stats_out), the normalising consumer sums those 256 partials per wave instead of re-reducing the row per block.

(An in-launch dependency scheme -- consecutive ops on two streams, op i+1 prefetching its packed weights while spinning on
op i's arrival counters -- was built and measured in round 2: bit-identical results but 2x SLOWER than plain stream order
(437 vs 820 tokens/s) and not robust under foreign load; see docs/history/DESIGN_rounds_1-5.md 4.1.1; the kernel lives in git history, commits b17792b..f84fc07.)
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .. import ops


class DecodeLayer:
    """The four (fused) quantised linears of one decoder layer + its two RMSNorm weights."""

    def __init__(self, qkv, o, gate_up, down, input_norm: torch.Tensor, post_norm: torch.Tensor,
                 o_bias: Optional[torch.Tensor] = None, down_bias: Optional[torch.Tensor] = None,
                 o_input_index: Optional[torch.Tensor] = None):
        self.qkv, self.o, self.gate_up, self.down = qkv, o, gate_up, down
        self.input_norm, self.post_norm = input_norm, post_norm
        # tensor parallel only: the FULL-layer bias of the row-parallel o / down projections (added once, after the reduction)
        self.o_bias, self.down_bias = o_bias, down_bias
        # tensor parallel + act-order o_proj only (utils.tp.shard_gptq_row(act_order="global_sort")["input_index"]): the features of
        # the FULL attention output this rank's sorted rows consume
        self.o_input_index = o_input_index


def _lin_tensors(lin, dtype):
    """(qweight_t, meta, bias, scale_dtype, perm) of a post_init()ed HIP QuantLinear for activations of `dtype`.
    perm: the act-order (desc_act) permutation of the module's input features or None; the decode op applies it in the kernel."""
    if not getattr(lin, "_ready", False):
        raise RuntimeError("DecodeStep needs post_init()ed HIP QuantLinear modules")
    if hasattr(lin, "_runtime"):  # HipAwqLinear: constants depend on the compute dtype
        meta, bias = lin._runtime(dtype)
        return lin.qweight, meta, bias, dtype, None
    bias = lin._bias_for(dtype, lin.qweight.device)
    return lin.qweight, lin.meta, bias, lin._scale_dtype, getattr(lin, "perm", None)


class DecodeStep:
    def __init__(self, layers: Sequence[DecodeLayer], hidden: int, q_dim: int, dtype: torch.dtype, eps: float = 1e-5,
                 device: Optional[torch.device] = None, exact: bool = False):
        """exact: the opt-in exact-arithmetic dequant (GPTQHIP_GEMM_EXACT; not the reference's per-weight rounding chain)."""
        if not layers:
            raise ValueError("no layers")
        self.layers = list(layers)
        self.device = device or layers[0].qkv.qweight.device
        self.dtype = dtype
        dev = self.device
        inter2 = layers[0].gate_up.out_features
        qkv_n = layers[0].qkv.out_features
        self.x_in = torch.zeros(hidden, dtype=dtype, device=dev)      # the step's input (embedding of the new token)
        # residual stream: h1 / h2 of every layer get their own buffers (no reuse, no WAR reasoning)
        self.h = torch.zeros((len(self.layers), 2, hidden), dtype=dtype, device=dev)
        self.qkv_out = torch.zeros(qkv_n, dtype=dtype, device=dev)
        self.gu_out = torch.zeros(inter2, dtype=dtype, device=dev)   # gate|up, or silu(gate)*up in its first half (interleaved)
        # per-tile sums of squares of every residual-stream vector written by the chain
        self.stats = torch.zeros((len(self.layers), 2, -(-hidden // 16)), dtype=torch.float32, device=dev)
        self._keep = []   # tensors the raw-pointer structs refer to
        self.ops: List = []
        # one scratch for the whole chain (only narrow layers use it), sized for the largest request
        need = 0
        for L in self.layers:
            for lin in (L.qkv, L.o, L.gate_up, L.down):
                need = max(need, ops.workspace_bytes(1, lin.in_features, lin.out_features, lin.group_size, getattr(lin, "kernel_bits", lin.bits), False))
        self.workspace = torch.zeros(max(need, 1 << 20), dtype=torch.uint8, device=dev)
        h_in, st_in = self.x_in, None     # the step's input comes from outside the chain: no producer statistics
        for li, L in enumerate(self.layers):
            h1, h2 = self.h[li, 0], self.h[li, 1]
            st1, st2 = self.stats[li, 0], self.stats[li, 1]
            paired = bool(getattr(L.gate_up, "gate_up_interleaved", False))
            plan = [
                # (module, x, out, in_glue, norm weight, residual, out_glue, stats_in, stats_out)
                (L.qkv, h_in, self.qkv_out, ops.GLUE_RMSNORM, L.input_norm, None, ops.OUT_NONE, st_in, None),
                (L.o, self.qkv_out, h1, ops.GLUE_NONE, None, h_in, ops.OUT_NONE, None, st1),   # stand-in attention: a = q
                (L.gate_up, h1, self.gu_out, ops.GLUE_RMSNORM, L.post_norm, None,
                 ops.OUT_SILU_MUL_PAIRED if paired else ops.OUT_NONE, st1, None),
                (L.down, self.gu_out, h2, ops.GLUE_NONE if paired else ops.GLUE_SILU_MUL, None, h1, ops.OUT_NONE, None, st2),
            ]
            for j, (lin, x, out, glue, nw, res, oglue, s_in, s_out) in enumerate(plan):
                qw, meta, bias, sdt, perm = _lin_tensors(lin, dtype)
                K, N = lin.in_features, lin.out_features
                if j == 1 and K != q_dim:
                    raise ValueError("o_proj in_features must equal q_dim")
                if not ops.decode_supported(K, N, lin.group_size, perm is not None):
                    raise NotImplementedError(f"decode chain: layer shape K={K} N={N} group_size={lin.group_size} unsupported")
                self._keep.extend([qw, meta, bias, nw, perm])
                self.ops.append(ops.make_decode_op(x, qw, meta, bias, out, K, N, lin.group_size, getattr(lin, "kernel_bits", lin.bits), sdt, in_glue=glue,
                                                   norm_weight=nw, eps=eps, residual=res, workspace=self.workspace,
                                                   out_glue=oglue, stats_in=s_in, stats_out=s_out, perm=perm, exact=exact))
            h_in, st_in = h2, st2
        self.out = h_in
        self._seq = ops.bind_decode_seq(self.ops)

    def run(self) -> torch.Tensor:
        """Enqueue one decode step on the current stream (ONE host call for the whole chain).  Capture-safe: wrap in
        torch.cuda.graph() to replay a token as one graph launch."""
        with torch.cuda.device(self.device):
            ops.launch_decode_seq(self._seq, self.device)
        return self.out


def tp_layer_plan(gather_o_input: bool):
    """The tensor-parallel decode step of ONE decoder layer as symbolic steps over named buffers -- the orchestration TPDecodeStep binds to
    device pointers, and the one tests/test_tp_gloo.py executes on two gloo ranks with the oracle as local compute and dist.all_reduce as
    the exchange (same list, so the order of ops, what feeds what and where the reductions sit are tested without a GPU).

    Buffers: h_in / st_in (the residual stream entering the layer and its per-tile sums of squares), qkv_out, o_in (act-order o_proj
    shards only), partial (fp32 [hidden]), h1 / st1, gu_out, h2 / st2.
    Steps:  ("op", which, x, out, in_glue, norm, out_glue, stats_in)      which in qkv | o | gate_up | down
            ("ag", x_local, index, out)                                    all-gather the ranks' attention outputs, select this rank's rows' features
            ("ar", residual, bias, out, stats_out)                         out = act(residual + act(act(sum_r partial) + bias)), stats of out"""
    steps = [("op", "qkv", "h_in", "qkv_out", "rmsnorm", "w_in", "none", "st_in")]
    o_x = "qkv_out_local"                                   # stand-in attention: a_r = this rank's q columns
    if gather_o_input:
        steps.append(("ag", "qkv_out_local", "o_input_index", "o_in"))
        o_x = "o_in"
    steps += [("op", "o", o_x, "partial", "none", None, "partial_f32", None),
              ("ar", "h_in", "o_bias", "h1", "st1"),
              ("op", "gate_up", "h1", "gu_out", "rmsnorm", "w_post", "silu_mul_paired", "st1"),
              ("op", "down", "gu_out", "partial", "none", None, "partial_f32", None),
              ("ar", "h1", "down_bias", "h2", "st2")]
    return steps


class TPDecodeStep:
    """The same chain for one rank of a tensor-parallel group (Megatron split, utils/tp.py): qkv and gate_up are this rank's
    COLUMN shards (gate_up interleaved per shard), o and down its ROW shards.  Per layer 4 decode ops + 2 one-shot xGMI
    all-reduces (utils/xgmi_allreduce.py), every launch capture-safe, no host involvement between them:

        qkv_r = rmsnorm(h; w_in) @ Wqkv[:, shard r]            in_glue RMSNORM (stats from the previous all-reduce)
        p     = a_r @ Wo[shard r, :]                           OUT_PARTIAL_F32: unrounded fp32 partial sums
        h1    = h + act(sum_r p) (+ bias)                      all-reduce kernel: rank-ordered sum, rounding chain, residual,
                                                               per-tile sums of h1^2 for the next RMSNorm
        a_r   = silu(g_r) * u_r                                gate_up shard with the paired epilogue
        p     = a_r @ Wdown[shard r, :];  h2 = h1 + act(sum_r p)

    The rounding points are those of the reference's single-GPU module chain (round the full linear output once, then add the
    residual), so tp=N differs from tp=1 only by the fp32 association of the K-shards' partial sums.  Every rank holds the
    same h1 / h2 bit for bit (the reduction order is the rank order on every rank).

    Act-order (desc_act=True) checkpoints (the rule of gptqmodel/utils/marlin.py:296-305,368-372 -- row shards
    are cut from the globally group-sorted rows): the column shards keep their input permutation and the decode op applies it in
    the kernel; down_proj's permutation is folded into the column OWNERSHIP of gate / up (utils.tp.shard_mlp_act_order: exact, no
    exchange); o_proj's rows need attention-output features owned by other ranks, so ONE one-shot all-gather + select kernel
    (gptqhip_allgather_select) sits in front of it (DecodeLayer.o_input_index) -- 80 extra 16 KB exchanges per 70B token instead
    of the module path's all_gather + index_select per layer."""

    def __init__(self, layers: Sequence[DecodeLayer], hidden: int, q_dim_local: int, dtype: torch.dtype, comm, eps: float = 1e-5,
                 device: Optional[torch.device] = None):
        if not layers:
            raise ValueError("no layers")
        if hidden % 16 != 0 or hidden > comm.n_max:
            raise ValueError("hidden must be a multiple of 16 and fit the communicator")
        self.layers, self.comm, self.dtype = list(layers), comm, dtype
        self.device = dev = device or layers[0].qkv.qweight.device
        self.x_in = torch.zeros(hidden, dtype=dtype, device=dev)
        self.h = torch.zeros((len(self.layers), 2, hidden), dtype=dtype, device=dev)
        self.stats = torch.zeros((len(self.layers), 2, hidden // 16), dtype=torch.float32, device=dev)
        self.qkv_out = torch.zeros(layers[0].qkv.out_features, dtype=dtype, device=dev)
        self.gu_out = torch.zeros(layers[0].gate_up.out_features, dtype=dtype, device=dev)
        self.partial = torch.zeros(hidden, dtype=torch.float32, device=dev)
        need = 0
        for L in self.layers:
            for lin in (L.qkv, L.o, L.gate_up, L.down):
                need = max(need, ops.workspace_bytes(1, lin.in_features, lin.out_features, lin.group_size, getattr(lin, "kernel_bits", lin.bits), False))
        self.workspace = torch.zeros(max(need, 1 << 20), dtype=torch.uint8, device=dev)
        self._keep, self.steps = [], []   # steps: ("op", struct) | ("ar", residual, bias, out, stats_out) | ("ag", x_local, index, out)
        self.o_in = None                  # act-order o_proj shards: the gathered + selected attention-output features

        def bind(lin, x, out, glue, nw, oglue, s_in):
            qw, meta, bias, sdt, perm = _lin_tensors(lin, dtype)
            K, N = lin.in_features, lin.out_features
            if oglue == ops.OUT_PARTIAL_F32 and perm is not None:
                raise NotImplementedError("TPDecodeStep: a row-parallel shard must have sequential groups -- cut it with "
                                          "utils.tp.shard_gptq_row(act_order='global_sort') / shard_mlp_act_order")
            if oglue == ops.OUT_PARTIAL_F32 and bias is not None:
                raise ValueError("row-parallel shards must not carry a bias: pass the layer bias as DecodeLayer.o_bias / down_bias")
            if not ops.decode_supported(K, N, lin.group_size, perm is not None):
                raise NotImplementedError(f"decode chain: shard shape K={K} N={N} group_size={lin.group_size} unsupported")
            self._keep.extend([qw, meta, bias, nw, perm])
            self.steps.append(("op", ops.make_decode_op(x, qw, meta, bias, out, K, N, lin.group_size, getattr(lin, "kernel_bits", lin.bits), sdt, in_glue=glue,
                                                        norm_weight=nw, eps=eps, workspace=self.workspace, out_glue=oglue,
                                                        stats_in=s_in, perm=perm)))

        glue = {"rmsnorm": ops.GLUE_RMSNORM, "none": ops.GLUE_NONE}
        oglue = {"none": ops.OUT_NONE, "partial_f32": ops.OUT_PARTIAL_F32, "silu_mul_paired": ops.OUT_SILU_MUL_PAIRED}
        h_in, st_in = self.x_in, None
        for li, L in enumerate(self.layers):
            if not getattr(L.gate_up, "gate_up_interleaved", False):
                raise ValueError("TPDecodeStep needs gate_up shards fused with fuse_gate_up_interleaved")
            if L.o.in_features != q_dim_local or L.o.out_features != hidden or L.down.out_features != hidden:
                raise ValueError("row-parallel shard shapes do not match hidden / q_dim_local")
            buf = {"h_in": h_in, "st_in": st_in, "qkv_out": self.qkv_out, "qkv_out_local": self.qkv_out[:q_dim_local] if L.o_input_index is not None
                   else self.qkv_out, "partial": self.partial, "h1": self.h[li, 0], "st1": self.stats[li, 0], "gu_out": self.gu_out,
                   "h2": self.h[li, 1], "st2": self.stats[li, 1], "w_in": L.input_norm, "w_post": L.post_norm, "o_bias": L.o_bias,
                   "down_bias": L.down_bias, None: None}
            lins = {"qkv": L.qkv, "o": L.o, "gate_up": L.gate_up, "down": L.down}
            if L.o_input_index is not None:
                # act-order o_proj shard: its sorted rows consume features of the FULL attention output
                idx = L.o_input_index.to(device=dev, dtype=torch.int32).contiguous()
                if idx.numel() != q_dim_local:
                    raise ValueError("o_input_index must select q_dim_local features")
                if self.o_in is None:
                    self.o_in = torch.zeros(q_dim_local, dtype=dtype, device=dev)
                self._keep.append(idx)
                buf["o_input_index"], buf["o_in"] = idx, self.o_in
            for st in tp_layer_plan(L.o_input_index is not None):
                if st[0] == "op":
                    _, which, x, out, g, norm, og, s_in = st
                    bind(lins[which], buf[x], buf[out], glue[g], buf[norm], oglue[og], buf[s_in])
                elif st[0] == "ag":
                    self.steps.append(("ag", buf[st[1]], buf[st[2]], buf[st[3]]))
                else:
                    self.steps.append(("ar", buf[st[1]], buf[st[2]], buf[st[3]], buf[st[4]]))
            h_in, st_in = buf["h2"], buf["st2"]
        self.out = h_in

    def run(self) -> torch.Tensor:
        with torch.cuda.device(self.device):
            for st in self.steps:
                if st[0] == "op":
                    ops.launch_decode_op(st[1], self.device)
                elif st[0] == "ag":
                    self.comm.gather_select(st[1], st[2], out=st[3])
                else:
                    _, res, bias, out, stats = st
                    self.comm(self.partial, self.dtype, bias=bias, residual=res, out=out, stats_out=stats)
        return self.out

    def check(self) -> None:
        """Host-side health check at a synchronisation point (not inside a captured region): raises when a peer wait of this
        rank's communicator timed out (the affected outputs were poisoned with NaN by the kernel)."""
        self.comm.check_status()


__all__ = ["DecodeLayer", "DecodeStep", "TPDecodeStep", "tp_layer_plan"]
