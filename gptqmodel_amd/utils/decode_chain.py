"""Batch-1 decode step over a stack of Llama-style decoder layers as a CHAIN of decode ops (gptqhip_decode_linear).

The reference runs a decode step as HF's module graph: per layer 7 QuantLinear.forward calls (torch.py:302-347) with
RMSNorm / SiLU*mul / residual-add torch kernels between them.  On MI355X the per-launch fixed cost (dependent-kernel
boundary + ramp + tail, ~3.5 us) is ~45 % of such a token (profiles/r01_*).  This helper builds the same computation from
the modules' already-relayouted tensors as 4 ops per layer -- sibling projections fused (utils.model.fuse_siblings), glue
fused into the GEMV (csrc/gptqhip_gemv1.hip) -- and enqueues them either
  * serial:  one stream, ordinary stream-ordered launches, or
  * overlap: even ops on one stream, odd ops on a second; op i+1 prefetches its packed weights while op i computes and
    waits on op i's device-side arrival counters, so the HBM stream does not stop at op boundaries.  Both modes run the
    same kernels in the same arithmetic order: their results are bit-identical.
All ops are bound once (raw-pointer structs); a step is 4*L ctypes calls, or one HIP graph replay after capture.

Per layer (h = residual stream [hidden], activation dtype):
    qkv = rmsnorm(h; w_in) @ Wqkv                       in_glue RMSNORM
    a   = attention(qkv)                                 NOT a quantised linear: `attention` hook, or the stand-in "a = q"
    h1  = h + a @ Wo                                     residual epilogue
    gu  = rmsnorm(h1; w_post) @ Wgate_up                 in_glue RMSNORM
    h2  = h1 + (silu(gate) * up) @ Wdown                 in_glue SILU_MUL + residual epilogue
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .. import ops


class DecodeLayer:
    """The four (fused) quantised linears of one decoder layer + its two RMSNorm weights."""

    def __init__(self, qkv, o, gate_up, down, input_norm: torch.Tensor, post_norm: torch.Tensor):
        self.qkv, self.o, self.gate_up, self.down = qkv, o, gate_up, down
        self.input_norm, self.post_norm = input_norm, post_norm


def _lin_tensors(lin, dtype):
    """(qweight_t, meta, bias, scale_dtype) of a post_init()ed HIP QuantLinear for activations of `dtype`."""
    if not getattr(lin, "_ready", False):
        raise RuntimeError("DecodeStep needs post_init()ed HIP QuantLinear modules")
    if getattr(lin, "perm", None) is not None:
        raise NotImplementedError("act-order (desc_act) modules are not supported by the decode chain; use forward()")
    if hasattr(lin, "_runtime"):  # HipAwqLinear: constants depend on the compute dtype
        meta, bias = lin._runtime(dtype)
        return lin.qweight, meta, bias, dtype
    bias = lin._bias_for(dtype, lin.qweight.device)
    return lin.qweight, lin.meta, bias, lin._scale_dtype


class DecodeStep:
    def __init__(self, layers: Sequence[DecodeLayer], hidden: int, q_dim: int, dtype: torch.dtype, eps: float = 1e-5,
                 overlap: bool = True, device: Optional[torch.device] = None):
        if not layers:
            raise ValueError("no layers")
        self.layers = list(layers)
        self.device = device or layers[0].qkv.qweight.device
        self.dtype = dtype
        self.overlap = overlap
        dev = self.device
        inter2 = layers[0].gate_up.out_features
        qkv_n = layers[0].qkv.out_features
        self.x_in = torch.zeros(hidden, dtype=dtype, device=dev)      # the step's input (embedding of the new token)
        # residual stream: h1 / h2 of every layer get their own 2*hidden*2-byte buffers (no reuse, no WAR reasoning)
        self.h = torch.zeros((len(self.layers), 2, hidden), dtype=dtype, device=dev)
        self.qkv_out = torch.zeros(qkv_n, dtype=dtype, device=dev)
        self.gu_out = torch.zeros(inter2, dtype=dtype, device=dev)
        n_ops = 4 * len(self.layers)
        self.counters = torch.zeros((n_ops, ops.COUNTER_SHARDS), dtype=torch.int32, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._keep = []   # tensors the raw-pointer structs refer to
        self.ops: List = []
        h_in = self.x_in
        prev_blocks = 0
        for li, L in enumerate(self.layers):
            h1, h2 = self.h[li, 0], self.h[li, 1]
            plan = [
                (L.qkv, h_in, self.qkv_out, ops.GLUE_RMSNORM, L.input_norm, None),
                (L.o, self.qkv_out, h1, ops.GLUE_NONE, None, h_in),           # stand-in attention: a = q = qkv[:q_dim]
                (L.gate_up, h1, self.gu_out, ops.GLUE_RMSNORM, L.post_norm, None),
                (L.down, self.gu_out, h2, ops.GLUE_SILU_MUL, None, h1),
            ]
            for j, (lin, x, out, glue, nw, res) in enumerate(plan):
                i = 4 * li + j
                qw, meta, bias, sdt = _lin_tensors(lin, dtype)
                K, N = lin.in_features, lin.out_features
                if j == 1 and K != q_dim:
                    raise ValueError("o_proj in_features must equal q_dim")
                blocks = ops.decode_blocks(K, N, lin.group_size)
                if blocks == 0:
                    raise NotImplementedError(f"decode chain: layer shape K={K} N={N} group_size={lin.group_size} unsupported")
                wait = self.counters[i - 1] if (overlap and i > 0) else None
                signal = self.counters[i] if (overlap and i + 1 < n_ops) else None
                self._keep.extend([qw, meta, bias, nw])
                self.ops.append(ops.make_decode_op(x, qw, meta, bias, out, K, N, lin.group_size, lin.bits, sdt, in_glue=glue,
                                                   norm_weight=nw, eps=eps, residual=res, wait=wait, wait_total=prev_blocks,
                                                   signal=signal, status=self.status))
                prev_blocks = blocks
            h_in = h2
        self.out = h_in
        self._side = torch.cuda.Stream(device=dev) if overlap else None

    def run(self) -> torch.Tensor:
        """Enqueue one decode step on the current stream (and, in overlap mode, a forked side stream that is joined
        before returning).  Capture-safe: wrap in torch.cuda.graph() to replay a token as one graph launch."""
        with torch.cuda.device(self.device):
            if not self.overlap:
                for op in self.ops:
                    ops.launch_decode_op(op, self.device)
                return self.out
            main = torch.cuda.current_stream(self.device)
            self.counters.zero_()                       # memset node: arrival counters start every step at 0
            self._side.wait_stream(main)
            for i, op in enumerate(self.ops):
                if i % 2 == 0:
                    ops.launch_decode_op(op, self.device)
                else:
                    with torch.cuda.stream(self._side):
                        ops.launch_decode_op(op, self.device)
            main.wait_stream(self._side)
            return self.out

    def check_status(self) -> None:
        """Raises if a bounded dependency spin ever gave up (host sync: call outside the hot loop)."""
        if int(self.status.item()) != 0:
            raise RuntimeError("decode chain: a dependency wait timed out (status word set); results are invalid")


__all__ = ["DecodeLayer", "DecodeStep"]
