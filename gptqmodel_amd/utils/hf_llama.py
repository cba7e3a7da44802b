"""Llama-family decoder layers on the decode ops (the callers of the quantised linears are HF's
`LlamaAttention` / `LlamaMLP`; this is the optional module-graph rewrite around `gptqmodel_post_init`).

`fuse_siblings` (utils/model.py) already cuts a layer from 7 to 4 quantised launches.  At batch 1 the rest of the layer's
non-attention work is glue between those launches -- two RMSNorms (6 small kernels each in eager HF), SiLU, the gate*up
product and two residual adds: ~16 dependent launches per layer.  `fuse_llama_decoder_layers` keeps HF's module tree but

  * folds an act-order checkpoint's down_proj input permutation into gate / up's output columns (utils.model.
    fold_act_order_into_producers: exact, removes down_proj's activation gather), fuses q/k/v into one module (views stand in for the three projections, as with fuse_siblings) and gate/up into ONE
    module with the columns interleaved in blocks of 8 (utils.model.fuse_gate_up_interleaved);
  * gives every decoder layer a decode fast path: when the layer is called with at most SIXTEEN tokens (batch x q_len <= 16, eval:
    single-sequence decode, a few sequences, or speculative tokens of one), the layer runs as 4 decode ops (gptqhip_decode_linear: RMSNorm on the input of qkv / gate_up with the statistics handed
    over by the op that produced the residual stream, SiLU*mul in the gate_up epilogue, residual add + next statistics in
    the o / down epilogue) around HF's own rotary / KV-cache update / attention call;
  * gives it a prefill path for more tokens than that (eval, no grad): the two RMSNorms run as ONE HIP kernel each
    (ops.rmsnorm_gather instead of eager HF's six small kernels with fp32 temporaries) that, for act-order checkpoints, writes
    the normalised activations already in the q|k|v / gate|up kernels' row order -- those two linears then run without their
    own x-gather pass (o_proj keeps its pre-pass: its input comes from attention; down_proj's was folded away at load time).
    Training-mode calls, calls with forward hooks on the layer's sub-modules and anything unexpected take HF's original path
    through the same fused modules.

The fast path binds raw pointers once per layer (buffers owned by the layer), so a decode step is capture-safe and costs 4
ctypes calls per layer on the host.  It mirrors `LlamaAttention.forward` / `LlamaDecoderLayer.forward` of transformers >= 4.56 /
5.x: `past_key_values` keyword, `Cache.update(k, v, layer_idx)`, the decoder layer returning a bare tensor.
`fuse_llama_decoder_layers` VERIFIES that contract on the installed classes (signature + source inspection, `_hf_contract`)
and leaves a layer untouched (returns it in `skipped` with the reason) when anything does not match -- older releases (4.48 -
4.55: `past_key_value` in **kwargs, tuple return, cache_kwargs) would otherwise run with a silently stale KV cache.

Aliasing: the fast path returns a view of a per-layer buffer that the next decode step overwrites; when the decoder layer
itself carries forward hooks (transformers records `output_hidden_states` through them) a clone is returned instead.

Reference behaviour preserved: each op's arithmetic is the reference's TorchLinear.forward chain (torch.py:326-347) composed
with HF's LlamaRMSNorm / LlamaMLP / residual formulas -- tests/test_gpu_e2e_llama.py compares logits with the unfused model.
"""
from __future__ import annotations

import os
import types
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..nn_modules.qlinear import BaseQuantLinear
from .model import (FusedSiblingView, _FusedGroup, fold_act_order_into_producers, fuse_gate_up_interleaved,
                    fuse_quant_linears)


def _is_quant(m) -> bool:
    # the fast paths call the kernels directly (not module.forward): modules whose forward does more than the GEMM -- an adapter,
    # an online Hadamard rotation of the input (qlinear/__init__.py:134-135) -- keep HF's own layer code
    # (the marker, not isinstance: the overlay classes sit on the REFERENCE's GPTQQuantLinear / AWQuantLinear, not on this package's mirror)
    return ((isinstance(m, BaseQuantLinear) or getattr(type(m), "_GPTQHIP_KERNEL_CLASS", False)) and hasattr(m, "qweight")
            and getattr(m, "adapter", None) is None
            and not getattr(m, "online_full_had", False) and not getattr(m, "online_partial_had", False))


MAX_ROWS = 16   # tokens per call on the fast path (gptqhip_decode_op.M): a few sequences at q_len 1, or speculative tokens of one


class _LayerDecodeState:
    """Buffers + bound decode ops of one decoder layer (built lazily on the first fast-path call; ops per row count M)."""

    def __init__(self, layer, prev: Optional["_LayerDecodeState"], dtype: torch.dtype, workspace: torch.Tensor):
        attn, mlp = layer.self_attn, layer.mlp
        self.lins = (attn.fused_q_proj_k_proj_v_proj.fused, attn.o_proj, mlp.fused_gate_up.fused, mlp.down_proj)
        qkv, o, gu, down = self.lins
        dev = qkv.qweight.device
        hidden = qkv.in_features
        self.hidden, self.dtype, self.device, self.workspace = hidden, dtype, dev, workspace
        self.q_dim, self.kv_dim = attn.fused_q_proj_k_proj_v_proj.sizes[0], attn.fused_q_proj_k_proj_v_proj.sizes[1]
        R = MAX_ROWS
        self.x_in = torch.zeros((R, hidden), dtype=dtype, device=dev)          # used when the input is not the previous layer's h2
        self.qkv_out = torch.zeros((R, qkv.out_features), dtype=dtype, device=dev)
        self.attn_in = torch.zeros((R, o.in_features), dtype=dtype, device=dev)  # attention output, copied in (HF returns a fresh tensor)
        self.h1 = torch.zeros((R, hidden), dtype=dtype, device=dev)
        self.h2 = torch.zeros((R, hidden), dtype=dtype, device=dev)
        self.act = torch.zeros((R, gu.out_features // 2), dtype=dtype, device=dev)
        self.st1 = torch.zeros((R, -(-hidden // 16)), dtype=torch.float32, device=dev)
        self.st2 = torch.zeros((R, -(-hidden // 16)), dtype=torch.float32, device=dev)
        self.prev = prev
        self.eps_in = float(getattr(layer.input_layernorm, "variance_epsilon", 1e-6))
        self.eps_post = float(getattr(layer.post_attention_layernorm, "variance_epsilon", 1e-6))
        self.w_in = layer.input_layernorm.weight.detach().to(dtype).contiguous()
        self.w_post = layer.post_attention_layernorm.weight.detach().to(dtype).contiguous()
        self._keep = []
        self.ops = {}      # M -> (qkv_chain | None, qkv_first, o_chain | None, o_first, gate_up, down)

    def supported(self, M: int) -> bool:
        for lin in self.lins:
            if not ops.decode_supported(lin.in_features, lin.out_features, lin.group_size, getattr(lin, "perm", None) is not None, M):
                return False
        return True

    def ops_for(self, M: int):
        """The six bound ops for M rows (rows are packed at the front of the [MAX_ROWS, ..] buffers: row stride = feature
        count, so the M-row views are contiguous)."""
        if M in self.ops:
            return self.ops[M]
        qkv, o, gu, down = self.lins
        prev, dtype, ws = self.prev, self.dtype, self.workspace

        def bind(lin, x, out, **kw):
            from .decode_chain import _lin_tensors
            qw, meta, bias, sdt, perm = _lin_tensors(lin, dtype)
            self._keep.extend([qw, meta, bias, perm])
            return ops.make_decode_op(x, qw, meta, bias, out, lin.in_features, lin.out_features, lin.group_size, getattr(lin, "kernel_bits", lin.bits), sdt,
                                      workspace=ws, perm=perm, M=M, **kw)

        # qkv / o come in two flavours: chained to the previous layer's h2 (+ its statistics), or fed from x_in (first layer, or
        # whenever the caller hands over some other tensor)
        rms_in = dict(in_glue=ops.GLUE_RMSNORM, norm_weight=self.w_in, eps=self.eps_in)
        chain_ok = prev is not None
        bound = (
            bind(qkv, prev.h2, self.qkv_out, stats_in=prev.st2, **rms_in) if chain_ok else None,
            bind(qkv, self.x_in, self.qkv_out, **rms_in),
            bind(o, self.attn_in, self.h1, residual=prev.h2, stats_out=self.st1) if chain_ok else None,
            bind(o, self.attn_in, self.h1, residual=self.x_in, stats_out=self.st1),
            bind(gu, self.h1, self.act, in_glue=ops.GLUE_RMSNORM, norm_weight=self.w_post, eps=self.eps_post, stats_in=self.st1,
                 out_glue=ops.OUT_SILU_MUL_PAIRED),
            bind(down, self.act, self.h2, residual=self.h1, stats_out=self.st2),
        )
        # o_proj -> gate_up -> down_proj run back to back: one host call each way
        tail_chain = ops.bind_decode_seq([bound[2], bound[4], bound[5]]) if chain_ok else None
        tail_first = ops.bind_decode_seq([bound[3], bound[4], bound[5]])
        self.ops[M] = bound + (tail_chain, tail_first)
        return self.ops[M]


def _layer_forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_values=None, use_cache=False,
                   position_embeddings=None, **kwargs):
    """LlamaDecoderLayer.forward with the few-token fast path in front of HF's original."""
    fd = self._gptqhip_fused

    def original():
        return fd["orig_forward"](hidden_states, attention_mask=attention_mask, position_ids=position_ids,
                                  past_key_values=past_key_values, use_cache=use_cache,
                                  position_embeddings=position_embeddings, **kwargs)

    hidden = fd["hidden"]
    M = hidden_states.numel() // hidden
    if (M < 1 or hidden_states.dim() != 3 or hidden_states.shape[-1] != hidden or self.training
            or not hidden_states.is_cuda or position_embeddings is None or fd["disabled"]
            or (torch.is_grad_enabled() and hidden_states.requires_grad)
            or kwargs.get("output_attentions") or _has_hooks(self.self_attn, self.mlp, self.input_layernorm, self.post_attention_layernorm)
            or "attentions" in _capture_keys(self.self_attn)):
        # (user hooks / attention-weight capture on the sub-modules the fast paths bypass: HF's own code runs them)
        return original()
    if M > MAX_ROWS:
        if fd["prefill"] is not None and hidden_states.dtype in (torch.float16, torch.bfloat16):
            return _prefill_forward(self, fd, hidden_states, attention_mask, past_key_values, position_embeddings, kwargs)
        return original()
    st: Optional[_LayerDecodeState] = fd["state"]
    if st is None or st.dtype != hidden_states.dtype:
        if not all(getattr(lin, "_ready", False) for lin in (self.self_attn.fused_q_proj_k_proj_v_proj.fused, self.self_attn.o_proj,
                                                             self.mlp.fused_gate_up.fused, self.mlp.down_proj)) \
                or hidden % 16 != 0 or hidden // 16 > 512:
            fd["disabled"] = True
            return original()
        prev_layer = fd["prev"]
        prev_state = None
        if prev_layer is not None:
            prev_state = prev_layer._gptqhip_fused["state"]
            if prev_state is not None and prev_state.dtype != hidden_states.dtype:
                prev_state = None
        st = fd["state"] = _LayerDecodeState(self, prev_state, hidden_states.dtype, fd["workspace"]())
    if M not in st.ops and not st.supported(M):
        if M == 1:
            fd["disabled"] = True
        return original()
    op_qkv_chain, op_qkv_first, _, _, _, _, tail_chain, tail_first = st.ops_for(M)
    dev = hidden_states.device
    attn = self.self_attn
    bsz, q_len = hidden_states.shape[0], hidden_states.shape[1]
    chained = op_qkv_chain is not None and hidden_states.data_ptr() == st.prev.h2.data_ptr()
    with torch.cuda.device(dev):
        if chained:
            ops.launch_decode_op(op_qkv_chain, dev)
        else:
            st.x_in[:M].copy_(hidden_states.reshape(M, hidden))
            ops.launch_decode_op(op_qkv_first, dev)
        # ---- HF's attention between the projections (LlamaAttention.forward, projections removed) ----------------------------
        q_dim, kv_dim, hd = st.q_dim, st.kv_dim, attn.head_dim
        qkv = st.qkv_out[:M]
        q = qkv[:, :q_dim].reshape(bsz, q_len, -1, hd)
        k = qkv[:, q_dim:q_dim + kv_dim].reshape(bsz, q_len, -1, hd)
        if fd["qk_norm"]:          # Qwen3-style per-head RMSNorm on q / k (HF's own modules)
            q, k = attn.q_norm(q), attn.k_norm(k)
        q, k = q.transpose(1, 2), k.transpose(1, 2)
        v = qkv[:, q_dim + kv_dim:].reshape(bsz, q_len, -1, hd).transpose(1, 2)
        cos, sin = position_embeddings
        q, k = fd["rotary"](q, k, cos, sin)
        if past_key_values is not None:
            k, v = past_key_values.update(k, v, attn.layer_idx)
        interface = fd["interfaces"].get_interface(attn.config._attn_implementation, fd["eager_attention"])
        window = _sliding_window(attn) if fd["sliding_window"] else None
        if window is not None:     # Mistral / Qwen2 / Qwen3 hand their window to the attention interface
            kwargs = dict(kwargs, sliding_window=window)
        attn_out, _ = interface(attn, q, k, v, attention_mask, dropout=0.0, scaling=attn.scaling, **kwargs)
        st.attn_in[:M].copy_(attn_out.reshape(M, -1))
        # ---- o_proj + residual, MLP ---------------------------------------------------------------------------------------------
        ops.launch_decode_seq(tail_chain if chained else tail_first, dev)
    out = st.h2[:M].view(hidden_states.shape)
    # a view of the layer's persistent buffer: anything that RECORDS it across steps (transformers captures
    # output_hidden_states through forward hooks on the decoder layer) must get its own copy
    return out.clone() if (_has_hooks(self) or "hidden_states" in _capture_keys(self)) else out


def _is_hf_capture_hook(fn) -> bool:
    # transformers >= 4.56 installs its output recorders as PERMANENT forward hooks on every decoder layer / attention module the
    # first time the model runs (utils/output_capturing.py); they are inert unless a collector is active (_capture_keys)
    return getattr(fn, "__name__", "") == "output_capturing_hook" and "output_capturing" in getattr(fn, "__module__", "")


def _has_hooks(*mods) -> bool:
    """Forward (pre-)hooks other than transformers' own inert output recorders."""
    return any(m._forward_pre_hooks or any(not _is_hf_capture_hook(h) for h in m._forward_hooks.values()) for m in mods)


_COLLECTOR = [False]   # transformers' active-output collector (utils/output_capturing.py), resolved on first use; None: not available


def _capture_keys(module):
    """Output kinds transformers is recording during this call ("hidden_states", "attentions", ...): the keys of its active
    collector; when that machinery is not importable (other releases), every kind a recorder hook on `module` could serve."""
    if _COLLECTOR[0] is False:
        try:
            from transformers.utils.output_capturing import _active_collector
            _COLLECTOR[0] = _active_collector
        except Exception:  # noqa: BLE001
            _COLLECTOR[0] = None
    if _COLLECTOR[0] is None:
        return ("hidden_states", "attentions") if module._forward_hooks else ()
    try:
        col = _COLLECTOR[0].get()
        return () if col is None else tuple(col.keys())
    except Exception:  # noqa: BLE001
        return ("hidden_states", "attentions") if module._forward_hooks else ()


def _sliding_window(attn):
    """The window HF's own attention forward hands to the attention interface: Qwen2 / Qwen3 keep it as `self.sliding_window`,
    Mistral reads `getattr(self.config, "sliding_window", None)` (modeling_mistral.py), Llama passes none."""
    if hasattr(attn, "sliding_window"):
        return attn.sliding_window
    return getattr(attn.config, "sliding_window", None)


def _prefill_forward(self, fd, hidden_states, attention_mask, past_key_values, position_embeddings, kwargs):
    """LlamaDecoderLayer.forward for more than MAX_ROWS tokens (eval): HF's layer with the two RMSNorms as one HIP kernel each,
    fused with the act-order gather of the projection that consumes them (module docstring)."""
    attn, mlp = self.self_attn, self.mlp
    qkv_lin, gu_lin = attn.fused_q_proj_k_proj_v_proj.fused, mlp.fused_gate_up.fused
    group = attn.fused_q_proj_k_proj_v_proj
    dt = hidden_states.dtype
    pf = fd["prefill"]
    if pf.get("dtype") != dt:
        pf.update(dtype=dt, w_in=self.input_layernorm.weight.detach().to(dt).contiguous(),
                  w_post=self.post_attention_layernorm.weight.detach().to(dt).contiguous(),
                  eps_in=float(getattr(self.input_layernorm, "variance_epsilon", 1e-6)),
                  eps_post=float(getattr(self.post_attention_layernorm, "variance_epsilon", 1e-6)))
    bsz, q_len = hidden_states.shape[0], hidden_states.shape[1]
    h = hidden_states.contiguous()
    xn = ops.rmsnorm_gather(h, pf["w_in"], pf["eps_in"], getattr(qkv_lin, "perm", None))
    qkv = qkv_lin.forward_pregathered(xn)
    del xn
    q_dim, kv_dim, hd = group.sizes[0], group.sizes[1], attn.head_dim
    q = qkv[..., :q_dim].reshape(bsz, q_len, -1, hd)
    k = qkv[..., q_dim:q_dim + kv_dim].reshape(bsz, q_len, -1, hd)
    if fd["qk_norm"]:
        q, k = attn.q_norm(q), attn.k_norm(k)
    q, k = q.transpose(1, 2), k.transpose(1, 2)
    v = qkv[..., q_dim + kv_dim:].reshape(bsz, q_len, -1, hd).transpose(1, 2)
    cos, sin = position_embeddings
    q, k = fd["rotary"](q, k, cos, sin)
    if past_key_values is not None:
        k, v = past_key_values.update(k, v, attn.layer_idx)
    interface = fd["interfaces"].get_interface(attn.config._attn_implementation, fd["eager_attention"])
    window = _sliding_window(attn) if fd["sliding_window"] else None
    if window is not None:
        kwargs = dict(kwargs, sliding_window=window)
    attn_out, _ = interface(attn, q, k, v, attention_mask, dropout=0.0, scaling=attn.scaling, **kwargs)
    del qkv, q, k, v
    h1 = h + attn.o_proj(attn_out.reshape(bsz, q_len, -1).contiguous())
    xn = ops.rmsnorm_gather(h1, pf["w_post"], pf["eps_post"], getattr(gu_lin, "perm", None))
    y = gu_lin.forward_pregathered(xn)
    del xn
    yv = y.view(*y.shape[:-1], y.shape[-1] // 16, 2, 8)
    a = mlp.act_fn(yv[..., 0, :]) * yv[..., 1, :]
    del y, yv
    return h1 + mlp.down_proj(a.reshape(bsz, q_len, -1))


def _hf_contract(layer) -> Optional[str]:
    """None when the installed transformers' decoder layer / attention follow the calling convention the fast paths mirror
    (module docstring), else the reason.  Signature + source inspection of the CLASSES (the instance's forward may already be
    wrapped by accelerate / hooks)."""
    import inspect
    try:
        lsig = inspect.signature(type(layer).forward).parameters
        asig = inspect.signature(type(layer.self_attn).forward).parameters
        lsrc = inspect.getsource(type(layer).forward)
        asrc = inspect.getsource(type(layer.self_attn).forward)
    except (OSError, TypeError, ValueError) as e:
        return f"cannot inspect the installed transformers classes ({e})"
    for name in ("past_key_values", "position_embeddings"):
        if name not in lsig or name not in asig:
            return f"transformers' {type(layer).__name__}/{type(layer.self_attn).__name__}.forward take no `{name}` keyword (transformers >= 4.56 needed)"
    if "return hidden_states" not in lsrc or "outputs = (hidden_states,)" in lsrc or "return outputs" in lsrc:
        return f"{type(layer).__name__}.forward does not return the bare hidden-states tensor (transformers >= 4.56 needed)"
    if "cache_kwargs" in asrc or ".update(key_states, value_states, self.layer_idx)" not in asrc:
        return f"{type(layer.self_attn).__name__}.forward does not call Cache.update(k, v, layer_idx) (cache_kwargs / StaticCache-era API)"
    return None


def _mlp_forward(self, x):
    """LlamaMLP.forward through the interleaved gate|up module (prefill / batched path).  gate and up are read as strided
    views of the fused output ([.., inter/8, 2, 8]): no de-interleaving copies, the same two elementwise kernels as HF's
    `act_fn(gate) * up`."""
    y = self.fused_gate_up.fused(x)
    v = y.view(*y.shape[:-1], y.shape[-1] // 16, 2, 8)
    a = self.act_fn(v[..., 0, :]) * v[..., 1, :]
    return self.down_proj(a.reshape(*y.shape[:-1], y.shape[-1] // 2))


# decoder layers whose single-token arithmetic IS the Llama formula the fast path implements (RMSNorm `w * act(x * rsqrt(mean x^2
# + eps))`, rotary on q / k after the projections (Qwen3: after its per-head q_norm / k_norm, applied through HF's own modules),
# SiLU MLP, plain residual adds).  Architectures with (1 + w) norms or GELU (Gemma), post-norms (OLMo2), parallel residuals etc.
# are NOT in this list and are left untouched.
_KNOWN_ATTENTION = ("LlamaAttention", "MistralAttention", "Qwen2Attention", "Qwen3Attention")
_KNOWN_NORMS = ("LlamaRMSNorm", "MistralRMSNorm", "Qwen2RMSNorm", "Qwen3RMSNorm")


def fuse_llama_decoder_layers(model: nn.Module, allow_unknown: bool = False) -> Tuple[List[nn.Module], List[Tuple[nn.Module, str]]]:
    """Call BEFORE gptqmodel_post_init (the modules must still be in the checkpoint layout).  Returns (fused layers, skipped
    [(layer, reason)]).  Handles HF decoder layers of the Llama / Mistral / Qwen2 families (q/k/v biases are carried by the
    fused module); `allow_unknown=True` also takes look-alike layers of other classes -- only when you know their
    single-token arithmetic is Llama's."""
    try:
        from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
        import transformers.models.llama.modeling_llama as hf_llama
        rotary, eager_attention = hf_llama.apply_rotary_pos_emb, hf_llama.eager_attention_forward
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(f"fuse_llama_decoder_layers needs transformers' attention-interface API: {e}") from e
    layers = [m for m in model.modules() if all(hasattr(m, a) for a in ("self_attn", "mlp", "input_layernorm", "post_attention_layernorm"))]
    ws_holder = {}

    def workspace():
        # one scratch for every layer's ops (cross-block split-K slabs + arrival counters, zero-initialised; ops are serialised)
        if "t" not in ws_holder:
            need = 1 << 20
            for L in fused:
                for lin in (L.self_attn.fused_q_proj_k_proj_v_proj.fused, L.self_attn.o_proj, L.mlp.fused_gate_up.fused, L.mlp.down_proj):
                    need = max(need, ops.workspace_bytes(1, lin.in_features, lin.out_features, lin.group_size, getattr(lin, "kernel_bits", lin.bits), False))
            ws_holder["t"] = torch.zeros(need, dtype=torch.uint8, device=fused[0].self_attn.o_proj.qweight.device)
        return ws_holder["t"]

    fused: List[nn.Module] = []
    skipped: List[Tuple[nn.Module, str]] = []
    prev = None
    for layer in layers:
        attn, mlp = layer.self_attn, layer.mlp
        names_a, names_m = ("q_proj", "k_proj", "v_proj", "o_proj"), ("gate_proj", "up_proj", "down_proj")
        if not all(_is_quant(getattr(attn, n, None)) for n in names_a) or not all(_is_quant(getattr(mlp, n, None)) for n in names_m):
            skipped.append((layer, "projections are not (adapter-free) HIP quant modules"))
            prev = None
            continue
        if any(getattr(getattr(p, n), "_ready", False) for p, ns in ((attn, names_a), (mlp, names_m)) for n in ns):
            skipped.append((layer, "already post_init()ed"))
            prev = None
            continue
        act = getattr(mlp, "act_fn", None)
        if not (isinstance(act, nn.SiLU) or type(act).__name__ in ("SiLU", "SiLUActivation")):
            skipped.append((layer, "MLP activation is not SiLU"))
            prev = None
            continue
        if not all(hasattr(attn, a) for a in ("head_dim", "scaling", "layer_idx", "config")):
            skipped.append((layer, "attention module does not follow the attention-interface layout"))
            prev = None
            continue
        why = _hf_contract(layer)
        if why is not None:
            skipped.append((layer, why))
            prev = None
            continue
        known = (type(attn).__name__ in _KNOWN_ATTENTION and type(layer.input_layernorm).__name__ in _KNOWN_NORMS
                 and type(layer.post_attention_layernorm).__name__ in _KNOWN_NORMS)
        qk_norm = hasattr(attn, "q_norm") and hasattr(attn, "k_norm")
        if (qk_norm and type(attn).__name__ != "Qwen3Attention" and not allow_unknown) or (not known and not allow_unknown):
            skipped.append((layer, f"{type(attn).__name__} / {type(layer.input_layernorm).__name__}: not a known Llama-formula layer"))
            prev = None
            continue
        try:
            # 2 / 3 / 5 / 6 / 7-bit modules: the same code values in the 4- / 8-bit layout first (what post_init would do anyway);
            # everything below concatenates / interleaves / permutes whole 4- / 8-bit words
            for lin in [getattr(attn, n) for n in names_a] + [getattr(mlp, n) for n in names_m]:
                if hasattr(lin, "widen_in_place"):
                    lin.widen_in_place()
            qkv = fuse_quant_linears([attn.q_proj, attn.k_proj, attn.v_proj])
            # act-order checkpoints: down_proj's input permutation moves into gate / up's column order (exact; only integer
            # codes are re-ordered), so down_proj needs no activation gather -- neither the prefill pre-pass nor the in-kernel one
            fold_act_order_into_producers(mlp.down_proj, [mlp.gate_proj, mlp.up_proj])
            gu = fuse_gate_up_interleaved(mlp.gate_proj, mlp.up_proj)
        except NotImplementedError as e:
            skipped.append((layer, f"siblings cannot share a launch: {e}"))
            prev = None
            continue
        sizes = [attn.q_proj.out_features, attn.k_proj.out_features, attn.v_proj.out_features]
        in_f = attn.q_proj.in_features
        group = _FusedGroup(qkv, sizes)
        attn.fused_q_proj_k_proj_v_proj = group
        for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
            setattr(attn, n, FusedSiblingView(group, i, in_f, sizes[i]))
        # gate/up: the interleaved module serves both; the stand-ins only exist so that module walks still find two names
        gu_group = _FusedGroup(gu, [gu.out_features])
        mlp.fused_gate_up = gu_group
        del mlp.gate_proj, mlp.up_proj
        mlp.forward = types.MethodType(_mlp_forward, mlp)
        import sys
        amod = sys.modules.get(type(attn).__module__)     # the model family's own rotary / eager-attention functions
        layer._gptqhip_fused = {"hidden": in_f, "state": None, "prev": prev, "disabled": False, "workspace": workspace,
                                "orig_forward": layer.forward, "rotary": getattr(amod, "apply_rotary_pos_emb", rotary),
                                "eager_attention": getattr(amod, "eager_attention_forward", eager_attention),
                                # families whose attention forward hands a window to the interface (the VALUE is resolved per call)
                                "sliding_window": type(attn).__name__ != "LlamaAttention" and (
                                    hasattr(attn, "sliding_window") or hasattr(attn.config, "sliding_window")),
                                "qk_norm": qk_norm, "interfaces": ALL_ATTENTION_FUNCTIONS,
                                # prefill path (ops.rmsnorm_gather + forward_pregathered; GPTQ and AWQ modules both have
                                # it) for hidden sizes the norm kernel stages in LDS
                                # (GPTQHIP_HF_PREFILL=0: A/B switch, keeps HF's own layer code for prefill)
                                "prefill": ({} if hasattr(qkv, "forward_pregathered") and hasattr(gu, "forward_pregathered")
                                            and in_f % 8 == 0 and in_f <= 16384 and os.environ.get("GPTQHIP_HF_PREFILL", "1") != "0"
                                            else None)}
        layer.forward = types.MethodType(_layer_forward, layer)
        fused.append(layer)
        prev = layer
    return fused, skipped


def auto_fuse(model: nn.Module) -> int:
    """The hook `gptqmodel_post_init` calls FIRST (this package's utils.model.gptqmodel_post_init and, through
    integration/gptqmodel_overlay/utils/model.patch, the reference's own gptqmodel/utils/model.py:1281): when the model holds
    decoder layers whose seven projections are (adapter-free, not yet post_init()ed) HIP quant modules of a Llama-formula
    family, rewrite them with fuse_llama_decoder_layers so that what `GPTQModel.load()` returns decodes through four fused
    decode ops per layer instead of seven plugin forwards + ~16 glue kernels (bench.py: 385 -> ~950 tokens/s on the Llama-3-8B
    linear stack).  The reference's fast kernels get their speed inside post_init() / forward() without a model rewrite too
    (marlin.py:246-293 repacks there).

    Opt-out: GPTQHIP_AUTO_FUSE=0 in the environment, or `model._gptqhip_auto_fuse = False`.  Never raises: anything unexpected
    (transformers without the attention-interface API, unknown layer classes, siblings with different quantisation parameters)
    leaves the modules as they are and the plugin forward() path serves them.  Returns the number of fused layers."""
    if os.environ.get("GPTQHIP_AUTO_FUSE", "1") == "0" or getattr(model, "_gptqhip_auto_fuse", True) is False:
        return 0
    if not isinstance(model, nn.Module) or _is_quant(model):
        return 0
    try:
        has_layer = any(all(hasattr(m, a) for a in ("self_attn", "mlp", "input_layernorm", "post_attention_layernorm")) for m in model.modules())
        if not has_layer:
            return 0
        fused, skipped = fuse_llama_decoder_layers(model)
    except Exception as e:  # noqa: BLE001  (a convenience pass must never break a load)
        import warnings
        warnings.warn(f"gptqmodel_amd: decoder-layer fusion skipped ({type(e).__name__}: {e}); the plugin forward() path serves the model")
        return 0
    model._gptqhip_fused_layers = len(fused)
    model._gptqhip_skipped_layers = [(type(layer).__name__, why) for layer, why in skipped]
    return len(fused)


__all__ = ["fuse_llama_decoder_layers", "auto_fuse"]
