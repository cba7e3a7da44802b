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


_PLANES = {2: ((2, 0),), 3: ((2, 0), (1, 2)), 4: ((4, 0),), 5: ((4, 0), (1, 4)), 6: ((4, 0), (2, 4)), 7: ((4, 0), (2, 4), (1, 6)), 8: ((8, 0),)}


def _zero_fields(qzeros: torch.Tensor, bits: int, planar: bool) -> torch.Tensor:
    """qzeros int32 [G, N*bits/32] -> int64 [G, N] zero-points, continuous `bits`-bit fields of each 32-column group's bit stream or
    the planar layout (utils/planar_packing.py:7-24)."""
    g, cols = qzeros.shape
    w = (qzeros.to(torch.int64) & 0xFFFFFFFF).reshape(g, cols // bits, bits)
    out = torch.zeros((g, cols // bits, 32), dtype=torch.int64, device=qzeros.device)
    if planar:
        row = 0
        for width, offset in _PLANES[bits]:
            pf = 32 // width
            sh = torch.arange(pf, dtype=torch.int64, device=qzeros.device) * width
            codes = (w[:, :, row:row + width, None] >> sh) & ((1 << width) - 1)
            out |= codes.reshape(g, cols // bits, 32) << offset
            row += width
    else:
        for i in range(32):
            pos = bits * i
            wi, sh = pos // 32, pos % 32
            v = w[:, :, wi] >> sh
            if sh + bits > 32:
                v = v | (w[:, :, wi + 1] << (32 - sh))
            out[:, :, i] = v & ((1 << bits) - 1)
    return out.reshape(g, -1)


def _pack_zero_fields(z: torch.Tensor, bits: int, planar: bool) -> torch.Tensor:
    g, n = z.shape
    c = z.to(torch.int64).reshape(g, n // 32, 32)
    out = torch.zeros((g, n // 32, bits), dtype=torch.int64, device=z.device)
    if planar:
        row = 0
        for width, offset in _PLANES[bits]:
            pf = 32 // width
            plane = ((c >> offset) & ((1 << width) - 1)).reshape(g, n // 32, width, pf)
            sh = torch.arange(pf, dtype=torch.int64, device=z.device) * width
            out[:, :, row:row + width] = (plane << sh).sum(dim=-1)
            row += width
    else:
        for i in range(32):
            pos = bits * i
            wi, sh = pos // 32, pos % 32
            out[:, :, wi] |= (c[:, :, i] << sh) & 0xFFFFFFFF
            if sh + bits > 32:
                out[:, :, wi + 1] |= c[:, :, i] >> (32 - sh)
    out = out.reshape(g, -1) & 0xFFFFFFFF
    return torch.where(out >= 2 ** 31, out - 2 ** 32, out).to(torch.int32)


def unshift_v2_qzeros(qzeros: torch.Tensor, bits: int, planar=None) -> torch.Tensor:
    """Runtime (v2) zero-points -> the on-disk `format: gptq` ones (v1, zero - 1 in every field, modulo 2^bits): the inverse of
    shift_v1_qzeros, what the reference's writer applies (convert_gptq_v2_to_v1_format_module, utils/model.py:900-943): a word subtract for
    2 / 4 / 8 bits, the decoded values shifted by one for 3 / 5 / 6 / 7 bits and the planar layouts."""
    if bits not in _PLANES:
        raise NotImplementedError(f"v2->v1 conversion: bits={bits}")
    planar = bits in (5, 6, 7) if planar is None else bool(planar)
    if bits in (2, 4, 8) and not planar:          # the reference's word subtract (:910-931), the exact inverse of the word add
        sub = {2: 0x55555555, 4: 0x11111111, 8: 0x01010101}[bits]
        return qzeros - (sub - 2 ** 32 if sub >= 2 ** 31 else sub)
    z = (_zero_fields(qzeros, bits, planar) - 1) & ((1 << bits) - 1)
    return _pack_zero_fields(z, bits, planar)


def shift_v1_qzeros(qzeros: torch.Tensor, bits: int, planar=None) -> torch.Tensor:
    """v1 -> v2 zero-points for every bit width the reference converts (utils/model.py:750-844): 2 / 4 / 8 bits add one to every packed
    field with int32 wraparound (the word add of :756-831); 3 / 5 / 6 / 7 bits and every planar layout have fields that straddle words
    or planes, so the DECODED values are shifted by one modulo 2^bits and re-encoded (:767-775, :834-839)."""
    planar = bits in (5, 6, 7) if planar is None else bool(planar)
    if bits in (3, 5, 6, 7) or planar:
        if bits not in _PLANES:
            raise NotImplementedError(f"v1->v2 conversion: bits={bits}")
        z = (_zero_fields(qzeros, bits, planar) + 1) & ((1 << bits) - 1)
        return _pack_zero_fields(z, bits, planar)
    add = {2: 0x55555555, 4: 0x11111111, 8: 0x01010101}.get(bits)
    if add is None:
        raise NotImplementedError(f"v1->v2 conversion: bits={bits}")
    if add >= 2 ** 31:
        add -= 2 ** 32
    return qzeros + add  # int32 tensor add wraps like the reference's `+= 0b0001...`


def convert_gptq_v1_to_v2_format_module(module: BaseQuantLinear, bits: int, pack_dtype: torch.dtype = torch.int32):
    """v1 checkpoints store zero-1; the loader turns them into v2 zero-points before post_init (utils/model.py:750-844)."""
    if pack_dtype != torch.int32:
        raise NotImplementedError(f"v1->v2 conversion supports int32 words, got {pack_dtype}")
    module.qzeros.data = shift_v1_qzeros(module.qzeros.data, bits, planar=bool(getattr(module, "planar", False)))
    module.qzero_format(format=2)
    return module


def convert_gptq_v1_to_v2_format(model: nn.Module, bits: int, pack_dtype: torch.dtype = torch.int32):
    for m in model.modules():
        if isinstance(m, BaseQuantLinear) and getattr(m, "REQUIRES_FORMAT_V2", False) and hasattr(m, "qzero_format") \
                and m.qzero_format() == 1:
            convert_gptq_v1_to_v2_format_module(m, bits=m.bits, pack_dtype=pack_dtype)
    return model


def hf_convert_gptq_v1_to_v2_format(model: nn.Module, bits: int, qlinear_kernel=None, checkpoint_format: str = "gptq",
                                    meta=None):
    """HF/optimum entry (utils/model.py:730-748): convert only `gptq` (v1) checkpoints, and only when a loaded module
    needs v2 zeros.  Returns (model, converted)."""
    if str(checkpoint_format).lower() != "gptq":
        return model, False
    if not any(isinstance(m, BaseQuantLinear) and getattr(m, "REQUIRES_FORMAT_V2", False) for m in model.modules()):
        return model, False
    return convert_gptq_v1_to_v2_format(model, bits=bits), True


def hf_gptqmodel_post_init(model, use_act_order: bool = False, quantize_config=None, max_input_length=None):
    """HF/optimum entry (utils/model.py:1276-1278)."""
    return gptqmodel_post_init(model, use_act_order)


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


def dynamic_get(dynamic: Optional[Dict[str, Dict]], module_name: str):
    """Per-module override lookup of `QuantizeConfig.dynamic` (quantization/config.py:1614-1652, 1822-1854): patterns are
    tried in dict order and matched with re.match at the start of the qualified module name; "-:" prefix = negative
    match (returns False: leave the module unquantised), "+:" or no prefix = positive (returns a copy of the override
    dict); no match returns None."""
    if not dynamic:
        return None
    import re
    for pattern, overrides in dynamic.items():
        negative = pattern.startswith("-:")
        raw = pattern[2:] if pattern.startswith(("-:", "+:")) else pattern
        if re.match(raw, module_name):
            return False if negative else dict(overrides or {})
    return None


def make_quant(model: nn.Module, names: Iterable[str], bits: int, group_size: int, desc_act: bool, sym: bool,
               backend: BACKEND = BACKEND.AUTO, format: FORMAT = FORMAT.GPTQ, quant_method: METHOD = METHOD.GPTQ,
               device=DEVICE.ROCM, dtype: Optional[torch.dtype] = None,
               dynamic: Optional[Dict[str, Dict]] = None) -> List[Type[BaseQuantLinear]]:
    """Replace every nn.Linear whose qualified name is in `names` by the selected QuantLinear class
    (utils/model.py:398-472, 651-727).  NotImplementedError from a candidate => try the next one (AUTO).
    `dynamic`: per-module overrides of bits / group_size / desc_act / sym, or exclusion (utils/model.py:545-564)."""
    wanted = set(names)
    base = (bits, group_size, desc_act, sym)
    cache: Dict[tuple, List[Type[BaseQuantLinear]]] = {}

    def candidates_for(cfg):
        if cfg not in cache:
            cache[cfg] = select_quant_linear(bits=cfg[0], group_size=cfg[1], desc_act=cfg[2], sym=cfg[3], device=device,
                                             backend=backend, format=format, quant_method=quant_method, dtype=dtype,
                                             multi_select=True)
        return cache[cfg]

    candidates = candidates_for(base)
    modules: Dict[str, nn.Module] = dict(model.named_modules())
    for full_name in sorted(wanted):
        sub = modules.get(full_name)
        if sub is None or isinstance(sub, BaseQuantLinear):
            continue
        if not isinstance(sub, nn.Linear):
            raise ValueError(f"make_quant: `{full_name}` is {type(sub).__name__}, expected nn.Linear")
        cfg = base
        overrides = dynamic_get(dynamic, full_name)
        if overrides is False:  # negative match: this module stays a float nn.Linear
            continue
        if overrides:
            cfg = (overrides.get("bits", bits), overrides.get("group_size", group_size),
                   overrides.get("desc_act", desc_act), overrides.get("sym", sym))
        parent_name, _, child = full_name.rpartition(".")
        parent = modules[parent_name] if parent_name else model
        last = None
        for cls in candidates_for(cfg):
            try:
                create_quant_module(parent, child, cls, cfg[0], cfg[1], cfg[2], cfg[3], sub.in_features,
                                    sub.out_features, sub.bias is not None, full_name, backend, format, dtype)
                last = None
                break
            except NotImplementedError as e:
                last = e
        if last is not None:
            raise ValueError(f"No compatible quant module for `{full_name}`: {last}")
    return candidates


def gptqmodel_post_init(model: nn.Module, use_act_order: bool = False, **_kw) -> nn.Module:
    """Call post_init() on every QuantLinear once its tensors are on the device (utils/model.py:1281-1344).  Round 6: recognised
    decoder layers are first rewritten onto the fused decode ops (utils.hf_llama.auto_fuse; GPTQHIP_AUTO_FUSE=0 opts out), so the
    model a loader returns takes the fast path by itself."""
    if isinstance(model, BaseQuantLinear):
        model.post_init()
        return model
    from .hf_llama import auto_fuse
    auto_fuse(model)
    for m in model.modules():
        if isinstance(m, BaseQuantLinear):
            m.post_init()
    return model


# ---------------------------------------------------------------------------------------------------------
# Fused sibling projections: q/k/v and gate/up read the same x, so their checkpoint tensors are
# concatenated along N BEFORE post_init and run as ONE kernel launch (7 -> 4 launches per decoder layer, x read
# once).  HF decoder layers keep calling q_proj(x), k_proj(x), v_proj(x): each name is replaced by a view module
# that returns its column slice of the fused result computed once per distinct input tensor.
# ---------------------------------------------------------------------------------------------------------
class _FusedGroup(nn.Module):
    """One fused launch shared by its sibling views.  The fused output is cached for exactly ONE input tensor, identified
    by object identity while a strong reference keeps that tensor alive (id()/data_ptr() of a dead tensor get reused by the
    next decode step's activations), plus its in-place version counter where autograd tracks one.  The cache is dropped as
    soon as every sibling has been served, so nothing stale or large (a prefill's fused output) stays pinned."""

    def __init__(self, fused: BaseQuantLinear, sizes: List[int]):
        super().__init__()
        self.fused = fused
        self.sizes = list(sizes)
        self.offsets = [sum(sizes[:i]) for i in range(len(sizes))]
        self._x = None
        self._ver = None
        self._out = None
        self._served = set()

    @staticmethod
    def _version_of(x: torch.Tensor):
        try:
            return x._version
        except RuntimeError:  # inference tensors (torch.inference_mode) do not track a version counter
            return None

    def slice_for(self, x: torch.Tensor, index: int) -> torch.Tensor:
        ver = self._version_of(x)
        if self._out is None or self._x is not x or ver != self._ver or index in self._served:
            self._out = self.fused(x)
            self._x, self._ver = x, ver
            self._served = set()
        o = self.offsets[index]
        y = self._out[..., o:o + self.sizes[index]]
        self._served.add(index)
        if len(self._served) == len(self.sizes):
            self._x = self._out = self._ver = None
            self._served = set()
        return y


class FusedSiblingView(nn.Module):
    """Stands in for one sibling projection (e.g. `k_proj`) after fusion."""

    def __init__(self, group: _FusedGroup, index: int, in_features: int, out_features: int):
        super().__init__()
        self._group = [group]  # list: do not register the shared group as a child of every view
        self.index = index
        self.in_features = in_features
        self.out_features = out_features

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._group[0].slice_for(x, self.index)


def fuse_quant_linears(mods: List[BaseQuantLinear]) -> BaseQuantLinear:
    """Concatenate checkpoint-layout GPTQ/AWQ modules along N into one module of the same class.  Must run BEFORE
    post_init().  Raises NotImplementedError when the siblings cannot share one kernel launch (different
    quantisation parameters, or act-order with different permutations)."""
    m0 = mods[0]
    if m0.bits not in (4, 8):
        # (the column concatenation works on whole packed words: 4- and 8-bit fields; the widened bit widths keep separate launches)
        raise NotImplementedError(f"sibling fusion packs 4- and 8-bit columns only (bits={m0.bits}); not fusing")
    for m in mods[1:]:
        same = (type(m) is type(m0) and m.bits == m0.bits and m.group_size == m0.group_size
                and m.in_features == m0.in_features and m.sym == m0.sym and m.desc_act == m0.desc_act
                and m.scales.dtype == m0.scales.dtype and (m.bias is None) == (m0.bias is None))
        if not same:
            raise NotImplementedError("siblings differ in quantisation parameters; not fusing")
        g0, g1 = getattr(m0, "g_idx", None), getattr(m, "g_idx", None)
        if (g0 is None) != (g1 is None) or (g0 is not None and not torch.equal(g0, g1)):
            raise NotImplementedError("siblings have different g_idx (act-order); not fusing")
    if getattr(m0, "_ready", False):
        raise RuntimeError("fuse_quant_linears must be called before post_init()")
    n_total = sum(m.out_features for m in mods)
    fused = type(m0)(bits=m0.bits, group_size=m0.requested_group_size, sym=m0.sym, desc_act=m0.desc_act,
                     in_features=m0.in_features, out_features=n_total, bias=m0.bias is not None,
                     register_buffers=False, name="+".join(m.name for m in mods), adapter=None)
    # registered buffers (not plain attributes): model.to(device) / list_buffers() must reach the fused tensors
    fused.register_buffer("qweight", torch.cat([m.qweight for m in mods], dim=1).contiguous())
    fused.register_buffer("qzeros", torch.cat([m.qzeros for m in mods], dim=1).contiguous())
    fused.register_buffer("scales", torch.cat([m.scales for m in mods], dim=1).contiguous())
    if hasattr(m0, "g_idx"):
        fused.register_buffer("g_idx", m0.g_idx)
    if m0.bias is not None:
        fused.register_buffer("bias", torch.cat([m.bias for m in mods]).contiguous())
    else:
        fused.bias = None
    if getattr(m0, "format", None) is not None and hasattr(fused, "format"):
        fused.format = m0.format
    if hasattr(m0, "qzero_format"):
        fused.qzero_format(format=m0.qzero_format())
    fused.train(m0.training)
    return fused


def _interleave_cols(a: torch.Tensor, b: torch.Tensor, block: int) -> torch.Tensor:
    """[R, C] x 2 -> [R, 2C]: blocks of `block` columns of a and b alternate (a0..a7 b0..b7 a8..a15 ...).
    The result owns its storage (written through a strided view of a fresh tensor): a reshape()d torch.stack would be a VIEW
    whose base stays alive behind `tensor.data = ...` in post_init -- one extra copy of the packed weights per layer."""
    r, c = a.shape
    out = torch.empty((r, 2 * c), dtype=a.dtype, device=a.device)
    v = out.view(r, c // block, 2, block)
    v[:, :, 0, :] = a.reshape(r, c // block, block)
    v[:, :, 1, :] = b.reshape(r, c // block, block)
    return out


def fuse_gate_up_interleaved(gate: BaseQuantLinear, up: BaseQuantLinear) -> BaseQuantLinear:
    """gate_proj + up_proj -> ONE module whose output columns alternate in blocks of 8: [g0..g7, u0..u7, g8..g15, u8..u15, ...].
    Every 16-column MFMA tile of the kernel then holds both halves of 8 MLP neurons, so the batch-1 decode op can apply
    SiLU(gate) * up in its epilogue (GPTQHIP_OUT_SILU_MUL_PAIRED) -- once per element, by the block that produced it.
    Blocks of 8 keep the packed words whole (qzeros / AWQ qweight pack 8 four-bit columns per int32).  Must run BEFORE
    post_init().  forward() of the result returns the interleaved [.., 2*inter] tensor; `deinterleave_gate_up` undoes it."""
    for m in (gate, up):
        if getattr(m, "_ready", False):
            raise RuntimeError("fuse_gate_up_interleaved must be called before post_init()")
    if gate.bits not in (4, 8):
        raise NotImplementedError(f"gate/up interleaving packs 4- and 8-bit columns only (bits={gate.bits}); not fusing")
    same = (type(gate) is type(up) and gate.bits == up.bits and gate.group_size == up.group_size and gate.sym == up.sym
            and gate.in_features == up.in_features and gate.out_features == up.out_features and gate.desc_act == up.desc_act
            and gate.scales.dtype == up.scales.dtype and (gate.bias is None) == (up.bias is None))
    g0, g1 = getattr(gate, "g_idx", None), getattr(up, "g_idx", None)
    if not same or (g0 is None) != (g1 is None) or (g0 is not None and not torch.equal(g0, g1)):
        raise NotImplementedError("gate/up differ in quantisation parameters (or act-order permutation); not fusing")
    n = gate.out_features
    if n % 8 != 0:
        raise NotImplementedError("out_features must be a multiple of 8")
    fused = type(gate)(bits=gate.bits, group_size=gate.requested_group_size, sym=gate.sym, desc_act=gate.desc_act,
                       in_features=gate.in_features, out_features=2 * n, bias=gate.bias is not None,
                       register_buffers=False, name=f"{gate.name}|{up.name}", adapter=None)
    cpw = 32 // gate.bits                       # columns per packed word along N
    n_packed = gate.qzeros.shape[1]             # words per row of an N-packed tensor
    if gate.qweight.shape[1] == n:              # GPTQ: qweight is K-packed, one column per element
        fused.register_buffer("qweight", _interleave_cols(gate.qweight, up.qweight, 8))
    else:                                       # AWQ: qweight is N-packed like qzeros
        fused.register_buffer("qweight", _interleave_cols(gate.qweight, up.qweight, 8 // cpw))
    fused.register_buffer("qzeros", _interleave_cols(gate.qzeros, up.qzeros, 8 // cpw))
    assert fused.qzeros.shape[1] == 2 * n_packed
    fused.register_buffer("scales", _interleave_cols(gate.scales, up.scales, 8))
    if g0 is not None:
        fused.register_buffer("g_idx", g0)
    if gate.bias is not None:
        fused.register_buffer("bias", _interleave_cols(gate.bias[None], up.bias[None], 8)[0])
    else:
        fused.bias = None
    if hasattr(gate, "qzero_format"):
        fused.qzero_format(format=gate.qzero_format())
    if getattr(gate, "format", None) is not None and hasattr(fused, "format"):
        fused.format = gate.format
    fused.gate_up_interleaved = True
    fused.train(gate.training)
    return fused


def deinterleave_gate_up(y: torch.Tensor):
    """[.., 2*inter] output of a fuse_gate_up_interleaved module -> (gate [.., inter], up [.., inter])."""
    v = y.reshape(y.shape[:-1] + (y.shape[-1] // 16, 2, 8))
    return v[..., 0, :].reshape(y.shape[:-1] + (-1,)), v[..., 1, :].reshape(y.shape[:-1] + (-1,))


def _unpack_k(qweight: torch.Tensor, bits: int) -> torch.Tensor:
    """K-packed int32 [K*bits/32, N] -> codes int32 [K, N] (row 32/bits*r + j = bits j*bits.. of word r; qlinear/__init__.py:827-865)."""
    pf = 32 // bits
    sh = torch.arange(0, 32, bits, dtype=torch.int32, device=qweight.device).view(1, pf, 1)
    return ((qweight.unsqueeze(1) >> sh) & ((1 << bits) - 1)).reshape(qweight.shape[0] * pf, qweight.shape[1])


def _pack_k(codes: torch.Tensor, bits: int) -> torch.Tensor:
    pf = 32 // bits
    v = codes.reshape(codes.shape[0] // pf, pf, codes.shape[1]).to(torch.int64)
    sh = torch.arange(0, 32, bits, dtype=torch.int64, device=codes.device).view(1, pf, 1)
    w = (v << sh).sum(dim=1) & 0xFFFFFFFF
    return torch.where(w >= 2**31, w - 2**32, w).to(torch.int32)


def _unpack_n(q: torch.Tensor, bits: int) -> torch.Tensor:
    """N-packed int32 [R, N*bits/32] -> [R, N] (column 32/bits*c + j = bits j*bits.. of word c: the qzeros layout)."""
    pf = 32 // bits
    sh = torch.arange(0, 32, bits, dtype=torch.int32, device=q.device).view(1, 1, pf)
    return ((q.unsqueeze(2) >> sh) & ((1 << bits) - 1)).reshape(q.shape[0], q.shape[1] * pf)


def _pack_n(codes: torch.Tensor, bits: int) -> torch.Tensor:
    pf = 32 // bits
    v = codes.reshape(codes.shape[0], codes.shape[1] // pf, pf).to(torch.int64)
    sh = torch.arange(0, 32, bits, dtype=torch.int64, device=codes.device).view(1, 1, pf)
    w = (v << sh).sum(dim=2) & 0xFFFFFFFF
    return torch.where(w >= 2**31, w - 2**32, w).to(torch.int32)


def fold_act_order_into_producers(consumer: BaseQuantLinear, producers: List[BaseQuantLinear]) -> bool:
    """Remove an act-order (desc_act) checkpoint's INPUT permutation from `consumer` by re-ordering the OUTPUT columns of the
    quantised linears that produce its input through elementwise ops only -- Llama's down_proj(act(gate_proj(x)) * up_proj(x)):
    with perm = stable argsort(consumer.g_idx),

        consumer rows  k' <- perm[k']      (its packed rows are unpacked, gathered, re-packed; g_idx becomes k // group_size)
        producer cols  j' <- perm[j']      (qweight / scales columns gathered, packed zero-points re-packed, bias gathered)

    so  sum_k a[k] W[k,:]  is evaluated over the same terms in group order and the consumer needs NO gather any more: no x
    permutation pre-pass in prefill (one extra read + write of [M, K]), no in-kernel permutation at decode.  Exact: only
    integer codes move.  GPTQ checkpoint-layout modules BEFORE post_init(); returns False (nothing changed) when the consumer
    has no act-order permutation or the modules do not fit (producer out_features != consumer in_features, adapters, AWQ)."""
    g = getattr(consumer, "g_idx", None)
    if consumer.bits not in (4, 8):
        return False        # (rows are re-packed in 4- / 8-bit fields here; the other widths keep their gather)
    if g is None or g.numel() != consumer.in_features or getattr(consumer, "_ready", False):
        return False
    if any(getattr(p, "_ready", False) or p.out_features != consumer.in_features or getattr(p, "adapter", None) is not None
           or p.qweight.shape[1] != p.out_features for p in producers) or getattr(consumer, "adapter", None) is not None:
        return False
    if consumer.qweight.shape[1] != consumer.out_features:      # (AWQ packs along N: no act-order there)
        return False
    k, gs = consumer.in_features, consumer.group_size
    seq = torch.arange(k, device=g.device, dtype=torch.int64) // gs
    gl = g.long()
    if torch.equal(gl, seq):
        return False
    perm = torch.argsort(gl, stable=True)
    if not torch.equal(gl[perm], seq):
        return False                                            # unbalanced groups: the kernel layout cannot express it anyway
    bits = consumer.bits
    consumer.qweight.data = _pack_k(_unpack_k(consumer.qweight.data, bits)[perm], bits)
    consumer.g_idx.data = seq.to(g.dtype)
    for p in producers:
        pp = perm.to(p.qweight.device)
        p.qweight.data = p.qweight.data[:, pp].contiguous()
        p.scales.data = p.scales.data[:, pp].contiguous()
        p.qzeros.data = _pack_n(_unpack_n(p.qzeros.data, p.bits)[:, pp], p.bits)
        if getattr(p, "bias", None) is not None:
            p.bias.data = p.bias.data[pp].contiguous()
    return True


def fuse_siblings(parent: nn.Module, names: List[str]) -> Optional[_FusedGroup]:
    """Fuse parent.<names> (quant modules sharing their input) and replace them by views.  Returns the group, or None
    when fusion is not possible (the original modules are left untouched)."""
    mods = [getattr(parent, n) for n in names]
    if not all(isinstance(m, BaseQuantLinear) for m in mods) or any(m.adapter is not None for m in mods):
        return None
    try:
        fused = fuse_quant_linears(mods)
    except NotImplementedError:
        return None
    group = _FusedGroup(fused, [m.out_features for m in mods])
    setattr(parent, "fused_" + "_".join(names), group)  # registered once: post_init / .to() reach the fused module
    for i, (n, m) in enumerate(zip(names, mods)):
        setattr(parent, n, FusedSiblingView(group, i, m.in_features, m.out_features))
    return group


def dequantize_model(model: nn.Module, device="cpu", dtype: torch.dtype = torch.float16) -> nn.Module:
    """Replace every HIP QuantLinear by an nn.Linear holding its dequantised weights -- the mirror of the reference's
    dequantize_model (nn_modules/qlinear/torch.py:736-761, which does this for TorchLinear and moves the weights to CPU
    fp16).  The [K,N] weights come from the device dequant kernels (bit-exact with dequantize_weight()).  Fused sibling
    groups (fuse_siblings) are split back into one nn.Linear per original projection; decoder layers rewritten by
    utils.hf_llama.fuse_llama_decoder_layers get HF's own forward back, their interleaved gate|up module is de-interleaved into
    gate_proj / up_proj again (an act-order down_proj folded into them stays folded: gate / up columns and down_proj rows were
    re-ordered consistently, the MLP computes the same function)."""
    from ..nn_modules.qlinear.hip_awq import HipAwqLinear
    from ..nn_modules.qlinear.hip_gptq import HipGptqLinear, HipQuantEmbeddings

    def to_linear(w_kn: torch.Tensor, bias: Optional[torch.Tensor]) -> nn.Linear:
        lin = nn.Linear(w_kn.shape[0], w_kn.shape[1], bias=bias is not None)
        lin.weight = nn.Parameter(w_kn.T.detach().to(device=device, dtype=dtype).contiguous(), requires_grad=False)
        if bias is not None:
            lin.bias = nn.Parameter(bias.detach().to(device=device, dtype=dtype), requires_grad=False)
        return lin

    for layer in list(model.modules()):   # decoder layers of fuse_llama_decoder_layers: back to HF's own code path
        fd = getattr(layer, "_gptqhip_fused", None)
        if isinstance(fd, dict) and "orig_forward" in fd:
            layer.forward = fd["orig_forward"]
            del layer._gptqhip_fused
    for name, module in list(model.named_modules()):
        parent_name, _, child = name.rpartition(".")
        live = dict(model.named_modules())
        if parent_name and parent_name not in live:
            continue  # child of a fused group that was already replaced
        parent = live[parent_name] if parent_name else model
        if isinstance(module, _FusedGroup) and getattr(module.fused, "gate_up_interleaved", False):
            # fuse_gate_up_interleaved: columns alternate in blocks of 8 (g0..7 u0..7 g8..15 ...); there are no sibling views
            w = module.fused.dequantize_weight()
            b = module.fused.bias
            wg, wu = deinterleave_gate_up(w)
            bg, bu = (None, None) if b is None else deinterleave_gate_up(b[None])
            parent.gate_proj = to_linear(wg, None if bg is None else bg[0])
            parent.up_proj = to_linear(wu, None if bu is None else bu[0])
            parent.__dict__.pop("forward", None)      # the instance-level _mlp_forward: the class's own forward is back
            delattr(parent, child)
            continue
        if isinstance(module, _FusedGroup):
            w = module.fused.dequantize_weight()
            b = module.fused.bias
            views = {n: v for n, v in parent.named_children() if isinstance(v, FusedSiblingView) and v._group[0] is module}
            for vname, view in views.items():
                o, n_out = module.offsets[view.index], module.sizes[view.index]
                setattr(parent, vname, to_linear(w[:, o:o + n_out], None if b is None else b[o:o + n_out]))
            delattr(parent, child)
        elif isinstance(module, HipQuantEmbeddings):
            continue  # embeddings are not linears; the reference's dequantize_model leaves them alone too
        elif isinstance(module, (HipGptqLinear, HipAwqLinear)) and not isinstance(parent, _FusedGroup):
            setattr(parent, child, to_linear(module.dequantize_weight(), module.bias))
        elif isinstance(module, BaseQuantLinear) and not isinstance(parent, _FusedGroup):
            raise ValueError(f"dequantize_model: `{name}` is {type(module).__name__}; only HIP QuantLinear modules are "
                             "supported")
    cfg = getattr(model, "config", None)
    if cfg is not None and hasattr(cfg, "quantization_config"):
        del cfg.quantization_config
    return model
