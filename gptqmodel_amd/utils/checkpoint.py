"""On-disk GPTQ / AWQ checkpoints on either side of the quantised-linear path: the part of the reference's load flow that hands
tensors to the QuantLinear classes, and the inverse for tests / tools.

Reference flow being mirrored (gptqmodel/models/loader.py): `QuantizeConfig.from_pretrained` (quantization/config.py:3022-3043:
quantize_config.json, else quant_config.json, else config.json["quantization_config"]; key synonyms :1504-1525) ->
`make_quant` (loader.py:1092 -> utils/model.py:398) -> safetensors shards into the module buffers (loader.py:1646
`load_checkpoint_in_model...`) -> v1 -> v2 zero-point conversion for kernels with REQUIRES_FORMAT_V2 (loader.py:1658-1675,
utils/model.py:750-844) -> `gptqmodel_post_init` (loader.py:1804).  The model zoo, tokenizer handling, device maps, lazy /
offloaded loading and every other checkpoint format stay with the reference (out of scope); this file is the minimum
a caller needs to go from a checkpoint DIRECTORY to post_init()ed HIP modules inside an already constructed HF model, and it is what
tests/test_gpu_checkpoint.py round-trips.

On-disk conventions: GPTQ `format: "gptq"` stores zero-points minus one (v1); `gptq_v2` stores them as
is.  safetensors shards + `model.safetensors.index.json` (`weight_map`: tensor name -> shard file) as written by transformers.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, Iterable, List, Optional

import torch
import torch.nn as nn

from .backend import BACKEND
from .const import FORMAT, METHOD

QUANT_CONFIG_FILENAMES = ("quantize_config.json", "quant_config.json", "config.json")     # quantization/config.py:73-74
# quantization/config.py:1504-1525
_SYNONYMS = {"w_bit": "bits", "wbits": "bits", "q_group_size": "group_size", "version": "format", "checkpoint_format": "format",
             "quant_method": "method"}
_SYNONYMS_NEGATED = {"zero_point": "sym"}
SAFETENSORS_INDEX = "model.safetensors.index.json"
SAFETENSORS_SINGLE = "model.safetensors"


log = logging.getLogger("gptqmodel_amd")


def normalize_quantize_config(raw: Dict) -> Dict:
    """Canonical keys of a quantize_config payload (the subset that reaches the kernel constructor: SURVEY.md appendix A):
    bits, group_size, desc_act, sym, format, method, pack_dtype, dynamic, lm_head, meta.  Synonyms as in the reference; AWQ's
    `zero_point` is the NEGATION of sym; `is_marlin_format` is rejected like the reference does (config.py:67-68)."""
    if "is_marlin_format" in raw:
        raise ValueError("quantize_config: `is_marlin_format` is a hard-deprecated key (quantization/config.py:67-68)")
    cfg: Dict = {}
    for key, val in raw.items():
        if key in _SYNONYMS_NEGATED:
            cfg[_SYNONYMS_NEGATED[key]] = not bool(val)
        else:
            cfg[_SYNONYMS.get(key, key)] = val
    method = str(cfg.get("method", "gptq")).lower()
    fmt = cfg.get("format")
    if fmt is None:
        fmt = "gemm" if method == "awq" else "gptq"
    out = {
        "bits": int(cfg["bits"]),
        "group_size": int(cfg.get("group_size", 128)),
        "desc_act": bool(cfg.get("desc_act", False)),
        "sym": bool(cfg.get("sym", True)),
        "format": str(fmt).lower(),
        "method": method,
        "pack_dtype": str(cfg.get("pack_dtype", "int32")).replace("torch.", ""),
        "dynamic": cfg.get("dynamic"),
        "lm_head": bool(cfg.get("lm_head", False)),
        "meta": cfg.get("meta") or {},
    }
    ok_bits = (4,) if method == "awq" else (2, 3, 4, 5, 6, 7, 8)
    if out["bits"] not in ok_bits:
        # HipGptqLinear.SUPPORTS_BITS (2 / 3 / 5 / 6 / 7 are widened to the 4- / 8-bit kernel layout at post_init); AWQ is 4-bit
        raise ValueError(f"quantize_config: bits={out['bits']} is outside this backend ({method}: {ok_bits})")
    if out["group_size"] != -1 and out["group_size"] <= 0:
        raise ValueError(f"quantize_config: group_size={out['group_size']}")
    return out


def read_quantize_config(ckpt_dir: str) -> Dict:
    """quantize_config.json | quant_config.json | config.json["quantization_config"], first one found (config.py:3022-3043)."""
    for name in QUANT_CONFIG_FILENAMES:
        path = os.path.join(ckpt_dir, name)
        if os.path.exists(path):
            with open(path, "r", encoding="utf-8") as f:
                raw = json.load(f)
            if name == "config.json":
                if "quantization_config" not in raw:
                    continue
                raw = raw["quantization_config"]
            return normalize_quantize_config(raw)
    raise ValueError("no quantize_config.json, quant_config.json or config.json[quantization_config] in " + ckpt_dir)


def _shard_map(ckpt_dir: str) -> Dict[str, str]:
    """tensor name -> shard file (model.safetensors.index.json, or the single model.safetensors)."""
    from safetensors import safe_open
    idx = os.path.join(ckpt_dir, SAFETENSORS_INDEX)
    if os.path.exists(idx):
        with open(idx, "r", encoding="utf-8") as f:
            return dict(json.load(f)["weight_map"])
    single = os.path.join(ckpt_dir, SAFETENSORS_SINGLE)
    if not os.path.exists(single):
        raise ValueError(f"no {SAFETENSORS_INDEX} / {SAFETENSORS_SINGLE} in {ckpt_dir}")
    with safe_open(single, framework="pt") as f:
        return {k: SAFETENSORS_SINGLE for k in f.keys()}


def _written_by_v2_aware_quantizer(cfg: Dict) -> bool:
    """meta.quantizer names `gptqmodel:<version >= 0.9.0>` (quantization/config.py:2786-2792, MIN_VERSION_WITH_V2)."""
    vals = (cfg.get("meta") or {}).get("quantizer") or []
    if not isinstance(vals, list):
        vals = [vals]
    for val in vals:
        parts = str(val).split(":")
        if len(parts) >= 2 and parts[0].lower() == "gptqmodel":
            digits = []
            for tok in parts[1].split("."):
                num = "".join(ch for ch in tok if ch.isdigit())
                digits.append(int(num) if num else 0)
            if tuple(digits[:3] + [0] * (3 - len(digits[:3]))) >= (0, 9, 0):
                return True
    return False


def quantized_module_names(weight_map: Iterable[str]) -> List[str]:
    """Modules that are quantised in the checkpoint = those that own a `.qweight` tensor."""
    return sorted(k[: -len(".qweight")] for k in weight_map if k.endswith(".qweight"))


def load_quantized_checkpoint(model: nn.Module, ckpt_dir: str, device="cuda", backend: BACKEND = BACKEND.AUTO,
                              fuse_decoder_layers: bool = False, dtype: Optional[torch.dtype] = None) -> nn.Module:
    """Turn an already constructed HF model (any weights; typically built from the checkpoint's config.json) into the quantised
    model stored in `ckpt_dir`, on the HIP backend: make_quant -> every tensor of the safetensors shards into its parameter /
    buffer -> v1 -> v2 zero-points -> (optional) utils.hf_llama.fuse_llama_decoder_layers -> gptqmodel_post_init.  Raises on
    missing / unexpected / mis-shaped tensors -- a silently half-loaded model is worse than an error."""
    from safetensors import safe_open
    from .model import convert_gptq_v1_to_v2_format, gptqmodel_post_init, make_quant
    cfg = read_quantize_config(ckpt_dir)
    if cfg["method"] not in ("gptq", "awq"):
        raise NotImplementedError(f"quant method `{cfg['method']}` is outside this backend (GPTQ / AWQ only)")
    fmt = {"gptq": FORMAT.GPTQ, "gptq_v2": FORMAT.GPTQ_V2, "gptq_p": FORMAT.GPTQ_P, "gemm": FORMAT.GEMM}.get(cfg["format"])
    if fmt is None:
        raise NotImplementedError(f"checkpoint format `{cfg['format']}` is outside this backend (gptq, gptq_v2, gptq_p, AWQ gemm)")
    method = METHOD.GPTQ if cfg["method"] == "gptq" else METHOD.AWQ
    weight_map = _shard_map(ckpt_dir)
    names = quantized_module_names(weight_map)
    if not names:
        raise ValueError("the checkpoint holds no quantised (.qweight) tensors")
    if dtype is None:
        dtype = next((p.dtype for p in model.parameters() if p.dtype in (torch.float16, torch.bfloat16)), torch.float16)
    make_quant(model, names, bits=cfg["bits"], group_size=cfg["group_size"], desc_act=cfg["desc_act"], sym=cfg["sym"],
               backend=backend, format=fmt, quant_method=method, dynamic=cfg["dynamic"], dtype=dtype)
    # continuous and split-plane 3-bit words are not interchangeable, and a module is planar only if its constructor saw
    # `format=gptq_p`: fail loudly if a construction site dropped the format (the reference's check, utils/model.py:1316-1333)
    from ..nn_modules.qlinear import BaseQuantLinear
    for name, mod in model.named_modules():
        if isinstance(mod, BaseQuantLinear) and mod.bits == 3 and bool(getattr(mod, "planar", False)) != (fmt == FORMAT.GPTQ_P):
            raise ValueError(f"`{name}`: 3-bit module constructed with planar={bool(getattr(mod, 'planar', False))} but the "
                             f"checkpoint format is `{cfg['format']}`")
    model.to(device)
    targets = dict(model.named_parameters())
    targets.update(dict(model.named_buffers()))
    seen = set()
    by_file: Dict[str, List[str]] = {}
    for name, fname in weight_map.items():
        by_file.setdefault(fname, []).append(name)
    for fname, keys in sorted(by_file.items()):
        with safe_open(os.path.join(ckpt_dir, fname), framework="pt") as f:
            for key in keys:
                t = f.get_tensor(key)
                if key not in targets:
                    # extra keys the reference tolerates too (it loads through accelerate's non-strict load_checkpoint_in_model,
                    # models/loader.py:1646): AutoGPTQ-era files carry an all-zero `.bias` for every quantised Linear even when the
                    # model's Linear has bias=False, older ones a `rotary_emb.inv_freq` buffer.  Anything else is an error.
                    owner = key.rsplit(".", 1)[0]
                    if key.endswith(".rotary_emb.inv_freq") or (key.endswith(".bias") and owner in names and not bool(t.any())):
                        log.warning("load_quantized_checkpoint: ignoring checkpoint tensor `%s` (no counterpart in the model)", key)
                        continue
                    raise ValueError(f"unexpected checkpoint tensor `{key}`")
                dst = targets[key]
                if tuple(dst.shape) != tuple(t.shape):
                    raise ValueError(f"`{key}`: checkpoint shape {tuple(t.shape)} != module shape {tuple(dst.shape)}")
                with torch.no_grad():
                    dst.copy_(t.to(device=dst.device, dtype=dst.dtype if dst.dtype.is_floating_point else t.dtype))
                seen.add(key)
    quant_owned = [k for k in targets if any(k.startswith(n + ".") for n in names)]
    missing = [k for k in quant_owned if k not in seen and not k.endswith((".meta", ".perm"))]
    for k in [k for k in missing if k.endswith(".g_idx")]:
        # checkpoints written without act-order sometimes omit g_idx: the trivial mapping row k -> group k // group_size
        mod = model.get_submodule(k[: -len(".g_idx")])
        with torch.no_grad():
            targets[k].copy_((torch.arange(targets[k].numel(), device=targets[k].device) // mod.group_size).to(targets[k].dtype))
        missing.remove(k)
    if missing:
        raise ValueError(f"the checkpoint lacks tensors of quantised modules: {missing[:8]}")
    # every OTHER parameter must have been loaded as well (norms, embeddings, an untied lm_head): a silently random-initialised tensor
    # is worse than an error.  Tied parameters count as loaded when any alias was.  Buffers are NOT checked: models register persistent
    # buffers that checkpoints never hold (older transformers' rotary inv_freq, GPT-2 / NeoX attention masks) and the reference's
    # non-strict accelerate load accepts that.
    params = {n for n, _ in model.named_parameters()}
    by_storage: Dict[int, List[str]] = {}
    for k, t in targets.items():
        if t.numel() > 0:      # (zero-size tensors all report data_ptr() == 0: they alias nothing)
            by_storage.setdefault(t.data_ptr(), []).append(k)
    unloaded = [k for k in targets if k in params and k not in seen and k not in quant_owned
                and not any(a in seen for a in by_storage.get(targets[k].data_ptr(), []))]
    if unloaded:
        raise ValueError(f"the checkpoint lacks model tensors: {unloaded[:8]}")
    # ... but say which PERSISTENT buffers kept their initial values (running statistics would be wrong, rotary tables are fine)
    nonpersistent = {f"{mn}.{bn}" if mn else bn for mn, m in model.named_modules() for bn in getattr(m, "_non_persistent_buffers_set", ())}
    stale = [k for k, _ in model.named_buffers() if k in targets and k not in seen and k not in quant_owned and k not in nonpersistent
             and not any(a in seen for a in by_storage.get(targets[k].data_ptr(), []))]
    if stale:
        log.warning("load_quantized_checkpoint: persistent buffers not in the checkpoint keep their initial values: %s%s", stale[:8],
                    " ..." if len(stale) > 8 else "")
    if fmt == FORMAT.GPTQ and not cfg["sym"] and not _written_by_v2_aware_quantizer(cfg):
        # v1 files of asymmetric models from producers older than the v2-aware code base store zero-points this conversion would
        # shift by one: the reference refuses them (models/loader.py:1658-1663), so does this loader
        raise ValueError("loading a sym=False `format: gptq` (v1) checkpoint needs meta.quantizer = gptqmodel:>=0.9.0 "
                         "(models/loader.py:1658-1663)")
    if fmt == FORMAT.GPTQ:
        # on-disk v1 (zero - 1) -> runtime v2, for every kernel that asks for it (loader.py:1658-1675)
        convert_gptq_v1_to_v2_format(model, bits=cfg["bits"])
    if fuse_decoder_layers:
        from .hf_llama import fuse_llama_decoder_layers
        fuse_llama_decoder_layers(model)
    else:
        model._gptqhip_auto_fuse = False   # the caller said no: gptqmodel_post_init's own pass (utils.hf_llama.auto_fuse) stays off too
    gptqmodel_post_init(model, use_act_order=cfg["desc_act"])
    model.eval()
    return model


def _v2_to_v1_qzeros(qzeros: torch.Tensor, bits: int, planar=None) -> torch.Tensor:
    """Runtime (v2) zero-points -> on-disk `format: gptq` (v1, zero - 1): the reference writer's
    convert_gptq_v2_to_v1_format_module (utils/model.py:900-943), the exact inverse of the loader's conversion."""
    from .model import unshift_v2_qzeros
    return unshift_v2_qzeros(qzeros, bits, planar)


def save_quantized_checkpoint(model: nn.Module, ckpt_dir: str, quantize_config: Dict, max_shard_bytes: int = 1 << 30) -> List[str]:
    """Write `model` (HIP quant modules still in the CHECKPOINT layout, i.e. before post_init) in the on-disk layout:
    safetensors shards of at most max_shard_bytes + model.safetensors.index.json + quantize_config.json.  `format: "gptq"`
    stores v1 zero-points like the reference's writer.  Returns the shard file names.  (Saving a post_init()ed model is refused by
    the modules themselves: the kernel layout is not a checkpoint layout.)"""
    from safetensors.torch import save_file
    from ..nn_modules.qlinear import BaseQuantLinear
    cfg = normalize_quantize_config(quantize_config)
    # validated BEFORE anything in ckpt_dir is removed or written (a refused save must not cost a valid checkpoint already there): the
    # reference only accepts sym=False v1 files whose producer entry is `gptqmodel:>=0.9.0` (quantization/config.py:2790,
    # models/loader.py:1658-1663); a file stamped with this repo's own tag -- or with any producer both loaders refuse -- would be unreadable
    if cfg["format"] == "gptq" and not cfg["sym"] and not _written_by_v2_aware_quantizer(cfg):
        raise ValueError("saving sym=False with checkpoint_format=gptq (v1) needs an explicit meta.quantizer the reference recognises "
                         "(e.g. ['gptqmodel:<version >= 0.9.0>']), or save as gptq_v2")
    os.makedirs(ckpt_dir, exist_ok=True)
    state = {}
    v1_owners, planar_of, bits_of = set(), {}, {}
    for name, mod in model.named_modules():
        if isinstance(mod, BaseQuantLinear):
            if getattr(mod, "_ready", False):
                raise RuntimeError(f"`{name}` is already post_init()ed: save from the checkpoint-layout model")
            if getattr(mod, "source_bits", mod.bits) != mod.bits:
                raise RuntimeError(f"`{name}` was widened from {mod.source_bits} to {mod.bits} bits (layer fusion): save from the "
                                   f"checkpoint-layout model")
            if cfg["format"] == "gptq" and hasattr(mod, "qzero_format") and mod.qzero_format() == 2:
                v1_owners.add(name)
                planar_of[name] = bool(getattr(mod, "planar", False))
                bits_of[name] = int(mod.bits)
    for key, t in model.state_dict().items():
        t = t.detach()
        if key.endswith(".qzeros") and key[: -len(".qzeros")] in v1_owners:
            # the MODULE's width, not the config's: `dynamic` overrides make mixed-width checkpoints (utils/model.py:908)
            t = _v2_to_v1_qzeros(t, bits_of[key[: -len(".qzeros")]], planar_of[key[: -len(".qzeros")]])
        state[key] = t.to("cpu").contiguous()
    shards: List[Dict[str, torch.Tensor]] = [{}]
    size = 0
    for key in sorted(state):
        nbytes = state[key].numel() * state[key].element_size()
        if shards[-1] and size + nbytes > max_shard_bytes:
            shards.append({})
            size = 0
        shards[-1][key] = state[key]
        size += nbytes
    # a directory that already holds a save must not keep stale shards / a stale index (the reader prefers the index)
    for old in os.listdir(ckpt_dir):
        if old == SAFETENSORS_INDEX or old == SAFETENSORS_SINGLE or (old.startswith("model-") and old.endswith(".safetensors")):
            os.remove(os.path.join(ckpt_dir, old))
    files, weight_map = [], {}
    for i, sh in enumerate(shards):
        fname = SAFETENSORS_SINGLE if len(shards) == 1 else f"model-{i + 1:05d}-of-{len(shards):05d}.safetensors"
        save_file(sh, os.path.join(ckpt_dir, fname), metadata={"format": "pt"})
        files.append(fname)
        for key in sh:
            weight_map[key] = fname
    if len(shards) > 1:
        total = sum(t.numel() * t.element_size() for t in state.values())
        with open(os.path.join(ckpt_dir, SAFETENSORS_INDEX), "w", encoding="utf-8") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1)
    payload = {"bits": cfg["bits"], "group_size": cfg["group_size"], "desc_act": cfg["desc_act"], "sym": cfg["sym"],
               "lm_head": cfg["lm_head"], "quant_method": cfg["method"], "checkpoint_format": cfg["format"], "pack_dtype": cfg["pack_dtype"],
               "meta": dict(cfg["meta"], quantizer=cfg["meta"].get("quantizer", ["gptqmodel_amd:test-writer"]))}
    if cfg["dynamic"]:
        payload["dynamic"] = cfg["dynamic"]
    with open(os.path.join(ckpt_dir, "quantize_config.json"), "w", encoding="utf-8") as f:
        json.dump(payload, f, indent=1)
    return files


__all__ = ["read_quantize_config", "normalize_quantize_config", "load_quantized_checkpoint", "save_quantized_checkpoint",
           "quantized_module_names"]
