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
    qcfg = {"bits": 4, "group_size": 128, "desc_act": desc_act, "sym": False, "quant_method": "gptq", "checkpoint_format": "gptq",
            "meta": {"quantizer": ["gptqmodel:5.0.0"]}}     # (a producer entry the reference's v1 loader accepts; the writer insists on one)
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


@pytest.mark.parametrize("bits,fmt", [(3, "gptq"), (6, "gptq"), (2, "gptq_v2"), (5, "gptq_p"), (3, "gptq_p")])
def test_other_bit_widths_directory_round_trip(tmp_path, bits, fmt):
    """A 2- / 3- / 6-bit directory: modules in the checkpoint layout -> save (`format: gptq` stores the reference
    writer's v1 zero-points, utils/model.py:900-943) -> load into a fresh skeleton (make_quant picks HipGptqLinear, v1 -> v2 through the
    decoded values, post_init widens the codes) -> forward against the oracle's dequantised weights."""
    import numpy as np
    import torch.nn as nn
    from helpers import assert_forward_close, torch_to_f32
    from oracle import gptq_oracle as O
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.checkpoint import load_quantized_checkpoint, save_quantized_checkpoint
    from gptqmodel_amd.utils.const import FORMAT
    K, H, gs = 256, 512, 64
    rng = np.random.RandomState(bits)
    planar = True if (fmt == "gptq_p" and bits == 3) else None      # split-plane 3-bit words exist under gptq_p only

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.norm = nn.LayerNorm(K, elementwise_affine=True, bias=False)
            self.up = nn.Linear(K, H, bias=False)
            self.down = nn.Linear(H, K, bias=False)

        def forward(self, x):
            return self.down(self.up(self.norm(x)))

    def tensors(k, n):
        codes = rng.randint(0, 1 << bits, size=(k, n)).astype(np.uint8)
        zeros = rng.randint(1, 1 << bits, size=(k // gs, n)).astype(np.uint8)
        scales = O.round_to(rng.rand(k // gs, n).astype(np.float32) * 0.02 + 0.005, "fp16")
        return O.pack_rows_any(codes, bits, planar), O.pack_cols_any(zeros, bits, planar), scales, (np.arange(k) // gs).astype(np.int32)

    src = Net().half()
    want = {}
    for name, (k, n) in (("up", (K, H)), ("down", (H, K))):
        qw, qz, sc, gi = tensors(k, n)
        lin = HipGptqLinear(bits=bits, group_size=gs, sym=False, desc_act=False, in_features=k, out_features=n, bias=False,
                            format=FORMAT(fmt))
        assert bool(lin.planar) == (bits in (5, 6, 7) or bool(planar))
        lin.qweight, lin.qzeros, lin.scales, lin.g_idx = (torch.from_numpy(qw), torch.from_numpy(qz), torch.from_numpy(sc).half(),
                                                          torch.from_numpy(gi))
        lin.qzero_format(format=2)
        setattr(src, name, lin)
        want[name] = (qz, O.dequant_gptq(qw, qz, sc, gi, bits, planar=planar))
    ckpt = str(tmp_path / "ckpt")
    qcfg = {"bits": bits, "group_size": gs, "desc_act": False, "sym": False, "quant_method": "gptq", "checkpoint_format": fmt,
            "meta": {"quantizer": ["gptqmodel:5.0.0"]}}
    save_quantized_checkpoint(src, ckpt, qcfg)
    from safetensors.torch import load_file
    disk = load_file(os.path.join(ckpt, "model.safetensors"))
    z_disk = O.unpack_cols_any(disk["up.qzeros"].numpy(), bits, planar).astype(np.int32)
    z_live = O.unpack_cols_any(want["up"][0], bits, planar).astype(np.int32)
    assert np.array_equal(z_disk, (z_live - 1) & ((1 << bits) - 1) if fmt == "gptq" else z_live)

    torch.manual_seed(7)
    loaded = load_quantized_checkpoint(Net().half(), ckpt, device="cuda")
    assert all(isinstance(m, HipGptqLinear) and m._ready and m.kernel_bits == (4 if bits < 4 else 8) for m in (loaded.up, loaded.down))
    assert np.array_equal(loaded.up.dequantize_weight().float().cpu().numpy(), want["up"][1])
    x = torch.randn(5, K, device="cuda", dtype=torch.float16)
    with torch.no_grad():
        xn = loaded.norm(x)
        h = loaded.up(xn)
        y = loaded.down(h)
    assert_forward_close(torch_to_f32(h), O.matmul_round(xn.float().cpu().numpy(), want["up"][1], None, "fp16"), "fp16")
    assert_forward_close(torch_to_f32(y), O.matmul_round(h.float().cpu().numpy(), want["down"][1], None, "fp16"), "fp16")
