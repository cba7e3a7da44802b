#!/usr/bin/env python
"""Checkpoint-directory round trip on a small random-init HF Llama (no network needed):

  1. quantise every decoder nn.Linear to 4-bit g128 (RTN; --desc-act for an act-order checkpoint) and pack it on the device;
  2. WRITE the model in the on-disk GPTQ layout (`format: gptq` = v1 zero-points, sharded safetensors + index, quantize_config.json)
     with gptqmodel_amd.utils.checkpoint.save_quantized_checkpoint;
  3. build a fresh model from the directory's config.json and LOAD it the way the reference's loader does
     (make_quant -> safetensors into the buffers -> v1 -> v2 -> decoder-layer fusion -> gptqmodel_post_init:
     gptqmodel/models/loader.py:1092,1646-1675,1804) with load_quantized_checkpoint;
  4. generate.

    python examples/checkpoint_roundtrip.py [--dir /tmp/ckpt] [--desc-act]
"""
import argparse
import os
import sys
import tempfile

import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=None)
    ap.add_argument("--desc-act", action="store_true")
    args = ap.parse_args()
    from transformers import LlamaConfig, LlamaForCausalLM
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.checkpoint import load_quantized_checkpoint, read_quantize_config, save_quantized_checkpoint
    from gptqmodel_amd.utils.const import FORMAT
    from gptqmodel_amd.utils.model import make_quant
    from hf_llama_dropin import rtn

    dev = torch.device("cuda", 0)
    cfg = LlamaConfig(hidden_size=2048, intermediate_size=5632, num_hidden_layers=4, num_attention_heads=16, num_key_value_heads=4,
                      vocab_size=4096, max_position_embeddings=512, tie_word_embeddings=False)
    torch.manual_seed(0)
    with torch.device(dev):
        model = LlamaForCausalLM(cfg).to(torch.float16).eval()
    names = [n for n, m in model.named_modules() if isinstance(m, nn.Linear) and ".layers." in n]
    floats = dict(model.named_modules())
    floats = {n: floats[n] for n in names}
    make_quant(model, names, bits=4, group_size=128, desc_act=args.desc_act, sym=False, backend=BACKEND.AUTO, format=FORMAT.GPTQ,
               dtype=torch.float16)
    mods = dict(model.named_modules())
    perms = {}
    for n in names:
        lin, qm = floats[n], mods[n]
        if args.desc_act:
            key = (n.rsplit(".", 1)[0], lin.in_features)
            perms.setdefault(key, (torch.randperm(lin.in_features) // 128).to(torch.int32))
            g_idx = perms[key]
            order = torch.argsort(g_idx.long(), stable=True).to(dev)
            scales, zeros = rtn(lin.weight.data[:, order], 128, 4)
        else:
            g_idx = (torch.arange(lin.in_features) // 128).to(torch.int32)
            scales, zeros = rtn(lin.weight.data, 128, 4)
        qm.pack(lin, scales, zeros.clamp(min=1), g_idx)     # zero-points >= 1: representable in the v1 on-disk format
    out_dir = args.dir or tempfile.mkdtemp(prefix="gptqhip_ckpt_")
    files = save_quantized_checkpoint(model, out_dir, {"bits": 4, "group_size": 128, "desc_act": args.desc_act, "sym": False,
                                                        "quant_method": "gptq", "checkpoint_format": "gptq",
                                                        "meta": {"quantizer": ["gptqmodel:5.0.0"]}}, max_shard_bytes=16 << 20)
    model.config.save_pretrained(out_dir)
    print(f"wrote {len(files)} safetensors shard(s) + quantize_config.json to {out_dir}: {read_quantize_config(out_dir)}")
    del model, floats, mods
    torch.cuda.empty_cache()

    fresh = LlamaForCausalLM(LlamaConfig.from_pretrained(out_dir)).to(torch.float16)
    loaded = load_quantized_checkpoint(fresh, out_dir, device=dev, fuse_decoder_layers=True)
    n_q = sum(1 for m in loaded.modules() if type(m).__name__ == "HipGptqLinear")
    ids = torch.randint(0, cfg.vocab_size, (1, 16), device=dev)
    with torch.no_grad():
        out = loaded.generate(input_ids=ids, max_new_tokens=16, do_sample=False, pad_token_id=0)
    print(f"loaded: {n_q} HIP quant modules (decoder layers fused), generated {out.shape[1] - ids.shape[1]} tokens: {out[0, -8:].tolist()}")


if __name__ == "__main__":
    main()
