#!/usr/bin/env python
"""End-to-end drop-in demo on a random-init Hugging Face Llama (no checkpoint / network needed):

  1. build the model (Llama-3-8B shapes with --size 8b, a small one by default), weights created on the GPU;
  2. quantise every decoder nn.Linear to 4-bit g128 (RTN) and pack it ON THE DEVICE through the module's pack()
     (reference API: PackableQuantLinear.pack_block, qlinear/__init__.py:1036);
  3. swap the modules with make_quant (BACKEND.AUTO -> HipGptqLinear), gptqmodel_post_init (which itself moves the decoder layers onto the fused decode ops);
  4. greedy-decode with HF generate (eager; Python-bound) and with ONE captured HIP graph per decode step over a static
     KV cache (what a serving stack would do), and report tokens/s for both.

The numbers here include attention, norms, rotary, lm_head ... (plain PyTorch ops); bench.py isolates the quantised
linears, which is the path this repository owns."""
import argparse
import sys
import time
import os

import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

SIZES = {
    "tiny": dict(hidden_size=512, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2,
                 vocab_size=2048),
    "1b": dict(hidden_size=2048, intermediate_size=8192, num_hidden_layers=16, num_attention_heads=32, num_key_value_heads=8,
               vocab_size=128256),
    "8b": dict(hidden_size=4096, intermediate_size=14336, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=8,
               vocab_size=128256),
}


def rtn(weight, group_size, bits):
    n, k = weight.shape
    w = weight.float().reshape(n, k // group_size, group_size)
    wmax, wmin = w.amax(dim=2), w.amin(dim=2)
    maxq = (1 << bits) - 1
    scales = ((wmax - wmin).clamp(min=1e-5) / maxq).half().float()
    zeros = torch.round(-wmin / scales).clamp(0, maxq)
    return scales, zeros


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="tiny", choices=list(SIZES))
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--no-fuse", action="store_true")
    ap.add_argument("--batch", type=int, default=1, help="sequences decoded together (<= 16 stay on the decode-op fast path)")
    ap.add_argument("--siblings-only", action="store_true", help="fuse q/k/v and gate/up only (no decode-op layer fast path)")
    ap.add_argument("--quant-lm-head", action="store_true", help="also quantise lm_head (qcfg.lm_head upstream, loader.py:1376)")
    ap.add_argument("--desc-act", action="store_true", help="act-order checkpoint: g_idx = a random permutation per input tensor "
                                                             "(shared by q|k|v and by gate|up like a real GPTQ act-order checkpoint)")
    args = ap.parse_args()
    run(args.size, args.dtype, args.new_tokens, not args.no_fuse, args.quant_lm_head, verbose=True, decode_ops=not args.siblings_only,
        batch=args.batch, desc_act=args.desc_act)


def run(size="8b", dtype_name="fp16", new_tokens=64, fuse=True, quant_lm_head=False, verbose=False, decode_ops=True, batch=1,
        desc_act=False):
    """Returns {"eager_tokens_per_s", "graph_tokens_per_s" | None, "build_s", ...}; bench.py reports it as `e2e`."""
    import types
    args = types.SimpleNamespace(size=size, dtype=dtype_name, new_tokens=new_tokens, no_fuse=not fuse, quant_lm_head=quant_lm_head)
    res = {"model": f"random-init HF LlamaForCausalLM, Llama-3-{size} shapes, every decoder nn.Linear RTN-quantised to int4 g128 "
                    "and swapped through make_quant -> (layer fusion, see layer_path) -> gptqmodel_post_init; real attention / rotary / KV cache / lm_head",
           "new_tokens": new_tokens}
    say = print if verbose else (lambda *a, **k: None)
    from transformers import LlamaConfig, LlamaForCausalLM
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.const import FORMAT
    from gptqmodel_amd.utils.model import fuse_siblings, gptqmodel_post_init, make_quant

    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda", 0)
    cfg = LlamaConfig(max_position_embeddings=2048, tie_word_embeddings=False, **SIZES[args.size])
    torch.manual_seed(0)
    t0 = time.time()
    with torch.device(dev):
        model = LlamaForCausalLM(cfg).to(dtype).eval()
    names = [n for n, m in model.named_modules() if isinstance(m, nn.Linear) and
             (".layers." in n or (args.quant_lm_head and n == "lm_head"))]
    mods = dict(model.named_modules())
    floats = {n: mods[n] for n in names}
    make_quant(model, names, bits=4, group_size=128, desc_act=desc_act, sym=False, backend=BACKEND.AUTO, format=FORMAT.GPTQ,
               dtype=dtype)
    mods = dict(model.named_modules())
    res["desc_act"] = desc_act
    perms = {}
    for n in names:
        lin, qm = floats[n], mods[n]
        if desc_act:
            key = (n.rsplit(".", 1)[0], lin.in_features)          # siblings that share an input share the permutation
            if key not in perms:
                perms[key] = (torch.randperm(lin.in_features) // 128).to(torch.int32)
            g_idx = perms[key]
            order = torch.argsort(g_idx.long(), stable=True).to(lin.weight.device)
            scales, zeros = rtn(lin.weight.data[:, order], 128, 4)  # group g = the columns whose g_idx == g
        else:
            g_idx = (torch.arange(lin.in_features) // 128).to(torch.int32)
            scales, zeros = rtn(lin.weight.data, 128, 4)
        qm.pack(lin, scales, zeros, g_idx)
    del floats
    res["layer_path"] = "per-module launches"
    if args.no_fuse:
        model._gptqhip_auto_fuse = False      # opt out of gptqmodel_post_init's decoder-layer pass (same as GPTQHIP_AUTO_FUSE=0)
    elif not decode_ops:
        for layer in model.model.layers:
            fuse_siblings(layer.self_attn, ["q_proj", "k_proj", "v_proj"])
            fuse_siblings(layer.mlp, ["gate_proj", "up_proj"])
        res["layer_path"] = "fused siblings (q|k|v, gate|up), torch glue"
    # round 6: NO manual rewrite on the default path -- gptqmodel_post_init itself (utils.hf_llama.auto_fuse; the reference's own
    # gptqmodel_post_init does the same through integration/gptqmodel_overlay/utils/model.patch) puts every recognised decoder layer
    # on the four fused decode ops: this is what a loader returns
    gptqmodel_post_init(model)
    if getattr(model, "_gptqhip_fused_layers", 0):
        res["layer_path"] = (f"decode ops in {model._gptqhip_fused_layers} layers ({len(getattr(model, '_gptqhip_skipped_layers', []))} skipped), "
                             "applied by gptqmodel_post_init itself")
    torch.cuda.synchronize()
    del mods, qm, lin   # (the pre-fusion modules are no longer part of the model: release their checkpoint-layout tensors)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    nq = sum(1 for m in model.modules() if type(m).__name__ == "HipGptqLinear")
    res["build_s"] = time.time() - t0
    res["quant_launches_per_token"] = nq
    say(f"layer path: {res['layer_path']}")
    say(f"built + quantised + packed + repacked {len(names)} linears ({nq} launches/token) in {time.time() - t0:.1f} s; "
        f"GPU memory {torch.cuda.memory_allocated() / 2**30:.2f} GiB")

    ids = torch.randint(0, cfg.vocab_size, (batch, 16), device=dev)
    res["batch"] = batch
    with torch.no_grad():
        model.generate(input_ids=ids, max_new_tokens=4, do_sample=False, pad_token_id=0)  # warm-up
        torch.cuda.synchronize()
        t0 = time.time()
        out = model.generate(input_ids=ids, max_new_tokens=args.new_tokens, do_sample=False, pad_token_id=0)
        torch.cuda.synchronize()
        dt = time.time() - t0
    res["eager_tokens_per_s"] = batch * args.new_tokens / dt
    say(f"HF generate (eager, Python-bound): {batch * args.new_tokens / dt:.1f} tokens/s (batch {batch})")

    # prefill: one 2048-token prompt through the same modules (the MFMA-tiled kernel; HF's attention / norms around it)
    try:
        n_prompt = 2048 if args.size != "tiny" else 256
        pids = torch.randint(0, cfg.vocab_size, (1, n_prompt), device=dev)
        with torch.no_grad():
            for _ in range(2):
                model(input_ids=pids, use_cache=False, logits_to_keep=1)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(3):
                model(input_ids=pids, use_cache=False, logits_to_keep=1)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / 3
        res["prefill_tokens_per_s"] = n_prompt / dt
        res["prefill_prompt_tokens"] = n_prompt
        say(f"prefill of one {n_prompt}-token prompt: {n_prompt / dt:.0f} tokens/s ({dt * 1e3:.1f} ms)")
    except Exception as e:  # noqa: BLE001
        res["prefill_error"] = f"{type(e).__name__}: {str(e)[:200]}"
        say(f"prefill leg skipped: {type(e).__name__}: {e}")

    # one HIP graph per decode step over a static KV cache
    try:
        from transformers import StaticCache
        with torch.no_grad():
            cache = StaticCache(config=cfg, max_cache_len=256)
            pos = torch.arange(ids.shape[1], device=dev)
            o = model(input_ids=ids, past_key_values=cache, cache_position=pos, use_cache=True)
            tok = o.logits[:, -1:].argmax(-1)
            s_tok = tok.clone()
            s_pos = torch.tensor([ids.shape[1]], device=dev)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(2):  # warm-up outside capture
                    model(input_ids=s_tok, past_key_values=cache, cache_position=s_pos, use_cache=True)
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    lo = model(input_ids=s_tok, past_key_values=cache, cache_position=s_pos, use_cache=True).logits
                    nxt = lo[:, -1:].argmax(-1)
                torch.cuda.synchronize()
                t0 = time.time()
                for i in range(args.new_tokens):
                    g.replay()
                    s_tok.copy_(nxt)
                    s_pos.add_(1)
                stream.synchronize()
                dt = time.time() - t0
        res["graph_tokens_per_s"] = batch * args.new_tokens / dt
        say(f"graph-replayed decode step (static KV cache): {batch * args.new_tokens / dt:.1f} tokens/s (batch {batch})")
    except Exception as e:  # transformers API drift must not hide the eager result above
        res["graph_tokens_per_s"] = None
        res["graph_error"] = f"{type(e).__name__}: {str(e)[:200]}"
        say(f"graph-replayed decode skipped: {type(e).__name__}: {e}")
    del model
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
