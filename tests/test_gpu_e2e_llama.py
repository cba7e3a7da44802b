"""End-to-end drop-in check on the GPU: a random-init Hugging Face Llama is quantised (RTN, packed on the device),
its decoder nn.Linear modules are swapped for the HIP QuantLinear through the reference-style make_quant /
gptqmodel_post_init flow (gptqmodel/utils/model.py:398, :1281) and compared, prefill and KV-cache decode, with the same
model running dense fp16 weights = the dequantised checkpoint (what BACKEND.TORCH computes, torch.py:326-347)."""
import copy
import zlib

import pytest
import torch
import torch.nn as nn

from helpers import rel_err

pytestmark = pytest.mark.gpu

transformers = pytest.importorskip("transformers")


def _rtn(weight: torch.Tensor, group_size: int, bits: int):
    """Asymmetric round-to-nearest per (output row, K-group): scales/zeros as [out, G] like the reference quantiser hands
    them to pack()."""
    n, k = weight.shape
    w = weight.float().reshape(n, k // group_size, group_size)
    wmax, wmin = w.amax(dim=2), w.amin(dim=2)
    maxq = (1 << bits) - 1
    scales = ((wmax - wmin).clamp(min=1e-5) / maxq).half().float()
    zeros = torch.round(-wmin / scales).clamp(1, maxq)   # >= 1: representable in the v1 on-disk format (zero - 1 per field)
    return scales, zeros


def _get(model, name):
    mod = model
    for part in name.split("."):
        mod = getattr(mod, part)
    return mod


def _build(desc_act: bool, fuse, dtype, family="llama", gs=128, bits=4):
    from transformers import LlamaConfig, LlamaForCausalLM
    from gptqmodel_amd import ops
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.const import FORMAT
    from gptqmodel_amd.utils.model import fuse_siblings, gptqmodel_post_init, make_quant

    torch.manual_seed(7)
    # the decoder-layer fast path needs shapes inside the decode op's pipeline (K = 1024 -> 8 chunks, 2816 -> 22 padded to 24)
    dims = dict(hidden_size=512, intermediate_size=1408)
    if fuse in ("layers", "layers_dims_only"):   # (act-order in the kernel and the 2..4-row ops need whole 4-deep ring rounds: >= 16 chunks of K)
        dims = dict(hidden_size=2048, intermediate_size=5632)
    common = dict(num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, vocab_size=2048,
                  max_position_embeddings=128, tie_word_embeddings=False, **dims)
    if family == "qwen2":      # q/k/v carry a bias; same layer formula
        from transformers import Qwen2Config, Qwen2ForCausalLM
        dense = Qwen2ForCausalLM(Qwen2Config(**common)).to(dtype).cuda().eval()
        for layer in dense.model.layers:   # (random-init biases are zero: make them matter)
            for n in ("q_proj", "k_proj", "v_proj"):
                getattr(layer.self_attn, n).bias.data.normal_(0, 0.05)
    elif family == "qwen3":    # per-head q_norm / k_norm between the projections and rotary
        from transformers import Qwen3Config, Qwen3ForCausalLM
        dense = Qwen3ForCausalLM(Qwen3Config(head_dim=dims["hidden_size"] // 8, **common)).to(dtype).cuda().eval()
        for layer in dense.model.layers:
            layer.self_attn.q_norm.weight.data.normal_(1.0, 0.1)
            layer.self_attn.k_norm.weight.data.normal_(1.0, 0.1)
    elif family == "mistral":
        from transformers import MistralConfig, MistralForCausalLM
        dense = MistralForCausalLM(MistralConfig(sliding_window=64, **common)).to(dtype).cuda().eval()
    else:
        dense = LlamaForCausalLM(LlamaConfig(**common)).to(dtype).cuda().eval()
    quant = copy.deepcopy(dense)
    names = [n for n, m in dense.named_modules() if isinstance(m, nn.Linear) and ".layers." in n]
    assert len(names) == 14
    make_quant(quant, names, bits=bits, group_size=gs, desc_act=desc_act, sym=False, backend=BACKEND.AUTO,
               format=FORMAT.GPTQ, dtype=dtype)
    for name in names:
        lin, qm = _get(dense, name), _get(quant, name)
        assert isinstance(qm, HipGptqLinear)
        k = lin.in_features
        if desc_act:
            # one act-order permutation per input tensor: q/k/v (and gate/up) of a layer share it, as in real checkpoints
            gen_l = torch.Generator().manual_seed(zlib.crc32((name.rsplit(".", 1)[0] + str(k)).encode()))
            g_idx = (torch.randperm(k, generator=gen_l) // gs).to(torch.int32)
        else:
            g_idx = (torch.arange(k) // gs).to(torch.int32)
        # quantise the weight with its columns grouped by g_idx (group g = the columns whose g_idx == g)
        order = torch.argsort(g_idx.long(), stable=True).cuda()
        scales, zeros = _rtn(lin.weight.data[:, order], gs, bits)
        qm.pack(lin, scales, zeros, g_idx)             # the device packer, every bit width (continuous 2 / 3 / 4 / 8, planar 5 / 6 / 7)
        if bits in (4, 8):
            w = ops.dequant(qm.qweight, qm.qzeros, qm.scales, qm.g_idx, gs, bits, dtype)       # [K, N]
        else:
            w = qm.dequantize_weight().to(dtype)
        lin.weight.data.copy_(w.T)
    if fuse == "layers":
        from gptqmodel_amd.utils.hf_llama import fuse_llama_decoder_layers
        fused, skipped = fuse_llama_decoder_layers(quant)
        assert len(fused) == 2 and not skipped
    elif fuse == "layers_dims_only":
        pass   # the decode-op-friendly shapes, modules left unfused (tests/test_gpu_checkpoint.py fuses after loading from disk)
    elif fuse:
        for layer in quant.model.layers:
            assert fuse_siblings(layer.self_attn, ["q_proj", "k_proj", "v_proj"]) is not None
            assert fuse_siblings(layer.mlp, ["gate_proj", "up_proj"]) is not None
    import gptqmodel_amd.utils.model as _m
    _m.gptqmodel_post_init(quant)      # (looked up at call time: test_gpu_checkpoint.py swaps it to keep the checkpoint layout)
    return dense, quant


@pytest.mark.parametrize("desc_act,fuse,dtype", [(False, False, torch.float16), (False, True, torch.float16),
                                                  (True, True, torch.float16), (False, True, torch.bfloat16)])
def test_llama_prefill_and_decode_match_dense_dequantised_model(desc_act, fuse, dtype):
    dense, quant = _build(desc_act, fuse, dtype)
    tol = 2e-2 if dtype == torch.float16 else 6e-2
    torch.manual_seed(11)
    ids = torch.randint(0, 2048, (2, 24), device="cuda")
    with torch.no_grad():
        o_d = dense(input_ids=ids, use_cache=True)
        o_q = quant(input_ids=ids, use_cache=True)
        assert o_q.logits.dtype == dtype and o_q.logits.shape == o_d.logits.shape
        assert rel_err(o_q.logits.float().cpu().numpy(), o_d.logits.float().cpu().numpy()) < tol
        # teacher-forced KV-cache decode (M = batch = 2 rows per launch), both models fed the same tokens
        pk_d, pk_q = o_d.past_key_values, o_q.past_key_values
        tok = o_d.logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(6):
            s_d = dense(input_ids=tok, past_key_values=pk_d, use_cache=True)
            s_q = quant(input_ids=tok, past_key_values=pk_q, use_cache=True)
            assert rel_err(s_q.logits.float().cpu().numpy(), s_d.logits.float().cpu().numpy()) < tol
            pk_d, pk_q = s_d.past_key_values, s_q.past_key_values
            tok = s_d.logits[:, -1].argmax(-1, keepdim=True)


@pytest.mark.parametrize("desc_act,dtype,family", [(False, torch.float16, "llama"), (True, torch.float16, "llama"),
                                                   (False, torch.bfloat16, "llama"), (False, torch.float16, "qwen2"),
                                                   (False, torch.float16, "mistral"), (False, torch.float16, "qwen3")])
def test_llama_decoder_layers_on_decode_ops_match_dense(desc_act, dtype, family):
    """fuse_llama_decoder_layers: prefill (HF path through the fused modules) and single-token KV-cache decode (4 decode ops
    per layer around HF's attention) against the dense model holding the dequantised weights; the fast path must have run."""
    dense, quant = _build(desc_act, "layers", dtype, family)
    tol = 2e-2 if dtype == torch.float16 else 6e-2
    torch.manual_seed(13)
    ids = torch.randint(0, 2048, (1, 20), device="cuda")
    with torch.no_grad():
        o_d = dense(input_ids=ids, use_cache=True)
        o_q = quant(input_ids=ids, use_cache=True)
        assert rel_err(o_q.logits.float().cpu().numpy(), o_d.logits.float().cpu().numpy()) < tol
        assert all(L._gptqhip_fused["state"] is None for L in quant.model.layers)     # 20 tokens: not the decode ops ...
        # ... but the prefill path: ops.rmsnorm_gather (fused with the act-order gather) + forward_pregathered
        assert all(L._gptqhip_fused["prefill"].get("dtype") == dtype for L in quant.model.layers)
        pk_d, pk_q = o_d.past_key_values, o_q.past_key_values
        tok = o_d.logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(8):
            s_d = dense(input_ids=tok, past_key_values=pk_d, use_cache=True)
            s_q = quant(input_ids=tok, past_key_values=pk_q, use_cache=True)
            assert torch.isfinite(s_q.logits).all()
            assert rel_err(s_q.logits.float().cpu().numpy(), s_d.logits.float().cpu().numpy()) < tol
            pk_d, pk_q = s_d.past_key_values, s_q.past_key_values
            tok = s_d.logits[:, -1].argmax(-1, keepdim=True)
    if desc_act:   # down_proj's act-order permutation was folded into gate / up: no gather left there, q|k|v and o keep theirs
        assert all(L.mlp.down_proj.perm is None and L.self_attn.o_proj.perm is not None for L in quant.model.layers)
    states = [L._gptqhip_fused["state"] for L in quant.model.layers]
    assert all(st is not None for st in states) and not any(L._gptqhip_fused["disabled"] for L in quant.model.layers)
    assert states[1].prev is states[0]          # layer 1 consumes layer 0's residual stream + statistics in place
    # up to four tokens per call stay on the fast path: two sequences at q_len 1 ...
    with torch.no_grad():
        ids2 = torch.randint(0, 2048, (2, 1), device="cuda")
        assert rel_err(quant(input_ids=ids2).logits.float().cpu().numpy(), dense(input_ids=ids2).logits.float().cpu().numpy()) < tol
        assert desc_act or (2 in states[0].ops and 2 in states[1].ops)      # (no in-kernel permutation for M > 1: HF's path)
        # ... or three new tokens of one sequence on top of its KV cache (speculative-decoding verification step)
        o_d = dense(input_ids=ids, use_cache=True)
        o_q = quant(input_ids=ids, use_cache=True)
        nxt = torch.randint(0, 2048, (1, 3), device="cuda")
        s_d = dense(input_ids=nxt, past_key_values=o_d.past_key_values, use_cache=True)
        s_q = quant(input_ids=nxt, past_key_values=o_q.past_key_values, use_cache=True)
        assert rel_err(s_q.logits.float().cpu().numpy(), s_d.logits.float().cpu().numpy()) < tol
        assert desc_act or 3 in states[0].ops
        # six tokens (two sequences x three), nine, sixteen (two x eight): still the decode ops; seventeen: the prefill path
        for shape in ((2, 3), (1, 9), (3, 4), (2, 8)):
            idn = torch.randint(0, 2048, shape, device="cuda")
            assert rel_err(quant(input_ids=idn).logits.float().cpu().numpy(), dense(input_ids=idn).logits.float().cpu().numpy()) < tol
            assert desc_act or shape[0] * shape[1] in states[0].ops
        ids17 = torch.randint(0, 2048, (1, 17), device="cuda")
        assert rel_err(quant(input_ids=ids17).logits.float().cpu().numpy(), dense(input_ids=ids17).logits.float().cpu().numpy()) < tol
        assert 17 not in states[0].ops
        out = quant.generate(input_ids=ids[:, :8], max_new_tokens=6, do_sample=False, pad_token_id=0)
    assert out.shape == (1, 14)


@pytest.mark.parametrize("bits,desc_act", [(3, False), (5, True)])
def test_other_bit_widths_take_the_decode_ops(bits, desc_act):
    """3-bit (continuous) and 5-bit (planar) checkpoints under fuse_llama_decoder_layers: the modules are widened to the 4- / 8-bit
    layout up front (HipGptqLinear.widen_in_place: same code values), so sibling fusion, gate|up interleaving, act-order folding and
    the 4 decode ops per layer apply unchanged; against the dense model holding the dequantised weights."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    dense, quant = _build(desc_act, "layers", torch.float16, bits=bits)
    wide = 4 if bits < 4 else 8
    mods = [m for m in quant.modules() if isinstance(m, HipGptqLinear)]
    assert len(mods) == 8 and all(m.bits == wide and m.kernel_bits == wide and m._ready for m in mods)
    assert all(L.self_attn.o_proj.source_bits == bits for L in quant.model.layers)
    torch.manual_seed(17)
    ids = torch.randint(0, 2048, (1, 20), device="cuda")
    with torch.no_grad():
        o_d = dense(input_ids=ids, use_cache=True)
        o_q = quant(input_ids=ids, use_cache=True)
        assert rel_err(o_q.logits.float().cpu().numpy(), o_d.logits.float().cpu().numpy()) < 2e-2
        pk_d, pk_q = o_d.past_key_values, o_q.past_key_values
        tok = o_d.logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(4):
            s_d = dense(input_ids=tok, past_key_values=pk_d, use_cache=True)
            s_q = quant(input_ids=tok, past_key_values=pk_q, use_cache=True)
            assert rel_err(s_q.logits.float().cpu().numpy(), s_d.logits.float().cpu().numpy()) < 2e-2
            pk_d, pk_q = s_d.past_key_values, s_q.past_key_values
            tok = s_d.logits[:, -1].argmax(-1, keepdim=True)
    states = [L._gptqhip_fused["state"] for L in quant.model.layers]
    assert all(st is not None for st in states) and not any(L._gptqhip_fused["disabled"] for L in quant.model.layers)
    if desc_act:
        assert all(L.mlp.down_proj.perm is None for L in quant.model.layers)


def test_llama_decoder_layers_group_size_32_take_the_decode_ops():
    """A 32g checkpoint (a group constant per K-step) on the fused decoder layers: prefill path + decode ops, as for 128g."""
    dense, quant = _build(False, "layers", torch.float16, gs=32)
    torch.manual_seed(17)
    ids = torch.randint(0, 2048, (1, 20), device="cuda")
    with torch.no_grad():
        o_d, o_q = dense(input_ids=ids, use_cache=True), quant(input_ids=ids, use_cache=True)
        assert rel_err(o_q.logits.float().cpu().numpy(), o_d.logits.float().cpu().numpy()) < 2e-2
        pk_d, pk_q = o_d.past_key_values, o_q.past_key_values
        tok = o_d.logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(4):
            s_d = dense(input_ids=tok, past_key_values=pk_d, use_cache=True)
            s_q = quant(input_ids=tok, past_key_values=pk_q, use_cache=True)
            assert rel_err(s_q.logits.float().cpu().numpy(), s_d.logits.float().cpu().numpy()) < 2e-2
            pk_d, pk_q = s_d.past_key_values, s_q.past_key_values
            tok = s_d.logits[:, -1].argmax(-1, keepdim=True)
        ids6 = torch.randint(0, 2048, (2, 3), device="cuda")
        assert rel_err(quant(input_ids=ids6).logits.float().cpu().numpy(), dense(input_ids=ids6).logits.float().cpu().numpy()) < 2e-2
    states = [L._gptqhip_fused["state"] for L in quant.model.layers]
    assert all(st is not None and 1 in st.ops and 6 in st.ops for st in states)
    assert not any(L._gptqhip_fused["disabled"] for L in quant.model.layers)


def test_llama_generate_runs_on_quantised_model():
    dense, quant = _build(False, True, torch.float16)
    ids = torch.randint(0, 2048, (1, 8), device="cuda")
    with torch.no_grad():
        out = quant.generate(input_ids=ids, max_new_tokens=8, do_sample=False, pad_token_id=0)
    assert out.shape == (1, 16)
    n_launch_modules = sum(1 for m in quant.modules() if type(m).__name__ == "HipGptqLinear")
    assert n_launch_modules == 2 * 4  # fused qkv, o, fused gate_up, down per layer


def test_fused_decoder_layers_hidden_state_capture_and_dequantize_model():
    """(1) output_hidden_states across decode steps: the fast path returns a view of a per-layer buffer, so what transformers
    records through its forward hooks must be a copy -- the states of step 1 stay intact after step 2.  (2) forward hooks on a
    sub-module the fast path bypasses send the layer through HF's own code (the hook fires).  (3) dequantize_model on a model
    rewritten by fuse_llama_decoder_layers: HF's forward restored, the interleaved gate|up module split back into gate_proj /
    up_proj, logits equal to the dense twin."""
    from gptqmodel_amd.utils.model import dequantize_model
    dense, quant = _build(True, "layers", torch.float16)
    torch.manual_seed(3)
    ids = torch.randint(0, 2048, (1, 6), device="cuda")
    with torch.no_grad():
        o_q = quant(input_ids=ids, use_cache=True)
        o_d = dense(input_ids=ids, use_cache=True)
        tok = o_d.logits[:, -1].argmax(-1, keepdim=True)
        s1 = quant(input_ids=tok, past_key_values=o_q.past_key_values, use_cache=True, output_hidden_states=True)
        d1 = dense(input_ids=tok, past_key_values=o_d.past_key_values, use_cache=True, output_hidden_states=True)
        kept = [h.clone() for h in s1.hidden_states]
        tok2 = d1.logits[:, -1].argmax(-1, keepdim=True)
        quant(input_ids=tok2, past_key_values=s1.past_key_values, use_cache=True, output_hidden_states=True)
        assert all(torch.equal(a, b) for a, b in zip(kept, s1.hidden_states)), "recorded hidden states alias a reused buffer"
        for a, b in zip(s1.hidden_states, d1.hidden_states):
            assert rel_err(a.float().cpu().numpy(), b.float().cpu().numpy()) < 2e-2
        fired = []
        hk = quant.model.layers[0].mlp.register_forward_hook(lambda m, i, o: fired.append(1))
        quant(input_ids=tok)
        hk.remove()
        assert fired, "a forward hook on a bypassed sub-module must be honoured (HF path)"
        want = dense(input_ids=ids).logits
        dequantize_model(quant, device="cuda", dtype=torch.float16)
        assert not any(type(m).__name__ == "HipGptqLinear" for m in quant.modules())
        assert all(isinstance(L.mlp.gate_proj, nn.Linear) and isinstance(L.mlp.up_proj, nn.Linear) and not hasattr(L, "_gptqhip_fused")
                   for L in quant.model.layers)
        got = quant(input_ids=ids).logits
        assert rel_err(got.float().cpu().numpy(), want.float().cpu().numpy()) < 2e-2


def test_hf_contract_check_rejects_older_calling_conventions():
    """fuse_llama_decoder_layers verifies the transformers calling convention its fast paths mirror: classes with the
    4.48-4.55 convention (`past_key_value` in kwargs, tuple return, cache_kwargs) must be skipped with a reason, not run with a
    silently stale KV cache.  Pure host logic, but it needs the quant modules, hence a GPU test."""
    from gptqmodel_amd.utils import hf_llama

    class OldAttn(nn.Module):
        def forward(self, hidden_states, position_embeddings=None, attention_mask=None, past_key_value=None, **kwargs):
            cache_kwargs = {}
            return hidden_states, None

    class OldLayer(nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn = OldAttn()

        def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None, position_embeddings=None, **kwargs):
            outputs = (hidden_states,)
            return outputs

    assert "past_key_values" in hf_llama._hf_contract(OldLayer())
    dense, quant = _build(False, False, torch.float16)
    assert hf_llama._hf_contract(dense.model.layers[0]) is None


def test_mistral_fast_paths_hand_the_sliding_window_to_the_attention_interface():
    """MistralAttention has no `sliding_window` attribute: HF's own forward reads it from the config and passes it to the attention
    interface as a keyword (modeling_mistral.py).  The decode-op fast path and the prefill path must do the same -- with
    flash-attention the window is NOT encoded in the mask.  A recording attention function registered through
    transformers' AttentionInterface sees what every path hands over."""
    from transformers import AttentionInterface
    from transformers.modeling_utils import ALL_ATTENTION_FUNCTIONS
    dense, quant = _build(False, "layers", torch.float16, "mistral")
    assert not hasattr(quant.model.layers[0].self_attn, "sliding_window")
    seen = []
    sdpa = ALL_ATTENTION_FUNCTIONS["sdpa"]

    def capture(module, q, k, v, mask, **kw):
        seen.append((q.shape[-2], kw.get("sliding_window", "MISSING")))
        return sdpa(module, q, k, v, mask, **kw)

    AttentionInterface.register("gptqhip_capture_sw", capture)
    for m in (dense, quant):
        m.set_attn_implementation("gptqhip_capture_sw")
    ids = torch.randint(0, 2048, (1, 20), device="cuda")
    with torch.no_grad():
        for model in (dense, quant):
            seen.clear()
            o = model(input_ids=ids, use_cache=True)                       # 20 tokens: HF's layer / the prefill path
            model(input_ids=ids[:, :1], past_key_values=o.past_key_values, use_cache=True)   # 1 token: HF's layer / the decode ops
            assert [w for _, w in seen] == [64] * 4 and [n for n, _ in seen] == [20, 20, 1, 1], seen
    assert all(L._gptqhip_fused["state"] is not None and L._gptqhip_fused["prefill"].get("dtype") == torch.float16
               for L in quant.model.layers), "the fast paths must have run"


def test_awq_llama_decoder_layers_prefill_and_decode_match_dense():
    """An AWQ (gemm format) model through the same two fast paths: the prefill path (ops.rmsnorm_gather feeding
    HipAwqLinear.forward_pregathered) and the decode ops, against the dense model holding the AWQ-dequantised weights
    (packing_utils.py:106-121: (code - zero) * scale, rounded once)."""
    import numpy as np
    from transformers import LlamaConfig, LlamaForCausalLM
    from oracle import gptq_oracle as O
    from gptqmodel_amd.nn_modules.qlinear.hip_awq import HipAwqLinear
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.const import FORMAT, METHOD
    from gptqmodel_amd.utils.hf_llama import fuse_llama_decoder_layers
    from gptqmodel_amd.utils.model import gptqmodel_post_init, make_quant

    torch.manual_seed(5)
    dtype, gs = torch.float16, 128
    dense = LlamaForCausalLM(LlamaConfig(num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, vocab_size=2048,
                                         max_position_embeddings=128, tie_word_embeddings=False, hidden_size=2048,
                                         intermediate_size=5632)).to(dtype).cuda().eval()
    quant = copy.deepcopy(dense)
    names = [n for n, m in dense.named_modules() if isinstance(m, nn.Linear) and ".layers." in n]
    make_quant(quant, names, bits=4, group_size=gs, desc_act=False, sym=False, backend=BACKEND.AUTO, format=FORMAT.GEMM,
               quant_method=METHOD.AWQ, dtype=dtype)
    for name in names:
        lin, qm = _get(dense, name), _get(quant, name)
        assert isinstance(qm, HipAwqLinear)
        scales, zeros = _rtn(lin.weight.data, gs, 4)                                   # [N, G]
        n, k = lin.weight.shape
        w = lin.weight.data.float().reshape(n, k // gs, gs)
        codes = torch.clamp(torch.round(w / scales[:, :, None]) + zeros[:, :, None], 0, 15).reshape(n, k)
        qm.qweight.data = torch.from_numpy(O.pack_awq_cols(codes.T.contiguous().cpu().numpy().astype(np.uint8))).cuda()
        qm.qzeros.data = torch.from_numpy(O.pack_awq_cols(zeros.T.contiguous().cpu().numpy().astype(np.uint8))).cuda()
        qm.scales.data = scales.T.contiguous().to(dtype).cuda()
        deq = ((codes.reshape(n, k // gs, gs) - zeros[:, :, None]) * scales[:, :, None].half().float()).reshape(n, k)
        lin.weight.data.copy_(deq.to(dtype))
    quant = quant.cuda()
    fused, skipped = fuse_llama_decoder_layers(quant)
    assert len(fused) == 2 and not skipped, skipped
    gptqmodel_post_init(quant)
    tol = 2e-2
    ids = torch.randint(0, 2048, (1, 20), device="cuda")
    with torch.no_grad():
        o_d, o_q = dense(input_ids=ids, use_cache=True), quant(input_ids=ids, use_cache=True)
        assert rel_err(o_q.logits.float().cpu().numpy(), o_d.logits.float().cpu().numpy()) < tol
        assert all(L._gptqhip_fused["prefill"].get("dtype") == dtype for L in quant.model.layers)    # the prefill path ran
        pk_d, pk_q = o_d.past_key_values, o_q.past_key_values
        tok = o_d.logits[:, -1].argmax(-1, keepdim=True)
        for _ in range(4):
            s_d = dense(input_ids=tok, past_key_values=pk_d, use_cache=True)
            s_q = quant(input_ids=tok, past_key_values=pk_q, use_cache=True)
            assert rel_err(s_q.logits.float().cpu().numpy(), s_d.logits.float().cpu().numpy()) < tol
            pk_d, pk_q = s_d.past_key_values, s_q.past_key_values
            tok = s_d.logits[:, -1].argmax(-1, keepdim=True)
        ids8 = torch.randint(0, 2048, (2, 4), device="cuda")
        assert rel_err(quant(input_ids=ids8).logits.float().cpu().numpy(), dense(input_ids=ids8).logits.float().cpu().numpy()) < tol
    states = [L._gptqhip_fused["state"] for L in quant.model.layers]
    assert all(st is not None and 1 in st.ops and 8 in st.ops for st in states)
