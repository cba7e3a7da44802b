"""What the reference's loader flow yields on the MI355X with the overlay applied: an HF Llama whose decoder
nn.Linear modules go through the REFERENCE's own make_quant (gptqmodel/utils/model.py:398, BACKEND.AUTO on DEVICE.ROCM -> the overlay
classes) and the REFERENCE's own gptqmodel_post_init (utils/model.py:1281, + integration/gptqmodel_overlay/utils/model.patch) must come
out with every decoder layer on the four fused decode ops -- no manual fuse_llama_decoder_layers call -- and decode like the dense
dequantised model.  GPTQHIP_AUTO_FUSE=0 must leave the plugin modules alone."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["/root/reference", os.path.join(ROOT, "oracle", "_ref")]
SRC = next((s for s in SOURCES if os.path.isdir(os.path.join(s, "gptqmodel", "nn_modules", "qlinear"))), None)

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(SRC is None, reason="no reference tree and no oracle/_ref snapshot")]
pytest.importorskip("transformers")

_SCRIPT = r'''
import copy, json, os, sys
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch
import torch.nn as nn
assert torch.cuda.is_available()
torch.zeros(1, device="cuda:0")
os.environ["GPTQ_REFERENCE_ROOT"] = {tree!r}
from oracle.ref_import import load_reference
ref = load_reference()
os.environ.pop("CUDA_VISIBLE_DEVICES", None)
from transformers import LlamaConfig, LlamaForCausalLM
from gptqmodel.nn_modules.qlinear import hip as hipmod, GPTQQuantLinear
from gptqmodel.utils.backend import BACKEND
from gptqmodel.utils.model import make_quant, gptqmodel_post_init
from gptqmodel.quantization import FORMAT, METHOD
from gptqmodel.quantization.config import QuantizeConfig
from gptqmodel.models._const import DEVICE
from gptqmodel_amd import ops

dtype, gs = torch.float16, 128
torch.manual_seed(7)
cfg = LlamaConfig(num_hidden_layers=2, num_attention_heads=8, num_key_value_heads=2, vocab_size=2048, max_position_embeddings=128,
                  tie_word_embeddings=False, hidden_size=2048, intermediate_size=5632)
dense = LlamaForCausalLM(cfg).to(dtype).cuda().eval()


def get(model, name):
    for part in name.split("."):
        model = getattr(model, part)
    return model


def rtn(weight, group_size, bits=4):
    n, k = weight.shape
    w = weight.float().reshape(n, k // group_size, group_size)
    wmax, wmin = w.amax(dim=2), w.amin(dim=2)
    scales = ((wmax - wmin).clamp(min=1e-5) / 15).half().float()
    zeros = torch.round(-wmin / scales).clamp(1, 15)
    return scales, zeros


names = [n for n, m in dense.named_modules() if isinstance(m, nn.Linear) and ".layers." in n]
out = {{}}
for auto in ("1", "0"):
    os.environ["GPTQHIP_AUTO_FUSE"] = auto
    quant = copy.deepcopy(dense).cpu()
    qcfg = QuantizeConfig(bits=4, group_size=gs, desc_act=False, sym=False, method=METHOD.GPTQ, format=FORMAT.GPTQ_V2)
    make_quant(quant, qcfg, quant_result={{n: {{}} for n in names}}, backend=BACKEND.AUTO, lm_head_name="lm_head", device=DEVICE.ROCM,
               from_quantized=True, dtype=dtype)
    quant.cuda()
    made = set()
    for name in names:
        lin, qm = get(dense, name), get(quant, name)
        made.add(type(qm).__module__ + "." + type(qm).__name__)
        g_idx = (torch.arange(lin.in_features) // gs).to(torch.int32)
        scales, zeros = rtn(lin.weight.data, gs)
        qm.pack(lin, scales, zeros, g_idx)
        if auto == "1":
            lin.weight.data.copy_(ops.dequant(qm.qweight, qm.qzeros, qm.scales, qm.g_idx, gs, 4, dtype).T)
    gptqmodel_post_init(quant, use_act_order=False, quantize_config=qcfg)      # the REFERENCE's function
    rec = {{"made": sorted(made), "fused_layers": getattr(quant, "_gptqhip_fused_layers", None),
           "skipped": getattr(quant, "_gptqhip_skipped_layers", None),
           "layer_has_fast_path": [hasattr(l, "_gptqhip_fused") for l in quant.model.layers],
           "gate_proj_still_there": [hasattr(l.mlp, "gate_proj") for l in quant.model.layers]}}
    torch.manual_seed(11)
    ids = torch.randint(0, 2048, (1, 12), device="cuda")
    with torch.no_grad():
        o_d = dense(input_ids=ids, use_cache=True)
        o_q = quant(input_ids=ids, use_cache=True)
        rec["prefill_rel"] = float((o_q.logits.float() - o_d.logits.float()).abs().max() / o_d.logits.float().abs().max())
        nxt = o_d.logits[:, -1:].argmax(-1)
        rels = []
        kd, kq = o_d.past_key_values, o_q.past_key_values
        for _ in range(4):
            s_d = dense(input_ids=nxt, past_key_values=kd, use_cache=True)
            s_q = quant(input_ids=nxt, past_key_values=kq, use_cache=True)
            rels.append(float((s_q.logits.float() - s_d.logits.float()).abs().max() / s_d.logits.float().abs().max()))
            kd, kq = s_d.past_key_values, s_q.past_key_values
            nxt = s_d.logits[:, -1:].argmax(-1)
        rec["decode_rel"] = rels
    # the decode steps bound their ops lazily: one row, four launches per layer (qkv | o + gate_up + down as one host call)
    rec["decode_ops_per_layer"] = []
    for l in quant.model.layers:
        st = getattr(l, "_gptqhip_fused", {{}}).get("state") if hasattr(l, "_gptqhip_fused") else None
        rec["decode_ops_per_layer"].append(None if st is None else sorted(st.ops.keys()))
    out[auto] = rec
    del quant
    torch.cuda.empty_cache()
print("RESULT " + json.dumps(out))
'''


@pytest.fixture(scope="module")
def result(tmp_path_factory):
    tree = tmp_path_factory.mktemp("gptqmodel_overlaid_autofuse")
    shutil.copytree(os.path.join(SRC, "gptqmodel"), os.path.join(tree, "gptqmodel"), ignore=shutil.ignore_patterns("__pycache__"))
    subprocess.run([sys.executable, os.path.join(ROOT, "integration", "apply_overlay.py"), str(tree)], check=True)
    code = _SCRIPT.format(root=ROOT, tree=str(tree))
    env = {k: v for k, v in os.environ.items() if k not in ("CUDA_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "GPTQHIP_DISABLE", "GPTQHIP_AUTO_FUSE")}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500, env=env, cwd="/tmp")
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("RESULT ")]
    assert line, res.stdout[-2000:] + res.stderr[-6000:]
    return json.loads(line[-1][7:])


def test_reference_post_init_lands_every_layer_on_the_decode_ops(result):
    r = result["1"]
    assert r["made"] == ["gptqmodel.nn_modules.qlinear.hip.HipGptqLinear"], r["made"]       # the reference's selector picked the overlay class
    assert r["fused_layers"] == 2 and not r["skipped"], r
    assert r["layer_has_fast_path"] == [True, True] and r["gate_proj_still_there"] == [False, False], r
    # the single-token steps bound the M = 1 decode ops of both layers (the 12-token prompt took the <= 16-row ops too)
    assert all(ops_m is not None and 1 in ops_m for ops_m in r["decode_ops_per_layer"]), r
    assert r["prefill_rel"] < 2e-2 and max(r["decode_rel"]) < 2e-2, r


def test_opt_out_keeps_the_plugin_modules(result):
    r = result["0"]
    assert r["fused_layers"] is None and r["layer_has_fast_path"] == [False, False] and r["gate_proj_still_there"] == [True, True], r
    assert r["prefill_rel"] < 2e-2 and max(r["decode_rel"]) < 2e-2, r
