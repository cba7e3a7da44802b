"""The drop-in boundary against the REAL reference: copy /root/reference/gptqmodel to a temp tree, apply
integration/gptqmodel_overlay (backend.patch + nn_modules/qlinear/hip.py), import it through oracle/ref_import.py and
check that the reference's UNMODIFIED discovery / selection code treats the HIP classes as first-class kernels.

Needs the reference mounted (build container only): marked `reference`, skipped elsewhere.  No GPU: validate_once() is
monkeypatched the same way the reference's own selection tests patch device probes."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gptqmodel")), reason="reference not mounted")]

_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, {root!r})
os.environ["GPTQ_REFERENCE_ROOT"] = {tree!r}
os.environ["CUDA_VISIBLE_DEVICES"] = ""
for k, v in {env!r}.items():
    os.environ[k] = v
from oracle.ref_import import load_reference
ref = load_reference()
import torch
from gptqmodel.nn_modules.qlinear import hip as hipmod            # the overlay file, inside the reference package
from gptqmodel.nn_modules.qlinear import BaseQuantLinear, GPTQQuantLinear, AWQuantLinear
from gptqmodel.nn_modules.qlinear.torch import TorchLinear
from gptqmodel.nn_modules.qlinear.torch_awq import AwqTorchLinear
from gptqmodel.utils import importer
from gptqmodel.utils.backend import BACKEND, normalize_backend
from gptqmodel.quantization import FORMAT, METHOD
from gptqmodel.models._const import DEVICE
out = {{}}
H, A = hipmod.HipGptqLinear, hipmod.HipAwqLinear
out["bases"] = [issubclass(H, GPTQQuantLinear), issubclass(A, AWQuantLinear), issubclass(H, BaseQuantLinear)]
out["module"] = [H.__module__, A.__module__]
if {fake_device!r}:   # pretend the .so + a gfx950 device are usable (CPU container): only the device probe is replaced
    for c in (H, A):
        c.validate_once = classmethod(lambda cls: (True, None))
        c.cached_validate_once.cache_clear()
H.verify_supports_params(); A.verify_supports_params(); hipmod.HipQuantEmbeddings.verify_supports_params()
out["verify"] = True
kernels = importer.iter_quant_linear_kernels()
out["discovered"] = [H in kernels, A in kernels, hipmod.HipQuantEmbeddings in kernels]
auto = importer.AUTO_BACKEND_KERNEL_MAPPING
out["auto_first_gptq"] = next(iter(auto[METHOD.GPTQ][FORMAT.GPTQ].values())).__name__ if False else None
sel = lambda **kw: importer.select_quant_linear(bits=4, group_size=128, desc_act=False, sym=True, pack_dtype=torch.int32, **kw)
def name(f):
    try:
        r = f()
        return r.__name__ if isinstance(r, type) else [c.__name__ for c in r]
    except Exception as e:
        return "ERR:" + type(e).__name__ + ":" + str(e)[:120]
out["auto_gptq_rocm"] = name(lambda: sel(device=DEVICE.ROCM, backend=BACKEND.AUTO, format=FORMAT.GPTQ, quant_method=METHOD.GPTQ))
out["auto_gptq_v2_rocm"] = name(lambda: sel(device=DEVICE.ROCM, backend=BACKEND.AUTO, format=FORMAT.GPTQ_V2, quant_method=METHOD.GPTQ))
out["auto_awq_rocm"] = name(lambda: importer.select_quant_linear(bits=4, group_size=128, desc_act=False, sym=False, pack_dtype=torch.int32,
                            device=DEVICE.ROCM, backend=BACKEND.AUTO, format=FORMAT.GEMM, quant_method=METHOD.AWQ))
out["auto_gptq_cpu"] = name(lambda: sel(device=DEVICE.CPU, backend=BACKEND.AUTO, format=FORMAT.GPTQ, quant_method=METHOD.GPTQ))
out["multi_gptq_rocm"] = name(lambda: sel(device=DEVICE.ROCM, backend=BACKEND.AUTO, format=FORMAT.GPTQ, quant_method=METHOD.GPTQ, multi_select=True))
out["explicit_gptq_hip"] = name(lambda: sel(device=DEVICE.ROCM, backend=BACKEND.GPTQ_HIP, format=FORMAT.GPTQ, quant_method=METHOD.GPTQ))
out["explicit_awq_hip"] = name(lambda: importer.select_quant_linear(bits=4, group_size=128, desc_act=False, sym=False, pack_dtype=torch.int32,
                               device=DEVICE.ROCM, backend=BACKEND.AWQ_HIP, format=FORMAT.GEMM, quant_method=METHOD.AWQ))
out["kernel_for_backend"] = [importer.get_kernel_for_backend(BACKEND.GPTQ_HIP, METHOD.GPTQ, FORMAT.GPTQ).__name__,
                             importer.get_kernel_for_backend(BACKEND.AWQ_HIP, METHOD.AWQ, FORMAT.GEMM).__name__]
out["alias"] = [normalize_backend("hip", quant_method=METHOD.GPTQ).value, normalize_backend("hip", quant_method=METHOD.AWQ).value,
                normalize_backend("gptq_hip").value]
out["hf_select"] = name(lambda: importer.hf_select_quant_linear_v2(bits=4, group_size=128, desc_act=False, sym=True, format="gptq",
                        quant_method="gptq", device_map={{"": "cuda:0"}}, backend="auto", pack=False))
# group_size 16 is outside the HIP class' contract: AUTO must fall through to an upstream kernel, explicit selection must refuse
out["auto_g16"] = name(lambda: importer.select_quant_linear(bits=4, group_size=16, desc_act=False, sym=True, pack_dtype=torch.int32,
                       device=DEVICE.ROCM, backend=BACKEND.AUTO, format=FORMAT.GPTQ, quant_method=METHOD.GPTQ))
out["explicit_g16"] = name(lambda: importer.select_quant_linear(bits=4, group_size=16, desc_act=False, sym=True, pack_dtype=torch.int32,
                           device=DEVICE.ROCM, backend=BACKEND.GPTQ_HIP, format=FORMAT.GPTQ, quant_method=METHOD.GPTQ))
# the other bit widths of the reference's torch kernel are inside it (widened to the 4- / 8-bit kernel layout at post_init)
out["auto_bits"] = [name(lambda b=b: importer.select_quant_linear(bits=b, group_size=128, desc_act=False, sym=True, pack_dtype=torch.int32,
                    device=DEVICE.ROCM, backend=BACKEND.AUTO, format=FORMAT.GPTQ, quant_method=METHOD.GPTQ)) for b in (2, 3, 5, 6, 7)]
# constructing the class runs the REAL base-class __init__ + validate chain and registers the checkpoint buffers
m = H(bits=4, group_size=128, sym=True, desc_act=False, in_features=256, out_features=64, bias=True, register_buffers=True,
      format=FORMAT.GPTQ) if {fake_device!r} else None
if m is not None:
    out["auto_gptq_p"] = [name(lambda b=b: importer.select_quant_linear(bits=b, group_size=128, desc_act=False, sym=True, pack_dtype=torch.int32,
                           device=DEVICE.ROCM, backend=BACKEND.AUTO, format=FORMAT.GPTQ_P, quant_method=METHOD.GPTQ)) for b in (3, 5)]
    m3 = H(bits=3, group_size=128, sym=True, desc_act=False, in_features=256, out_features=64, bias=False, register_buffers=True,
           pack_dtype=torch.int32, backend=BACKEND.GPTQ_HIP)
    m6 = H(bits=6, group_size=128, sym=True, desc_act=False, in_features=256, out_features=64, bias=False, register_buffers=True,
           pack_dtype=torch.int32, backend=BACKEND.GPTQ_HIP)
    out["other_bits"] = [list(m3.qweight.shape), list(m3.qzeros.shape), m3.kernel_bits, bool(m3.planar),
                         list(m6.qweight.shape), list(m6.qzeros.shape), m6.kernel_bits, bool(m6.planar)]
    out["buffers"] = sorted(n for n, _ in m.named_buffers())
    out["shapes"] = [list(m.qweight.shape), list(m.qzeros.shape), list(m.scales.shape), list(m.g_idx.shape)]
    out["requires_v2"] = [bool(m.REQUIRES_FORMAT_V2), m.qzero_format()]
    try:
        m.eval(); m.train(True); out["train"] = "no error"
    except NotImplementedError:
        out["train"] = "NotImplementedError"
    a = A(bits=4, group_size=128, sym=False, desc_act=False, in_features=256, out_features=64, bias=False, register_buffers=True)
    out["awq_shapes"] = [list(a.qweight.shape), list(a.qzeros.shape), list(a.scales.shape)]
print("RESULT " + json.dumps(out))
'''


@pytest.fixture(scope="module")
def overlaid_tree(tmp_path_factory):
    tree = tmp_path_factory.mktemp("gptqmodel_overlaid")
    shutil.copytree(os.path.join(REF, "gptqmodel"), os.path.join(tree, "gptqmodel"),
                    ignore=shutil.ignore_patterns("__pycache__"))
    shutil.copytree(os.path.join(REF, "tests"), os.path.join(tree, "tests"), ignore=shutil.ignore_patterns("__pycache__", "models"))
    assert subprocess.run([sys.executable, os.path.join(ROOT, "integration", "apply_overlay.py"), str(tree), "--check"]).returncode == 1
    subprocess.run([sys.executable, os.path.join(ROOT, "integration", "apply_overlay.py"), str(tree)], check=True)
    assert subprocess.run([sys.executable, os.path.join(ROOT, "integration", "apply_overlay.py"), str(tree), "--check"]).returncode == 0
    return str(tree)


def _run(tree, fake_device=True, env=None):
    import json
    code = _SCRIPT.format(root=ROOT, tree=tree, fake_device=fake_device, env=env or {})
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    line = [l for l in res.stdout.splitlines() if l.startswith("RESULT ")]
    assert line, res.stdout[-2000:] + res.stderr[-4000:]
    return json.loads(line[-1][7:])


def test_overlay_classes_are_first_class_reference_kernels(overlaid_tree):
    r = _run(overlaid_tree)
    assert r["bases"] == [True, True, True]
    assert r["module"] == ["gptqmodel.nn_modules.qlinear.hip"] * 2
    assert r["verify"] is True
    assert r["discovered"] == [True, True, False]      # embeddings opt out of backend discovery
    # BACKEND.AUTO on ROCm: the HIP classes win (priority 120); elsewhere nothing changes
    assert r["auto_gptq_rocm"] == "HipGptqLinear" and r["auto_gptq_v2_rocm"] == "HipGptqLinear"
    assert r["auto_awq_rocm"] == "HipAwqLinear"
    assert r["auto_gptq_cpu"] not in ("HipGptqLinear",) and not r["auto_gptq_cpu"].startswith("ERR")
    assert r["multi_gptq_rocm"][0] == "HipGptqLinear" and "TorchLinear" in r["multi_gptq_rocm"]
    # explicit backends resolve to exactly one class
    assert r["explicit_gptq_hip"] == "HipGptqLinear" and r["explicit_awq_hip"] == "HipAwqLinear"
    assert r["kernel_for_backend"] == ["HipGptqLinear", "HipAwqLinear"]
    assert r["alias"] == ["gptq_hip", "awq_hip", "gptq_hip"]
    assert r["hf_select"] == "HipGptqLinear"
    # outside the contract: AUTO falls through, explicit refuses with the reference's ValueError
    assert r["auto_g16"] not in ("HipGptqLinear",) and not r["auto_g16"].startswith("ERR")
    assert r["explicit_g16"].startswith("ERR:ValueError")
    assert r["auto_bits"] == ["HipGptqLinear"] * 5 and r["auto_gptq_p"] == ["HipGptqLinear"] * 2
    assert r["other_bits"] == [[24, 64], [2, 6], 4, False, [48, 64], [2, 12], 8, True]
    # the REAL base class registered the checkpoint contract
    assert r["buffers"] == ["bias", "g_idx", "qweight", "qzeros", "scales"]
    assert r["shapes"] == [[32, 64], [2, 8], [2, 64], [256]]
    assert r["requires_v2"] == [True, 1] and r["train"] == "NotImplementedError"
    assert r["awq_shapes"] == [[256, 8], [2, 8], [2, 64]]


def test_overlay_soft_fails_to_upstream_kernels(overlaid_tree):
    """GPTQHIP_DISABLE=1 (or, as in this container, no usable device): validate_once() reports the kernels unavailable
    and BACKEND.AUTO falls through to the reference's own TorchLinear / AwqTorchLinear; explicit selection raises."""
    for env in ({"GPTQHIP_DISABLE": "1"}, {}):
        r = _run(overlaid_tree, fake_device=False, env=env)
        assert r["discovered"] == [True, True, False]
        assert r["auto_gptq_rocm"] == "TorchLinear", r
        assert r["auto_awq_rocm"] == "AwqTorchLinear", r
        assert r["explicit_gptq_hip"].startswith("ERR:ValueError")


def test_reference_selection_tests_still_pass_with_overlay(overlaid_tree):
    """The reference's own backend-naming / hierarchy tests run unchanged against the overlaid tree."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['GPTQ_REFERENCE_ROOT'] = %r; os.environ['CUDA_VISIBLE_DEVICES'] = '';"
            "from oracle.ref_import import load_reference; load_reference(); import pytest;"
            "sys.exit(pytest.main(['-q', '-x', '-p', 'no:cacheprovider', %r, %r]))") % (
        ROOT, overlaid_tree, os.path.join(overlaid_tree, "tests", "test_backend_naming.py"),
        os.path.join(overlaid_tree, "tests", "kernels", "test_qlinear_hierarchy.py"))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=overlaid_tree)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]


def test_reference_quantize_config_parses_what_the_repo_writes(tmp_path):
    """The quantize_config.json written by utils.checkpoint.save_quantized_checkpoint (and the legacy-alias forms its reader
    accepts) parsed by the REAL reference's QuantizeConfig.from_pretrained (quantization/config.py:3022): the keys that reach the
    kernel constructor must agree field by field with utils.checkpoint.read_quantize_config."""
    import json
    script = r'''
import json, os, sys
sys.path.insert(0, {root!r})
os.environ["CUDA_VISIBLE_DEVICES"] = ""
from oracle.ref_import import load_reference
load_reference()
from gptqmodel.quantization.config import QuantizeConfig
from gptqmodel_amd.utils.checkpoint import read_quantize_config
out = []
for d in {dirs!r}:
    q = QuantizeConfig.from_pretrained(d)
    ours = read_quantize_config(d)
    fmt = getattr(q.format, "value", q.format)
    meth = getattr(q.method, "value", q.method) if hasattr(q, "method") else getattr(q.quant_method, "value", q.quant_method)
    out.append({{"ref": [int(q.bits), int(q.group_size), bool(q.desc_act), bool(q.sym), str(fmt).lower(), str(meth).lower()],
                 "ours": [ours["bits"], ours["group_size"], ours["desc_act"], ours["sym"], ours["format"], ours["method"]]}})
print("RESULT " + json.dumps(out))
'''
    payloads = [
        {"bits": 4, "group_size": 128, "desc_act": True, "sym": False, "lm_head": False, "quant_method": "gptq", "checkpoint_format": "gptq",
         "pack_dtype": "int32", "meta": {"quantizer": ["gptqmodel:5.0.0"]}},                                 # what the writer emits
        {"bits": 8, "group_size": 32, "desc_act": False, "sym": True, "quant_method": "gptq", "checkpoint_format": "gptq_v2"},
        {"w_bit": 4, "q_group_size": 64, "zero_point": True, "version": "gemm", "quant_method": "awq"},       # AutoAWQ-style keys
    ]
    dirs = []
    for i, pl in enumerate(payloads):
        d = tmp_path / f"c{i}"
        d.mkdir()
        (d / "quantize_config.json").write_text(json.dumps(pl))
        dirs.append(str(d))
    r = subprocess.run([sys.executable, "-c", script.format(root=ROOT, dirs=dirs)], capture_output=True, text=True, timeout=600, cwd="/tmp")
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")), None)
    assert line is not None, (r.stdout[-800:], r.stderr[-1500:])
    for res in json.loads(line[7:]):
        assert res["ref"] == res["ours"], res
