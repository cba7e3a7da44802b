"""Import shim for the REAL reference (ModelCloud/GPTQModel) mounted read-only at /root/reference.

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py (to generate tests/golden/*) and by the
`-m "not gpu"` tests that cross-check the oracle against the live reference when it is mounted.
Nothing under gptqmodel_amd/ imports this.  /root/reference does not exist on the GPU box: there the shim falls back to
`oracle/_ref/` (the files this import executes, copied by oracle/make_ref_snapshot.py; git-ignored, travels with a push),
which bench.py's cpu_baseline leg uses to time the reference's own modules.

Recipe: the reference's `gptqmodel/__init__.py` eagerly imports the model zoo, which
needs packages absent here (tokenicer, defuser, torchvision ...).  We pre-seed `sys.modules` with
shell packages for `gptqmodel` and `gptqmodel.models` so only the leaf modules we need execute, and put
three tiny stubs (pcre, logbar, device_smi) on sys.path.
"""
import os
import sys
import types

_SNAPSHOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")   # oracle/make_ref_snapshot.py (git-ignored)


def _pick_root() -> str:
    env = os.environ.get("GPTQ_REFERENCE_ROOT")
    if env:
        return env
    if os.path.isdir("/root/reference/gptqmodel"):
        return "/root/reference"
    return _SNAPSHOT   # the GPU box: only the files the shim executes, for timing the reference's CPU path


REF_ROOT = _pick_root()
REF_PKG = os.path.join(REF_ROOT, "gptqmodel")
REF_IS_SNAPSHOT = os.path.abspath(REF_ROOT) == os.path.abspath(_SNAPSHOT)
_SHIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shim")


def reference_available() -> bool:
    return os.path.isdir(REF_PKG)


def _shell(name: str, path: str) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


_loaded = None


def load_reference():
    """Returns a namespace with the reference's TorchLinear / AwqTorchLinear / selector symbols."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REF_ROOT}")
    if "gptqmodel" in sys.modules and not getattr(sys.modules["gptqmodel"], "_refshim", False):
        raise RuntimeError("a real `gptqmodel` is already imported; refusing to shadow it")
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
    if _SHIM_DIR not in sys.path:
        sys.path.insert(0, _SHIM_DIR)
    root = _shell("gptqmodel", REF_PKG)
    root.DEBUG_ON = False
    root._refshim = True
    _shell("gptqmodel.models", os.path.join(REF_PKG, "models"))

    from gptqmodel.nn_modules.qlinear.torch import TorchLinear
    from gptqmodel.nn_modules.qlinear.torch_awq import AwqTorchLinear
    from gptqmodel.quantization import FORMAT, METHOD
    from gptqmodel.quantization.awq.utils.packing_utils import dequantize_gemm
    from gptqmodel.utils.backend import BACKEND
    from gptqmodel.utils.importer import select_quant_linear
    from gptqmodel.utils.model import convert_gptq_v1_to_v2_format_module
    from gptqmodel.models._const import DEVICE

    ns = types.SimpleNamespace(
        TorchLinear=TorchLinear,
        AwqTorchLinear=AwqTorchLinear,
        FORMAT=FORMAT,
        METHOD=METHOD,
        BACKEND=BACKEND,
        DEVICE=DEVICE,
        select_quant_linear=select_quant_linear,
        dequantize_gemm=dequantize_gemm,
        convert_gptq_v1_to_v2_format_module=convert_gptq_v1_to_v2_format_module,
    )
    _loaded = ns
    return ns
