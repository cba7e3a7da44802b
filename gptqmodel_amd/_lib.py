"""ctypes binding of libgptqhip.so (C ABI in include/gptqhip.h).

The library is built in-tree (gptqmodel_amd/csrc/libgptqhip.so) so that it travels with the source
snapshot.  There is NO fallback: if the shared object is missing or fails to load, `load()` raises --
the product path must fail loudly rather than silently run something else.

Precedent for a ctypes C-ABI kernel plugin inside the reference: `_GGMLBridge`
(gptqmodel/nn_modules/qlinear/gguf_cpp.py:88).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
# GPTQHIP_LIB: load another build of the library (dev A/B builds under tests/dev/ablate/; same ABI check as the product build)
LIB_PATH = os.environ.get("GPTQHIP_LIB") or os.path.join(CSRC_DIR, "libgptqhip.so")
ABI_VERSION = 19

# every symbol include/gptqhip.h declares: name -> (restype, argtypes)
_c = ctypes
_vp, _i, _sz = _c.c_void_p, _c.c_int, _c.c_size_t
SIGNATURES = {
    "gptqhip_abi_version": (_i, []),
    "gptqhip_last_error": (_c.c_char_p, []),
    "gptqhip_device_info": (_i, [_i, _c.POINTER(_i), _c.POINTER(_sz), _c.c_char_p, _i]),
    "gptqhip_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "gptqhip_tiled_words": (_sz, [_i, _i, _i]),
    "gptqhip_meta_words": (_sz, [_i, _i, _i]),
    "gptqhip_repack_tiled": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "gptqhip_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "gptqhip_decode_linear": (_i, [_vp, _vp]),
    "gptqhip_decode_supported": (_i, [_i, _i, _i, _i, _i]),
    "gptqhip_plan_describe": (_i, [_i, _i, _i, _i, _i, _i, _c.c_char_p, _i]),
    "gptqhip_decode_linear_seq": (_i, [_vp, _i, _vp]),
    "gptqhip_comm_bytes": (_sz, [_i, _i]),
    "gptqhip_comm_alloc": (_i, [_sz, _c.POINTER(_vp), _c.c_char_p]),
    "gptqhip_comm_open": (_i, [_c.c_char_p, _c.POINTER(_vp)]),
    "gptqhip_comm_close": (_i, [_vp]),
    "gptqhip_comm_free": (_i, [_vp]),
    "gptqhip_comm_status": (_i, [_vp, _c.POINTER(_c.c_uint32)]),
    "gptqhip_comm_set_timeout": (_i, [_vp, _c.c_uint]),
    "gptqhip_allreduce_oneshot": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "gptqhip_allgather_select": (_i, [_vp, _c.POINTER(_vp), _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "gptqhip_dequant": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "gptqhip_dequant_tiled": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "gptqhip_repack_awq": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "gptqhip_widen_codes": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gptqhip_embedding": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "gptqhip_pack_gptq": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "gptqhip_pack_gptq_host": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i]),
    "gptqhip_gather_cols": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "gptqhip_rmsnorm_gather": (_i, [_vp, _vp, _vp, _vp, _i, _i, _c.c_float, _i, _vp]),
    "gptqhip_set_tuning": (_i, [_i, _i, _i]),
    "gptqhip_set_decode_form": (_i, [_i]),
}

class DecodeOp(ctypes.Structure):
    """struct gptqhip_decode_op (include/gptqhip.h)."""
    _fields_ = [("qweight_t", _vp), ("meta", _vp), ("bias", _vp), ("x", _vp), ("norm_weight", _vp), ("residual", _vp),
                ("out", _vp), ("workspace", _vp), ("workspace_bytes", _sz), ("stats_in", _vp), ("stats_out", _vp),
                ("perm", _vp), ("eps", _c.c_float),
                ("K", _i), ("N", _i), ("group_size", _i), ("bits", _i), ("act_dtype", _i), ("scale_dtype", _i),
                ("in_glue", _i), ("out_glue", _i), ("stats_n", _i), ("flags", _i), ("M", _i)]


_lock = threading.Lock()
_lib = None


def build(verbose: bool = False) -> str:
    """Compile libgptqhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError(f"building libgptqhip.so failed (exit {res.returncode}):\n{res.stderr[-4000:]}")
    return LIB_PATH


def load() -> ctypes.CDLL:
    """Load the library once per process (a lock, like the reference's loader lock gptqmodel/extension.py:121)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libgptqhip.so not found at {LIB_PATH}; run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C gptqmodel_amd/csrc`.  There is no CPU/PyTorch fallback for the HIP backend."
            )
        # Import torch FIRST: libgptqhip.so needs libamdhip64.so.7 and must bind to the HIP runtime torch already
        # loaded (its bundled copy has that SONAME).  Loading ours first would map /opt/rocm's runtime as a second
        # copy next to torch's and the two would not share devices/streams.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        got = lib.gptqhip_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libgptqhip.so ABI version {got} != expected {ABI_VERSION}; rebuild it")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().gptqhip_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
