# Drop-in file for the UPSTREAM tree: copy to gptqmodel/nn_modules/qlinear/hip.py (integration/apply_overlay.py does it,
# together with utils/backend.patch).  The reference discovers QuantLinear kernels by importing every module of this
# package and walking BaseQuantLinear.__subclasses__() (gptqmodel/utils/importer.py:110-127,169-179), so nothing else has
# to be registered: BACKEND.AUTO on DEVICE.ROCM then resolves to HipGptqLinear / HipAwqLinear (priority 120), explicit
# BACKEND.GPTQ_HIP / BACKEND.AWQ_HIP resolve to exactly one class each, and GPTQModel.load() is a drop-in.
#
# The classes are built on the reference's OWN GPTQQuantLinear / AWQuantLinear (validate chain, buffer registration,
# adapter hook, rotation hook, qzero_format, train() guard all come from upstream); the MI355X-specific part --
# post_init() relayout, forward() through libgptqhip.so, dequantize_weight(), pack() -- is the same code object the
# gptqmodel_amd package uses for its own classes (gptqmodel_amd/nn_modules/qlinear/hip_impl.py).
# If gptqmodel_amd is not installed the import fails with ImportError, which the discovery loop skips by design.
from types import SimpleNamespace

from gptqmodel_amd.nn_modules.qlinear.hip_impl import make_hip_classes

from ...adapter.adapter import Lora
from ...models._const import DEVICE, PLATFORM
from ...quantization import FORMAT, METHOD
from ...utils.backend import BACKEND
from . import AWQuantLinear, GPTQQuantLinear


if not hasattr(BACKEND, "GPTQ_HIP") or not hasattr(BACKEND, "AWQ_HIP"):
    raise ImportError("gptqmodel.utils.backend.BACKEND has no GPTQ_HIP / AWQ_HIP members: apply "
                      "integration/gptqmodel_overlay/utils/backend.patch")

HipGptqLinear, HipQuantEmbeddings, HipAwqLinear = make_hip_classes(
    SimpleNamespace(GPTQQuantLinear=GPTQQuantLinear, AWQuantLinear=AWQuantLinear, BACKEND=BACKEND, DEVICE=DEVICE,
                    FORMAT=FORMAT, METHOD=METHOD, PLATFORM=PLATFORM, Lora=Lora),
    __name__,
)

__all__ = ["HipGptqLinear", "HipQuantEmbeddings", "HipAwqLinear"]
