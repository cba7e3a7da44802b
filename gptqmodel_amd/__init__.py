"""gptqmodel_amd -- MI355X (gfx950) native GPTQ/AWQ dequant-matmul QuantLinear backend.

Only the hot path of ModelCloud/GPTQModel is implemented here: hand-written HIP kernels
behind a C ABI (include/gptqhip.h), and the host-side mirror of the reference's QuantLinear / BACKEND
plugin interface so the classes drop into `GPTQModel.load()`.
"""
__version__ = "0.1.0"
