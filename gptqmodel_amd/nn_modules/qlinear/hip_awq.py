"""HipAwqLinear -- BACKEND.AWQ_HIP on this package's mirror of the reference plugin contract (implementation shared with
the upstream-tree overlay: hip_impl.make_hip_classes; created together with HipGptqLinear so each class exists once)."""
from .hip_gptq import _HipAwqLinear as HipAwqLinear

__all__ = ["HipAwqLinear"]
