"""Minimal adapter contract (reference gptqmodel/adapter/adapter.py:100-173): the kernel keeps the hook
`if self.adapter: out = self.adapter.apply(x=x, out=out)` (torch.py:344-345) and leaves the rank-r update to
torch -- it is outside the quantised hot path."""
from __future__ import annotations

import torch


class Adapter:
    def __init__(self, rank: int = None, path: str = None):
        self.rank = rank
        self.path = path

    def apply(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def post_init(self, weight_key: str, device: torch.device, **kwargs):
        pass

    def optimize(self, *a, **k):
        pass


class Lora(Adapter):
    """out += (x @ A) @ B   (adapter.py:148-173)"""

    def __init__(self, rank: int, path: str = None, lora_A: torch.Tensor = None, lora_B: torch.Tensor = None):
        super().__init__(rank, path)
        self.lora_A = lora_A
        self.lora_B = lora_B

    def apply(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        if x.dtype != self.lora_A.dtype or x.device != self.lora_A.device:
            self.lora_A = self.lora_A.to(device=x.device, dtype=x.dtype)
            self.lora_B = self.lora_B.to(device=x.device, dtype=x.dtype)
        if out.dim() > x.dim() and out.shape[0] > 1:
            shape = out.shape
            out = out.view(-1, out.shape[-1])
            out.add_((x @ self.lora_A) @ self.lora_B)
            return out.view(shape)
        return out.add_((x @ self.lora_A) @ self.lora_B)

    def post_init(self, weight_key: str, device: torch.device, lora_A=None, lora_B=None):
        if lora_A is not None and lora_B is not None:
            self.lora_A, self.lora_B = lora_A.to(device), lora_B.to(device)
        elif self.lora_A is not None:
            self.lora_A, self.lora_B = self.lora_A.to(device), self.lora_B.to(device)
