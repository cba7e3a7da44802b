"""Shared helpers for the parity tests: golden-fixture decoding and oracle <-> torch conversions."""
import numpy as np
import torch

from oracle import gptq_oracle as O

TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


def bits_to_torch(bits: np.ndarray, dtype: str, device="cpu") -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).copy()).view(TDT[dtype])
    return t.to(device)


def torch_to_bits(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits_to_f32(bits: np.ndarray, dtype: str) -> np.ndarray:
    if dtype == "fp16":
        return bits.view(np.float16).astype(np.float32)
    return O.bf16_from_bits(bits)


def torch_to_f32(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """max |a-b| / max |b|  -- the 'relative error of the output vector' gate (SURVEY.md §7 hard part i)."""
    return float(np.abs(a - b).max() / max(float(np.abs(b).max()), 1e-12))


def synth_gptq(seed, bits, k, n, gs, desc_act=False, sym=False, scale_dtype="fp16"):
    """Seeded synthetic GPTQ-v2 tensors following BASELINE.md §2 / SURVEY.md §8d."""
    rng = np.random.RandomState(seed)
    pf = 32 // bits
    g = k // gs
    qweight = rng.randint(-2**31, 2**31, size=(k // pf, n), dtype=np.int64).astype(np.int32)
    if sym:
        word = sum(((1 << (bits - 1)) << (bits * j)) for j in range(pf))
        qzeros = np.full((g, n // pf), word, dtype=np.uint32).view(np.int32)
    else:
        qzeros = rng.randint(-2**31, 2**31, size=(g, n // pf), dtype=np.int64).astype(np.int32)
    scales = O.round_to(rng.rand(g, n).astype(np.float32) * 0.01 + 0.005, scale_dtype)
    g_idx = ((rng.permutation(k) if desc_act else np.arange(k)) // gs).astype(np.int32)
    return qweight, qzeros, scales, g_idx


def f32_to_torch(a: np.ndarray, dtype: str, device="cpu") -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TDT[dtype]).to(device)
