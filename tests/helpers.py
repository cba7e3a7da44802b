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


# element-wise tolerances of the reference's own forward test (tests/test_torch_kernel_accuracy.py:111-125):
#   torch.allclose(out, ref, atol=5e-3 if fp16 else 3e-2, rtol=1e-2)
REF_ATOL = {"fp16": 5e-3, "bf16": 3e-2}
REF_RTOL = 1e-2
# norm-wise bar: north_star's <= 1e-3 relative fp16 error; bf16: 8e-3 = one bf16 ulp (2^-7) of the largest output, what a kernel that repeats the
# reference's rounding chain meets.  (The reference's own kernel table, tests/kernels/test_gptq.py:229-266, 353-360, uses 0.008 differently: as the
# atol of an ELEMENT-WISE torch.isclose(ref, out, rtol=0.15, atol=0.008) -- REF_KERNEL_* below; its torch-kernel accuracy test compares against an
# exact-arithmetic fp32 product with allclose(atol=3e-2, rtol=1e-2), tests/test_torch_kernel_accuracy.py:111-125 -- REF_ATOL / REF_RTOL.)
NORM_TOL = {"fp16": 1e-3, "bf16": 8e-3}
# exact-arithmetic decode form on bf16 activations (decode form 5): the output is the correctly rounded exact sum, the reference's is the rounded sum of
# per-weight-rounded products -- they differ by the reference's own rounding noise, i.e. up to one bf16 ulp per rounding step (two with a bias / residual)
NORM_TOL_BF16_EXACT = 2.0 ** -6
REF_KERNEL_RTOL, REF_KERNEL_ATOL = 0.15, {"fp16": 0.004, "bf16": 0.008}     # test_gptq.py: isclose(rtol=0.15, atol=<per backend: Marlin / ExllamaV2 values>)


def decode_norm_tol(act: str, bits: int = 4):
    """Norm-wise bar of a batch-1 call (gptqhip_gemm at M = 1, gptqhip_decode_linear at M = 1) under the process-default decode form: 4-bit codes
    take the exact-arithmetic form 5 (fp16: 1e-3 as everywhere; bf16 activations: one bf16 ulp per rounding step, NORM_TOL_BF16_EXACT);
    GPTQHIP_DECODE_BITFAITHFUL=1 and 8-bit codes keep the reference's rounding chain and the plain bars."""
    import os
    if act == "bf16" and bits == 4 and os.environ.get("GPTQHIP_DECODE_BITFAITHFUL", "0") in ("", "0"):
        return NORM_TOL_BF16_EXACT
    return NORM_TOL[act]


def assert_forward_close(got: np.ndarray, ref: np.ndarray, act: str, tag=None, strict_atol: bool = False, norm_tol: float = None):
    """Both gates on a rounded forward output: (1) the norm-wise relative error of the output (north_star), and
    (2) the reference's own ELEMENT-WISE assertion with its atol/rtol.

    The reference states its atol (5e-3 fp16 / 3e-2 bf16) on ITS test data -- x = randn * 0.5, K = 64, outputs of magnitude <~ 1, compared
    with an exact-arithmetic fp32 product (tests/test_torch_kernel_accuracy.py:111-125).  On a vector whose largest output is S the same
    gate is atol * max(1, S / 5): an absolute 5e-3 on outputs of several hundred (the stress-scale cases of test_gpu_decode_chain.py) is
    2e-5 of the output scale, which only a kernel that repeats the reference's per-weight rounding bit for bit can meet on the elements
    that cancel to ~0 -- it measures the reference's OWN rounding noise, not the kernel's error.  strict_atol=True keeps the unscaled atol:
    used for the bit-faithful decode forms (gptqhip_set_decode_form 0 / 4) and everywhere the outputs stay at the reference's scale."""
    got = np.asarray(got, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float32)
    assert got.shape == ref.shape, (got.shape, ref.shape, tag)
    e = rel_err(got, ref)
    nt = NORM_TOL[act] if norm_tol is None else norm_tol
    assert e <= nt, (f"norm-wise rel err {e:.3e} > {nt}", tag)
    # the reference's kernel-table gate (element-wise isclose with rtol 0.15), at the output's scale like the gate below
    katol = REF_KERNEL_ATOL[act] * max(1.0, float(np.abs(ref).max()) / 5.0)
    assert not (np.abs(got - ref) > katol + REF_KERNEL_RTOL * np.abs(ref)).any(), ("outside the reference's kernel-table isclose gate", tag)
    atol = REF_ATOL[act] * (1.0 if strict_atol else max(1.0, float(np.abs(ref).max()) / 5.0))
    bad = np.abs(got - ref) > atol + REF_RTOL * np.abs(ref)
    assert not bad.any(), (f"{int(bad.sum())} of {bad.size} elements outside atol={atol:.3g} rtol={REF_RTOL}; "
                           f"worst |d|={float(np.abs(got - ref).max()):.3e}", tag)


def synth_gptq(seed, bits, k, n, gs, desc_act=False, sym=False, scale_dtype="fp16"):
    """Seeded synthetic GPTQ-v2 tensors following BASELINE.md §2 / SURVEY.md §8d."""
    rng = np.random.RandomState(seed)
    g = k // gs
    # (any word pattern is a valid encoding at every bit width / layout: every bit belongs to some code)
    qweight = rng.randint(-2**31, 2**31, size=(k * bits // 32, n), dtype=np.int64).astype(np.int32)
    if sym:
        qzeros = O.pack_cols_any(np.full((g, n), 1 << (bits - 1), dtype=np.uint8), bits)
    else:
        qzeros = rng.randint(-2**31, 2**31, size=(g, n * bits // 32), dtype=np.int64).astype(np.int32)
    scales = O.round_to(rng.rand(g, n).astype(np.float32) * 0.01 + 0.005, scale_dtype)
    g_idx = ((rng.permutation(k) if desc_act else np.arange(k)) // gs).astype(np.int32)
    return qweight, qzeros, scales, g_idx


# ----------------------------------------------------------------------------------------------------------
# FULL-SIZE reference fixtures (tests/golden/full_*.npz, written by oracle/make_golden.py:make_fullsize): the packed
# tensors of a BASELINE-sized layer are far too big to commit, so a fixture stores the SEED + the reference's outputs
# (+ a few dequantised rows + a checksum of the regenerated inputs); both the generator and the tests rebuild the
# inputs with this one function (numpy legacy RandomState: the stream is frozen across numpy versions).
# ----------------------------------------------------------------------------------------------------------
def synth_full_case(kind, seed, bits, k, n, gs, desc_act, sym, scale_dtype, act, m):
    """-> dict(qweight, qzeros, scales (f32, exactly representable in scale_dtype), g_idx | None, x (f32, in act)).
    kind "gptq": checkpoint-layout v2 tensors; kind "awq": AWQ GEMM-layout tensors (no g_idx, asymmetric zeros)."""
    rng = np.random.RandomState(seed)
    g = k // gs
    if kind == "gptq":
        qweight, qzeros, scales, g_idx = synth_gptq(seed, bits, k, n, gs, desc_act=desc_act, sym=sym,
                                                    scale_dtype=scale_dtype)
        rng = np.random.RandomState(seed + 7919)
    else:
        qweight = rng.randint(-2**31, 2**31, size=(k, n // 8), dtype=np.int64).astype(np.int32)
        qzeros = rng.randint(-2**31, 2**31, size=(g, n // 8), dtype=np.int64).astype(np.int32)
        scales = O.round_to(rng.rand(g, n).astype(np.float32) * 0.01 + 0.005, scale_dtype)
        g_idx = None
    x = O.round_to(rng.randn(m, k).astype(np.float32) * 0.5, act)
    return {"qweight": qweight, "qzeros": qzeros, "scales": scales, "g_idx": g_idx, "x": x}


def inputs_checksum(case) -> np.ndarray:
    """Cheap order-sensitive checksum of the regenerated inputs (uint64[4]): detects any RNG / rounding drift."""
    out = []
    for key in ("qweight", "qzeros", "scales", "x"):
        a = np.ascontiguousarray(case[key])
        u = a.view(np.uint32).astype(np.uint64).reshape(-1)
        w = (np.arange(u.size, dtype=np.uint64) % np.uint64(65521)) + np.uint64(1)
        out.append(np.uint64((u * w).sum()))
    return np.array(out, dtype=np.uint64)


def f32_to_torch(a: np.ndarray, dtype: str, device="cpu") -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TDT[dtype]).to(device)


# ----------------------------------------------------------------------------------------------------------
# numpy model of the MFMA-tile-major kernel layout (gptqmodel_amd/csrc/gptqhip_device.h) -- the checker for the
# device repack kernel (integer work: compared bit-exactly)
# ----------------------------------------------------------------------------------------------------------
def np_repack_tiled(qweight, qzeros, scales_bits, perm, group_size, bits):
    pf = 32 // bits
    codes = O.unpack_rows(qweight, bits).astype(np.uint32)          # [K, N]
    zeros = O.unpack_cols(qzeros, bits).astype(np.uint32)           # [G, N]
    K, N = codes.shape
    G = K // group_size
    if perm is not None:
        codes = codes[np.asarray(perm).astype(np.int64)]
    chunks, tiles = -(-K // 128), -(-N // 16)
    Kp, Np = chunks * 128, tiles * 16
    full = np.zeros((Kp, Np), dtype=np.uint32)
    full[:K, :N] = codes
    if Kp > K:
        full[K:, :N] = zeros[G - 1][None, :]                          # padded rows dequantise to exactly 0
    steps = 4
    if bits == 4:
        # word[tile, chunk, lane=(rq,c), j]: rows 128*chunk + 32*j + 8*rq + e, e at bit 4*(e>>1) + 16*(e&1)
        v = full.reshape(chunks, steps, 4, 8, tiles, 16)              # [chunk, j, rq, e, tile, c]
        sh = np.array([4 * (e >> 1) + 16 * (e & 1) for e in range(8)], dtype=np.uint32)
        words = np.bitwise_or.reduce(v << sh[None, None, None, :, None, None], axis=3)   # [chunk, j, rq, tile, c]
        words = words.transpose(3, 0, 2, 4, 1)                         # [tile, chunk, rq, c, j]
        qw_t = np.ascontiguousarray(words).reshape(-1)
    else:
        # word[tile, chunk, h, lane, jj]: j = 2h + (jj>>1), half = jj&1, rows ... + 4*half + e, e at 8*(e>>1)+16*(e&1)
        v = full.reshape(chunks, 2, 2, 4, 2, 4, tiles, 16)            # [chunk, h, jlo, rq, half, e, tile, c]
        sh = np.array([8 * (e >> 1) + 16 * (e & 1) for e in range(4)], dtype=np.uint32)
        words = np.bitwise_or.reduce(v << sh[None, None, None, None, None, :, None, None], axis=5)
        words = words.transpose(5, 0, 1, 3, 6, 2, 4)                   # [tile, chunk, h, rq, c, jlo, half]
        qw_t = np.ascontiguousarray(words).reshape(-1)
    meta = np.zeros((tiles, G, 16), dtype=np.uint32)
    sb = np.zeros((G, Np), dtype=np.uint32)
    sb[:, :N] = scales_bits.astype(np.uint32)
    zz = np.zeros((G, Np), dtype=np.uint32)
    zz[:, :N] = zeros
    m = sb | ((np.uint32(0xE400) | zz) << np.uint32(16))
    meta = m.reshape(G, tiles, 16).transpose(1, 0, 2)
    return qw_t.view(np.int32), np.ascontiguousarray(meta).reshape(-1).view(np.int32)


def shared_gpu_wait_ms(world: int, default_ms: int = 10000) -> int:
    """Peer-wait bound for the multi-process tests whose ranks all run on ONE GPU.  With 4 / 8 processes (+ the pytest process)
    the GPU scheduler time-slices them -- a collective step costs a rotation of its quanta (measured: ~4.5 ms per all-reduce at
    world 8 vs 56 us at world 2) and an occasional rotation takes seconds -- so the product's 10 s bound, sized for one process
    per GPU, can expire although nothing is lost.  Those tests wait longer instead of failing on the scheduler."""
    return default_ms if world < 4 else max(default_ms, 120000)
