"""CPU ORACLE for the GPTQ/AWQ int4/int8 grouped dequant-matmul path  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module, and
only as the checker / reported baseline.  The product (gptqmodel_amd/) never imports it and fails
loudly when the HIP library is missing.

It is an independent numpy restatement of the reference algorithm (ModelCloud/GPTQModel), each function
citing the reference file:line it follows (paths relative to /root/reference/):

  GPTQ  gptqmodel/nn_modules/qlinear/torch.py          TorchLinear  (BACKEND.TORCH / GPTQ_TORCH)
        gptqmodel/nn_modules/qlinear/__init__.py       buffer contract, generic dequant
        gptqmodel/utils/model.py:750-844               v1 -> v2 qzeros conversion
  AWQ   gptqmodel/nn_modules/qlinear/torch_awq.py      AwqTorchLinear (BACKEND.TORCH_AWQ)
        gptqmodel/quantization/awq/utils/packing_utils.py

PARITY PIN (see DESIGN.md "Oracle"): checked against
  * tests/golden/q4_kat_1024.npz   -- the reference's own known-answer vector tests/q4_reference.py
    (recipe tests/test_q4_exllama_v2.py:65-88), and
  * tests/golden/ref_*.npz         -- outputs of the real reference TorchLinear / AwqTorchLinear run in
    the build container by oracle/make_golden.py (dequant stage bit-exact, forward within 1 output ulp).

Rounding contract restated:
  W[k,n]  = round_to(scales.dtype)( float(scales[g,n]) * (code[k,n] - zero[g,n]) )   one rounding
  W'      = round_to(x.dtype)(W)                  (second rounding only when scales fp16 and x bf16)
  y[m,n]  = round_to(x.dtype)( sum_k float(x[m,k]) * float(W'[k,n]) )  (+ bias, rounded again)
The contraction itself is aten matmul in the reference (accumulation order unpinned); the oracle
accumulates in fp32 (numpy sgemm) -- parity for that stage is "within tolerance", see tests.
"""
from __future__ import annotations

import numpy as np

AWQ_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)          # packing_utils.py:9
AWQ_REVERSE_ORDER = (0, 4, 1, 5, 2, 6, 3, 7)  # packing_utils.py:10

FP16 = "fp16"
BF16 = "bf16"


# ----------------------------------------------------------------------------------------------
# bf16 helpers (numpy has no bfloat16): bf16 values are carried as float32 arrays that are exactly
# representable in bf16; `round_bf16` is round-to-nearest-even like torch's .to(torch.bfloat16).
# ----------------------------------------------------------------------------------------------
def round_bf16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    rounding = ((u >> 16) & 1) + 0x7FFF
    r = (((u + rounding) >> 16) << 16).astype(np.uint32)
    out = r.view(np.float32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.nan
    return out


def bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 array (already bf16-representable) -> uint16 bit patterns."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def round_to(x: np.ndarray, dtype: str) -> np.ndarray:
    """Round an fp32 array to `dtype` and return it as float32 (exactly representable)."""
    if dtype == FP16:
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)
    if dtype == BF16:
        return round_bf16(x)
    raise ValueError(dtype)


# ----------------------------------------------------------------------------------------------
# GPTQ
# ----------------------------------------------------------------------------------------------
def unpack_rows(qweight: np.ndarray, bits: int) -> np.ndarray:
    """qweight int32 [K/pf, N] -> codes uint8 [K, N]; code(8r+j, n) = (word >> bits*j) & maxq.
    Follows torch.py:706-714 (`_right_shift_unpack` of qweight.unsqueeze(1) by wf[1,pf,1], & maxq)
    and the layout contract qlinear/__init__.py:827-865."""
    assert bits in (2, 4, 8)
    pf = 32 // bits
    w = np.ascontiguousarray(qweight).view(np.uint32)
    shifts = (np.arange(pf, dtype=np.uint32) * bits)[None, :, None]
    codes = (w[:, None, :] >> shifts) & np.uint32((1 << bits) - 1)
    return codes.reshape(w.shape[0] * pf, w.shape[1]).astype(np.uint8)


def unpack_cols(qzeros: np.ndarray, bits: int) -> np.ndarray:
    """qzeros int32 [G, N/pf] -> zeros uint8 [G, N]; zero(g, pf*c+j) = (word >> bits*j) & maxq.
    Follows torch.py:465-478 (`_stream_decode_qzeros`)."""
    assert bits in (2, 4, 8)
    pf = 32 // bits
    w = np.ascontiguousarray(qzeros).view(np.uint32)
    shifts = (np.arange(pf, dtype=np.uint32) * bits)[None, None, :]
    z = (w[:, :, None] >> shifts) & np.uint32((1 << bits) - 1)
    return z.reshape(w.shape[0], w.shape[1] * pf).astype(np.uint8)


# ---- the other bit widths of the generic dequantize_weight (qlinear/__init__.py:947-999) -------------------
# continuous 3-bit: 32 codes in three int32 words, code i at bit 3 i of the 96-bit little-endian group -- the reference special-cases
# the two codes that straddle a word boundary (i = 10: bits 30-31 of word 0 + bit 0 of word 1; i = 21: bit 31 of word 1 + bits 0-1 of
# word 2; qlinear/__init__.py:982-991).  Planar ("split-plane", utils/planar_packing.py:7-24; always for 5 / 6 / 7 bits): per 32 codes
# `bits` words, first the low plane's words, then the higher planes; inside a plane of width w, word i holds codes [i*32/w, (i+1)*32/w)
# at shifts w*j.
PLANES = {2: ((2, 0),), 3: ((2, 0), (1, 2)), 4: ((4, 0),), 5: ((4, 0), (1, 4)), 6: ((4, 0), (2, 4)), 7: ((4, 0), (2, 4), (1, 6)), 8: ((8, 0),)}
PLANAR_ONLY_BITS = (5, 6, 7)


def is_planar(bits: int, planar=None) -> bool:
    """5 / 6 / 7 bits only exist planar; 3 bits are planar only under FORMAT.GPTQ_P (qlinear/__init__.py:766-773)."""
    return bits in PLANAR_ONLY_BITS if planar is None else bool(planar)


def _unpack_groups(words: np.ndarray, bits: int, planar: bool) -> np.ndarray:
    """words uint32 [blocks, bits, cols] (the `bits` words of every group of 32 codes) -> codes uint32 [blocks, 32, cols]."""
    blocks, _, cols = words.shape
    out = np.zeros((blocks, 32, cols), dtype=np.uint32)
    if planar:
        row = 0
        for width, offset in PLANES[bits]:
            pf = 32 // width
            sh = (np.arange(pf, dtype=np.uint32) * width)[None, None, :, None]
            codes = (words[:, row:row + width, None, :] >> sh) & np.uint32((1 << width) - 1)      # [blocks, width, pf, cols]
            out |= codes.reshape(blocks, 32, cols) << np.uint32(offset)
            row += width
        return out
    stream = words.astype(np.uint64)
    for i in range(32):
        pos = bits * i
        w, sh = pos // 32, pos % 32
        v = stream[:, w] >> np.uint64(sh)
        if sh + bits > 32:
            v = v | (stream[:, w + 1] << np.uint64(32 - sh))
        out[:, i] = (v & np.uint64((1 << bits) - 1)).astype(np.uint32)
    return out


def unpack_rows_any(qweight: np.ndarray, bits: int, planar=None) -> np.ndarray:
    """qweight int32 [K*bits/32, N] -> codes uint8 [K, N] for every bit width the reference's generic path reads."""
    if bits in (2, 4, 8) and not is_planar(bits, planar):
        return unpack_rows(qweight, bits)
    w = np.ascontiguousarray(qweight).view(np.uint32)
    assert w.shape[0] % bits == 0
    return _unpack_groups(w.reshape(w.shape[0] // bits, bits, w.shape[1]), bits, is_planar(bits, planar)).reshape(-1, w.shape[1]).astype(np.uint8)


def unpack_cols_any(qzeros: np.ndarray, bits: int, planar=None) -> np.ndarray:
    """qzeros int32 [G, N*bits/32] -> zero-points uint8 [G, N] (the column-packed twin of unpack_rows_any)."""
    if bits in (2, 4, 8) and not is_planar(bits, planar):
        return unpack_cols(qzeros, bits)
    return unpack_rows_any(np.ascontiguousarray(np.ascontiguousarray(qzeros).T), bits, planar).T.copy()


def _pack_groups(codes: np.ndarray, bits: int, planar: bool) -> np.ndarray:
    """codes [blocks, 32, cols] -> words uint32 [blocks, bits, cols] (inverse of _unpack_groups; test-tensor synthesis only)."""
    blocks, _, cols = codes.shape
    c = codes.astype(np.uint64)
    out = np.zeros((blocks, bits, cols), dtype=np.uint64)
    if planar:
        row = 0
        for width, offset in PLANES[bits]:
            pf = 32 // width
            plane = ((c >> np.uint64(offset)) & np.uint64((1 << width) - 1)).reshape(blocks, width, pf, cols)
            sh = (np.arange(pf, dtype=np.uint64) * np.uint64(width))[None, None, :, None]
            out[:, row:row + width] = np.bitwise_or.reduce(plane << sh, axis=2)
            row += width
    else:
        for i in range(32):
            pos = bits * i
            w, sh = pos // 32, pos % 32
            out[:, w] |= (c[:, i] << np.uint64(sh)) & np.uint64(0xFFFFFFFF)
            if sh + bits > 32:
                out[:, w + 1] |= c[:, i] >> np.uint64(32 - sh)
    return (out & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def pack_rows_any(codes: np.ndarray, bits: int, planar=None) -> np.ndarray:
    if bits in (2, 4, 8) and not is_planar(bits, planar):
        return pack_rows(codes, bits)
    k, n = codes.shape
    assert k % 32 == 0
    return _pack_groups(codes.reshape(k // 32, 32, n), bits, is_planar(bits, planar)).reshape(-1, n).view(np.int32)


def pack_cols_any(zeros: np.ndarray, bits: int, planar=None) -> np.ndarray:
    if bits in (2, 4, 8) and not is_planar(bits, planar):
        return pack_cols(zeros, bits)
    return np.ascontiguousarray(pack_rows_any(np.ascontiguousarray(zeros.T), bits, planar).T)


def widen_codes(qweight: np.ndarray, qzeros: np.ndarray, bits: int, planar=None):
    """Checkpoint tensors of any bit width -> the SAME codes and zero-points in the continuous 4-bit (bits <= 4) or 8-bit layout the
    HIP kernels read: W = scale * (code - zero) is unchanged, only the field width grows.  (The layout model of gptqhip_widen_codes.)"""
    wide = 4 if bits <= 4 else 8
    return pack_rows(unpack_rows_any(qweight, bits, planar), wide), pack_cols(unpack_cols_any(qzeros, bits, planar), wide), wide


def convert_v1_to_v2_qzeros(qzeros: np.ndarray, bits: int, planar=None) -> np.ndarray:
    if bits in (3, 5, 6, 7) or is_planar(bits, planar):
        # fields that straddle words or planes: shift the DECODED values by one, modulo 2^bits, and re-encode
        # (utils/model.py:767-775,834-839 -> model_dequant._shift_gptq_qzeros)
        z = (unpack_cols_any(qzeros, bits, planar).astype(np.uint32) + 1) & ((1 << bits) - 1)
        return pack_cols_any(z.astype(np.uint8), bits, planar)
    return _convert_v1_to_v2_words(qzeros, bits)


def _convert_v1_to_v2_words(qzeros: np.ndarray, bits: int) -> np.ndarray:
    """GPTQ v1 checkpoints store zero-1; the loader adds 0x11111111 (4-bit) / 0x01010101 (8-bit) per
    int32 word with wraparound.  utils/model.py:814-831."""
    add = {2: 0x55555555, 4: 0x11111111, 8: 0x01010101}[bits]
    return (np.ascontiguousarray(qzeros).view(np.uint32) + np.uint32(add)).view(np.int32)


def normalize_g_idx(g_idx: np.ndarray, groups: int) -> np.ndarray:
    """Negative g_idx wraps by +G like python/torch indexing does in `scales[g_idx]` (torch.py:717)."""
    g = np.asarray(g_idx).astype(np.int64)
    g = np.where(g < 0, g + groups, g)
    if g.min() < 0 or g.max() >= groups:
        raise IndexError("g_idx out of range")  # torch raises IndexError too
    return g


def dequant_gptq(qweight, qzeros, scales_f32, g_idx, bits: int, scale_dtype: str = FP16, planar=None) -> np.ndarray:
    """[K,N] weights as float32 holding values exactly representable in `scale_dtype`.
    W = scales[g_idx] * (code - zeros[g_idx])   torch.py:716-717  == qlinear/__init__.py:1001-1003.
    (code - zero) is an int8/int16 subtraction (exact), the product of an fp16|bf16 scale and an integer
    |.|<=255 is exact in fp32, so rounding the fp32 product once reproduces torch's fp16/bf16 multiply."""
    codes = unpack_rows_any(qweight, bits, planar).astype(np.int32)      # planar: FORMAT.GPTQ_P (3 bits; 5 / 6 / 7 always are)
    zeros = unpack_cols_any(qzeros, bits, planar).astype(np.int32)
    scales_f32 = np.asarray(scales_f32, dtype=np.float32)
    g = normalize_g_idx(g_idx, scales_f32.shape[0])
    w = scales_f32[g] * (codes - zeros[g]).astype(np.float32)
    return round_to(w, scale_dtype)


def matmul_round(x_f32: np.ndarray, w_f32: np.ndarray, bias_f32, act_dtype: str) -> np.ndarray:
    """out = round(x @ W) (+ bias, rounded again: `out.add_(bias)` on a tensor of x.dtype, torch.py:337-342)."""
    acc = np.asarray(x_f32, np.float32) @ np.asarray(w_f32, np.float32)
    out = round_to(acc, act_dtype)
    if bias_f32 is not None:
        out = round_to(out + np.asarray(bias_f32, np.float32)[None, :], act_dtype)
    return out


def forward_gptq(x_f32, qweight, qzeros, scales_f32, g_idx, bits: int, bias_f32=None,
                 act_dtype: str = FP16, scale_dtype: str = FP16, planar=None) -> np.ndarray:
    """TorchLinear._forward_eager (torch.py:326-347): dequantize, cast weights to x.dtype, matmul, bias."""
    w = dequant_gptq(qweight, qzeros, scales_f32, g_idx, bits, scale_dtype, planar)
    if act_dtype != scale_dtype:
        w = round_to(w, act_dtype)  # torch.py:331-335  weights.to(dtype=x.dtype)
    x2 = np.asarray(x_f32, np.float32).reshape(-1, x_f32.shape[-1])
    out = matmul_round(x2, w, bias_f32, act_dtype)
    return out.reshape(x_f32.shape[:-1] + (w.shape[1],))


# ----------------------------------------------------------------------------------------------
# AWQ (FORMAT.GEMM)
# ----------------------------------------------------------------------------------------------
def unpack_awq_cols(q: np.ndarray, bits: int = 4) -> np.ndarray:
    """[R, N/8] int32 -> uint8 [R, N] in LOGICAL column order.
    packing_utils.py:13-27 (`_unpack_columnwise`: slot 8c+i = nibble i) followed by
    packing_utils.py:43-57 (`reverse_awq_order`: logical col 8c+j = slot 8c+REV[j]) and the & 0xF of :113-114."""
    assert bits == 4
    w = np.ascontiguousarray(q).view(np.uint32)
    shifts = (np.arange(8, dtype=np.uint32) * 4)[None, None, :]
    slots = ((w[:, :, None] >> shifts) & np.uint32(0xF)).astype(np.uint8)  # [R, N/8, 8] nibble i
    logical = slots[:, :, list(AWQ_REVERSE_ORDER)]
    return logical.reshape(w.shape[0], w.shape[1] * 8)


def dequant_awq(qweight, qzeros, scales_f32, group_size: int, compute_dtype: str = FP16) -> np.ndarray:
    """packing_utils.py:106-121 `dequantize_gemm`: (iweight - izeros.repeat(g)) * scales.repeat(g).
    scales are first cast to the compute dtype (torch_awq.py:149-155), the product is rounded once."""
    codes = unpack_awq_cols(qweight).astype(np.int32)
    zeros = unpack_awq_cols(qzeros).astype(np.int32)
    s = round_to(np.asarray(scales_f32, np.float32), compute_dtype)
    k = codes.shape[0]
    if group_size == -1:
        group_size = k
    g = np.arange(k) // group_size
    w = (codes - zeros[g]).astype(np.float32) * s[g]
    return round_to(w, compute_dtype)


def forward_awq(x_f32, qweight, qzeros, scales_f32, group_size: int, bias_f32=None,
                act_dtype: str = FP16) -> np.ndarray:
    """AwqTorchLinear.forward (torch_awq.py:157-195): out-of-place bias add in compute dtype."""
    w = dequant_awq(qweight, qzeros, scales_f32, group_size, act_dtype)
    x2 = np.asarray(x_f32, np.float32).reshape(-1, x_f32.shape[-1])
    bias = None if bias_f32 is None else round_to(np.asarray(bias_f32, np.float32), act_dtype)
    out = matmul_round(x2, w, bias, act_dtype)
    return out.reshape(x_f32.shape[:-1] + (w.shape[1],))


def awq_to_gptq_layout(qweight, qzeros, bits: int = 4):
    """packing_utils.py:90-103 `unpack_reorder_pack`: AWQ [K,N/8] interleaved -> K-packed sequential
    qweight [K/8,N] + N-packed sequential qzeros [G,N/8] (zeros as-is, no +-1).  Checker for the
    device repack kernel."""
    codes = unpack_awq_cols(qweight)
    zeros = unpack_awq_cols(qzeros)
    return pack_rows(codes, bits), pack_cols(zeros, bits)


# ----------------------------------------------------------------------------------------------
# packers for synthetic fixtures (inverse of the unpackers above; contract of
# qlinear/__init__.py:1197-1323 python pack path -- only the bit layout, no quantisation)
# ----------------------------------------------------------------------------------------------
def pack_rows(codes: np.ndarray, bits: int) -> np.ndarray:
    pf = 32 // bits
    k, n = codes.shape
    assert k % pf == 0
    c = codes.astype(np.uint32).reshape(k // pf, pf, n)
    shifts = (np.arange(pf, dtype=np.uint32) * bits)[None, :, None]
    return np.bitwise_or.reduce(c << shifts, axis=1).astype(np.uint32).view(np.int32)


def pack_cols(zeros: np.ndarray, bits: int) -> np.ndarray:
    pf = 32 // bits
    g, n = zeros.shape
    assert n % pf == 0
    z = zeros.astype(np.uint32).reshape(g, n // pf, pf)
    shifts = (np.arange(pf, dtype=np.uint32) * bits)[None, None, :]
    return np.bitwise_or.reduce(z << shifts, axis=2).astype(np.uint32).view(np.int32)


def pack_awq_cols(vals: np.ndarray) -> np.ndarray:
    """[R,N] logical uint8 -> AWQ int32 [R,N/8]: nibble i of word c holds column 8c+AWQ_ORDER[i]
    (torch_awq.py:134-139)."""
    r, n = vals.shape
    v = vals.astype(np.uint32).reshape(r, n // 8, 8)[:, :, list(AWQ_ORDER)]
    shifts = (np.arange(8, dtype=np.uint32) * 4)[None, None, :]
    return np.bitwise_or.reduce(v << shifts, axis=2).astype(np.uint32).view(np.int32)


def quantize_pack_gptq(weight_f32: np.ndarray, scales_f32: np.ndarray, zeros: np.ndarray, g_idx: np.ndarray, bits: int, planar=None):
    """The reference packer's quantise-and-pack step: weight [N,K] fp32, scales [G,N] fp32, zeros [G,N] ints, g_idx [K]
    -> (qweight int32 [K*bits/32, N], qzeros int32 [G, N*bits/32]).
    q = clamp(rint((w + zero*scale) / scale), 0, maxq) in fp32, scale==0 -> 1e-6, negative g_idx wraps by +G
    (gptqmodel_ext/pack_block_cpu.cpp:105-143; python path qlinear/__init__.py:1197-1228).  rint = round-half-even
    like std::nearbyint / torch.round."""
    w = np.ascontiguousarray(weight_f32, dtype=np.float32)
    s = np.ascontiguousarray(scales_f32, dtype=np.float32)
    z = np.asarray(zeros).astype(np.float32)
    g = normalize_g_idx(g_idx, s.shape[0])
    off = (z * s).astype(np.float32)                       # scale_zeros = zeros * scales   (fp32)
    sk = s[g].T.astype(np.float32)                         # [N, K]
    sk = np.where(sk == 0.0, np.float32(1e-6), sk)
    q = np.rint(((w + off[g].T) / sk).astype(np.float32))
    q = np.clip(q, 0, (1 << bits) - 1).astype(np.uint8)    # [N, K]
    return pack_rows_any(q.T.copy(), bits, planar), pack_cols_any(np.asarray(zeros).astype(np.uint8), bits, planar)


def act_order_perm(g_idx: np.ndarray) -> np.ndarray:
    """Stable argsort of g_idx: the row order in which the backend stores act-order weights
    (semantics of torch_fused.py:121-151 / utils/marlin.py:368-372)."""
    return np.argsort(np.asarray(g_idx).astype(np.int64), kind="stable").astype(np.int32)


# ----------------------------------------------------------------------------------------------
# torch-CPU restatement: the same op sequence as BACKEND.TORCH, multi-threaded aten, used ONLY as the
# timed `cpu_baseline` ("port") in bench.py and cross-checked against the numpy oracle in tests.
# ----------------------------------------------------------------------------------------------
def torch_cpu_dequant_gptq(qweight, qzeros, scales, g_idx, bits: int):
    """[K,N] weights in scales.dtype: the aten op sequence of torch.py:700-717 (shift-unpack to int8, mask, gather by g_idx,
    (w - z) * s).  This is the function upstream wraps in torch.compile in post_init (torch.py:215-216,259)."""
    import torch
    pf = 32 // bits
    maxq = (1 << bits) - 1
    ddt = torch.int16 if bits == 8 else torch.int8
    sh = torch.arange(0, 32, bits, dtype=torch.int32)
    z = torch.bitwise_and(torch.bitwise_right_shift(qzeros.unsqueeze(2).expand(-1, -1, pf), sh.view(1, 1, pf)).to(ddt), maxq)
    z = z.reshape(scales.shape)
    w = torch.bitwise_and(torch.bitwise_right_shift(qweight.unsqueeze(1).expand(-1, pf, -1), sh.view(1, pf, 1)).to(ddt), maxq)
    w = w.reshape(w.shape[0] * pf, w.shape[2])
    g = g_idx.long()
    return scales[g] * (w - z[g])


def torch_cpu_forward_gptq(x, qweight, qzeros, scales, g_idx, bits: int, bias=None, dequant=torch_cpu_dequant_gptq):
    """x [M,K] fp16|bf16 CPU tensor; packed tensors as in the checkpoint (v2 zeros).
    Same aten op sequence as torch.py:700-717 + 326-347: dequant (above; `dequant` may be its torch.compile'd form), cast,
    matmul, bias."""
    import torch
    weights = dequant(qweight, qzeros, scales, g_idx, bits)
    if weights.dtype != x.dtype:
        weights = weights.to(x.dtype)
    out = torch.matmul(x, weights)
    if bias is not None:
        out.add_(bias.to(out.dtype))
    return out


def torch_cpu_dequant_awq(qweight, qzeros, scales, group_size: int, bits: int = 4):
    """[K,N] weights in scales.dtype: the aten op sequence of `dequantize_gemm` (quantization/awq/utils/packing_utils.py:106-121):
    column-wise shift-unpack of qweight [K,N/8] and qzeros [G,N/8] to int8 (:14-30), the AWQ_REVERSE_ORDER column gather (:46-61,
    order :10), the overflow mask, repeat_interleave of scales / zeros over the group (:117-118), (w - z) * s."""
    import torch
    pf = 32 // bits
    sh = torch.arange(0, 32, bits)
    iw = torch.bitwise_right_shift(qweight[:, :, None], sh[None, None, :]).to(torch.int8).view(qweight.shape[0], -1)
    iz = torch.bitwise_right_shift(qzeros[:, :, None], sh[None, None, :]).to(torch.int8).view(qzeros.shape[0], -1)
    rev = torch.arange(iw.shape[-1], dtype=torch.int32).view(-1, pf)[:, AWQ_REVERSE_ORDER].reshape(-1)
    iw, iz = iw[:, rev], iz[:, rev]
    iw = torch.bitwise_and(iw, (1 << bits) - 1)
    iz = torch.bitwise_and(iz, (1 << bits) - 1)
    return (iw - iz.repeat_interleave(group_size, dim=0)) * scales.repeat_interleave(group_size, dim=0)


def torch_cpu_forward_awq(x, qweight, qzeros, scales, group_size: int, bias=None, bits: int = 4):
    """x [M,K] fp16|bf16 CPU tensor, AWQ GEMM-layout tensors.  Same aten op sequence as AwqTorchLinear.forward
    (nn_modules/qlinear/torch_awq.py:157-195): scales cast to the compute dtype (:149-155), dequantize_gemm, contiguous weight in
    the compute dtype, matmul, out-of-place bias add.  Timed as the C4 `cpu_baseline` leg in bench.py ("port")."""
    import torch
    if scales.dtype != x.dtype:
        scales = scales.to(x.dtype)
    weight = torch_cpu_dequant_awq(qweight, qzeros, scales, group_size, bits)
    if weight.dtype != x.dtype or not weight.is_contiguous():
        weight = weight.to(x.dtype).contiguous()
    out = torch.matmul(x, weight)
    if bias is not None:
        out = out + bias.to(out.dtype)
    return out


# ----------------------------------------------------------------------------------------------
# Decoder-layer glue between the quantised linears -- NOT part of the reference repo: these restate what the reference's
# CALLER (HF transformers LlamaDecoderLayer: LlamaRMSNorm.forward, LlamaMLP.forward, the two residual adds) computes between
# QuantLinear.forward calls, in the activation dtype, so the fused decode ops (gptqhip_decode_linear) have a checker.
# ----------------------------------------------------------------------------------------------
def rmsnorm_ref(h_f32: np.ndarray, weight_f32: np.ndarray, eps: float, act_dtype: str) -> np.ndarray:
    """LlamaRMSNorm: h32 = h.float(); var = mean(h32^2); (h32 * rsqrt(var + eps)).to(dtype) * weight  (rounded)."""
    h = np.asarray(h_f32, np.float32)
    var = np.mean(h.astype(np.float64) ** 2, axis=-1, keepdims=True).astype(np.float32)
    inv = (1.0 / np.sqrt(var.astype(np.float64) + eps)).astype(np.float32)
    return round_to(np.asarray(weight_f32, np.float32) * round_to(h * inv, act_dtype), act_dtype)


def silu_mul_ref(gate_f32: np.ndarray, up_f32: np.ndarray, act_dtype: str) -> np.ndarray:
    """LlamaMLP: act_fn(gate) * up with act_fn = SiLU evaluated in fp32 (aten computes half / bfloat16 SiLU in float opmath:
    x / (1 + exp(-x)) with the fp32 exp, so exp(-x) overflows to inf for x < -88.7 and the result is -0, not a 1e-37 denormal-ish
    value a float64 evaluation would produce) and rounded to the activation dtype."""
    g = np.asarray(gate_f32, np.float32)
    with np.errstate(over="ignore"):
        s = round_to(g / (np.float32(1.0) + np.exp(-g)), act_dtype)
    return round_to(s * np.asarray(up_f32, np.float32), act_dtype)


def residual_add_ref(res_f32: np.ndarray, y_f32: np.ndarray, act_dtype: str) -> np.ndarray:
    """hidden = residual + hidden in the activation dtype (one rounding)."""
    return round_to(np.asarray(res_f32, np.float32) + np.asarray(y_f32, np.float32), act_dtype)
