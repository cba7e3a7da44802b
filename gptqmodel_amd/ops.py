"""Thin torch-tensor wrappers over the C ABI (raw device pointers + the current HIP stream).

PyTorch is plumbing here (device memory, streams); all arithmetic happens in libgptqhip.so.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import torch

from . import _lib

FP16, BF16 = 0, 1
_DT = {torch.float16: FP16, torch.bfloat16: BF16}

# per (device index, stream handle) zero-initialised scratch, grown on demand
# (precedent: ExllamaV2 per-device ScratchSpace, gptqmodel/utils/model.py:1304-1313)
_workspaces: Dict[Tuple[int, int], torch.Tensor] = {}
# Outgrown workspaces are RETIRED, never freed: a HIP graph captured earlier still holds the old pointer and relies on
# its arrival counters staying zero, so the allocator must not hand that memory to anyone else.
_retired: list = []
# gptqhip_workspace_bytes per problem signature (it re-runs the launch planners: not free on the eager decode path)
_need_cache: Dict[Tuple[int, int, int, int, int, int], int] = {}


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream(device: torch.device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


try:  # raw current-stream handle without building a torch.cuda.Stream object (~1.5 us saved per eager launch)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover - older / newer torch
    _raw_stream = None


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("gptqmodel_amd HIP ops need tensors on a ROCm device (got a CPU tensor); "
                               "there is no CPU fallback")


def device_info(device: int = 0):
    lib = _lib.load()
    cu = ctypes.c_int(0)
    hbm = ctypes.c_size_t(0)
    arch = ctypes.create_string_buffer(64)
    rc = lib.gptqhip_device_info(device, ctypes.byref(cu), ctypes.byref(hbm), arch, 64)
    _lib.check(rc, "gptqhip_device_info")
    return {"cu_count": cu.value, "hbm_bytes": hbm.value, "arch": arch.value.decode()}


def _dev_index(device: torch.device) -> int:
    return device.index if device.index is not None else torch.cuda.current_device()


def _stream_handle(dev_index: int) -> int:
    if _raw_stream is not None:
        return _raw_stream(dev_index)
    return torch.cuda.current_stream(dev_index).cuda_stream


def workspace_for(device: torch.device, nbytes: int, stream_handle: Optional[int] = None) -> torch.Tensor:
    idx = _dev_index(device)
    key = (idx, _stream_handle(idx) if stream_handle is None else stream_handle)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if ws is not None:
            _retired.append(ws)
        # zero-filled: the split-K arrival counters must start at 0 (kernels reset them after use)
        ws = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def reserve_workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Pre-size the current stream's workspace (e.g. for the largest prefill) BEFORE capturing a HIP graph, so that no
    later call has to replace the buffer the graph points at."""
    return workspace_for(device, nbytes)


def workspace_bytes(M: int, K: int, N: int, group_size: int, bits: int, has_perm: bool) -> int:
    key = (M, K, N, group_size, bits, 1 if has_perm else 0)
    need = _need_cache.get(key)
    if need is None:
        need = int(_lib.load().gptqhip_workspace_bytes(*key))
        if need == 0:
            raise RuntimeError(f"gptqhip_workspace_bytes: unsupported problem M={M} K={K} N={N} group_size={group_size} bits={bits}")
        if len(_need_cache) > 4096:
            _need_cache.clear()
        _need_cache[key] = need
    return need


def repack_tiled(qweight: Optional[torch.Tensor], qzeros: torch.Tensor, scales: torch.Tensor,
                 perm: Optional[torch.Tensor], group_size: int, bits: int):
    """Checkpoint layout (qweight [K*bits/32,N], qzeros [G,N*bits/32], scales [G,N]) -> (qweight_t, meta) in the
    MFMA-tile-major kernel layout (include/gptqhip.h).  One-time, on device, in post_init.
    qweight=None rebuilds only `meta` (returns (None, meta))."""
    lib = _lib.load()
    _require_cuda(qweight, qzeros, scales, perm)
    if scales.dtype not in _DT:
        raise RuntimeError(f"repack_tiled: unsupported scales dtype {scales.dtype}")
    pf = 32 // bits
    G, N = scales.shape
    K = G * group_size
    if qweight is not None:
        if tuple(qweight.shape) != (K // pf, N):
            raise RuntimeError(f"repack_tiled: qweight shape {tuple(qweight.shape)} != {(K // pf, N)}")
        qweight = qweight.contiguous()
    qzeros, scales = qzeros.contiguous(), scales.contiguous()
    nw = lib.gptqhip_tiled_words(K, N, bits)
    nm = lib.gptqhip_meta_words(K, N, group_size)
    if nw == 0 or nm == 0:
        raise RuntimeError(f"repack_tiled: unsupported shape K={K} N={N} group_size={group_size} bits={bits}")
    qw_t = torch.empty(nw, dtype=torch.int32, device=scales.device) if qweight is not None else None
    meta = torch.empty(nm, dtype=torch.int32, device=scales.device)
    with torch.cuda.device(scales.device):
        rc = lib.gptqhip_repack_tiled(_ptr(qweight), _ptr(qzeros), _ptr(scales), _ptr(perm), _ptr(qw_t), _ptr(meta),
                                      K, N, group_size, bits, _stream(scales.device))
    _lib.check(rc, "gptqhip_repack_tiled")
    return qw_t, meta


def gemm(x: torch.Tensor, qweight_t: torch.Tensor, meta: torch.Tensor, bias: Optional[torch.Tensor],
         perm: Optional[torch.Tensor], N: int, group_size: int, bits: int, scale_dtype: torch.dtype,
         out: Optional[torch.Tensor] = None, partial_f32: bool = False, exact_bf16: bool = False) -> torch.Tensor:
    """out[M,N] = x[M,K] @ dequant(qweight_t, meta) (+bias)  via gptqhip_gemm (tiled layout).
    partial_f32=True returns the unrounded float32 accumulators (tensor-parallel partial sums, no bias).
    exact_bf16=True opts in to GPTQHIP_GEMM_EXACT_BF16 (bf16 decode without the per-weight bf16 rounding; see
    include/gptqhip.h)."""
    # Hot eager path (HF generate calls this once per quantised linear per token): every line here is host time in front
    # of a 5-15 us kernel (tests/dev/eager_overhead.py), so no generic loops, no Stream / c_void_p objects, no device context
    # switch unless the tensor lives on another device than the current one.
    lib = _lib.load()
    if not (x.is_cuda and qweight_t.is_cuda and meta.is_cuda):
        raise RuntimeError("gptqmodel_amd HIP ops need tensors on a ROCm device (got a CPU tensor); "
                           "there is no CPU fallback")
    if x.dim() != 2 or not x.is_contiguous():
        raise RuntimeError("gemm: x must be a contiguous [M,K] tensor")
    adt, sdt = _DT.get(x.dtype), _DT.get(scale_dtype)
    if adt is None or sdt is None:
        raise RuntimeError(f"gemm: unsupported dtypes x={x.dtype} scales={scale_dtype}")
    if bias is not None and bias.dtype != x.dtype:
        raise RuntimeError("gemm: bias dtype must equal activation dtype")
    M, K = x.shape
    if partial_f32 and bias is not None:
        raise RuntimeError("gemm: partial_f32 excludes bias (add it once after the all-reduce)")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if partial_f32 else x.dtype, device=x.device)
    if M == 0:
        return out
    idx = x.device.index
    switch = idx != torch.cuda.current_device()
    if switch:
        prev = torch.cuda.current_device()
        torch.cuda.set_device(idx)
    try:
        sh = _stream_handle(idx)
        ws = workspace_for(x.device, workspace_bytes(M, K, N, group_size, bits, perm is not None), sh)
        rc = lib.gptqhip_gemm(x.data_ptr(), qweight_t.data_ptr(), meta.data_ptr(), 0 if perm is None else perm.data_ptr(),
                              0 if bias is None else bias.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N,
                              group_size, bits, adt, sdt, (1 if partial_f32 else 0) | (2 if exact_bf16 else 0), sh)
    finally:
        if switch:
            torch.cuda.set_device(prev)
    if rc != 0:
        _lib.check(rc, "gptqhip_gemm")
    return out


GLUE_NONE, GLUE_RMSNORM, GLUE_SILU_MUL = 0, 1, 2
OUT_NONE, OUT_SILU_MUL_PAIRED, OUT_PARTIAL_F32 = 0, 1, 2


def decode_supported(K: int, N: int, group_size: int, has_perm: bool = False, M: int = 1) -> bool:
    """True when gptqhip_decode_linear handles a [K,N] layer (regular pipeline; has_perm: with an act-order permutation
    applied in the kernel; M: rows 1..16), else use gemm()."""
    return bool(_lib.load().gptqhip_decode_supported(K, N, group_size, 1 if has_perm else 0, M))


def make_decode_op(x: torch.Tensor, qweight_t: torch.Tensor, meta: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor,
                   K: int, N: int, group_size: int, bits: int, scale_dtype: torch.dtype, in_glue: int = GLUE_NONE,
                   norm_weight: Optional[torch.Tensor] = None, eps: float = 1e-5, residual: Optional[torch.Tensor] = None,
                   workspace: Optional[torch.Tensor] = None, out_glue: int = OUT_NONE,
                   stats_in: Optional[torch.Tensor] = None, stats_out: Optional[torch.Tensor] = None,
                   perm: Optional[torch.Tensor] = None, M: int = 1, exact: bool = False) -> "_lib.DecodeOp":
    """Fill a struct gptqhip_decode_op (include/gptqhip.h) from tensors.  The struct only holds raw pointers: the caller
    keeps the tensors alive (DecodeStep does) -- binding once and re-launching costs no per-call Python work.
    `workspace`: the stream's scratch (workspace_for); taken from the CURRENT stream when omitted.
    `perm`: the module's act-order permutation (int32 [K]); applied to the glued input inside the kernel.
    `M`: rows (1..16): x [M,K], out / residual [M,N], stats_in [M, K/16], stats_out [M, ceil(N/16)], all contiguous.
    `exact`: the opt-in exact-arithmetic dequant (GPTQHIP_GEMM_EXACT, include/gptqhip.h)."""
    _require_cuda(x, qweight_t, meta, bias, out, norm_weight, residual, workspace, stats_in, stats_out, perm)
    if not 1 <= M <= 16:
        raise RuntimeError("decode op: M must be 1..16")
    if perm is not None and (perm.dtype != torch.int32 or perm.numel() != K or not perm.is_contiguous()):
        raise RuntimeError("decode op: perm must be a contiguous int32 [K] tensor")
    want_out = torch.float32 if out_glue == OUT_PARTIAL_F32 else x.dtype
    if x.dtype not in _DT or out.dtype != want_out or scale_dtype not in _DT:
        raise RuntimeError(f"decode op: unsupported dtypes x={x.dtype} out={out.dtype} scales={scale_dtype}")
    need = (2 * K if in_glue == GLUE_SILU_MUL else K) * M
    n_out = (N // 2 if out_glue == OUT_SILU_MUL_PAIRED else N) * M
    if x.numel() < need or not x.is_contiguous() or out.numel() < n_out or not out.is_contiguous():
        raise RuntimeError(f"decode op: x needs >= {need} contiguous elements (has {x.numel()}), out >= {n_out}")
    tiles_in, tiles_out = -(-K // 16), -(-N // 16)
    for t, n, what in ((stats_in, M * tiles_in, "stats_in"), (stats_out, M * tiles_out, "stats_out")):
        if t is not None and (t.dtype != torch.float32 or t.numel() < n or not t.is_contiguous()):
            raise RuntimeError(f"decode op: {what} must be a contiguous float32 tensor with >= {n} elements")
    for t, n, what in ((bias, N, "bias"), (residual, M * N, "residual"), (norm_weight, K, "norm_weight")):
        if t is not None and (t.dtype != x.dtype or t.numel() < n or not t.is_contiguous()):
            raise RuntimeError(f"decode op: {what} must be a contiguous {x.dtype} tensor with >= {n} elements")
    if in_glue == GLUE_RMSNORM and norm_weight is None:
        raise RuntimeError("decode op: GLUE_RMSNORM needs norm_weight")
    if workspace is None:
        with torch.cuda.device(x.device):
            workspace = workspace_for(x.device, workspace_bytes(M, K, N, group_size, bits, False))
    p = lambda t: 0 if t is None else t.data_ptr()
    return _lib.DecodeOp(p(qweight_t), p(meta), p(bias), p(x), p(norm_weight), p(residual), p(out), p(workspace),
                         workspace.numel(), p(stats_in), p(stats_out), p(perm), float(eps), K, N, group_size, bits, _DT[x.dtype],
                         _DT[scale_dtype], int(in_glue), int(out_glue), tiles_in if stats_in is not None else 0,
                         2 if exact else 0, int(M))


def launch_decode_op(op: "_lib.DecodeOp", device: torch.device) -> None:
    """Enqueue one decode op on the CURRENT stream of `device` (gptqhip_decode_linear)."""
    rc = _lib.load().gptqhip_decode_linear(ctypes.byref(op), _stream(device))
    _lib.check(rc, "gptqhip_decode_linear")


def bind_decode_seq(op_list):
    """ctypes array of pointers to bound decode ops, for launch_decode_seq (keeps the structs alive through the array)."""
    arr = (ctypes.POINTER(_lib.DecodeOp) * len(op_list))(*[ctypes.pointer(o) for o in op_list])
    arr._ops = list(op_list)
    return arr


def launch_decode_seq(seq, device: torch.device) -> None:
    """Enqueue a dependent run of bound decode ops with ONE host call (gptqhip_decode_linear_seq)."""
    rc = _lib.load().gptqhip_decode_linear_seq(ctypes.cast(seq, ctypes.c_void_p), len(seq), _stream(device))
    _lib.check(rc, "gptqhip_decode_linear_seq")


def decode_linear(x: torch.Tensor, qweight_t: torch.Tensor, meta: torch.Tensor, bias: Optional[torch.Tensor], K: int,
                  N: int, group_size: int, bits: int, scale_dtype: torch.dtype, **kw) -> torch.Tensor:
    """One-shot convenience wrapper (tests): out[N] of a batch-1 decode op with optional fused glue."""
    out = kw.pop("out", None)
    if out is None:
        M = int(kw.get("M", 1))
        n_out = N // 2 if kw.get("out_glue", OUT_NONE) == OUT_SILU_MUL_PAIRED else N
        out = torch.empty(n_out if M == 1 else (M, n_out), dtype=torch.float32 if kw.get("out_glue") == OUT_PARTIAL_F32 else x.dtype,
                          device=x.device)
    with torch.cuda.device(x.device):
        op = make_decode_op(x, qweight_t, meta, bias, out, K, N, group_size, bits, scale_dtype, **kw)
        launch_decode_op(op, x.device)
    return out


def dequant(qweight, qzeros, scales, g_idx, group_size: int, bits: int, out_dtype=None) -> torch.Tensor:
    """[K,N] weights from the CHECKPOINT layout (bit-exact with the reference's dequantize_weight())."""
    lib = _lib.load()
    _require_cuda(qweight, qzeros, scales, g_idx)
    pf = 32 // bits
    K, N = qweight.shape[0] * pf, qweight.shape[1]
    out_dtype = out_dtype or scales.dtype
    out = torch.empty((K, N), dtype=out_dtype, device=qweight.device)
    with torch.cuda.device(qweight.device):
        rc = lib.gptqhip_dequant(_ptr(qweight), _ptr(qzeros), _ptr(scales), _ptr(g_idx), _ptr(out), K, N, group_size,
                                 bits, _DT[scales.dtype], _DT[out_dtype], _stream(qweight.device))
    _lib.check(rc, "gptqhip_dequant")
    return out


def dequant_tiled(qweight_t, meta, perm, K: int, N: int, group_size: int, bits: int, scale_dtype, out_dtype=None):
    """[K,N] weights from the TILED layout; rows come back in checkpoint order when perm is given."""
    lib = _lib.load()
    _require_cuda(qweight_t, meta, perm)
    out_dtype = out_dtype or scale_dtype
    out = torch.empty((K, N), dtype=out_dtype, device=qweight_t.device)
    with torch.cuda.device(qweight_t.device):
        rc = lib.gptqhip_dequant_tiled(_ptr(qweight_t), _ptr(meta), _ptr(perm), _ptr(out), K, N, group_size, bits,
                                       _DT[scale_dtype], _DT[out_dtype], _stream(qweight_t.device))
    _lib.check(rc, "gptqhip_dequant_tiled")
    return out


PLANAR_ONLY_BITS = (5, 6, 7)


def kernel_bits(bits: int) -> int:
    """Field width of the layout the kernels read for a checkpoint of `bits`-bit codes (gptqhip_widen_codes)."""
    return 4 if bits <= 4 else 8


def widen_codes(qweight: torch.Tensor, qzeros: torch.Tensor, bits: int, planar: Optional[bool] = None):
    """(qweight, qzeros) of a 2 / 3 / 5 / 6 / 7-bit checkpoint -> the same codes in the continuous 4- or 8-bit layout + that width."""
    lib = _lib.load()
    _require_cuda(qweight, qzeros)
    if planar is None:
        planar = bits in PLANAR_ONLY_BITS
    K, N, G = qweight.shape[0] * 32 // bits, qweight.shape[1], qzeros.shape[0]
    wide = kernel_bits(bits)
    qw = torch.empty((K * wide // 32, N), dtype=torch.int32, device=qweight.device)
    qz = torch.empty((G, N * wide // 32), dtype=torch.int32, device=qweight.device)
    with torch.cuda.device(qweight.device):
        rc = lib.gptqhip_widen_codes(_ptr(qweight.contiguous()), _ptr(qzeros.contiguous()), _ptr(qw), _ptr(qz), K, N, G, bits,
                                     1 if planar else 0, _stream(qweight.device))
    _lib.check(rc, "gptqhip_widen_codes")
    return qw, qz, wide


def repack_awq(qweight_awq: torch.Tensor, qzeros_awq: torch.Tensor):
    lib = _lib.load()
    _require_cuda(qweight_awq, qzeros_awq)
    K, N = qweight_awq.shape[0], qweight_awq.shape[1] * 8
    G = qzeros_awq.shape[0]
    qw = torch.empty((K // 8, N), dtype=torch.int32, device=qweight_awq.device)
    qz = torch.empty_like(qzeros_awq)
    with torch.cuda.device(qweight_awq.device):
        rc = lib.gptqhip_repack_awq(_ptr(qweight_awq.contiguous()), _ptr(qzeros_awq.contiguous()), _ptr(qw), _ptr(qz),
                                    K, N, G, _stream(qweight_awq.device))
    _lib.check(rc, "gptqhip_repack_awq")
    return qw, qz


def embedding(ids: torch.Tensor, qweight_t: torch.Tensor, meta: torch.Tensor, inv_perm: Optional[torch.Tensor],
              K: int, N: int, group_size: int, bits: int, scale_dtype: torch.dtype) -> torch.Tensor:
    """Rows ids[...] of the dequantised [K,N] matrix (tiled layout) in the scales dtype; raises IndexError on ids
    outside [0, K) like torch.nn.functional.embedding."""
    lib = _lib.load()
    _require_cuda(ids, qweight_t, meta, inv_perm)
    flat = ids.reshape(-1).to(torch.int64).contiguous()
    T = flat.numel()
    out = torch.empty((T, N), dtype=scale_dtype, device=qweight_t.device)
    status = torch.zeros(1, dtype=torch.int32, device=qweight_t.device)
    with torch.cuda.device(qweight_t.device):
        for t0 in range(0, T, 65535):
            t1 = min(T, t0 + 65535)
            rc = lib.gptqhip_embedding(_ptr(flat[t0:t1]), _ptr(qweight_t), _ptr(meta), _ptr(inv_perm), _ptr(out[t0:t1]),
                                       _ptr(status), t1 - t0, K, N, group_size, bits, _DT[scale_dtype],
                                       _stream(qweight_t.device))
            _lib.check(rc, "gptqhip_embedding")
    if int(status.item()) != 0:
        raise IndexError("index out of range in quantised embedding lookup")
    return out.reshape(tuple(ids.shape) + (N,))


def pack_gptq(weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, g_idx: torch.Tensor, bits: int, planar: Optional[bool] = None):
    """Device quantise-and-pack: weight [N,K], scales [G,N], zeros [G,N], g_idx [K] -> (qweight, qzeros) in the
    checkpoint layout, bit-exact with the reference's pack_block.  bits 2..8; planar: split-plane words (default: the bit
    width's own layout -- 5 / 6 / 7 bits are planar, 3 bits only under FORMAT.GPTQ_P)."""
    lib = _lib.load()
    _require_cuda(weight, scales, zeros, g_idx)
    N, K = weight.shape
    G = scales.shape[0]
    if planar is None:
        planar = bits in PLANAR_ONLY_BITS
    w = weight.to(torch.float32).contiguous()
    s = scales.to(torch.float32).contiguous()
    z = zeros.to(torch.int32).contiguous()
    gi = g_idx.to(torch.int32).contiguous()
    if gi.numel() != K:
        raise ValueError(f"g_idx length {gi.numel()} != in_features {K}")
    gn = torch.where(gi < 0, gi + G, gi)
    if gn.numel() and (int(gn.min()) < 0 or int(gn.max()) >= G):
        raise IndexError(f"pack_gptq: g_idx values out of range (groups={G})")
    qweight = torch.empty((K * bits // 32, N), dtype=torch.int32, device=weight.device)
    qzeros = torch.empty((G, N * bits // 32), dtype=torch.int32, device=weight.device)
    with torch.cuda.device(weight.device):
        rc = lib.gptqhip_pack_gptq(_ptr(w), _ptr(s), _ptr(z), _ptr(gi), _ptr(qweight), _ptr(qzeros), K, N, G, bits,
                                   1 if planar else 0, _stream(weight.device))
    _lib.check(rc, "gptqhip_pack_gptq")
    return qweight, qzeros


def pack_gptq_host(weight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, g_idx: torch.Tensor, bits: int,
                   threads: int = 0, planar: Optional[bool] = None):
    """Host (CPU tensors) quantise-and-pack through the library's threaded C++ packer -- the equivalent of the
    reference's native pack_block_cpu.  Same contract as pack_gptq; needs no GPU."""
    lib = _lib.load()
    for t in (weight, scales, zeros, g_idx):
        if t.is_cuda:
            raise RuntimeError("pack_gptq_host takes CPU tensors (use pack_gptq on the device)")
    N, K = weight.shape
    G = scales.shape[0]
    if planar is None:
        planar = bits in PLANAR_ONLY_BITS
    w = weight.to(torch.float32).contiguous()
    s = scales.to(torch.float32).contiguous()
    z = zeros.to(torch.int32).contiguous()
    gi = g_idx.to(torch.int32).contiguous()
    if gi.numel() != K:
        raise ValueError(f"g_idx length {gi.numel()} != in_features {K}")
    qweight = torch.empty((K * bits // 32, N), dtype=torch.int32)
    qzeros = torch.empty((G, N * bits // 32), dtype=torch.int32)
    rc = lib.gptqhip_pack_gptq_host(_ptr(w), _ptr(s), _ptr(z), _ptr(gi), _ptr(qweight), _ptr(qzeros), K, N, G, bits,
                                    1 if planar else 0, threads)
    _lib.check(rc, "gptqhip_pack_gptq_host")
    return qweight, qzeros


def gather_cols(x: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _require_cuda(x, perm)
    M, K = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.gptqhip_gather_cols(_ptr(x), _ptr(perm), _ptr(out), M, K, _stream(x.device))
    _lib.check(rc, "gptqhip_gather_cols")
    return out


def rmsnorm_gather(h: torch.Tensor, weight: torch.Tensor, eps: float, perm: Optional[torch.Tensor] = None,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """HF LlamaRMSNorm over the last dimension fused with an act-order gather: out[.., k'] = weight[p] * act(h32[.., p] * rsqrt(mean
    h32^2 + eps)), p = perm[k'] (perm None: plain RMSNorm).  Feed the result to HipGptqLinear.forward_pregathered of the q|k|v or
    gate|up module whose `perm` this is: the linear then needs no gather pass of its own."""
    lib = _lib.load()
    _require_cuda(h, weight)
    if h.dtype not in _DT or weight.dtype != h.dtype or not h.is_contiguous() or not weight.is_contiguous():
        raise RuntimeError("rmsnorm_gather: h and weight must be contiguous fp16 / bf16 tensors of the same dtype")
    K = h.shape[-1]
    M = h.numel() // K
    if weight.numel() != K or (perm is not None and (perm.dtype != torch.int32 or perm.numel() != K or not perm.is_contiguous())):
        raise RuntimeError("rmsnorm_gather: weight / perm must have K elements (perm int32)")
    if out is None:
        out = torch.empty_like(h)
    elif out.shape != h.shape or out.dtype != h.dtype or not out.is_contiguous() or out.data_ptr() == h.data_ptr():
        raise RuntimeError("rmsnorm_gather: out must be a distinct contiguous tensor like h")
    with torch.cuda.device(h.device):
        rc = lib.gptqhip_rmsnorm_gather(_ptr(h), _ptr(weight), _ptr(perm), _ptr(out), M, K, float(eps), _DT[h.dtype], _stream(h.device))
    _lib.check(rc, "gptqhip_rmsnorm_gather")
    return out


def plan_describe(M: int, K: int, N: int, group_size: int, bits: int = 4, has_perm: bool = False) -> str:
    """Kernel family + launch geometry gptqhip_gemm would pick for this call (host logic; needs no GPU)."""
    import ctypes
    buf = ctypes.create_string_buffer(256)
    _lib.check(_lib.load().gptqhip_plan_describe(M, K, N, group_size, bits, 1 if has_perm else 0, buf, 256), "gptqhip_plan_describe")
    return buf.value.decode()


def set_tuning(force_split_k: int = 0, force_kernel: int = 0, force_waves: int = 0) -> None:
    _need_cache.clear()  # forced plans change the workspace layout
    _lib.check(_lib.load().gptqhip_set_tuning(force_split_k, force_kernel, force_waves), "gptqhip_set_tuning")


def set_decode_form(form: int = -1) -> None:
    """Batch-1 decode form for the calling thread (include/gptqhip.h gptqhip_set_decode_form documents all six): 5 = preload kernel + raw codes
    as fp16 denormals (the default for fp16 and bf16 activations), 3 = preload + group-factored dequant, 4 = preload + the reference's per-weight
    rounding (bit-faithful; the default for fp16 activations with bf16 scales), 0 / 2 = the rounds 1-5 kernel (bit-faithful / group-factored), 1 = the LDS-DMA stream kernel (opt-in, slower),
    -1 = the process default (GPTQHIP_DECODE_BITFAITHFUL=1 in the environment makes that 4 for every dtype)."""
    _lib.check(_lib.load().gptqhip_set_decode_form(form), "gptqhip_set_decode_form")
