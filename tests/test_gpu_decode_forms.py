"""GPU tests of the batch-1 decode FORMS (include/gptqhip.h gptqhip_set_decode_form; round 6):
  0  skinny_kernel, the reference's per-weight rounding            (bit-faithful)
  4  skinny1_kernel (preload), the reference's per-weight rounding (bit-faithful; the bf16 default)
  3  skinny1_kernel (preload) + group-factored dequant             (the fp16 default: exact (q - z), scale per chunk in fp32)
  2  skinny_kernel + group-factored dequant
  1  decode_stream_kernel (LDS-DMA weight ring, offsets / zero-points taken out per chunk)   -- opt-in, measured slower
  5  skinny1_kernel (preload) + RAW codes: the 4-bit codes enter the MFMA as fp16 denormals (one AND per two weights), zero-points through the
     chunk's activation sum
Every form against the numpy oracle (TorchLinear semantics, oracle/gptq_oracle.py) on the decode ops a Llama layer runs, with the glue the
decode chain uses; the bit-faithful forms under the UNSCALED element-wise gate of the reference's own test."""
import numpy as np
import pytest
import torch

from helpers import NORM_TOL_BF16_EXACT, assert_forward_close, f32_to_torch, synth_gptq, torch_to_f32
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FORMS = {0: True, 4: True, 3: False, 2: False, 1: False, 5: False}     # form -> strict element-wise atol


@pytest.fixture(scope="module")
def ops():
    from gptqmodel_amd import ops as _ops
    yield _ops
    _ops.set_decode_form(-1)


def _tiled(ops, qweight, qzeros, scales, gs, sdt="fp16"):
    sc = f32_to_torch(scales, sdt, DEV)
    return ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, None, gs, 4) + (sc,)


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("K,N,gs", [(4096, 4096, 128), (4096, 6144, 128), (14336, 4096, 128), (4096, 1024, 128), (8192, 1024, 256),
                                    (11008, 4096, 128), (4096, 512, 4096)])
@pytest.mark.parametrize("act,sdt", [("fp16", "fp16"), ("bf16", "bf16"), ("fp16", "bf16")])
def test_plain_op_every_form_vs_oracle(ops, form, K, N, gs, act, sdt):
    if act == "bf16" and form in (1, 2, 3):
        pytest.skip("forms 1 / 2 / 3 exist for fp16 activations only (bf16: form 4 = bit-faithful default, form 5 = exact arithmetic through the f16 matrix pipe)")
    nt = NORM_TOL_BF16_EXACT if (act == "bf16" and form == 5) else None
    qweight, qzeros, scales, g_idx = synth_gptq(500 + K // 128 + N // 16 + gs, 4, K, N, gs, scale_dtype=sdt)
    rng = np.random.RandomState(5)
    x = O.round_to(rng.randn(1, K).astype(np.float32) * 0.5, act)
    bias = O.round_to(rng.randn(N).astype(np.float32) * 0.1, act)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs, sdt)
    ref = O.forward_gptq(x, qweight, qzeros, scales, g_idx, 4, bias, act, sdt)
    ops.set_decode_form(form)
    try:
        out = ops.decode_linear(f32_to_torch(x[0], act, DEV), qw_t, meta, f32_to_torch(bias, act, DEV), K, N, gs, 4, sc.dtype)
        gen = ops.gemm(f32_to_torch(x, act, DEV), qw_t, meta, f32_to_torch(bias, act, DEV), None, N, gs, 4, sc.dtype)
        torch.cuda.synchronize()
    finally:
        ops.set_decode_form(-1)
    assert_forward_close(torch_to_f32(out)[None], ref, act, tag=("decode op", form), strict_atol=FORMS[form], norm_tol=nt)
    assert_forward_close(torch_to_f32(gen), ref, act, tag=("gemm M=1", form), strict_atol=FORMS[form], norm_tol=nt)
    assert torch.equal(out, gen[0]), "the plugin path (gptqhip_gemm at M = 1) and the decode op must run the same form"


@pytest.mark.parametrize("form", sorted(FORMS))
def test_bit_faithful_forms_agree_bit_for_bit(ops, form):
    """Forms 0 and 4 share the rounding chain and the summation order of a tile's K range: identical bits on a plain op."""
    if not FORMS[form]:
        pytest.skip("exact-arithmetic form")
    K, N, gs = 4096, 4096, 128
    qweight, qzeros, scales, g_idx = synth_gptq(77, 4, K, N, gs)
    x = f32_to_torch(O.round_to(np.random.RandomState(6).randn(K).astype(np.float32), "fp16"), "fp16", DEV)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs)
    outs = {}
    try:
        for f in (0, form):
            ops.set_decode_form(f)
            outs[f] = ops.decode_linear(x, qw_t, meta, None, K, N, gs, 4, sc.dtype).clone()
    finally:
        ops.set_decode_form(-1)
    torch.cuda.synchronize()
    if form == 4:   # (16 x 2 chunks in form 0 vs 8 x 4 in the preload form on this shape: same chunks, different wave grouping)
        assert_forward_close(torch_to_f32(outs[4])[None], torch_to_f32(outs[0])[None], "fp16", strict_atol=True)
    else:
        assert torch.equal(outs[0], outs[form])


@pytest.mark.parametrize("form", sorted(FORMS))
@pytest.mark.parametrize("act", ["fp16", "bf16"])
@pytest.mark.parametrize("inter", [1024, 8192])      # 8192: 1024 column tiles = four per CU -> skinny1p_kernel
def test_layer_ops_with_glue_every_form(ops, form, act, inter):
    """The four ops of a decoder layer the way the chain runs them (RMSNorm from producer statistics, residual + stats_out, paired
    SiLU*mul epilogue), reference-scale activations, every form against the oracle's composition of the same steps."""
    if act == "bf16" and form in (1, 2, 3):
        pytest.skip("fp16-only forms")
    nt = NORM_TOL_BF16_EXACT if (act == "bf16" and form == 5) else None
    gs, hidden = 128, 4096
    rng = np.random.RandomState(31)
    h = O.round_to(rng.randn(hidden).astype(np.float32) * 0.5, act)
    w = O.round_to(1.0 + rng.randn(hidden).astype(np.float32) * 0.1, act)
    a_in = O.round_to(rng.randn(hidden).astype(np.float32) * 0.3, act)
    strict = FORMS[form]
    ops.set_decode_form(form)
    try:
        # o_proj-like: h1 = h + a @ Wo, statistics out
        qw_o, qz_o, sc_o, gi_o = synth_gptq(43, 4, hidden, hidden, gs)
        qwo_t, meta_o, sco = _tiled(ops, qw_o, qz_o, sc_o, gs)
        stats = torch.zeros(hidden // 16, dtype=torch.float32, device=DEV)
        h1 = ops.decode_linear(f32_to_torch(a_in, act, DEV), qwo_t, meta_o, None, hidden, hidden, gs, 4, sco.dtype,
                               residual=f32_to_torch(h, act, DEV), stats_out=stats)
        h1_np = torch_to_f32(h1)
        y_o = O.forward_gptq(a_in[None], qw_o, qz_o, sc_o, gi_o, 4, None, act, "fp16")
        assert_forward_close(h1_np[None], O.residual_add_ref(h[None], y_o, act), act, tag=("residual+stats", form), strict_atol=strict, norm_tol=nt)
        assert np.allclose(stats.cpu().numpy(), (h1_np.astype(np.float64) ** 2).reshape(-1, 16).sum(axis=1), rtol=1e-5)
        # gate_up-like: a = silu(g) * u, g|u = rmsnorm(h1) @ Wgu, columns interleaved in blocks of 8; statistics in (and, separately, none)
        qweight, qzeros, scales, g_idx = synth_gptq(41, 4, hidden, 2 * inter, gs)
        order = np.stack([np.arange(inter).reshape(-1, 8), (inter + np.arange(inter)).reshape(-1, 8)], axis=1).reshape(-1)
        qw_i, sc_i = np.ascontiguousarray(qweight[:, order]), np.ascontiguousarray(scales[:, order])
        qz_i = O.pack_cols(O.unpack_cols(qzeros, 4)[:, order], 4)
        qwi_t, meta_i, sci = _tiled(ops, qw_i, qz_i, sc_i, gs)
        xn1 = O.rmsnorm_ref(h1_np, w, 1e-5, act)
        gu1 = O.forward_gptq(xn1[None], qweight, qzeros, scales, g_idx, 4, None, act, "fp16")[0]
        a_ref = O.silu_mul_ref(gu1[:inter], gu1[inter:], act)
        for st_in in (stats, None):
            a_dev = torch.zeros(2 * inter, dtype=h1.dtype, device=DEV)
            ops.decode_linear(h1, qwi_t, meta_i, None, hidden, 2 * inter, gs, 4, sci.dtype, out=a_dev, in_glue=ops.GLUE_RMSNORM,
                              norm_weight=f32_to_torch(w, act, DEV), eps=1e-5, out_glue=ops.OUT_SILU_MUL_PAIRED, stats_in=st_in)
            assert_forward_close(torch_to_f32(a_dev)[None, :inter], a_ref[None], act, tag=("rmsnorm + paired silu", form, st_in is not None),
                                 strict_atol=strict, norm_tol=nt)
        torch.cuda.synchronize()
    finally:
        ops.set_decode_form(-1)


@pytest.mark.parametrize("K,N", [(4096, 4096), (14336, 4096), (4096, 6144)])
def test_group_factored_default_is_no_further_from_exact_arithmetic_than_the_reference(ops, K, N):
    """The fp16 default (form 3) leaves the reference's per-weight rounding: y = sum_g s_g * sum_k x_k (q_k - z_g) with the integers exact and the
    scale applied once per group in fp32.  That is the exact value of the reference's expression up to fp32 accumulation; the reference's own output
    differs from it by its per-weight fp16 rounding (2^-12 relative per weight).  Pinned here against float64 arithmetic on the integer codes:
    the default's distance from the exact result is no larger than the reference chain's (oracle = TorchLinear semantics), on BASELINE's shapes."""
    gs = 128
    qweight, qzeros, scales, g_idx = synth_gptq(77 + K // 128 + N // 16, 4, K, N, gs, scale_dtype="fp16")
    rng = np.random.RandomState(11)
    x = O.round_to(rng.randn(1, K).astype(np.float32) * 0.5, "fp16")
    codes = O.unpack_rows(qweight, 4).astype(np.int64)
    zeros = O.unpack_cols(qzeros, 4).astype(np.int64)
    g = O.normalize_g_idx(g_idx, scales.shape[0])
    w_exact = np.asarray(scales, np.float64)[g] * (codes - zeros[g]).astype(np.float64)
    y_exact = x.astype(np.float64) @ w_exact                                   # no rounding anywhere
    y_ref = O.forward_gptq(x, qweight, qzeros, scales, g_idx, 4, None, "fp16", "fp16").astype(np.float64)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs, "fp16")
    outs = {}
    for form in (3, 4, 5):
        ops.set_decode_form(form)
        try:
            outs[form] = torch_to_f32(ops.decode_linear(f32_to_torch(x[0], "fp16", DEV), qw_t, meta, None, K, N, gs, 4, sc.dtype)).astype(np.float64)[None]
            torch.cuda.synchronize()
        finally:
            ops.set_decode_form(-1)
    scale = np.abs(y_exact).max()
    err_ref = np.abs(y_ref - y_exact).max() / scale
    err_f3 = np.abs(outs[3] - y_exact).max() / scale
    err_f4 = np.abs(outs[4] - y_exact).max() / scale
    rms = lambda a: float(np.sqrt(np.mean((a - y_exact) ** 2)) / scale)   # noqa: E731
    # both are dominated by the final rounding of y to fp16 (2^-11 relative); the default must not be worse than the reference chain
    assert err_f3 <= err_ref * 1.05 + 1e-6, (err_f3, err_ref)
    assert rms(outs[3]) <= rms(y_ref) * 1.02 + 1e-7, (rms(outs[3]), rms(y_ref))
    assert err_f4 <= err_ref * 1.5 + 1e-6, (err_f4, err_ref)                    # bit-faithful form: the reference's own noise level
    assert err_f3 <= 1e-3 and err_ref <= 1e-3
    err_f5 = np.abs(outs[5] - y_exact).max() / scale
    assert err_f5 <= err_ref * 1.05 + 1e-6 and rms(outs[5]) <= rms(y_ref) * 1.02 + 1e-7, (err_f5, err_ref, rms(outs[5]), rms(y_ref))


@pytest.mark.parametrize("form", [3, 4, 5])
@pytest.mark.parametrize("glue", ["rmsnorm", "none"])
@pytest.mark.parametrize("N", [8192, 16384, 17600])
def test_persistent_tile_variant_every_preload_form(ops, form, glue, N):
    """skinny1p_kernel (layers with >= 2 column tiles per CU, no bias / residual: the fused gate_up) -- 8192 columns = 512 tiles = two per block;
    16384 columns = 1024 tiles = four per block; 17600 columns = 1100 tiles, not a multiple of the CU count: one tile per block (skinny1_kernel)."""
    K, gs, act = 4096, 128, "fp16"
    qweight, qzeros, scales, g_idx = synth_gptq(91, 4, K, N, gs)
    rng = np.random.RandomState(17)
    h = O.round_to(rng.randn(K).astype(np.float32) * 0.5, act)
    w = O.round_to(1.0 + rng.randn(K).astype(np.float32) * 0.1, act)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs)
    x_ref = O.rmsnorm_ref(h, w, 1e-5, act) if glue == "rmsnorm" else h
    ref = O.forward_gptq(x_ref[None], qweight, qzeros, scales, g_idx, 4, None, act, "fp16")
    ops.set_decode_form(form)
    try:
        kw = dict(in_glue=ops.GLUE_RMSNORM, norm_weight=f32_to_torch(w, act, DEV), eps=1e-5) if glue == "rmsnorm" else {}
        out = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, None, K, N, gs, 4, sc.dtype, **kw)
        torch.cuda.synchronize()
    finally:
        ops.set_decode_form(-1)
    assert_forward_close(torch_to_f32(out)[None], ref, act, tag=("persistent tiles", form, glue), strict_atol=FORMS[form])


@pytest.mark.parametrize("case", ["positive_large", "tiny", "zero_point_extremes", "one_hot"])
def test_raw_code_form_adversarial_inputs(ops, case):
    """Form 5 takes the zero-points out through the chunk's activation sum (y = s (sum x q - z sum x)) and feeds the codes as fp16 denormals:
    inputs built to stress exactly that -- same-sign activations of large magnitude (sum x q and z sum x nearly cancel), activations near
    the fp16 denormal range, all zero-points at 0 / 15, a single non-zero activation -- against float64 arithmetic and the reference chain."""
    K, N, gs = 4096, 4096, 128
    qweight, qzeros, scales, g_idx = synth_gptq(313, 4, K, N, gs, scale_dtype="fp16")
    rng = np.random.RandomState(23)
    if case == "positive_large":
        x = np.abs(rng.randn(1, K).astype(np.float32)) * 40.0 + 20.0
    elif case == "tiny":
        x = rng.randn(1, K).astype(np.float32) * 3e-4
    elif case == "one_hot":
        x = np.zeros((1, K), np.float32)
        x[0, 1234] = 3.0
    else:
        x = rng.randn(1, K).astype(np.float32) * 0.5
        z = np.where(rng.rand(K // gs, N) < 0.5, 0, 15).astype(np.int32)
        qzeros = O.pack_cols(z, 4)
    x = O.round_to(x, "fp16")
    codes = O.unpack_rows(qweight, 4).astype(np.int64)
    zeros = O.unpack_cols(qzeros, 4).astype(np.int64)
    g = O.normalize_g_idx(g_idx, scales.shape[0])
    y_exact = x.astype(np.float64) @ (np.asarray(scales, np.float64)[g] * (codes - zeros[g]).astype(np.float64))
    y_ref = O.forward_gptq(x, qweight, qzeros, scales, g_idx, 4, None, "fp16", "fp16").astype(np.float64)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs, "fp16")
    ops.set_decode_form(5)
    try:
        out = torch_to_f32(ops.decode_linear(f32_to_torch(x[0], "fp16", DEV), qw_t, meta, None, K, N, gs, 4, sc.dtype)).astype(np.float64)[None]
        torch.cuda.synchronize()
    finally:
        ops.set_decode_form(-1)
    assert np.isfinite(out).all()
    scale = np.abs(y_exact).max()
    err, err_ref = np.abs(out - y_exact).max() / scale, np.abs(y_ref - y_exact).max() / scale
    assert err <= max(err_ref * 1.05, 2.0 ** -11) + 1e-7, (case, err, err_ref)       # no further from exact arithmetic than the reference (or one fp16 rounding)
    assert np.abs(out - y_ref).max() / np.abs(y_ref).max() <= 1e-3, case              # north_star's bar against the reference itself


@pytest.mark.parametrize("K,N,sdt", [(4096, 4096, "bf16"), (14336, 4096, "bf16"), (4096, 6144, "fp16")])
@pytest.mark.parametrize("xscale", [0.5, 3000.0, 1e-3])
def test_bf16_raw_code_form_vs_exact_arithmetic(ops, K, N, sdt, xscale):
    """Form 5 on bf16 activations: the wave converts its bf16 x pieces to fp16 exactly (divided by a power of two taken from its largest |x|) and
    runs the raw-code path of the f16 matrix pipe; the output is the bf16 rounding of the exact sum.  Against float64 arithmetic on the integer
    codes it is no further away than the reference chain (TorchLinear: every weight rounded to bf16 first), at ordinary, large (beyond fp16's
    range without the power of two) and tiny activation scales; against the reference itself it stays inside the reference's own element-wise gates."""
    gs = 128
    qweight, qzeros, scales, g_idx = synth_gptq(611 + K // 128 + N // 16, 4, K, N, gs, scale_dtype=sdt)
    rng = np.random.RandomState(29)
    x = O.round_to(rng.randn(1, K).astype(np.float32) * xscale, "bf16")
    if xscale > 100:
        x[0, 7] = float(O.round_to(np.array([2.0e5], np.float32), "bf16")[0])          # one outlier far beyond fp16's 65504
    codes = O.unpack_rows(qweight, 4).astype(np.int64)
    zeros = O.unpack_cols(qzeros, 4).astype(np.int64)
    g = O.normalize_g_idx(g_idx, scales.shape[0])
    y_exact = x.astype(np.float64) @ (np.asarray(scales, np.float64)[g] * (codes - zeros[g]).astype(np.float64))
    y_ref = O.forward_gptq(x, qweight, qzeros, scales, g_idx, 4, None, "bf16", sdt)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs, sdt)
    ops.set_decode_form(5)
    try:
        out = torch_to_f32(ops.decode_linear(f32_to_torch(x[0], "bf16", DEV), qw_t, meta, None, K, N, gs, 4, sc.dtype))[None]
        torch.cuda.synchronize()
    finally:
        ops.set_decode_form(-1)
    assert np.isfinite(out).all()
    scale = np.abs(y_exact).max()
    err, err_ref = np.abs(out - y_exact).max() / scale, np.abs(y_ref.astype(np.float64) - y_exact).max() / scale
    assert err <= max(err_ref * 1.05, 2.0 ** -8) + 1e-7, (err, err_ref)      # one bf16 rounding of the exact sum, never worse than the reference chain
    assert_forward_close(out, y_ref, "bf16", tag=("bf16 form 5", K, N, sdt, xscale), norm_tol=NORM_TOL_BF16_EXACT)


@pytest.mark.parametrize("form", [4, 5])
@pytest.mark.parametrize("act", ["fp16", "bf16"])
@pytest.mark.parametrize("glue", ["rmsnorm+stats", "none"])
def test_persistent_tile_variant_with_act_order(ops, form, act, glue):
    """desc_act=True checkpoints on a many-tile layer (8192 columns = two tiles per CU): skinny1p_kernel stages the glued x row once per block and
    every wave gathers its chunks' elements by the permutation (RMSNorm with producer statistics, or no glue) -- against the oracle with g_idx."""
    K, N, gs = 4096, 8192, 128
    qweight, qzeros, scales, g_idx = synth_gptq(733, 4, K, N, gs, desc_act=True)
    perm = torch.from_numpy(np.argsort(g_idx, kind="stable").astype(np.int32)).to(DEV)
    sc = f32_to_torch(scales, "fp16", DEV)
    qw_t, meta = ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, perm, gs, 4)
    rng = np.random.RandomState(19)
    h = O.round_to(rng.randn(K).astype(np.float32) * 1.5, act)
    w = O.round_to(1.0 + rng.randn(K).astype(np.float32) * 0.1, act)
    kw, x_ref = {}, h
    if glue != "none":
        st_in = torch.from_numpy((h.astype(np.float64) ** 2).reshape(-1, 16).sum(axis=1).astype(np.float32)).to(DEV)
        kw = dict(in_glue=ops.GLUE_RMSNORM, norm_weight=f32_to_torch(w, act, DEV), eps=1e-5, stats_in=st_in)
        x_ref = O.rmsnorm_ref(h, w, 1e-5, act)
    ref = O.forward_gptq(x_ref[None], qweight, qzeros, scales, g_idx, 4, None, act, "fp16")
    ops.set_decode_form(form)
    try:
        out = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, None, K, N, gs, 4, sc.dtype, perm=perm, **kw)
        torch.cuda.synchronize()
    finally:
        ops.set_decode_form(-1)
    nt = NORM_TOL_BF16_EXACT if (act == "bf16" and form == 5) else None
    assert_forward_close(torch_to_f32(out)[None], ref, act, tag=("persistent + act-order", form, act, glue), strict_atol=FORMS[form], norm_tol=nt)
