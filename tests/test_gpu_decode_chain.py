"""GPU tests of the batch-1 decode op (gptqhip_decode_linear: the skinny kernel's M = 1 pipeline with fused decoder-layer
glue): the GEMV against the oracle, the fused glue against the HF-semantics restatement in the oracle, and a chain of
dependent ops (DecodeStep) against the same step run as separate launches (plugin forward() + torch glue)."""
import numpy as np
import pytest
import torch

from helpers import assert_forward_close, decode_norm_tol, f32_to_torch, rel_err, synth_gptq, torch_to_bits, torch_to_f32
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TDT = {"fp16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def ops():
    from gptqmodel_amd import ops as _ops
    return _ops


def _tiled(ops, qweight, qzeros, scales, gs, bits, sdt="fp16"):
    sc = f32_to_torch(scales, sdt, DEV)
    return ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, None, gs, bits) + (sc,)


@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (8192, 1024), (8192, 10240),
                                 (28672, 512), (4096, 1000), (4096, 1024)])
@pytest.mark.parametrize("act,bits", [("fp16", 4), ("bf16", 4), ("fp16", 8)])
def test_decode_op_plain_vs_oracle(ops, K, N, act, bits):
    gs = 128
    qweight, qzeros, scales, g_idx = synth_gptq(100 + K // 128 + N // 8, bits, K, N, gs)
    rng = np.random.RandomState(3)
    x = O.round_to(rng.randn(1, K).astype(np.float32) * 0.5, act)
    bias = O.round_to(rng.randn(N).astype(np.float32) * 0.1, act)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs, bits)
    assert ops.decode_supported(K, N, gs)
    out = ops.decode_linear(f32_to_torch(x[0], act, DEV), qw_t, meta, f32_to_torch(bias, act, DEV), K, N, gs, bits, sc.dtype)
    torch.cuda.synchronize()
    ref = O.forward_gptq(x, qweight, qzeros, scales, g_idx, bits, bias, act, "fp16")
    assert_forward_close(torch_to_f32(out)[None], ref, act, norm_tol=decode_norm_tol(act, bits))
    # and against the general kernel of the product path at M = 1 (same rounding chain, different accumulation order)
    gen = ops.gemm(f32_to_torch(x, act, DEV), qw_t, meta, f32_to_torch(bias, act, DEV), None, N, gs, bits, sc.dtype)
    assert_forward_close(torch_to_f32(out)[None], torch_to_f32(gen), act, norm_tol=decode_norm_tol(act, bits))


@pytest.mark.parametrize("act", ["fp16", "bf16"])
def test_decode_op_fused_glue_vs_hf_semantics(ops, act):
    """RMSNorm prologue, SiLU*mul prologue and residual epilogue against the numpy restatement of HF's LlamaRMSNorm /
    LlamaMLP / residual adds composed with the oracle's GEMV."""
    gs, hidden, inter = 128, 4096, 14336
    rng = np.random.RandomState(11)
    # (a) gate_up = rmsnorm(h; w) @ Wgu
    qweight, qzeros, scales, g_idx = synth_gptq(41, 4, hidden, 2 * inter, gs)
    h = O.round_to(rng.randn(hidden).astype(np.float32) * 2.0, act)
    w = O.round_to(1.0 + rng.randn(hidden).astype(np.float32) * 0.1, act)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs, 4)
    gu = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, None, hidden, 2 * inter, gs, 4, sc.dtype,
                           in_glue=ops.GLUE_RMSNORM, norm_weight=f32_to_torch(w, act, DEV), eps=1e-5)
    xn = O.rmsnorm_ref(h, w, 1e-5, act)
    gu_ref = O.forward_gptq(xn[None], qweight, qzeros, scales, g_idx, 4, None, act, "fp16")
    assert_forward_close(torch_to_f32(gu)[None], gu_ref, act, tag="rmsnorm", norm_tol=decode_norm_tol(act))
    # (b) h2 = h + (silu(gate) * up) @ Wdown   -- fed with the DEVICE's gate|up so only this op is under test
    qweight2, qzeros2, scales2, g_idx2 = synth_gptq(42, 4, inter, hidden, gs)
    qw2, meta2, sc2 = _tiled(ops, qweight2, qzeros2, scales2, gs, 4)
    bias = O.round_to(rng.randn(hidden).astype(np.float32) * 0.1, act)
    h2 = ops.decode_linear(gu, qw2, meta2, f32_to_torch(bias, act, DEV), inter, hidden, gs, 4, sc2.dtype,
                           in_glue=ops.GLUE_SILU_MUL, residual=f32_to_torch(h, act, DEV))
    gu_np = torch_to_f32(gu)
    a = O.silu_mul_ref(gu_np[:inter], gu_np[inter:], act)
    y = O.forward_gptq(a[None], qweight2, qzeros2, scales2, g_idx2, 4, bias, act, "fp16")
    h2_ref = O.residual_add_ref(h[None], y, act)
    assert_forward_close(torch_to_f32(h2)[None], h2_ref, act, tag="silu_mul+residual", norm_tol=decode_norm_tol(act))
    # (c) the same two ops the way the decode chain runs them: the producer of h hands over the per-tile sums of h^2
    #     (stats_out -> stats_in), gate|up columns interleaved in blocks of 8 with the SiLU*mul in the producer's epilogue
    qw_o, qz_o, sc_o, gi_o = synth_gptq(43, 4, hidden, hidden, gs)
    qwo_t, meta_o, sco = _tiled(ops, qw_o, qz_o, sc_o, gs, 4)
    a_in = O.round_to(rng.randn(hidden).astype(np.float32) * 0.5, act)
    stats = torch.zeros(hidden // 16, dtype=torch.float32, device=DEV)
    h1 = ops.decode_linear(f32_to_torch(a_in, act, DEV), qwo_t, meta_o, None, hidden, hidden, gs, 4, sco.dtype,
                           residual=f32_to_torch(h, act, DEV), stats_out=stats)
    h1_np = torch_to_f32(h1)
    y_o = O.forward_gptq(a_in[None], qw_o, qz_o, sc_o, gi_o, 4, None, act, "fp16")
    assert_forward_close(h1_np[None], O.residual_add_ref(h[None], y_o, act), act, tag="residual+stats", norm_tol=decode_norm_tol(act))
    want_stats = (h1_np.astype(np.float64) ** 2).reshape(-1, 16).sum(axis=1)
    assert np.allclose(stats.cpu().numpy(), want_stats, rtol=1e-5), "stats_out must be the per-tile sums of out^2"
    gate_cols, up_cols = np.arange(inter), inter + np.arange(inter)
    order = np.stack([gate_cols.reshape(-1, 8), up_cols.reshape(-1, 8)], axis=1).reshape(-1)       # g0..7 u0..7 g8..15 ...
    qw_i = np.ascontiguousarray(qweight[:, order])
    sc_i = np.ascontiguousarray(scales[:, order])
    zz = O.unpack_cols(qzeros, 4)[:, order]
    qz_i = O.pack_cols(zz, 4)
    qwi_t, meta_i, sci = _tiled(ops, qw_i, qz_i, sc_i, gs, 4)
    a_dev = torch.zeros(2 * inter, dtype=TDT[act], device=DEV)
    ops.decode_linear(h1, qwi_t, meta_i, None, hidden, 2 * inter, gs, 4, sci.dtype, out=a_dev, in_glue=ops.GLUE_RMSNORM,
                      norm_weight=f32_to_torch(w, act, DEV), eps=1e-5, out_glue=ops.OUT_SILU_MUL_PAIRED, stats_in=stats)
    xn1 = O.rmsnorm_ref(h1_np, w, 1e-5, act)
    gu1 = O.forward_gptq(xn1[None], qweight, qzeros, scales, g_idx, 4, None, act, "fp16")[0]
    a_ref = O.silu_mul_ref(gu1[:inter], gu1[inter:], act)
    assert_forward_close(torch_to_f32(a_dev)[None, :inter], a_ref[None], act, tag="stats_in rmsnorm + paired silu*mul", norm_tol=decode_norm_tol(act))


@pytest.mark.parametrize("act", ["fp16", "bf16"])
def test_decode_op_act_order_in_kernel_perm_with_glue(ops, act):
    """desc_act=True checkpoints: the decode op applies the act-order permutation inside the kernel to the GLUED input
    (RMSNorm from producer statistics and from an in-block reduction; residual + stats_out epilogue) -- against the oracle's
    forward with the checkpoint's g_idx."""
    gs, bits = 128, 4
    rng = np.random.RandomState(21)
    for K, N, with_stats in ((4096, 6144, True), (4096, 1024, False), (14336, 4096, False), (8192, 10240, True)):
        qweight, qzeros, scales, g_idx = synth_gptq(300 + K // 128 + N // 16, bits, K, N, gs, desc_act=True)
        perm = torch.from_numpy(np.argsort(g_idx, kind="stable").astype(np.int32)).to(DEV)
        sc = f32_to_torch(scales, "fp16", DEV)
        qw_t, meta = ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, perm, gs, bits)
        h = O.round_to(rng.randn(K).astype(np.float32) * 1.5, act)
        w = O.round_to(1.0 + rng.randn(K).astype(np.float32) * 0.1, act)
        res = O.round_to(rng.randn(N).astype(np.float32), act)
        st_in = None
        if with_stats:
            st_in = torch.from_numpy((h.astype(np.float64) ** 2).reshape(-1, 16).sum(axis=1).astype(np.float32)).to(DEV)
        st_out = torch.zeros(-(-N // 16), dtype=torch.float32, device=DEV)
        out = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, None, K, N, gs, bits, sc.dtype, in_glue=ops.GLUE_RMSNORM,
                                norm_weight=f32_to_torch(w, act, DEV), eps=1e-5, residual=f32_to_torch(res, act, DEV),
                                stats_in=st_in, stats_out=st_out, perm=perm)
        xn = O.rmsnorm_ref(h, w, 1e-5, act)
        y = O.forward_gptq(xn[None], qweight, qzeros, scales, g_idx, bits, None, act, "fp16")
        ref = O.residual_add_ref(res[None], y, act)
        got = torch_to_f32(out)
        assert_forward_close(got[None], ref, act, tag=(K, N, with_stats), norm_tol=decode_norm_tol(act))
        assert np.allclose(st_out.cpu().numpy(), (got.astype(np.float64) ** 2).reshape(-1, 16).sum(axis=1), rtol=1e-5)
        # plain (no glue) with the permutation == the plugin path's batch-1 kernel
        plain = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, None, K, N, gs, bits, sc.dtype, perm=perm)
        gen = ops.gemm(f32_to_torch(h[None], act, DEV), qw_t, meta, None, perm, N, gs, bits, sc.dtype)
        assert torch.equal(plain, gen[0])
    with pytest.raises(RuntimeError, match="in-kernel act-order"):   # row too long for the LDS-resident variant
        K, N = 28672, 512
        qweight, qzeros, scales, g_idx = synth_gptq(9, bits, K, N, gs, desc_act=True)
        perm = torch.from_numpy(np.argsort(g_idx, kind="stable").astype(np.int32)).to(DEV)
        sc = f32_to_torch(scales, "fp16", DEV)
        qw_t, meta = ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, perm, gs, bits)
        ops.decode_linear(torch.zeros(K, dtype=TDT[act], device=DEV), qw_t, meta, None, K, N, gs, bits, sc.dtype, perm=perm)


_ROWS_SHAPES = ((4096, 6144, True, False), (4096, 4096, False, False), (4096, 512, True, False), (11008, 1024, False, False),
                (4096, 2048, True, True))


def _check_decode_rows(ops, M, act, gs, shapes=_ROWS_SHAPES):
    bits = 4
    rng = np.random.RandomState(31 + M)
    for K, N, with_stats, paired in shapes:
        qweight, qzeros, scales, g_idx = synth_gptq(700 + K // 128 + N // 16, bits, K, N, gs)
        if paired:   # interleave gate|up columns in blocks of 8 (what fuse_gate_up_interleaved stores)
            inter = N // 2
            order = np.stack([np.arange(inter).reshape(-1, 8), inter + np.arange(inter).reshape(-1, 8)], axis=1).reshape(-1)
        sc = f32_to_torch(scales, "fp16", DEV)
        if paired:
            qw_i = np.ascontiguousarray(qweight[:, order])
            qz_i = O.pack_cols(O.unpack_cols(qzeros, 4)[:, order], 4)
            sc_i = f32_to_torch(np.ascontiguousarray(scales[:, order]), "fp16", DEV)
            qw_t, meta = ops.repack_tiled(torch.from_numpy(qw_i).to(DEV), torch.from_numpy(qz_i).to(DEV), sc_i, None, gs, bits)
        else:
            qw_t, meta = ops.repack_tiled(torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV), sc, None, gs, bits)
        h = O.round_to(rng.randn(M, K).astype(np.float32) * (1.0 + np.arange(M)[:, None]), act)    # rows of different scale
        w = O.round_to(1.0 + rng.randn(K).astype(np.float32) * 0.1, act)
        res = O.round_to(rng.randn(M, N).astype(np.float32), act)
        bias = O.round_to(rng.randn(N).astype(np.float32) * 0.1, act)
        st_in = None
        if with_stats:
            st_in = torch.from_numpy((h.astype(np.float64) ** 2).reshape(M, -1, 16).sum(axis=2).astype(np.float32)).to(DEV)
        xn = np.stack([O.rmsnorm_ref(h[m], w, 1e-5, act) for m in range(M)])
        y = O.forward_gptq(xn, qweight, qzeros, scales, g_idx, bits, None if paired else bias, act, "fp16")
        if paired:
            out = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, None, K, N, gs, bits, sc.dtype, in_glue=ops.GLUE_RMSNORM,
                                    norm_weight=f32_to_torch(w, act, DEV), eps=1e-5, stats_in=st_in,
                                    out_glue=ops.OUT_SILU_MUL_PAIRED, M=M).reshape(M, -1)
            ref = np.stack([O.silu_mul_ref(y[m, :inter], y[m, inter:], act) for m in range(M)])
            assert out.shape == (M, inter)
            assert_forward_close(torch_to_f32(out), ref, act, tag=(K, N, M, "paired"))
        else:
            st_out = torch.zeros((M, -(-N // 16)), dtype=torch.float32, device=DEV)
            out = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, f32_to_torch(bias, act, DEV), K, N, gs, bits, sc.dtype,
                                    in_glue=ops.GLUE_RMSNORM, norm_weight=f32_to_torch(w, act, DEV), eps=1e-5,
                                    residual=f32_to_torch(res, act, DEV), stats_in=st_in, stats_out=st_out, M=M).reshape(M, -1)
            ref = O.residual_add_ref(res, y, act)
            got = torch_to_f32(out)
            assert_forward_close(got, ref, act, tag=(K, N, M, with_stats))
            assert np.allclose(st_out.cpu().numpy(), (got.astype(np.float64) ** 2).reshape(M, -1, 16).sum(axis=2), rtol=1e-5)
            # no glue at all == the plugin path's kernel on the same rows
            plain = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, None, K, N, gs, bits, sc.dtype, M=M)
            gen = ops.gemm(f32_to_torch(h, act, DEV), qw_t, meta, None, None, N, gs, bits, sc.dtype)
            assert torch.equal(plain.reshape(M, -1), gen)
    return qw_t, meta, sc


@pytest.mark.parametrize("M", [2, 3, 4, 5, 7, 8, 9, 12, 13, 16])
@pytest.mark.parametrize("act", ["fp16", "bf16"])
def test_decode_op_rows_2_to_16(ops, M, act):
    """The decode op on up to sixteen rows (a few sequences, or speculative tokens of one): per-row RMSNorm statistics (from the
    producer and reduced in the kernel), per-row residual + stats_out, the paired SiLU*mul epilogue, bias, cross-block split-K
    and a padded plan -- each against the oracle composed with HF's glue formulas, row by row."""
    qw_t, meta, sc = _check_decode_rows(ops, M, act, 128)
    with pytest.raises(RuntimeError, match="1..16"):
        ops.decode_linear(torch.zeros((17, 4096), dtype=TDT[act], device=DEV), qw_t, meta, None, 4096, 2048, 128, 4, sc.dtype, M=17)


@pytest.mark.parametrize("gs", [32, 64])
@pytest.mark.parametrize("M", [1, 3, 8, 16])
@pytest.mark.parametrize("act", ["fp16", "bf16"])
def test_decode_op_with_glue_group_size_32_64(ops, M, act, gs):
    """group_size 32 / 64 (a group constant per 32-row K-step instead of one per 128-row chunk) through the decode op with all of
    its glue: same oracle composition as the 128-group test (round 3: the glue variants used to exist for group_size % 128 == 0 only,
    so a 32g / 64g checkpoint fell back to per-module launches + torch glue)."""
    assert ops.decode_supported(4096, 6144, gs, False, M)
    assert not ops.decode_supported(4096, 6144, gs, True, 1)      # the in-kernel permutation needs one group constant per chunk
    _check_decode_rows(ops, M, act, gs, shapes=((4096, 6144, True, False), (4096, 4096, False, False), (4096, 2048, True, True),
                                                 (11008, 1024, False, False)))


def test_decode_op_rejects_unsupported_shapes(ops):
    # K not a multiple of the 128-row chunk: not on the decode op's pipeline (gptqhip_gemm's general path takes it)
    qweight, qzeros, scales, _ = synth_gptq(1, 4, 160, 64, 32)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, 32, 4)
    assert not ops.decode_supported(160, 64, 32)
    with pytest.raises(RuntimeError, match="outside the decode op's regular pipeline"):
        ops.decode_linear(torch.zeros(160, dtype=torch.float16, device=DEV), qw_t, meta, None, 160, 64, 32, 4, sc.dtype)
    # a tiny layer (K = 256: two chunks, four column tiles, a group constant per K-step) IS on it since round 4 (short-K plans: the
    # only ring round is mostly padding) -- and right
    qweight, qzeros, scales, g_idx = synth_gptq(1, 4, 256, 64, 64)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, 64, 4)
    assert ops.decode_supported(256, 64, 64)
    x = O.round_to(np.random.RandomState(4).randn(256).astype(np.float32) * 0.5, "fp16")
    y = ops.decode_linear(f32_to_torch(x, "fp16", DEV), qw_t, meta, None, 256, 64, 64, 4, sc.dtype)
    assert_forward_close(torch_to_f32(y)[None], O.forward_gptq(x[None], qweight, qzeros, scales, g_idx, 4, None, "fp16", "fp16"), "fp16")
    with pytest.raises(RuntimeError, match="norm_weight"):
        ops.decode_linear(torch.zeros(256, dtype=torch.float16, device=DEV), qw_t, meta, None, 256, 64, 64, 4, sc.dtype,
                          in_glue=ops.GLUE_RMSNORM)


def _make_stack(n_layers, hidden, inter, q_dim, kv_dim, dtype, seed=0, interleave=True, desc_act=False):
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.decode_chain import DecodeLayer
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed)
    gidx = {}

    def g_idx_for(k):
        # one act-order permutation per input width (siblings that share an input share it, as in real checkpoints)
        if k not in gidx:
            if desc_act:
                gidx[k] = (torch.randperm(k, device=DEV, generator=gen) // 128).to(torch.int32)
            else:
                gidx[k] = torch.arange(k, device=DEV, dtype=torch.int32) // 128
        return gidx[k]

    def lin(k, n):
        m = HipGptqLinear(bits=4, group_size=128, sym=True, desc_act=desc_act, in_features=k, out_features=n, bias=False,
                          register_buffers=False)
        w = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=DEV, generator=gen)
        # code 0 -> 8: codes symmetric around the sym zero-point 8, i.e. zero-mean weights like a real checkpoint (with
        # plain uniform codes every linear has a DC gain of -0.005*K and the fp16 residual stream overflows in 3 layers)
        m.qweight = w | (((~(w | (w >> 1) | (w >> 2) | (w >> 3))) & 0x11111111) << 3)
        m.qzeros = torch.full((k // 128, n // 8), -2004318072, dtype=torch.int32, device=DEV)
        m.scales = (torch.rand((k // 128, n), device=DEV, generator=gen) * 0.01 + 0.005).to(dtype)
        m.g_idx = g_idx_for(k)
        m.bias = None
        m.qzero_format(format=2)
        m.eval()
        m.post_init()
        return m

    def raw(k, n):
        m = HipGptqLinear(bits=4, group_size=128, sym=True, desc_act=desc_act, in_features=k, out_features=n, bias=False,
                          register_buffers=False)
        w = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=DEV, generator=gen)
        m.qweight = w | (((~(w | (w >> 1) | (w >> 2) | (w >> 3))) & 0x11111111) << 3)
        m.qzeros = torch.full((k // 128, n // 8), -2004318072, dtype=torch.int32, device=DEV)
        m.scales = (torch.rand((k // 128, n), device=DEV, generator=gen) * 0.01 + 0.005).to(dtype)
        m.g_idx = g_idx_for(k)
        m.bias = None
        m.qzero_format(format=2)
        m.eval()
        return m

    from gptqmodel_amd.utils.model import fuse_gate_up_interleaved
    layers = []
    for li in range(n_layers):
        nw = lambda: (1.0 + 0.1 * torch.randn(hidden, device=DEV, generator=gen)).to(dtype)
        gate, up = raw(hidden, inter), raw(hidden, inter)
        if interleave == "all" or (interleave and li % 2 == 0):    # default: both fusion layouts in one stack
            gu = fuse_gate_up_interleaved(gate, up)
        else:
            gu = raw(hidden, 2 * inter)     # (a fresh per-width g_idx lookup: same tensor as gate / up)
            gu.qweight = torch.cat([gate.qweight, up.qweight], dim=1).contiguous()
            gu.scales = torch.cat([gate.scales, up.scales], dim=1).contiguous()
        for m in (gate, up, gu):
            m.post_init()
        L = DecodeLayer(lin(hidden, q_dim + 2 * kv_dim), lin(q_dim, hidden), gu, lin(inter, hidden), nw(), nw())
        L.gate, L.up = gate, up           # separate projections: what the unfused reference step runs
        layers.append(L)
    return layers


def _reference_step(layers, x_in, q_dim, inter, eps):
    """The same step from separate launches: torch glue (HF formulas) + the general product kernel HipGptqLinear.forward."""
    h = x_in.clone()
    dt = h.dtype

    def rms(v, w):
        v32 = v.float()
        return w * (v32 * torch.rsqrt(v32.pow(2).mean(-1, keepdim=True) + eps)).to(dt)

    for L in layers:
        qkv = L.qkv(rms(h, L.input_norm)[None])[0]
        h = h + L.o(qkv[None, :q_dim])[0]
        xn = rms(h, L.post_norm)[None]
        h = h + L.down(torch.nn.functional.silu(L.gate(xn)) * L.up(xn))[0]
    return h


@pytest.mark.parametrize("dtype,desc_act", [(torch.float16, False), (torch.bfloat16, False), (torch.float16, True)])
def test_chain_tracks_unfused_reference_and_replays_identically(dtype, desc_act):
    from gptqmodel_amd.utils.decode_chain import DecodeStep
    hidden, inter, q_dim, kv_dim, n_layers = 4096, 14336, 4096, 1024, 3
    layers = _make_stack(n_layers, hidden, inter, q_dim, kv_dim, dtype, desc_act=desc_act)
    if desc_act:
        assert all(L.qkv.perm is not None and L.down.perm is not None for L in layers)
    step = DecodeStep(layers, hidden, q_dim, dtype)
    gen = torch.Generator(device=DEV)
    gen.manual_seed(5)
    xs = [(torch.randn(hidden, device=DEV, generator=gen) * 0.5).to(dtype) for _ in range(3)]
    want = []
    for x in xs:
        step.x_in.copy_(x)
        want.append(step.run().clone())
        ref = _reference_step(layers, x, q_dim, inter, 1e-5)
        assert torch.isfinite(want[-1]).all() and torch.isfinite(ref).all()
        # fused vs unfused: same math, different launches / glue kernels through 12 dependent linears
        assert rel_err(torch_to_f32(want[-1]), torch_to_f32(ref)) <= (4e-3 if dtype == torch.float16 else 3e-2)
    # graph replay of the step: deterministic, tracks the changing input
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step.run()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = step.run()
        for i in range(30):
            step.x_in.copy_(xs[i % 3], non_blocking=True)
            g.replay()
            assert torch.equal(out, want[i % 3]), f"graph replay {i} differs from the eager step"
        s.synchronize()


@pytest.mark.parametrize("act,bits,sdt", [("fp16", 8, "fp16"), ("bf16", 4, "bf16"), ("bf16", 8, "fp16")])
def test_decode_op_glue_variants_ragged_n_bias(ops, act, bits, sdt):
    """Glue on the less-travelled instantiations: 8-bit codes, bf16 scales, bias + residual + stats_out together, N that is not a
    multiple of 16 (half-filled last tile), in-block RMSNorm statistics (no producer)."""
    gs, K, N = 128, 4096, 1000
    qweight, qzeros, scales, g_idx = synth_gptq(77 + bits, bits, K, N, gs, scale_dtype=sdt)
    rng = np.random.RandomState(5)
    h = O.round_to(rng.randn(K).astype(np.float32) * 3.0, act)
    w = O.round_to(1.0 + rng.randn(K).astype(np.float32) * 0.1, act)
    res = O.round_to(rng.randn(N).astype(np.float32), act)
    bias = O.round_to(rng.randn(N).astype(np.float32) * 0.1, act)
    qw_t, meta, sc = _tiled(ops, qweight, qzeros, scales, gs, bits, sdt)
    stats = torch.zeros(-(-N // 16), dtype=torch.float32, device=DEV)
    out = ops.decode_linear(f32_to_torch(h, act, DEV), qw_t, meta, f32_to_torch(bias, act, DEV), K, N, gs, bits, sc.dtype,
                            in_glue=ops.GLUE_RMSNORM, norm_weight=f32_to_torch(w, act, DEV), eps=1e-6,
                            residual=f32_to_torch(res, act, DEV), stats_out=stats)
    xn = O.rmsnorm_ref(h, w, 1e-6, act)
    y = O.forward_gptq(xn[None], qweight, qzeros, scales, g_idx, bits, bias, act, sdt)
    ref = O.residual_add_ref(res[None], y, act)
    got = torch_to_f32(out)
    assert_forward_close(got[None], ref, act, norm_tol=decode_norm_tol(act, bits))
    pad = np.zeros(len(stats) * 16, dtype=np.float64)
    pad[:N] = got.astype(np.float64) ** 2
    assert np.allclose(stats.cpu().numpy(), pad.reshape(-1, 16).sum(axis=1), rtol=1e-5)
