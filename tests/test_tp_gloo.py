"""Tensor-parallel path on CPU: world_size 2, gloo, one process per rank.  The shard math (column/row slicing of the
packed checkpoint tensors, g_idx rebasing), the collective wiring (one fp32 all-reduce per row-parallel layer) and the
rounding chain are exercised with the ORACLE as each rank's local compute (the HIP kernel needs a GPU; the same
RowParallel/ColumnParallel modules wrap HipGptqLinear there)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from helpers import synth_gptq
from oracle import gptq_oracle as O

from gptqmodel_amd.utils import tp


class OracleLocal(nn.Module):
    """Checker stand-in for a post_init'ed HipGptqLinear shard: same forward()/forward_partial() surface."""

    def __init__(self, t, bits, gs):
        super().__init__()
        self.t, self.bits, self.gs = t, bits, gs

    def _w(self):
        t = self.t
        return O.dequant_gptq(t["qweight"].numpy(), t["qzeros"].numpy(), t["scales"].float().numpy(),
                              t["g_idx"].numpy(), self.bits)

    def forward_partial(self, x):
        return torch.from_numpy(x.float().numpy() @ self._w())

    def forward(self, x):
        b = None if self.t.get("bias") is None else self.t["bias"].float().numpy()
        return torch.from_numpy(O.matmul_round(x.float().numpy(), self._w(), b, "fp16")).half()


def _tensors(seed, bits, K, N, gs, bias=True):
    qweight, qzeros, scales, g_idx = synth_gptq(seed, bits, K, N, gs)
    t = {"qweight": torch.from_numpy(qweight), "qzeros": torch.from_numpy(qzeros),
         "scales": torch.from_numpy(scales).half(), "g_idx": torch.from_numpy(g_idx)}
    t["bias"] = torch.from_numpy(np.random.RandomState(seed).randn(N).astype(np.float32) * 0.1).half() if bias else None
    return t


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bits, gs, H, I, M = 4, 64, 256, 512, 3
        up = _tensors(1, bits, H, I, gs)      # column parallel (like gate/up)
        down = _tensors(2, bits, I, H, gs)    # row parallel (like down)
        x = torch.from_numpy(O.round_to(np.random.RandomState(7).randn(M, H).astype(np.float32) * 0.5, "fp16")).half()

        col = tp.ColumnParallelQuantLinear(OracleLocal(tp.shard_gptq_column(up, rank, world, bits), bits, gs))
        row = tp.RowParallelQuantLinear(OracleLocal(tp.shard_gptq_row(down, rank, world, bits, gs), bits, gs),
                                        bias=down["bias"])
        h_local = col(x)                                  # [M, I/world], no communication
        y = row(h_local)                                  # one all-reduce

        # single-process reference of the same two layers
        h_full = OracleLocal(up, bits, gs)(x)
        n0, n1 = rank * I // world, (rank + 1) * I // world
        assert torch.equal(h_local, h_full[:, n0:n1]), "column shard must equal the full layer's column slice bit for bit"
        y_full = OracleLocal(down, bits, gs)(h_full)
        err = (y.float() - y_full.float()).abs().max().item() / y_full.float().abs().max().item()
        assert err <= 1e-3, err
        # gather_output variant reproduces the full output exactly
        col_g = tp.ColumnParallelQuantLinear(OracleLocal(tp.shard_gptq_column(up, rank, world, bits), bits, gs),
                                             gather_output=True)
        assert torch.equal(col_g(x), h_full)
        # act-order (desc_act=True) row-parallel layer: rows sorted by group globally, then sliced; the sharded activation is
        # all-gathered and each rank selects the input features its rows need
        down_a = dict(down)
        down_a["g_idx"] = torch.from_numpy((np.random.RandomState(11).permutation(I) // gs).astype(np.int32))
        sh = tp.shard_gptq_row(down_a, rank, world, bits, gs, act_order="global_sort")
        row_a = tp.RowParallelQuantLinear(OracleLocal(sh, bits, gs), bias=down_a["bias"], input_index=sh["input_index"])
        y_a = row_a(h_local)
        y_a_full = OracleLocal(down_a, bits, gs)(h_full)
        err_a = (y_a.float() - y_a_full.float()).abs().max().item() / y_a_full.float().abs().max().item()
        assert err_a <= 1e-3, err_a
        # prefill-sized message: the two-step exchange (all-to-all, rank-ordered sum, one rounding + bias, 16-bit all-gather) against the
        # fp32 all-reduce path -- same partials, same single rounding; M = 1031 rows (not a multiple of the world size: padded slab)
        xb = torch.from_numpy(O.round_to(np.random.RandomState(17).randn(1031, I // world).astype(np.float32) * 0.5, "fp16")).half()
        loc = OracleLocal(tp.shard_gptq_row(down, rank, world, bits, gs), bits, gs)
        assert 1031 * H * 4 >= tp.TWO_STEP_MIN_BYTES
        y2 = tp.RowParallelQuantLinear(loc, bias=down["bias"])(xb)
        y1 = tp.RowParallelQuantLinear(loc, bias=down["bias"], two_step=False)(xb)
        assert y2.shape == y1.shape == (1031, H) and y2.dtype == torch.float16
        assert torch.equal(y2, y1), "two ranks: a + b in rank order == the all-reduce's sum, so the paths must agree bit for bit"
        both = [torch.empty_like(y2) for _ in range(world)]
        dist.all_gather(both, y2)
        assert torch.equal(both[0], both[1])
        ret[rank] = err
    finally:
        dist.destroy_process_group()


def test_tp2_column_then_row_matches_single_process():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]


def test_shard_shapes_and_constraints():
    t = _tensors(3, 4, 8192 // 8, 1024, 128)  # K=1024, N=1024
    K, N = 1024, 1024
    for world in (1, 2, 4, 8):
        for r in range(world):
            c = tp.shard_gptq_column(t, r, world, 4)
            assert c["qweight"].shape == (K // 8, N // world) and c["qzeros"].shape == (K // 128, N // world // 8)
            assert c["scales"].shape == (K // 128, N // world) and c["g_idx"].shape == (K,)
            rw = tp.shard_gptq_row(t, r, world, 4, 128)
            assert rw["qweight"].shape == (K // 8 // world, N) and rw["scales"].shape == (K // 128 // world, N)
            assert rw["bias"] is None and int(rw["g_idx"].min()) == 0 and int(rw["g_idx"].max()) == K // 128 // world - 1
    # shards tile the full matrix exactly
    full = O.dequant_gptq(t["qweight"].numpy(), t["qzeros"].numpy(), t["scales"].float().numpy(), t["g_idx"].numpy(), 4)
    cols = [tp.shard_gptq_column(t, r, 4, 4) for r in range(4)]
    wc = np.concatenate([O.dequant_gptq(c["qweight"].numpy(), c["qzeros"].numpy(), c["scales"].float().numpy(),
                                         c["g_idx"].numpy(), 4) for c in cols], axis=1)
    assert np.array_equal(wc, full)
    rows = [tp.shard_gptq_row(t, r, 4, 4, 128) for r in range(4)]
    wr = np.concatenate([O.dequant_gptq(c["qweight"].numpy(), c["qzeros"].numpy(), c["scales"].float().numpy(),
                                         c["g_idx"].numpy(), 4) for c in rows], axis=0)
    assert np.array_equal(wr, full)
    with pytest.raises(ValueError):
        tp.shard_gptq_row(t, 0, 16, 4, 128)       # 1024/16 = 64 rows < one group
    with pytest.raises(ValueError):
        tp.shard_gptq_column(_tensors(4, 4, 256, 40, 64), 0, 8, 4)   # 40/8 = 5 columns, not a multiple of 8
    act = dict(t)
    act["g_idx"] = torch.from_numpy((np.random.RandomState(0).permutation(K) // 128).astype(np.int32))
    with pytest.raises(NotImplementedError):
        tp.shard_gptq_row(act, 0, 2, 4, 128)
    # act_order="global_sort": the shards tile the GROUP-SORTED matrix; input_index says which input features each owns
    full_act = O.dequant_gptq(act["qweight"].numpy(), act["qzeros"].numpy(), act["scales"].float().numpy(), act["g_idx"].numpy(), 4)
    perm = O.act_order_perm(act["g_idx"].numpy())
    for world in (2, 4):
        sh = [tp.shard_gptq_row(act, r, world, 4, 128, act_order="global_sort") for r in range(world)]
        assert np.array_equal(np.concatenate([c["input_index"].numpy() for c in sh]), perm)
        ws = np.concatenate([O.dequant_gptq(c["qweight"].numpy(), c["qzeros"].numpy(), c["scales"].float().numpy(),
                                            c["g_idx"].numpy(), 4) for c in sh], axis=0)
        assert np.array_equal(ws, full_act[perm])
    # bit-level helpers round-trip
    assert torch.equal(tp._pack_rows(tp._unpack_rows(t["qweight"], 4), 4), t["qweight"])
    w8 = torch.from_numpy(np.random.RandomState(1).randint(-2**31, 2**31, size=(16, 8), dtype=np.int64).astype(np.int32))
    assert torch.equal(tp._pack_rows(tp._unpack_rows(w8, 8), 8), w8)
    # Llama-3-70B shapes satisfy the constraints up to TP=8
    for (k, n) in [(8192, 8192), (8192, 1024), (8192, 28672), (28672, 8192)]:
        for world in (1, 2, 4, 8):
            tp._bounds(n, 0, world, 8, "N")
            tp._bounds(k, 0, world, 128, "K")


def test_awq_shards_tile_the_matrix():
    rng = np.random.RandomState(9)
    K, N, gs = 256, 128, 64
    t = {"qweight": torch.from_numpy(rng.randint(-2**31, 2**31, size=(K, N // 8), dtype=np.int64).astype(np.int32)),
         "qzeros": torch.from_numpy(rng.randint(-2**31, 2**31, size=(K // gs, N // 8), dtype=np.int64).astype(np.int32)),
         "scales": torch.from_numpy(rng.rand(K // gs, N).astype(np.float32) * 0.01 + 0.005).half(), "bias": None}
    full = O.dequant_awq(t["qweight"].numpy(), t["qzeros"].numpy(), t["scales"].float().numpy(), gs)
    wc = np.concatenate([O.dequant_awq(c["qweight"].numpy(), c["qzeros"].numpy(), c["scales"].float().numpy(), gs)
                         for c in (tp.shard_awq_column(t, r, 2) for r in range(2))], axis=1)
    wr = np.concatenate([O.dequant_awq(c["qweight"].numpy(), c["qzeros"].numpy(), c["scales"].float().numpy(), gs)
                         for c in (tp.shard_awq_row(t, r, 2, gs) for r in range(2))], axis=0)
    assert np.array_equal(wc, full) and np.array_equal(wr, full)


def test_shard_mlp_act_order_needs_no_exchange():
    """down_proj's act-order permutation folded into the column OWNERSHIP of gate / up (tp.shard_mlp_act_order): every rank's
    (gate_r, up_r) produce exactly the intermediate features its group-sorted down_proj rows consume, in that order -- so the
    sum over ranks of down_r(act(gate_r(x)) * up_r(x)) is the full MLP over the same terms (dequantised shards tile the
    permuted matrices bit for bit), with no all-gather (gptqmodel/utils/marlin.py:296-305,368-372)."""
    bits, gs, H, I = 4, 64, 256, 512
    gate, up = _tensors(21, bits, H, I, gs), _tensors(22, bits, H, I, gs)
    # gate / up themselves are act-order checkpoints on their INPUT side (shared permutation, like real GPTQ siblings)
    gi = torch.from_numpy((np.random.RandomState(5).permutation(H) // gs).astype(np.int32))
    gate["g_idx"], up["g_idx"] = gi, gi.clone()
    down = _tensors(23, bits, I, H, gs, bias=False)
    down["g_idx"] = torch.from_numpy((np.random.RandomState(6).permutation(I) // gs).astype(np.int32))
    deq = lambda t: O.dequant_gptq(t["qweight"].numpy(), t["qzeros"].numpy(), t["scales"].float().numpy(), t["g_idx"].numpy(), bits)
    wg, wu, wd = deq(gate), deq(up), deq(down)
    perm = O.act_order_perm(down["g_idx"].numpy())
    for world in (2, 4):
        sh = [tp.shard_mlp_act_order(gate, up, down, r, world, bits, gs) for r in range(world)]
        k = I // world
        for r, (g_r, u_r, d_r) in enumerate(sh):
            assert "input_index" not in d_r
            assert np.array_equal(d_r["g_idx"].numpy(), np.arange(k) // gs)
            cols = perm[r * k:(r + 1) * k]
            assert np.array_equal(deq(g_r), wg[:, cols]) and np.array_equal(deq(u_r), wu[:, cols])
            assert np.array_equal(deq(d_r), wd[cols])
            assert torch.equal(g_r["bias"], gate["bias"][torch.from_numpy(cols)])
        # the MLP through the shards == the full MLP (fp32 association of the K-shards aside)
        x = O.round_to(np.random.RandomState(8).randn(2, H).astype(np.float32) * 0.5, "fp16")
        full = O.matmul_round(O.silu_mul_ref(O.matmul_round(x, wg, gate["bias"].float().numpy(), "fp16"),
                                             O.matmul_round(x, wu, up["bias"].float().numpy(), "fp16"), "fp16"), wd, None, "fp16")
        part = sum(O.silu_mul_ref(O.matmul_round(x, deq(g_r), g_r["bias"].float().numpy(), "fp16"),
                                  O.matmul_round(x, deq(u_r), u_r["bias"].float().numpy(), "fp16"), "fp16") @ deq(d_r)
                   for g_r, u_r, d_r in sh)
        assert np.abs(O.round_to(part, "fp16") - full).max() <= 1e-3 * np.abs(full).max()
    # a down_proj without act-order degenerates to the plain contiguous split
    down_seq = _tensors(23, bits, I, H, gs, bias=False)
    g_r, u_r, d_r = tp.shard_mlp_act_order(gate, up, down_seq, 1, 2, bits, gs)
    assert np.array_equal(deq(g_r), wg[:, I // 2:]) and np.array_equal(deq(d_r), deq(down_seq)[I // 2:])


def test_shard_helpers_refuse_words_they_would_cut():
    """2 / 3 / 5 / 6 / 7-bit words are sharded after widening (same code values in 4- / 8-bit fields): slicing them directly would cut
    fields in two, so the helpers refuse instead."""
    import pytest
    import torch
    from gptqmodel_amd.utils import tp
    t = {"qweight": torch.zeros((24, 64), dtype=torch.int32), "qzeros": torch.zeros((2, 6), dtype=torch.int32),
         "scales": torch.ones((2, 64), dtype=torch.float16), "g_idx": torch.arange(256, dtype=torch.int32) // 128, "bias": None}
    for fn, args in ((tp.shard_gptq_column, (t, 0, 2, 3)), (tp.shard_gptq_row, (t, 0, 2, 3, 128)),
                     (tp.select_gptq_columns, (t, torch.arange(32), 3))):
        with pytest.raises(NotImplementedError, match="widen"):
            fn(*args)


# ---------------------------------------------------------------------------------------------------------------------
# Round 6: the ORCHESTRATION of the tensor-parallel decode step -- utils.decode_chain.tp_layer_plan, the very
# list TPDecodeStep binds to device pointers -- executed on two gloo ranks with the oracle as local compute and dist.all_reduce /
# dist.all_gather as the exchange, against the single-process chain oracle over the same shards.
# ---------------------------------------------------------------------------------------------------------------------
def _run_tp_plan(rank, world, layers, shards, x, eps=1e-5, act="fp16"):
    """Execute tp_layer_plan for every layer on this rank: numpy buffers by name, ops = the oracle's linear on the rank's shard with the
    decode op's glue semantics (include/gptqhip.h gptqhip_decode_op), "ar" = fp32 SUM all-reduce + the one-shot kernel's epilogue
    (one rounding of the sum, + bias, residual add, per-16 sums of squares), "ag" = all_gather + index select."""
    from chain_oracle import deq
    from gptqmodel_amd.utils.decode_chain import tp_layer_plan
    h_in, st_in = x.copy(), None
    for li, L in enumerate(layers):
        sh = shards[rank][li]
        buf = {"h_in": h_in, "st_in": st_in, "w_in": L["w_in"], "w_post": L["w_post"], "o_bias": None, "down_bias": None,
               "o_input_index": sh["o_index"], None: None}
        W = {"qkv": np.concatenate([deq(sh[n]) for n in ("q", "k", "v")], axis=1), "o": deq(sh["o"]),
             "gate": deq(sh["gate"]), "up": deq(sh["up"]), "down": deq(sh["down"])}
        q_local = sh["q"]["qweight"].shape[1]
        for st in tp_layer_plan(sh["o_index"] is not None):
            if st[0] == "op":
                _, which, xn, out, g, norm, og, s_in = st
                xv = buf[xn]
                if g == "rmsnorm":
                    if buf[s_in] is not None:      # the producer's statistics must BE the sums of squares of the vector they travel with
                        assert np.allclose(buf[s_in].sum(), (xv.astype(np.float64) ** 2).sum(), rtol=1e-5)
                    xv = O.rmsnorm_ref(xv, buf[norm], eps, act)
                if which == "gate_up":
                    assert og == "silu_mul_paired"
                    y = O.silu_mul_ref(O.matmul_round(xv[None], W["gate"], None, act), O.matmul_round(xv[None], W["up"], None, act), act)[0]
                elif og == "partial_f32":
                    y = (xv[None].astype(np.float32) @ W[which])[0]             # unrounded fp32 partial sums of this rank's K-shard
                else:
                    y = O.matmul_round(xv[None], W[which], None, act)[0]
                buf[out] = y
                if which == "qkv":
                    buf["qkv_out_local"] = y[:q_local]
            elif st[0] == "ag":
                _, xn, idx, out = st
                parts = [torch.empty(q_local, dtype=torch.float32) for _ in range(world)]
                dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(buf[xn], dtype=np.float32)))
                buf[out] = torch.cat(parts).numpy()[buf[idx]]
            else:
                _, res, bias, out, stats = st
                t = torch.from_numpy(np.ascontiguousarray(buf["partial"], dtype=np.float32))
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                y = O.round_to(t.numpy(), act)
                if buf[bias] is not None:
                    y = O.round_to(y + buf[bias], act)
                buf[out] = O.residual_add_ref(buf[res][None], y[None], act)[0]
                buf[stats] = (buf[out].astype(np.float64) ** 2).reshape(-1, 16).sum(axis=1).astype(np.float32)
        h_in, st_in = buf["h2"], buf["st2"]
    return h_in


def _tp_plan_worker(rank, world, port, desc_act, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from chain_oracle import build_layers, oracle_chain
        hidden, inter, q_dim, kv_dim, gs = 512, 1024, 512, 128, 64
        layers, shards = build_layers(world, 2, hidden, inter, q_dim, kv_dim, gs, desc_act, seed=7)
        x = O.round_to(np.random.RandomState(3).randn(hidden).astype(np.float32) * 0.5, "fp16")
        got = _run_tp_plan(rank, world, layers, shards, x)
        want = oracle_chain(x, layers, "fp16", 1e-5)          # the single-process composition over the same shards, rank-ordered sums
        # two ranks: a + b is the all-reduce's sum whatever its order, so the distributed run must equal the composition bit for bit
        ret[rank] = (bool(np.array_equal(got, want)), float(np.abs(got - want).max()), float(np.abs(want).max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("desc_act", [False, True])
def test_tp2_decode_step_plan_on_gloo_matches_the_chain_oracle(desc_act):
    world = 2
    port = 31500 + (os.getpid() % 2000) + (7 if desc_act else 0)
    ret = mp.Manager().dict()
    mp.spawn(_tp_plan_worker, args=(world, port, desc_act, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    for r in (0, 1):
        equal, err, scale = ret[r]
        assert equal, (r, err, scale)
        assert scale > 0.1
