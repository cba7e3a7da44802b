"""CPU checks of the chain oracle (tests/chain_oracle.py) and of the tensor-parallel shard functions THROUGH it: the oracle
composition of a two-layer decode chain over TP = 2 / 4 shards (column shards, globally group-sorted row shards with o_proj's input
index, down_proj's act-order permutation folded into the column ownership of gate / up) must compute the TP = 1 chain up to the
fp32 association of the K-shards' partial sums."""
import numpy as np
import pytest

from chain_oracle import build_layers, oracle_chain
from helpers import rel_err
from oracle import gptq_oracle as O


@pytest.mark.parametrize("desc_act", [False, True])
def test_tp_shards_compose_to_the_single_rank_chain(desc_act):
    hidden, inter, q_dim, kv_dim, gs = 512, 1024, 512, 128, 64
    ref_layers, _ = build_layers(1, 2, hidden, inter, q_dim, kv_dim, gs, desc_act, seed=7)
    x = O.round_to(np.random.RandomState(3).randn(hidden).astype(np.float32) * 0.5, "fp16")
    ref = oracle_chain(x, ref_layers, "fp16", 1e-5)
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.1
    for world in (2, 4):
        layers, shards = build_layers(world, 2, hidden, inter, q_dim, kv_dim, gs, desc_act, seed=7)
        for L in layers:
            for r in L["ranks"]:
                assert (r["o_index"] is not None) == desc_act
                assert np.array_equal(r["down"]["g_idx"], np.arange(inter // world) // gs)      # folded: sequential groups
                assert np.array_equal(r["o"]["g_idx"], np.arange(q_dim // world) // gs)
        got = oracle_chain(x, layers, "fp16", 1e-5)
        assert rel_err(got, ref) <= 2e-3, (world, rel_err(got, ref))
