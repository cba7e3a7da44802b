"""The C-ABI library loads on a CPU-only box and exports every symbol include/gptqhip.h declares; argument
validation returns error codes + messages without touching a GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from gptqmodel_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gptqhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gptqhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("gptqhip_gemm", "gptqhip_repack_tiled", "gptqhip_repack_awq", "gptqhip_dequant",
                 "gptqhip_dequant_tiled", "gptqhip_widen_codes", "gptqhip_workspace_bytes", "gptqhip_last_error", "gptqhip_device_info"):
        assert must in syms


def test_every_declared_symbol_is_exported_and_bound(lib):
    from gptqmodel_amd import _lib
    syms = declared_symbols()
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding table must mirror include/gptqhip.h exactly"
    for s in syms:
        assert hasattr(lib, s), f"libgptqhip.so does not export {s}"
    assert lib.gptqhip_abi_version() == _lib.ABI_VERSION


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(ROOT, "include", "gptqhip.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S).lower()
    assert "at::Tensor" not in text and "#include <hip" not in text  # callers need no HIP/torch headers


def test_size_helpers(lib):
    # 4-bit: one 1 KiB block (256 words) per (16-column tile, 128-row chunk); ragged shapes round up
    assert lib.gptqhip_tiled_words(4096, 4096, 4) == 4096 * 4096 // 8
    assert lib.gptqhip_tiled_words(4096, 4096, 8) == 4096 * 4096 // 4
    assert lib.gptqhip_tiled_words(96, 8, 4) == 256
    assert lib.gptqhip_meta_words(4096, 4096, 128) == 32 * 4096
    assert lib.gptqhip_meta_words(4096, 4096, 96) == 0  # K % group_size != 0
    assert lib.gptqhip_tiled_words(4096, 4096, 3) == 0
    assert lib.gptqhip_workspace_bytes(1, 4096, 4096, 128, 4, 0) >= 64 * 1024
    assert lib.gptqhip_workspace_bytes(8, 4096, 4096, 128, 4, 1) >= lib.gptqhip_workspace_bytes(8, 4096, 4096, 128, 4, 0) + 8 * 4096 * 2
    assert lib.gptqhip_workspace_bytes(0, 4096, 4096, 128, 4, 0) == 0


def test_decode_op_shape_predicate_is_host_logic(lib):
    """gptqhip_decode_supported is the planner's predicate (no GPU): every layer shape of the BASELINE models, and the common
    checkpoint shapes whose K / 128 has no waves x ring-depth factorisation (padded last ring round), are on the decode op's
    pipeline; ragged K is not (it takes gptqhip_gemm's general path)."""
    yes = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096), (8192, 10240), (8192, 8192), (8192, 57344), (28672, 8192),
           (11008, 4096), (4096, 22016), (18944, 3584), (3584, 37888), (5120, 27648), (13824, 5120), (1536, 8960), (8960, 1536)]
    for K, N in yes:
        assert lib.gptqhip_decode_supported(K, N, 128, 0, 1) == 1, (K, N)
    # group_size 32 / 64 (a constant per 32-row K-step): on the pipeline since round 3, without the in-kernel permutation
    for gs in (32, 64):
        assert all(lib.gptqhip_decode_supported(K, N, gs, 0, M) == 1 for K, N in yes[:4] for M in (1, 4, 16)), gs
        assert lib.gptqhip_decode_supported(4096, 4096, gs, 1, 1) == 0
    assert lib.gptqhip_decode_supported(4096, 4096, 16, 0, 1) == 0      # not a supported group size at all
    assert lib.gptqhip_decode_supported(4000, 4096, 32, 0, 1) == 0      # K % 128 != 0
    assert lib.gptqhip_decode_supported(4096, 4096, 4096, 0, 1) == 1    # one group for the whole K (group_size = -1 checkpoints)
    assert lib.gptqhip_decode_supported(14336, 4096, 14336, 0, 1) == 1
    assert lib.gptqhip_decode_supported(0, 4096, 128, 0, 1) == 0
    # act-order in the kernel: 4-deep ring only, and the x row must fit in LDS next to the wave slots
    assert lib.gptqhip_decode_supported(4096, 4096, 128, 1, 1) == 1 and lib.gptqhip_decode_supported(14336, 4096, 128, 1, 1) == 1
    assert lib.gptqhip_decode_supported(1024, 1024, 128, 1, 1) == 0 and lib.gptqhip_decode_supported(28672, 8192, 128, 1, 1) == 0
    # up to sixteen rows (no permutation there); seventeen are gptqhip_gemm's business
    assert lib.gptqhip_decode_supported(4096, 28672, 128, 0, 4) == 1 and lib.gptqhip_decode_supported(14336, 4096, 128, 0, 8) == 1
    for K, N in yes[:8]:
        assert lib.gptqhip_decode_supported(K, N, 128, 0, 9) == 1 and lib.gptqhip_decode_supported(K, N, 128, 0, 16) == 1, (K, N)
    assert lib.gptqhip_decode_supported(4096, 4096, 128, 0, 17) == 0 and lib.gptqhip_decode_supported(4096, 4096, 128, 1, 2) == 0


def test_prefill_planner_invariants_over_a_grid_of_shapes(lib):
    """plan_tiled (through gptqhip_plan_describe, forced to the prefill kernel) over a grid of M x layer shapes: whatever the launch model
    picks must be launchable -- a tile height that exists, at least 4 chunks of K per split block, at most 64 MiB of fp32 slabs, one round
    of blocks when K is split -- and gptqhip_workspace_bytes must cover the slabs."""
    assert lib.gptqhip_set_tuning(0, 2, 0) == 0
    try:
        for K, N in ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (8192, 10240), (28672, 8192), (8192, 57344), (1024, 512),
                     (4096, 128256), (11008, 4096), (2048, 2048)):
            for M in (17, 33, 64, 65, 100, 128, 200, 256, 384, 512, 777, 1024, 1536, 2048, 4096, 9000, 65536):
                buf = ctypes.create_string_buffer(256)
                assert lib.gptqhip_plan_describe(M, K, N, 128, 4, 0, buf, 256) == 0
                words = buf.value.decode().split()
                assert words[0] == "tiled", (M, K, N, words)
                pl = dict(kv.split("=") for kv in words[1:])
                bm, s, tail, bn = int(pl["bm"]), int(pl["splits"]), int(pl["tail_cols"]), int(pl.get("bn", 256))
                chunks, nbx = -(-K // 128), -(-N // bn)
                # 128-column blocks (one column tile per wave): 4-bit / one constant per chunk, up to 1024 rows (round 5; above 512 rows
                # at the three heights the calibration sweep covered)
                assert bn == 256 or (bn == 128 and M <= 1024 and bm <= 128 and tail == 0 and (M <= 512 or bm in (64, 96, 128))), (M, K, N, pl)
                # tile heights: 64 / 128 / 256 everywhere; 32..112 in steps of 16 for 4-bit / one constant per chunk up to 1024 rows (round 5)
                assert bm in ((32, 48, 64, 80, 96, 112, 128, 256) if M <= 1024 else (64, 128, 256)), (M, K, N, pl)
                assert 1 <= s <= max(1, chunks // 4) and 0 <= tail < nbx, (M, K, N, pl)
                for gs_other, bits_other in ((64, 4), (128, 8)):     # ... and never for the variants that have no such instantiation
                    b2 = ctypes.create_string_buffer(256)
                    assert lib.gptqhip_plan_describe(M, K, N, gs_other, bits_other, 0, b2, 256) == 0
                    p2 = dict(kv.split("=") for kv in b2.value.decode().split()[1:])
                    assert int(p2["bm"]) in (64, 128, 256) and "bn" not in p2, (M, K, N, b2.value)
                if s > 1:
                    assert s * M * N * 4 <= 64 << 20 and nbx * -(-M // bm) * s <= 256 and tail == 0, (M, K, N, pl)
                    assert lib.gptqhip_workspace_bytes(M, K, N, 128, 4, 0) >= s * M * N * 4, (M, K, N, pl)
    finally:
        assert lib.gptqhip_set_tuning(0, 0, 0) == 0


def test_kernel_family_crossover_is_host_logic(lib):
    """gptqhip_plan_describe = the planner's decisions without a GPU: the measured crossover between the decode kernel (one / several
    row tiles, one / several column tiles per block) and the MFMA-tiled prefill kernel (docs/history/DESIGN_rounds_1-5.md 4.1.1, profiles/r03_mid_m_sweep.txt,
    r03_wide_layers.txt).  Pinned here so that a planner edit that silently re-routes a regime shows up on the CPU."""
    def d(M, K, N, gs=128, bits=4, perm=0):
        buf = ctypes.create_string_buffer(256)
        assert lib.gptqhip_plan_describe(M, K, N, gs, bits, perm, buf, 256) == 0, lib.gptqhip_last_error()
        return dict(kv.split("=") for kv in buf.value.decode().split()[1:]) | {"family": buf.value.decode().split()[0]}
    # batch-1 decode: one row tile, one column tile, counted-wait pipeline
    for K, N in ((4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (8192, 57344)):
        p = d(1, K, N)
        assert p["family"] == "skinny" and p["mt"] == "1" and p["nt"] == "1" and p["regular"] == "1" and p["launches"] == "1"
    # 17..64 rows on narrow layers: ONE launch, two / four row tiles, 8 waves (512-thread instantiations)
    assert d(32, 4096, 4096) | {} == d(32, 4096, 4096) and d(32, 4096, 4096)["mt"] == "2" and d(32, 4096, 4096)["waves"] == "8"
    p = d(64, 4096, 4096)
    assert p["family"] == "skinny" and p["mt"] == "4" and p["launches"] == "1" and p["splits"] == "1"
    assert d(65, 4096, 4096)["family"] == "tiled" and d(48, 4096, 4096, bits=8)["family"] == "tiled"
    # K-heavy layers (K >= 10240, 4-bit g128): the prefill kernel from 17 rows since round 5 (14336x4096 13.0 vs 15.8 us at 17 rows);
    # 8192-deep layers keep the decode kernel up to 32 rows, 8-bit weights and group_size 64 keep the old 32-row rule
    assert d(16, 14336, 4096)["family"] == "skinny" and d(17, 14336, 4096)["family"] == "tiled" and d(40, 14336, 4096)["family"] == "tiled"
    assert d(32, 8192, 1024)["family"] == "skinny" and d(32, 14336, 4096, bits=8)["family"] == "skinny" and d(32, 14336, 4096, gs=64)["family"] == "skinny"
    # wide layers: the wide form (4 / 2 column tiles per block) from 5 rows, up to 16 rows everywhere, up to 32 where measured ahead
    assert d(4, 4096, 28672)["nt"] == "1" and d(5, 4096, 28672)["nt"] == "4" and d(16, 4096, 28672)["nt"] == "4"
    assert d(32, 4096, 28672)["family"] == "skinny" and d(32, 4096, 28672)["nt"] == "4" and d(33, 4096, 28672)["family"] == "tiled"
    assert d(16, 4096, 6144)["nt"] == "2" and d(16, 4096, 8192)["nt"] == "2" and d(16, 4096, 4096)["nt"] == "1"
    assert d(16, 4096, 128256)["nt"] == "4" and d(32, 4096, 128256)["family"] == "tiled"               # lm_head: tiled from 17 rows
    assert d(16, 8192, 10240)["family"] == "skinny" and d(24, 8192, 10240)["family"] == "tiled"       # K >= 8192: tiled from 17 rows ...
    assert d(24, 8192, 8192)["family"] == "skinny" and d(24, 8192, 8192)["nt"] == "2"                   # ... unless the wide form fits one round
    assert d(48, 4096, 6144)["family"] == "tiled" and d(32, 4096, 6144)["family"] == "skinny"          # 33..64 rows in one launch: N < 6144 only
    # mid M: tile height and split factor come from the launch model together (profiles/r03_tiled_planner.txt); since round 5 the height
    # moves in steps of 16 rows up to 512 rows (profiles/r05_midm_heights_sweep*.txt): a 160-row batch is two 80-row tiles, a 96-row
    # batch one 96-row tile, 72 / 136 rows on the reference benchmark's 4096x11008 layer one / two 80-row tiles (128 / 192 rows before)
    # ... and so does the block width: 128-column blocks (one column tile per wave, `bn=128`) where twice the blocks spare the launch its
    # split-K slabs -- the reference benchmark's 4096x11008 layer at 72 / 128 / 136 rows runs WITHOUT split-K (4 / 4 / 2 slabs before)
    assert d(160, 4096, 6144).get("bn") == "128" and d(160, 4096, 6144)["splits"] == "1"
    assert d(1280, 14336, 4096)["bm"] == "256" and d(1280, 14336, 4096)["splits"] == "3" and "bn" not in d(1280, 14336, 4096)
    assert d(96, 8192, 57344)["bm"] == "96" and d(96, 8192, 57344)["splits"] == "1"
    for m_ in (72, 128, 136):
        assert d(m_, 4096, 11008).get("bn") == "128" and d(m_, 4096, 11008)["splits"] == "1", (m_, d(m_, 4096, 11008))
    assert -(-72 // int(d(72, 4096, 11008)["bm"])) * int(d(72, 4096, 11008)["bm"]) <= 96 and d(136, 4096, 11008)["bm"] == "80"
    assert d(128, 4096, 4096).get("bn") == "128" and int(d(128, 4096, 4096)["splits"]) <= 4
    assert d(1025, 4096, 4096)["bm"] in ("64", "128", "256") and "bn" not in d(1025, 4096, 4096)
    assert d(768, 4096, 4096).get("bn") == "128" and d(768, 4096, 4096)["splits"] == "1"      # 640 / 768 rows on 4096^2: 35.0 / 36.4 -> 29.7 / 31.5 us
    assert d(160, 4096, 6144, gs=64)["bm"] in ("64", "128", "256") and "bn" not in d(160, 4096, 6144, gs=64)
    # prefill: 256- / 128- / 64-row tiles
    assert d(8192, 4096, 4096)["bm"] == "256" and d(128, 4096, 4096, bits=8)["bm"] == "64"
    # act-order: in-kernel permutation at one row, a gather pass otherwise
    assert d(1, 4096, 4096, perm=1)["gather"] == "0" and d(1, 4096, 4096, perm=1)["depth"] == "4"
    assert d(8, 4096, 4096, perm=1)["gather"] == "1" and d(2048, 4096, 4096, perm=1)["gather"] == "1"
    assert lib.gptqhip_plan_describe(0, 4096, 4096, 128, 4, 0, ctypes.create_string_buffer(8), 8) == -22


def test_comm_buffer_sizes_and_collective_argument_checks(lib):
    """The one-shot collectives' host logic: buffer size = header + 2 x 8 all-reduce slots + 2 x 8 all-gather slots, argument
    validation before anything touches a GPU."""
    assert lib.gptqhip_comm_bytes(9, 8192) == 0 and lib.gptqhip_comm_bytes(2, 0) == 0 and lib.gptqhip_comm_bytes(2, 70000) == 0
    b1, b2 = lib.gptqhip_comm_bytes(2, 8192), lib.gptqhip_comm_bytes(8, 8192)
    assert b1 == b2 >= 2 * 8 * 8192 * 4 + 2 * 8 * 8192 * 2          # sized for the maximum world: any group size shares a layout
    one = ctypes.c_void_p(256)
    peers = (ctypes.c_void_p * 2)(256, 512)
    assert lib.gptqhip_allreduce_oneshot(one, peers, 0, 2, 1001, 8192, None, None, one, None, 0, None) == -22     # n % 4
    assert lib.gptqhip_allreduce_oneshot(one, peers, 2, 2, 1024, 8192, None, None, one, None, 0, None) == -22     # rank >= world
    assert lib.gptqhip_allreduce_oneshot(ctypes.c_void_p(260), peers, 0, 2, 1024, 8192, None, None, one, None, 0, None) == -22   # alignment
    assert b"aligned" in lib.gptqhip_last_error()
    assert lib.gptqhip_allgather_select(one, peers, 0, 2, 1001, 8192, None, 2002, one, 0, None) == -22             # n_local % 8
    assert lib.gptqhip_allgather_select(one, peers, 0, 2, 1024, 8192, None, 1000, one, 0, None) == -22             # no index: whole vector
    assert lib.gptqhip_rmsnorm_gather(one, one, None, one, 4, 4096, 1e-5, 0, None) == -22                          # out aliases h
    assert lib.gptqhip_rmsnorm_gather(one, ctypes.c_void_p(512), None, ctypes.c_void_p(1024), 4, 4100, 1e-5, 0, None) == -22
    assert lib.gptqhip_rmsnorm_gather(one, one, None, one, 0, 4096, 1e-5, 0, None) == 0                            # empty batch


def test_argument_validation_reports_errors_without_a_gpu(lib):
    EINVAL = -22
    one = ctypes.c_void_p(16)  # non-null dummy; validation fails before any dereference
    rc = lib.gptqhip_gemm(None, one, one, None, None, one, one, 1 << 20, 1, 4096, 4096, 128, 4, 0, 0, 0, None)
    assert rc == EINVAL and b"null" in lib.gptqhip_last_error()
    rc = lib.gptqhip_gemm(one, one, one, None, None, one, one, 1 << 20, 1, 4096, 4096, 100, 4, 0, 0, 0, None)
    assert rc == EINVAL and b"group_size" in lib.gptqhip_last_error()
    rc = lib.gptqhip_gemm(one, one, one, None, None, one, one, 1 << 20, 1, 4096, 4096, 128, 3, 0, 0, 0, None)
    assert rc == EINVAL and b"bits" in lib.gptqhip_last_error()
    rc = lib.gptqhip_gemm(one, one, one, None, None, one, one, 1 << 20, 1, 4100, 4096, 128, 4, 0, 0, 0, None)
    assert rc == EINVAL
    rc = lib.gptqhip_gemm(one, one, one, None, None, one, one, 16, 1, 4096, 4096, 128, 4, 0, 0, 0, None)
    assert rc == -12 and b"workspace" in lib.gptqhip_last_error()  # ENOMEM
    # empty batch is a no-op success (the reference returns an empty tensor)
    assert lib.gptqhip_gemm(None, None, None, None, None, None, None, 0, 0, 4096, 4096, 128, 4, 0, 0, 0, None) == 0
    rc = lib.gptqhip_repack_tiled(one, one, one, None, None, one, 4096, 4096, 128, 4, None)
    assert rc == EINVAL  # qweight without qweight_t
    assert lib.gptqhip_device_info(0, None, None, None, 0) in (0, -19)  # ENODEV on a CPU-only box


def test_comm_buffer_sizing_and_argument_checks(lib):
    """One-shot all-reduce entry points: sizes and validation without a GPU."""
    one = ctypes.c_void_p(16)
    assert lib.gptqhip_comm_bytes(8, 8192) >= 2 * 8 * 8192 * 4
    assert lib.gptqhip_comm_bytes(9, 8192) == 0 and lib.gptqhip_comm_bytes(2, 0) == 0 and lib.gptqhip_comm_bytes(2, 1 << 20) == 0
    peers = (ctypes.c_void_p * 2)(16, 16)
    assert lib.gptqhip_allreduce_oneshot(one, peers, 0, 2, 1001, 8192, None, None, one, None, 0, None) == -22   # n % 4
    assert lib.gptqhip_allreduce_oneshot(one, peers, 2, 2, 1024, 8192, None, None, one, None, 0, None) == -22   # rank >= world
    assert lib.gptqhip_allreduce_oneshot(one, peers, 0, 2, 16384, 8192, None, None, one, None, 0, None) == -22  # n > n_max
    assert lib.gptqhip_allreduce_oneshot(one, peers, 0, 2, 1024, 8192, None, None, one, None, 7, None) == -22   # dtype tag
    assert lib.gptqhip_comm_open(None, None) == -22


def test_missing_library_fails_loudly(monkeypatch):
    from gptqmodel_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgptqhip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.load()


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from gptqmodel_amd import ops
    x = torch.zeros((1, 128), dtype=torch.float16)
    w = torch.zeros(256, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(x, w, w, None, None, 16, 128, 4, torch.float16)


def test_host_packer_bit_exact_with_reference_pack_block(lib):
    """gptqhip_pack_gptq_host (threaded C++, no GPU) vs the golden produced by the reference's pack_block."""
    import numpy as np
    import torch
    from conftest import load_golden
    from gptqmodel_amd import ops
    g = load_golden("ref_pack.npz")
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_qweight")})
    for t in tags:
        bits = int(t.split("_")[0][1:])
        for threads in (1, 3):
            qw, qz = ops.pack_gptq_host(torch.from_numpy(g[t + "_weight"]), torch.from_numpy(g[t + "_scales"]),
                                        torch.from_numpy(g[t + "_zeros"]), torch.from_numpy(g[t + "_g_idx"]), bits, threads)
            assert np.array_equal(qw.numpy(), g[t + "_qweight"]) and np.array_equal(qz.numpy(), g[t + "_qzeros"]), t
    # the other bit widths / layouts (continuous 2 / 3, split-plane 3, planar 5 / 6 / 7, 4 / 8 under gptq_p)
    g = load_golden("ref_pack_bits.npz")
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_qweight")})
    assert len(tags) == 8
    for t in tags:
        bits, planar = int(t.split("_")[0][1:]), t.endswith("_p")
        assert bool(int(g[t + "_planar"])) == (bits in (5, 6, 7) or (bits == 3 and planar))
        for threads in (1, 3):
            qw, qz = ops.pack_gptq_host(torch.from_numpy(g[t + "_weight"]), torch.from_numpy(g[t + "_scales"]),
                                        torch.from_numpy(g[t + "_zeros"]), torch.from_numpy(g[t + "_g_idx"]), bits, threads, planar=planar)
            assert np.array_equal(qw.numpy(), g[t + "_qweight"]) and np.array_equal(qz.numpy(), g[t + "_qzeros"]), t
    with pytest.raises(RuntimeError, match="out of range"):
        ops.pack_gptq_host(torch.zeros(32, 32), torch.ones(1, 32), torch.zeros(1, 32), torch.full((32,), 5), 4)
    with pytest.raises(RuntimeError, match="only planar"):
        ops.pack_gptq_host(torch.zeros(32, 32), torch.ones(1, 32), torch.zeros(1, 32), torch.zeros(32), 5, planar=False)


def test_host_packer_negative_g_idx_and_clamp_like_the_reference_tests(lib):
    """The reference's tests/test_pack.py restated for the C++ host packer: negative g_idx entries wrap by +G (:157-173) and
    reconstructed codes saturate BEFORE the integer conversion (:175-238: raw codes -8..23 plus +-1e20), here at every bit width."""
    import numpy as np
    import torch
    from gptqmodel_amd import ops
    from oracle import gptq_oracle as O
    torch.manual_seed(5)
    K, N, gs = 128, 32, 32
    w = torch.randn(N, K) * 0.05
    scales = torch.rand(K // gs, N) * 0.01 + 0.005
    zeros = torch.randint(0, 16, (K // gs, N), dtype=torch.int32)
    g_idx = (torch.arange(K) // gs).to(torch.int32)
    g_neg = g_idx.clone()
    g_neg[::7] -= K // gs
    a = ops.pack_gptq_host(w, scales, zeros, g_idx, 4, 2)
    b = ops.pack_gptq_host(w, scales, zeros, g_neg, 4, 2)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for bits in (2, 3, 4, 5, 6, 7, 8):
        max_q, zero = 2 ** bits - 1, 2 ** (bits - 1)
        raw = torch.arange(-8, 24, dtype=torch.float32).view(32, 1).expand(32, 32).contiguous()     # [in, out]
        raw[0].fill_(-1e20)
        raw[-1].fill_(1e20)
        qw, qz = ops.pack_gptq_host((raw - zero).T.contiguous(), torch.ones(1, 32), torch.full((1, 32), zero, dtype=torch.int32),
                                    torch.zeros(32, dtype=torch.int32), bits, 1)
        codes = O.unpack_rows_any(qw.numpy(), bits)
        assert np.array_equal(codes, raw.clamp(0, max_q).numpy().astype(np.uint8)), bits
        assert np.array_equal(O.unpack_cols_any(qz.numpy(), bits), np.full((1, 32), zero, dtype=np.uint8))
