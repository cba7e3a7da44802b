"""GPU tests of the plugin classes (HipGptqLinear / HipAwqLinear) behind the BACKEND selector -- the product path
the reference's callers use (make_quant -> post_init -> forward)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden_files, golden_planar, load_golden
from helpers import assert_forward_close, bits_to_f32, bits_to_torch, f32_to_torch, rel_err, synth_gptq, torch_to_bits, torch_to_f32
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_auto_select_and_forward_gptq_v1_checkpoint():
    """v1 checkpoint (zero-1 on disk) -> make_quant(AUTO) -> v1->v2 conversion -> post_init -> forward."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.const import FORMAT, METHOD
    from gptqmodel_amd.utils.model import convert_gptq_v1_to_v2_format, gptqmodel_post_init, make_quant

    K, N, gs = 512, 256, 128
    qweight, qzeros_v2, scales, g_idx = synth_gptq(5, 4, K, N, gs)
    qzeros_v1 = (qzeros_v2.view(np.uint32) - np.uint32(0x11111111)).view(np.int32)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(K, N, bias=False)

    model = Block()
    picked = make_quant(model, ["proj"], bits=4, group_size=gs, desc_act=False, sym=False, backend=BACKEND.AUTO,
                        format=FORMAT.GPTQ, quant_method=METHOD.GPTQ)
    assert picked[0] is HipGptqLinear and isinstance(model.proj, HipGptqLinear)
    sd = {"proj.qweight": torch.from_numpy(qweight), "proj.qzeros": torch.from_numpy(qzeros_v1),
          "proj.scales": f32_to_torch(scales, "fp16"), "proj.g_idx": torch.from_numpy(g_idx)}
    model.load_state_dict(sd)
    model = model.to(DEV)
    convert_gptq_v1_to_v2_format(model, bits=4)
    assert model.proj.qzero_format() == 2
    gptqmodel_post_init(model)
    x = O.round_to(np.random.RandomState(0).randn(2, 3, K).astype(np.float32) * 0.5, "fp16")
    out = model.proj(f32_to_torch(x, "fp16", DEV))
    assert out.shape == (2, 3, N)
    ref = O.forward_gptq(x, qweight, qzeros_v2, scales, g_idx, 4)
    assert_forward_close(torch_to_f32(out), ref, "fp16")
    model.eval()
    with pytest.raises(NotImplementedError):  # inference-only kernel (SUPPORTS_TRAINING=False, qlinear/__init__.py:463-483)
        model.proj.train(True)


def test_gptq_module_act_order_dequantize_weight_bit_exact():
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    g = load_golden("ref_gptq_g128_actorder_fp16.npz")
    K, N = g["qweight"].shape[0] * 8, g["qweight"].shape[1]
    lin = HipGptqLinear(bits=4, group_size=128, sym=False, desc_act=True, in_features=K, out_features=N, bias=False)
    lin.qweight = torch.from_numpy(g["qweight"])
    lin.qzeros = torch.from_numpy(g["qzeros"])
    lin.scales = bits_to_torch(g["scales"], "fp16")
    lin.g_idx = torch.from_numpy(g["g_idx"])
    lin.qzero_format(format=2)
    lin = lin.to(DEV).eval()
    lin.post_init()
    assert lin.perm is not None
    w = lin.dequantize_weight()
    assert np.array_equal(torch_to_bits(w), g["w_ref"].reshape(K, N))
    out = lin(bits_to_torch(g["x"], "fp16", DEV))
    assert_forward_close(torch_to_f32(out), bits_to_f32(g["out_ref"], "fp16"), "fp16")


@pytest.mark.parametrize("name", ["ref_awq_g128_bias_fp16.npz", "ref_awq_g128_bf16.npz"])
def test_awq_module(name):
    from gptqmodel_amd.nn_modules.qlinear.hip_awq import HipAwqLinear
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.const import DEVICE, FORMAT, METHOD
    from gptqmodel_amd.utils.importer import select_quant_linear
    g = load_golden(name)
    act, sdt, gs = str(g["act"]), str(g["scale_dtype"]), int(g["group_size"])
    K, N = g["qweight"].shape[0], g["qweight"].shape[1] * 8
    cls = select_quant_linear(bits=4, group_size=gs, desc_act=False, sym=False, device=DEVICE.ROCM,
                              backend=BACKEND.HIP, format=FORMAT.GEMM, quant_method=METHOD.AWQ)
    assert cls is HipAwqLinear
    lin = cls(bits=4, group_size=gs, sym=False, desc_act=False, in_features=K, out_features=N, bias=bool(g["bias"].size))
    lin.qweight = torch.from_numpy(g["qweight"])
    lin.qzeros = torch.from_numpy(g["qzeros"])
    lin.scales = bits_to_torch(g["scales"], sdt)
    if g["bias"].size:
        lin.bias = bits_to_torch(g["bias"], sdt)
    lin = lin.to(DEV).eval()
    lin.post_init()
    out = lin(bits_to_torch(g["x"], act, DEV))
    assert_forward_close(torch_to_f32(out), bits_to_f32(g["out_ref"], act), act)


def test_graph_capture_replay_matches_eager():
    """The decode loop is replayed as a HIP graph in bench.py: capture must be legal and deterministic."""
    from gptqmodel_amd import ops
    K, N, gs = 4096, 4096, 128
    qweight, qzeros, scales, _ = synth_gptq(9, 4, K, N, gs)
    qw, qz = torch.from_numpy(qweight).to(DEV), torch.from_numpy(qzeros).to(DEV)
    sc = f32_to_torch(scales, "fp16", DEV)
    x = torch.randn((1, K), device=DEV, dtype=torch.float16)
    qw_t, meta = ops.repack_tiled(qw, qz, sc, None, gs, 4)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eager = ops.gemm(x, qw_t, meta, None, None, N, gs, 4, sc.dtype).clone()
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = ops.gemm(x, qw_t, meta, None, None, N, gs, 4, sc.dtype)
        for _ in range(3):
            graph.replay()
        s.synchronize()
    assert torch.equal(out, eager)


def test_fused_siblings_match_separate_projections():
    """q/k/v fused along N (utils.model.fuse_siblings) must reproduce the three separate kernels' outputs."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.model import FusedSiblingView, fuse_siblings, gptqmodel_post_init

    K, gs = 1024, 128
    sizes = {"q_proj": 512, "k_proj": 128, "v_proj": 128}

    def build():
        blk = nn.Module()
        for i, (name, n) in enumerate(sizes.items()):
            qweight, qzeros, scales, g_idx = synth_gptq(100 + i, 4, K, n, gs)
            lin = HipGptqLinear(bits=4, group_size=gs, sym=False, desc_act=False, in_features=K, out_features=n,
                                bias=True, name=name)
            lin.qweight = torch.from_numpy(qweight)
            lin.qzeros = torch.from_numpy(qzeros)
            lin.scales = f32_to_torch(scales, "fp16")
            lin.g_idx = torch.from_numpy(g_idx)
            lin.bias = f32_to_torch(np.random.RandomState(i).randn(n).astype(np.float32) * 0.1, "fp16")
            lin.qzero_format(format=2)
            setattr(blk, name, lin)
        return blk.to(DEV).eval()

    ref_blk, fused_blk = build(), build()
    gptqmodel_post_init(ref_blk)
    group = fuse_siblings(fused_blk, list(sizes))
    assert group is not None and isinstance(fused_blk.k_proj, FusedSiblingView)
    gptqmodel_post_init(fused_blk)
    x = torch.randn(2, 3, K, device=DEV, dtype=torch.float16) * 0.5
    for name, n in sizes.items():
        a, b = getattr(ref_blk, name)(x), getattr(fused_blk, name)(x)
        assert b.shape == (2, 3, n)
        assert rel_err(torch_to_f32(b), torch_to_f32(a)) <= 1e-3
    # one fused launch served all three views of the same input tensor, and nothing stays pinned afterwards
    assert group._out is None and group._x is None


def test_row_parallel_partials_on_one_gpu():
    """TP row-parallel math with the real kernel: two K-shards of one layer, fp32 partial sums (PARTIAL_F32) added
    like the all-reduce does, then the reference rounding chain == the unsharded module."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils import tp

    K, N, gs, M = 2048, 512, 128, 5
    qweight, qzeros, scales, g_idx = synth_gptq(77, 4, K, N, gs)
    t = {"qweight": torch.from_numpy(qweight), "qzeros": torch.from_numpy(qzeros),
         "scales": f32_to_torch(scales, "fp16"), "g_idx": torch.from_numpy(g_idx), "bias": None}
    bias = f32_to_torch(np.random.RandomState(1).randn(N).astype(np.float32) * 0.1, "fp16", DEV)

    def module(tt, k):
        lin = HipGptqLinear(bits=4, group_size=gs, sym=False, desc_act=False, in_features=k, out_features=N, bias=False)
        lin.qweight, lin.qzeros, lin.scales, lin.g_idx = tt["qweight"], tt["qzeros"], tt["scales"], tt["g_idx"]
        lin.qzero_format(format=2)
        lin = lin.to(DEV).eval()
        lin.post_init()
        return lin

    x = O.round_to(np.random.RandomState(2).randn(M, K).astype(np.float32) * 0.5, "fp16")
    xt = f32_to_torch(x, "fp16", DEV)
    shards = [module(tp.shard_gptq_row(t, r, 2, 4, gs), K // 2) for r in range(2)]
    partial = sum(sh.forward_partial(xt[:, r * K // 2:(r + 1) * K // 2].contiguous()) for r, sh in enumerate(shards))
    assert partial.dtype == torch.float32
    y_tp = partial.to(torch.float16) + bias
    ref = O.forward_gptq(x, qweight, qzeros, scales, g_idx, 4, torch_to_f32(bias))
    assert_forward_close(torch_to_f32(y_tp), ref, "fp16")
    row = tp.RowParallelQuantLinear(shards[0], bias=None)  # world size 1 / no process group: no collective
    assert row(xt[:, :K // 2].contiguous()).dtype == torch.float16


def test_device_packer_bit_exact_and_roundtrip():
    """gptqhip_pack_gptq == the reference packer (golden from pack_block) bit for bit; and a module packed on the
    device then run through post_init/forward reproduces x @ W_quantised."""
    from gptqmodel_amd import ops
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    g = load_golden("ref_pack.npz")
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files if k.endswith("_qweight")})
    for t in tags:
        bits = int(t.split("_")[0][1:])
        qw, qz = ops.pack_gptq(torch.from_numpy(g[t + "_weight"]).to(DEV), torch.from_numpy(g[t + "_scales"]).to(DEV),
                               torch.from_numpy(g[t + "_zeros"]).to(DEV), torch.from_numpy(g[t + "_g_idx"]).to(DEV), bits)
        assert np.array_equal(qw.cpu().numpy(), g[t + "_qweight"]), t
        assert np.array_equal(qz.cpu().numpy(), g[t + "_qzeros"]), t
    g2 = load_golden("ref_pack_bits.npz")      # every other width / layout the reference's pack_block writes
    for t in sorted({k.rsplit("_", 1)[0] for k in g2.files if k.endswith("_qweight")}):
        bits, planar = int(t.split("_")[0][1:]), t.endswith("_p")
        qw, qz = ops.pack_gptq(torch.from_numpy(g2[t + "_weight"]).to(DEV), torch.from_numpy(g2[t + "_scales"]).to(DEV),
                               torch.from_numpy(g2[t + "_zeros"]).to(DEV), torch.from_numpy(g2[t + "_g_idx"]).to(DEV), bits, planar=planar)
        assert np.array_equal(qw.cpu().numpy(), g2[t + "_qweight"]), t
        assert np.array_equal(qz.cpu().numpy(), g2[t + "_qzeros"]), t
    # larger random problem vs the oracle packer + end-to-end through the module API
    K, N, gs, bits = 1024, 256, 128, 4
    rng = np.random.RandomState(3)
    lin = nn.Linear(K, N, bias=True)
    scales = torch.rand(N, K // gs) * 0.01 + 0.005
    zeros = torch.randint(0, 16, (N, K // gs)).float()
    g_idx = torch.arange(K, dtype=torch.int32) // gs
    mod = HipGptqLinear(bits=bits, group_size=gs, sym=False, desc_act=False, in_features=K, out_features=N, bias=True,
                        register_buffers=False)
    mod.pack_block(lin, scales.clone(), zeros.clone(), g_idx.clone())
    e_qw, e_qz = O.quantize_pack_gptq(lin.weight.detach().numpy(), scales.T.contiguous().numpy(),
                                      zeros.T.contiguous().numpy().astype(np.int32), g_idx.numpy(), bits)
    assert np.array_equal(mod.qweight.cpu().numpy(), e_qw) and np.array_equal(mod.qzeros.cpu().numpy(), e_qz)
    mod.eval()
    mod.post_init()
    x = O.round_to(rng.randn(3, K).astype(np.float32) * 0.5, "fp16")
    out = mod(f32_to_torch(x, "fp16", DEV))
    ref = O.forward_gptq(x, e_qw, e_qz, torch_to_f32(mod.scales), g_idx.numpy(), bits,
                         torch_to_f32(mod.bias))
    assert_forward_close(torch_to_f32(out), ref, "fp16")
    with pytest.raises(IndexError):
        ops.pack_gptq(lin.weight.detach().to(DEV), scales.T.contiguous().to(DEV), zeros.T.contiguous().to(DEV),
                      torch.full((K,), 99, dtype=torch.int32, device=DEV), bits)


@pytest.mark.parametrize("bits", [4, 8, 3, 6])
def test_device_packer_equals_host_packer_at_layer_size(bits):
    """16.8 M codes of a 4096 x 4096 layer through both packers: every word equal.  (Small fixtures cannot see a one-in-tens-of-
    thousands divergence: the device packer once fused (zero * scale) + weight into an fma -- the reference rounds twice -- and
    flipped 1 code of 32768 at 8 bits; the host packer is the reference's arithmetic, `profiles/r04_host_packer_speed.txt`.)"""
    from gptqmodel_amd import ops
    K = N = 4096
    gs = 128
    gen = torch.Generator().manual_seed(40 + bits)
    w = torch.empty(N, K).uniform_(-0.1, 0.1, generator=gen)
    scales = torch.empty(K // gs, N).uniform_(0.0004, 0.02, generator=gen)
    zeros = torch.randint(0, 1 << bits, (K // gs, N), generator=gen, dtype=torch.int32)
    g_idx = (torch.randperm(K, generator=gen) // gs).to(torch.int32)
    hw, hz = ops.pack_gptq_host(w, scales, zeros, g_idx, bits)
    dw, dz = ops.pack_gptq(w.to(DEV), scales.to(DEV), zeros.to(DEV), g_idx.to(DEV), bits)
    assert torch.equal(dw.cpu(), hw) and torch.equal(dz.cpu(), hz)


@pytest.mark.parametrize("bits", [2, 3, 4, 5, 6, 7, 8])
def test_pack_clamps_reconstructed_codes_like_the_reference_test(bits):
    """tests/test_pack.py:175-238 for the device packer through the class: raw codes -8..23 plus +-1e20 saturate to [0, maxq] before
    the integer conversion; dequantize_weight() of the packed module is exactly (clamped code - zero)."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    max_q, zero = 2 ** bits - 1, 2 ** (bits - 1)
    raw = torch.arange(-8, 24, dtype=torch.float32).view(32, 1).expand(32, 32).contiguous()
    raw[0].fill_(-1e20)
    raw[-1].fill_(1e20)
    linear = nn.Linear(32, 32, bias=False)
    linear.weight.data.copy_((raw - zero).T)
    q = HipGptqLinear(bits=bits, group_size=32, sym=True, desc_act=False, in_features=32, out_features=32, bias=False)
    q.pack_block(linear, torch.ones(32, 1), torch.full((32, 1), zero, dtype=torch.int32), torch.zeros(32, dtype=torch.int32))
    expected = (raw.clamp(0, max_q) - zero).to(torch.float16)
    assert torch.equal(q.dequantize_weight().cpu(), expected)
    q.eval()
    q.post_init()
    assert torch.equal(q.dequantize_weight().cpu(), expected)


@pytest.mark.parametrize("bits,desc_act", [(4, False), (4, True), (8, False), (3, True), (6, False)])
def test_quant_embeddings_match_reference_semantics(bits, desc_act):
    """HipQuantEmbeddings.forward(ids) == F.embedding(ids, dequantize_weight()) -- the reference's TorchQuantEmbeddings
    (torch.py:764-797) -- bit for bit, without materialising the table."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipQuantEmbeddings
    V, D, gs = 1024, 256, 128   # in_features = num_embeddings, out_features = dim
    qweight, qzeros, scales, g_idx = synth_gptq(64 + bits, bits, V, D, gs, desc_act=desc_act)
    emb = HipQuantEmbeddings(bits=bits, group_size=gs, sym=False, desc_act=desc_act, in_features=V, out_features=D, bias=False)
    emb.qweight, emb.qzeros = torch.from_numpy(qweight), torch.from_numpy(qzeros)
    emb.scales, emb.g_idx = f32_to_torch(scales, "fp16"), torch.from_numpy(g_idx)
    emb.qzero_format(format=2)
    emb = emb.to(DEV).eval()
    emb.post_init()
    ids = torch.tensor([[0, 1, 127, 128], [1023, 512, 5, 5]], device=DEV)
    out = emb(ids)
    assert out.shape == (2, 4, D) and out.dtype == torch.float16
    table = O.dequant_gptq(qweight, qzeros, scales, g_idx, bits)           # [V, D], bit-exact with the reference
    exp = table[ids.cpu().numpy()]
    assert np.array_equal(torch_to_f32(out), exp)
    assert np.array_equal(torch_to_bits(emb.dequantize_weight()), table.astype(np.float16).view(np.uint16))
    with pytest.raises(IndexError):
        emb(torch.tensor([V], device=DEV))


def test_dequantize_model_replaces_modules_incl_fused_groups():
    """dequantize_model (mirror of nn_modules/qlinear/torch.py:736-761): dense nn.Linear twins reproduce the quantised
    modules' outputs (same dequantised weights), fused q/k/v groups are split back."""
    import torch.nn as nn
    from gptqmodel_amd.utils.model import dequantize_model, fuse_siblings, gptqmodel_post_init, make_quant

    class Attn(nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = nn.Linear(256, 128, bias=True)
            self.k_proj = nn.Linear(256, 64, bias=True)
            self.v_proj = nn.Linear(256, 64, bias=True)
            self.o_proj = nn.Linear(128, 256, bias=False)
    torch.manual_seed(3)
    blk = Attn().half().cuda()
    floats = {n: m for n, m in blk.named_children()}
    make_quant(blk, list(floats), bits=4, group_size=64, desc_act=False, sym=False)
    for n, lin in floats.items():
        qm = getattr(blk, n)
        w = lin.weight.data.float().reshape(lin.out_features, lin.in_features // 64, 64)
        scales = ((w.amax(2) - w.amin(2)).clamp(min=1e-5) / 15).half().float()
        zeros = torch.round(-w.amin(2) / scales).clamp(0, 15)
        qm.pack(lin, scales, zeros, (torch.arange(lin.in_features) // 64).to(torch.int32))
    assert fuse_siblings(blk, ["q_proj", "k_proj", "v_proj"]) is not None
    gptqmodel_post_init(blk)
    x = (torch.randn(5, 256, device="cuda") * 0.5).half()
    xo = (torch.randn(5, 128, device="cuda") * 0.5).half()
    ref = {n: getattr(blk, n)(x).float().cpu() for n in ("q_proj", "k_proj", "v_proj")}
    ref["o_proj"] = blk.o_proj(xo).float().cpu()
    dequantize_model(blk, device="cuda")
    assert all(type(getattr(blk, n)) is nn.Linear for n in ("q_proj", "k_proj", "v_proj", "o_proj"))
    assert not any("fused" in n for n, _ in blk.named_children())
    for n in ("q_proj", "k_proj", "v_proj"):
        got = getattr(blk, n)(x).float().cpu()
        assert (got - ref[n]).abs().max() <= 2e-3 * ref[n].abs().max() + 1e-3
    assert (blk.o_proj(xo).float().cpu() - ref["o_proj"]).abs().max() <= 2e-3 * ref["o_proj"].abs().max() + 1e-3


def test_lora_adapter_hook_adds_low_rank_update():
    """The adapter hook of the reference forward (torch.py:344-345; adapter/adapter.py:148-173): out += (x @ A) @ B on top
    of the kernel's result, for 2-D and batched inputs."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.adapter import Lora
    K, N, gs, r = 256, 128, 64, 8
    qweight, qzeros, scales, g_idx = synth_gptq(21, 4, K, N, gs)
    torch.manual_seed(1)
    A = (torch.randn(K, r) * 0.05).half()
    B = (torch.randn(r, N) * 0.05).half()

    def build(adapter):
        lin = HipGptqLinear(bits=4, group_size=gs, sym=False, desc_act=False, in_features=K, out_features=N, bias=False,
                            adapter=adapter, register_buffers=False)
        lin.qweight = torch.from_numpy(qweight).cuda()
        lin.qzeros = torch.from_numpy(qzeros).cuda()
        lin.scales = f32_to_torch(scales, "fp16", "cuda")
        lin.g_idx = torch.from_numpy(g_idx).cuda()
        lin.bias = None
        lin.qzero_format(format=2)
        lin.eval()
        lin.post_init()
        return lin

    plain, lora = build(None), build(Lora(rank=r, lora_A=A, lora_B=B))
    for shape in ((3, K), (2, 5, K)):
        x = (torch.randn(*shape, device="cuda") * 0.5).half()
        want = plain(x).float() + ((x.reshape(-1, K) @ A.cuda()) @ B.cuda()).float().reshape(*shape[:-1], N)
        got = lora(x).float()
        assert got.shape == want.shape
        assert (got - want).abs().max() <= 2e-3 * want.abs().max() + 2e-3


def test_v1_checkpoint_reaching_post_init_unconverted_is_converted_there():
    """make_quant(format=GPTQ) -> load_state_dict -> gptqmodel_post_init WITHOUT the explicit
    v1->v2 conversion step must not run with every zero-point off by one: post_init converts (the reference loader
    always converts before post_init, models/loader.py:1658-1675).  A second post_init is a no-op."""
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.const import FORMAT, METHOD
    from gptqmodel_amd.utils.model import gptqmodel_post_init, make_quant
    K, N, gs = 512, 256, 128
    qweight, qzeros_v2, scales, g_idx = synth_gptq(15, 4, K, N, gs)
    qzeros_v1 = (qzeros_v2.view(np.uint32) - np.uint32(0x11111111)).view(np.int32)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.proj = nn.Linear(K, N, bias=False)

    model = Block()
    make_quant(model, ["proj"], bits=4, group_size=gs, desc_act=False, sym=False, backend=BACKEND.AUTO,
               format=FORMAT.GPTQ, quant_method=METHOD.GPTQ)
    model.load_state_dict({"proj.qweight": torch.from_numpy(qweight), "proj.qzeros": torch.from_numpy(qzeros_v1),
                           "proj.scales": f32_to_torch(scales, "fp16"), "proj.g_idx": torch.from_numpy(g_idx)})
    model = model.to(DEV)
    gptqmodel_post_init(model)
    assert model.proj.qzero_format() == 2
    gptqmodel_post_init(model)  # idempotent
    x = O.round_to(np.random.RandomState(0).randn(3, K).astype(np.float32) * 0.5, "fp16")
    ref = O.forward_gptq(x, qweight, qzeros_v2, scales, g_idx, 4)
    assert_forward_close(torch_to_f32(model.proj(f32_to_torch(x, "fp16", DEV))), ref, "fp16")


def test_derived_tensors_are_buffers_and_state_dict_refuses_kernel_layout():
    """meta / perm are non-persistent buffers (module.to() moves them, list_buffers() lists them);
    after post_init `qweight` holds tile-major words, so state_dict() must refuse instead of writing an unloadable
    checkpoint."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    K, N, gs = 512, 128, 128
    qweight, qzeros, scales, g_idx = synth_gptq(3, 4, K, N, gs, desc_act=True)
    lin = HipGptqLinear(bits=4, group_size=gs, sym=False, desc_act=True, in_features=K, out_features=N, bias=False,
                        register_buffers=True)
    lin.load_state_dict({"qweight": torch.from_numpy(qweight), "qzeros": torch.from_numpy(qzeros),
                         "scales": f32_to_torch(scales, "fp16"), "g_idx": torch.from_numpy(g_idx)})
    assert set(lin.state_dict()) == {"qweight", "qzeros", "scales", "g_idx"}   # checkpoint layout: saving is fine
    lin.qzero_format(format=2)
    lin = lin.to(DEV).eval()
    lin.post_init()
    bufs = dict(lin.named_buffers())
    assert bufs["meta"].is_cuda and bufs["perm"].is_cuda and len(lin.list_buffers()) == 6
    with pytest.raises(RuntimeError, match="tile-major"):
        lin.state_dict()
    x = O.round_to(np.random.RandomState(1).randn(2, K).astype(np.float32) * 0.5, "fp16")
    ref = O.forward_gptq(x, qweight, qzeros, scales, g_idx, 4)
    assert_forward_close(torch_to_f32(lin(f32_to_torch(x, "fp16", DEV))), ref, "fp16")


def test_awq_forward_partial_matches_unrounded_product():
    """HipAwqLinear implements forward_partial (row-parallel AWQ shards need it)."""
    from gptqmodel_amd.nn_modules.qlinear.hip_awq import HipAwqLinear
    g = load_golden("ref_awq_g128_fp16.npz")
    K, N = g["qweight"].shape[0], g["qweight"].shape[1] * 8
    lin = HipAwqLinear(bits=4, group_size=128, sym=False, desc_act=False, in_features=K, out_features=N, bias=False)
    lin.qweight = torch.from_numpy(g["qweight"])
    lin.qzeros = torch.from_numpy(g["qzeros"])
    lin.scales = bits_to_torch(g["scales"], "fp16")
    lin = lin.to(DEV).eval()
    lin.post_init()
    x = bits_to_torch(g["x"], "fp16", DEV)
    part = lin.forward_partial(x)
    assert part.dtype == torch.float32
    w = bits_to_f32(g["w_ref"], "fp16").reshape(K, N).astype(np.float64)
    ref = bits_to_f32(g["x"], "fp16").astype(np.float64) @ w
    assert rel_err(part.cpu().numpy(), ref.astype(np.float32)) <= 1e-5
    assert torch.equal(part.to(torch.float16), lin(x))  # one rounding of the same accumulators


OTHER_BITS_GOLDEN = [n for n in golden_files("ref_gptq_w") if n[len("ref_gptq_w")] in "23567"]


@pytest.mark.parametrize("name", OTHER_BITS_GOLDEN)
def test_other_bit_widths_through_the_plugin_class(name):
    """2- / 3-bit (continuous) and 5- / 6- / 7-bit (planar) checkpoints through HipGptqLinear -- post_init widens the
    codes to the 4- / 8-bit kernel layout (same values), dequantize_weight() is BIT-EXACT with the reference's generic
    dequantize_weight (gptqmodel/nn_modules/qlinear/__init__.py:977-1100) before and after post_init, forward() is inside the
    reference's tolerances, a v1 checkpoint (zero-1 on disk) converts to the same tensors, AUTO selection picks the class."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd.utils.backend import BACKEND
    from gptqmodel_amd.utils.const import DEVICE, FORMAT, METHOD
    from gptqmodel_amd.utils.importer import select_quant_linear
    from gptqmodel_amd.utils.model import convert_gptq_v1_to_v2_format_module
    g = load_golden(name)
    bits, act, sdt, gs = int(g["bits"]), str(g["act"]), str(g["scale_dtype"]), int(g["group_size"])
    K, N = g["g_idx"].shape[0], g["scales"].shape[1]
    desc = not np.array_equal(g["g_idx"], np.arange(K) // gs)
    # FORMAT.GPTQ_P: what the reference's config declares for 5 / 6 / 7-bit checkpoints (quantization/config.py:2660-2676) and for
    # split-plane 3-bit ones; the zero-points are stored as they are (no v1 shift)
    planar3 = bool(golden_planar(g))
    ckpt_format = FORMAT.GPTQ_P if (planar3 or bits in (5, 6, 7)) else FORMAT.GPTQ_V2
    cls = select_quant_linear(bits=bits, group_size=gs, desc_act=desc, sym=False, device=DEVICE.ROCM, backend=BACKEND.AUTO,
                              format=ckpt_format, quant_method=METHOD.GPTQ)
    assert cls is HipGptqLinear

    def module(qzeros, fmt):
        lin = cls(bits=bits, group_size=gs, sym=False, desc_act=desc, in_features=K, out_features=N, bias=bool(g["bias"].size),
                  format=FORMAT.GPTQ if fmt == 1 else ckpt_format)
        if g["bias"].size:
            lin.bias = bits_to_torch(g["bias"], act)
        assert tuple(lin.qweight.shape) == g["qweight"].shape and tuple(lin.qzeros.shape) == g["qzeros"].shape
        assert lin.kernel_bits == (4 if bits <= 4 else 8) and lin.planar == (bits in (5, 6, 7) or (planar3 and fmt == 2))
        lin.qweight, lin.qzeros = torch.from_numpy(g["qweight"]), torch.from_numpy(qzeros)
        lin.scales, lin.g_idx = bits_to_torch(g["scales"], sdt), torch.from_numpy(g["g_idx"])
        lin.qzero_format(format=fmt)
        return lin.to(DEV).eval()

    lin = module(g["qzeros"], 2)
    assert np.array_equal(torch_to_bits(lin.dequantize_weight()), g["w_ref"].reshape(K, N))     # checkpoint layout
    lin.post_init()
    assert (lin.perm is not None) == desc
    assert np.array_equal(torch_to_bits(lin.dequantize_weight()), g["w_ref"].reshape(K, N))     # kernel layout, act-order undone
    out = lin(bits_to_torch(g["x"], act, DEV))
    torch.cuda.synchronize()
    assert_forward_close(torch_to_f32(out), bits_to_f32(g["out_ref"], act), act)

    if planar3:        # (a v1 file is FORMAT.GPTQ = continuous 3-bit words: nothing to convert in a gptq_p checkpoint)
        return
    if bits == 2:      # what the reference's writer stores (utils/model.py:910-911: a WORD subtract, fields borrow from each other)
        qz1 = (g["qzeros"].view(np.uint32) - np.uint32(0x55555555)).view(np.int32)
    else:              # 3 / 5 / 6 / 7 bits: the decoded zero-points minus one, modulo 2^bits (:912-939)
        z1 = (O.unpack_cols_any(g["qzeros"], bits).astype(np.int32) - 1) & ((1 << bits) - 1)
        qz1 = O.pack_cols_any(z1.astype(np.uint8), bits)
    v1 = module(qz1, 1)
    convert_gptq_v1_to_v2_format_module(v1, bits=bits, pack_dtype=torch.int32)
    assert v1.qzero_format() == 2 and np.array_equal(v1.qzeros.cpu().numpy(), g["qzeros"])
    v1.post_init()
    assert torch.equal(v1(bits_to_torch(g["x"], act, DEV)), out)



def test_three_bit_needs_multiples_of_32():
    """3-bit packs 32 codes into three words (qlinear/__init__.py:1001-1043): in / out features must be multiples of 32."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    with pytest.raises(NotImplementedError):
        HipGptqLinear(bits=3, group_size=128, sym=False, desc_act=False, in_features=256, out_features=72, bias=False)
