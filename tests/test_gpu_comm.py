"""One-shot all-reduce (gptqhip_allreduce_oneshot / utils.xgmi_allreduce.OneShotAllReduce): the protocol -- IPC handle exchange,
peer-mapped pushes, arrival flags, rank-ordered reduction, fused rounding / bias / residual, device-side epochs under graph replay
-- driven by TWO PROCESSES that share GPU 0 through real IPC mappings (the test boxes have one GPU; across physical GPUs the same
code path runs over xGMI, which this test cannot cover)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _expected(parts, bias, residual, dtype):
    s = parts[0].clone()
    for p in parts[1:]:
        s = s + p                       # rank order, fp32
    y = s.to(dtype)
    if bias is not None:
        y = (y.float() + bias.float()).to(dtype)
    if residual is not None:
        y = (residual.float() + y.float()).to(dtype)
    return y


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from helpers import shared_gpu_wait_ms
    os.environ["GPTQHIP_COMM_TIMEOUT_MS"] = str(shared_gpu_wait_ms(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gptqmodel_amd.utils.xgmi_allreduce import OneShotAllReduce
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        n_max = 8192
        comm = OneShotAllReduce(n_max, dev)
        ok = True
        for it, (n, dtype, use_b, use_r) in enumerate([(8192, torch.float16, False, False), (8192, torch.float16, True, True),
                                                         (4096, torch.bfloat16, True, False), (1000, torch.float16, False, True),
                                                         (8192, torch.float16, False, True)] * (3 if world < 8 else 1)):
            g = torch.Generator().manual_seed(1000 * it + rank)
            part = torch.randn(n, generator=g) * 3.0
            gb = torch.Generator().manual_seed(77 + it)
            bias = (torch.randn(n, generator=gb) * 0.1).to(dtype) if use_b else None
            res = torch.randn(n, generator=gb).to(dtype) if use_r else None
            allp = [torch.empty(n) for _ in range(world)]
            dist.all_gather(allp, part)
            want = _expected(allp, bias, res, dtype)
            got = comm(part.to(dev), out_dtype=dtype, bias=None if bias is None else bias.to(dev),
                       residual=None if res is None else res.to(dev))
            torch.cuda.synchronize()
            ok = ok and torch.equal(got.cpu(), want)
        # graph replay: epochs live in device memory, so a captured launch keeps working; inputs change under it
        n, dtype = 8192, torch.float16
        part_dev = torch.zeros(n, device=dev)
        res_dev = torch.zeros(n, device=dev, dtype=dtype)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            comm(part_dev, out_dtype=dtype, residual=res_dev)
            s.synchronize()
            dist.barrier()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                out = comm(part_dev, out_dtype=dtype, residual=res_dev)
            for it in range(25 if world < 8 else 8):
                g = torch.Generator().manual_seed(5000 + 10 * it + rank)
                part = torch.randn(n, generator=g)
                res = torch.randn(n, generator=torch.Generator().manual_seed(9000 + it)).to(dtype)
                allp = [torch.empty(n) for _ in range(world)]
                dist.all_gather(allp, part)
                part_dev.copy_(part)
                res_dev.copy_(res)
                gr.replay()
                s.synchronize()
                ok = ok and torch.equal(out.cpu(), _expected(allp, None, res, dtype))
        comm.check_status()
        # through the tensor-parallel module: RowParallelQuantLinear(comm=...) == all-reduce path, bit for bit
        from gptqmodel_amd.utils import tp

        class Local(torch.nn.Module):
            def __init__(self, w):
                super().__init__()
                self.w = w

            def forward_partial(self, x):
                return x.float() @ self.w

        gw = torch.Generator().manual_seed(31 + rank)
        w = (torch.randn(256, 4096, generator=gw) * 0.05).to(dev)
        x = (torch.randn(1, 256, generator=torch.Generator().manual_seed(3 + rank)) * 0.5).half().to(dev)
        bias = (torch.randn(4096, generator=torch.Generator().manual_seed(8)) * 0.1).half().to(dev)
        y_comm = tp.RowParallelQuantLinear(Local(w), bias=bias, comm=comm)(x)
        parts = [torch.empty(1, 4096) for _ in range(world)]
        dist.all_gather(parts, (x.float() @ w).cpu())
        want = _expected([p[0] for p in parts], bias.cpu(), None, torch.float16)
        ok = ok and torch.equal(y_comm.cpu()[0], want)
        comm.check_status()
        comm.close()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_oneshot_allreduce_processes_sharing_one_gpu(world):
    """world = 4 / 8 exercise what only an 8-GPU node would otherwise reach first: 8 flags per block, 7 peer mappings, every
    `tid < world` lane, the rank-ordered sum over 8 slots."""
    port = 29700 + (os.getpid() % 2000) + 7 * world
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def _tp_chain_worker(rank, world, port, ret):
    """TPDecodeStep on 2 ranks (column shards -> partial-f32 row shards -> one-shot all-reduce with residual + RMSNorm statistics)
    against the same sharded computation from separate launches: plugin forward() / forward_partial() + torch glue."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from helpers import shared_gpu_wait_ms
    os.environ["GPTQHIP_COMM_TIMEOUT_MS"] = str(shared_gpu_wait_ms(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from test_gpu_decode_chain import _make_stack
        from helpers import rel_err, torch_to_f32
        from gptqmodel_amd.utils.decode_chain import TPDecodeStep
        from gptqmodel_amd.utils.xgmi_allreduce import OneShotAllReduce
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dtype, eps = torch.float16, 1e-5
        hidden, inter, q_dim, kv_dim, n_layers = 4096, 14336, 4096, 1024, 2
        # every process builds BOTH ranks' shards (seeded), runs its own through the chain and all of them through the reference
        shards = [_make_stack(n_layers, hidden, inter // world, q_dim // world, kv_dim // world, dtype, seed=50 + r, interleave="all")
                  for r in range(world)]
        for r in range(1, world):
            for li in range(n_layers):   # RMSNorm weights are replicated, not sharded
                shards[r][li].input_norm, shards[r][li].post_norm = shards[0][li].input_norm, shards[0][li].post_norm
        comm = OneShotAllReduce(hidden, dev)
        step = TPDecodeStep(shards[rank], hidden, q_dim // world, dtype, comm, eps=eps)

        def rms(v, w):
            v32 = v.float()
            return w * (v32 * torch.rsqrt(v32.pow(2).mean(-1, keepdim=True) + eps)).to(dtype)

        def reference(x):
            h = x.clone()
            for li in range(n_layers):
                xn = rms(h, shards[0][li].input_norm)[None]
                part = sum(S[li].o.forward_partial(S[li].qkv(xn)[:, :q_dim // world]) for S in shards)
                h = h + part[0].to(dtype)
                xn = rms(h, shards[0][li].post_norm)[None]
                part = sum(S[li].down.forward_partial(torch.nn.functional.silu(S[li].gate(xn)) * S[li].up(xn)) for S in shards)
                h = h + part[0].to(dtype)
            return h

        ok = True
        xs = [(torch.randn(hidden, generator=torch.Generator().manual_seed(40 + i)) * 0.5).to(dtype).to(dev) for i in range(3)]
        want = []
        for x in xs:
            step.x_in.copy_(x)
            got = step.run().clone()
            torch.cuda.synchronize()
            ref = reference(x)
            ok = ok and bool(torch.isfinite(got).all()) and rel_err(torch_to_f32(got), torch_to_f32(ref)) <= 4e-3
            both = [torch.empty(hidden, dtype=dtype) for _ in range(world)]
            dist.all_gather(both, got.cpu())
            ok = ok and all(torch.equal(both[0], b) for b in both)   # rank-ordered reduction: every rank holds the same bits
            want.append(got)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            step.run()
            s.synchronize()
            dist.barrier()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                out = step.run()
            for i in range(12):
                step.x_in.copy_(xs[i % 3], non_blocking=True)
                gr.replay()
                s.synchronize()
                ok = ok and torch.equal(out, want[i % 3])
        comm.check_status()
        comm.close()
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_tp_decode_chain_processes_sharing_one_gpu(world):
    """Llama-3-8B-shaped layers at TP = 2 and TP = 8 (shards 512 + 128 + 128 query / key / value columns, 1792 MLP columns,
    o_proj K = 512, down_proj K = 1792 = 14 x 128)."""
    port = 31700 + (os.getpid() % 2000) + 7 * world
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_tp_chain_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


def test_oneshot_allreduce_world1_is_the_rounding_chain():
    from gptqmodel_amd.utils.xgmi_allreduce import OneShotAllReduce
    dev = torch.device("cuda", 0)
    comm = OneShotAllReduce(4096, dev)
    part = torch.randn(4096, device=dev) * 2.0
    res = torch.randn(4096, device=dev).half()
    bias = (torch.randn(4096, device=dev) * 0.1).half()
    got = comm(part, out_dtype=torch.float16, bias=bias, residual=res)
    assert torch.equal(got, _expected([part], bias, res, torch.float16))
    with pytest.raises(RuntimeError, match="elements"):
        comm(torch.zeros(8192, device=dev))
    comm.check_status()
    # gather + select with an index outside the concatenation: that element is NaN and the status word says so
    xl = torch.randn(512, device=dev).half()
    idx = torch.tensor([0, 511, 512, -1, 7], dtype=torch.int32, device=dev)
    sel = comm.gather_select(xl, idx)
    torch.cuda.synchronize()
    assert torch.equal(sel[[0, 1, 4]], xl[[0, 511, 7]]) and bool(torch.isnan(sel[[2, 3]]).all())
    with pytest.raises(RuntimeError, match="index outside"):
        comm.check_status()
    comm.close()
