"""bench.py --model llama3-70b: BASELINE.json configs[4] -- Llama-3-70B GPTQ int4 g128 batch-1 decode, tensor parallel over
all ranks of the node (launched by torchrun, one process per GPU, RCCL over xGMI).

Megatron split (gptqmodel_amd/utils/tp.py): qkv / gate_up column-parallel (N / tp, no communication), o / down
row-parallel (K / tp): each rank's kernel returns UNROUNDED fp32 partial sums, ONE all-reduce(sum) per row-parallel layer
(2 per decoder layer, 160 per token, 32 KB each at M=1 -- latency-bound), then the reference's single rounding.
The linears of a token are chained through the layer glue like the 8B headline (true data dependencies).
value = tokens/s of the ONE token stream the TP group produces (strong scaling).  tp=1 uses the decode chain.
"""
from __future__ import annotations

import json
import os
import time

import torch


def _run_tp1_chain(args, dev):
    """One GPU holds the whole 70B model (35.6 GB packed): the decode chain with fused glue, like the 8B headline."""
    import bench as B
    from gptqmodel_amd.utils.decode_chain import DecodeStep
    cfg = B.LLAMA3_70B
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321)
    layers = B.build_stack(cfg, lambda k, n: B.make_gptq(k, n, 128, dev, gen, dtype), dev, gen, dtype)
    step = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
    step.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
    stream = torch.cuda.Stream(device=dev)
    t0 = time.perf_counter()
    ms, g = B.time_graph(step.run, stream, args.steps, args.warmup)
    if not torch.isfinite(step.out).all():
        raise SystemExit("bench_tp: non-finite activations")
    n_launch = cfg["layers"] * 4
    step_bytes, step_flops = B.model_bytes_flops(cfg)
    gbs = step_bytes / (ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "llama3_70b_gptq_int4_g128_decode_linear_stack_tokens_per_s", "value": 1e3 / ms, "unit": "tokens/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
        "config": {"workload": "Llama-3-70B GPTQ int4 g128 batch=1 decode: 560 quantised linears per token (x 80 layers) as a dependent "
                               "chain with the layer glue fused into the GEMVs, M=1, TP=1 (35.6 GB of packed weights on one GPU)",
                   "parallelism": "tp1", "launches_per_step": n_launch, "graph": True, "weight_bytes_per_token": step_bytes},
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / B.HBM_PEAK_GBS,
                     "traffic": None, "traffic_source": "none", "kernel": "gptqhip::skinny_kernel<...,GLUE> (decode op)",
                     "bytes_per_launch": step_bytes / n_launch, "avg_launch_us": ms * 1e3 / n_launch},
        "gemm_tflops_equiv": step_flops / (ms * 1e-3) / 1e12}), flush=True)


def run_70b(args, rank, local_rank, world, dev, dist):
    import bench as B
    from gptqmodel_amd.utils.tp import _bounds
    cfg = B.LLAMA3_70B
    tp = world
    if tp == 1:
        return _run_tp1_chain(args, dev)
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    gs = 128
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    h, inter, q, kv = cfg["hidden"], cfg["inter"], cfg["q"], cfg["kv"]
    for total, mult, what in ((q + 2 * kv, 8, "qkv out"), (2 * inter, 8, "gate_up out"), (q, gs, "o_proj in"), (inter, gs, "down in")):
        _bounds(total, 0, tp, mult, what)
    # shards generated directly at their sharded shapes: [K, N/tp] column-parallel, [K/tp, N] row-parallel
    shapes = [(h, (q + 2 * kv) // tp), (q // tp, h), (h, 2 * inter // tp), (inter // tp, h)]
    layers = []
    eps = 1e-5
    for _ in range(cfg["layers"]):
        lins = [B.make_gptq(k, n, gs, dev, gen, dtype) for k, n in shapes]
        nw = [(1.0 + 0.1 * torch.randn(h, device=dev, generator=gen)).to(dtype) for _ in range(2)]
        layers.append((lins, nw))
    x0 = (torch.randn(h, device=dev, generator=gen) * 0.5).to(dtype)
    if dist is not None:
        dist.broadcast(x0, 0)
    qs, inter_s = q // tp, inter // tp
    # decode messages (hidden * 4 B = 32 KB) go through the one-shot peer-to-peer all-reduce (one kernel, rounding + residual
    # fused, capture-safe: csrc/gptqhip_comm.hip); RCCL all-reduce is the fallback
    comm = None
    if tp > 1 and not os.environ.get("GPTQHIP_BENCH_RCCL"):
        try:
            from gptqmodel_amd.utils.xgmi_allreduce import OneShotAllReduce
            comm = OneShotAllReduce(h, dev)
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"bench_tp: one-shot all-reduce unavailable ({str(e)[:160]}); using RCCL all_reduce", flush=True)
            comm = None
        flag = torch.tensor([1 if comm is not None else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)     # all ranks or none
        if int(flag.item()) == 0:
            comm = None

    chain = None
    if comm is not None and not os.environ.get("GPTQHIP_BENCH_TP_MODULES"):
        # the product path for TP decode: 4 decode ops (glue fused, fp32 partials from the row shards) + 2 one-shot all-reduce
        # kernels (residual + RMSNorm statistics fused) per layer, one graph per token (utils/decode_chain.TPDecodeStep)
        from gptqmodel_amd.utils.decode_chain import DecodeLayer, TPDecodeStep
        dl = []
        for (qkv, o, gu, down), (w_in, w_post) in layers:
            gu.gate_up_interleaved = True   # random columns: this rank's gate|up shard counts as interleaved in blocks of 8
            dl.append(DecodeLayer(qkv, o, gu, down, w_in, w_post))
        chain = TPDecodeStep(dl, h, q // tp, dtype, comm, eps=eps)
        chain.x_in.copy_(x0)

    def rms(v, w):
        v32 = v.float()
        return w * (v32 * torch.rsqrt(v32.pow(2).mean(-1, keepdim=True) + eps)).to(dtype)

    def reduce_add(part, hcur):
        """hidden = residual + act(sum over ranks of the fp32 partials): the reference's single rounding, then the residual add."""
        if comm is not None:
            return comm(part, out_dtype=dtype, residual=hcur.contiguous())
        if tp > 1:
            dist.all_reduce(part)                                  # RCCL over xGMI
        return hcur + part.to(dtype)

    def token_step():
        if chain is not None:
            return chain.run()
        hcur = x0[None]
        for (qkv, o, gu, down), (w_in, w_post) in layers:
            a = qkv(rms(hcur, w_in))[:, :qs]                       # this rank's query heads stand in for its attention output
            part = o.forward_partial(a.contiguous())               # fp32 partial sums over this rank's K-shard
            hcur = reduce_add(part, hcur)
            g = gu(rms(hcur, w_post))
            part = down.forward_partial((torch.nn.functional.silu(g[:, :inter_s]) * g[:, inter_s:]).contiguous())
            hcur = reduce_add(part, hcur)
        return hcur

    stream = torch.cuda.Stream(device=dev)
    graph = None
    with torch.cuda.stream(stream):
        token_step()
        stream.synchronize()
        if not args.no_graph:
            try:   # RCCL collectives are capturable; fall back to eager launches if this build refuses
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    token_step()
                graph = g
            except Exception as e:  # noqa: BLE001
                if rank == 0:
                    print(f"bench_tp: graph capture with RCCL failed ({str(e)[:120]}); eager", flush=True)
                graph = None

    def run(n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                if graph is not None:
                    graph.replay()
                else:
                    token_step()

    run(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tmax = torch.tensor([wall], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    with torch.cuda.stream(stream):
        finite = bool(torch.isfinite(token_step()).all())    # every rank takes part (the step contains all-reduces)
    if comm is not None:
        comm.check_status()
    if rank == 0:
        ms = wall * 1e3 / args.steps
        step_bytes, step_flops = B.model_bytes_flops(cfg)
        n_launch = cfg["layers"] * 4
        if not finite:
            raise SystemExit("bench_tp: non-finite activations")
        gbs = step_bytes / tp / (ms * 1e-3) / 1e9
        out = {
            "metric": "llama3_70b_gptq_int4_g128_decode_linear_stack_tokens_per_s", "value": args.steps / wall, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": f"Llama-3-70B GPTQ int4 g128 batch=1 decode: 560 quantised linears per token (x 80 layers), M=1, "
                                   f"tensor parallel TP={tp} (column qkv/gate_up, row o/down + fp32 all-reduce), true data "
                                   "dependencies through the layer glue, random packed weights",
                       "parallelism": f"tp{tp}", "launches_per_step": n_launch, "graph": graph is not None,
                       "path": "TPDecodeStep (decode ops + fused all-reduce)" if chain is not None else "modules + torch glue",
                       "weight_bytes_per_token": step_bytes, "allreduce_per_token": 2 * cfg["layers"] if tp > 1 else 0,
                       "allreduce": ("one-shot peer-to-peer kernel (gptqhip_allreduce_oneshot)" if comm is not None else
                                     ("RCCL all_reduce" if tp > 1 else "none"))},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / B.HBM_PEAK_GBS,
                         "traffic": None, "traffic_source": "none", "kernel": "gptqhip::skinny_kernel (per rank)",
                         "bytes_per_launch": step_bytes / tp / n_launch, "avg_launch_us": ms * 1e3 / n_launch},
            "gemm_tflops_equiv": step_flops / (ms * 1e-3) / 1e12,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
