"""Tensor-parallel legs of bench.py: BASELINE.json configs[4] -- Llama-3-70B GPTQ int4 g128 batch-1 decode, tensor parallel over
all ranks of the node (one process per GPU, RCCL over xGMI).

Megatron split (gptqmodel_amd/utils/tp.py): qkv / gate_up column-parallel (N / tp, no communication), o / down
row-parallel (K / tp): each rank's kernel returns UNROUNDED fp32 partial sums, ONE all-reduce(sum) per row-parallel layer
(2 per decoder layer, 160 per token, 32 KB each at M=1 -- latency-bound), then the reference's single rounding.
The linears of a token are chained through the layer glue like the 8B headline (true data dependencies).
value = tokens/s of the ONE token stream the TP group produces (strong scaling).  tp=1 uses the decode chain.

Two entry points:
  tp_decode_entry(...)   every rank of the group calls it; returns the configs[] object {"config": "C5", "tp": N, ...} -- bench.py
                         appends it to the 8B headline line whenever it runs on N > 1 ranks, so ONE scaling sweep of the default
                         command also yields the 70B strong-scaling curve the metric names;
  run_70b(...)           `bench.py --model llama3-70b`: the same measurement as the headline of its own JSON line.

All-reduce selection: the one-shot peer-to-peer kernel (csrc/gptqhip_comm.hip) is used only after OneShotAllReduce.self_test()
has compared it bit for bit with the process group's own collective on THIS machine (the build boxes have one GPU: the first
contact with real xGMI links happens here); otherwise -- IPC mapping unavailable, self-test mismatch or time-out -- the step
falls back to module calls + dist.all_reduce (RCCL), graph-captured when the build allows.  The JSON says which path ran.
"""
from __future__ import annotations

import json
import os
import time

import torch


def _build_70b_layers(cfg, shapes, dev, gen, dtype, gs=128):
    """80 layers x 4 (sharded) modules with distinct packed words: one Philox draw per distinct shape, later layers reuse the words
    rotated + xored (distinct bytes in HBM; 35 GB of Philox output would dominate the leg)."""
    import bench as B
    base = {}

    def mk(k, n):
        key = (k, n)
        if key not in base:
            base[key] = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev, generator=gen)
            return B.make_gptq(k, n, gs, dev, gen, dtype, derive_from=base[key].clone())
        salt = int(torch.randint(1, 2**31 - 1, (1,), generator=gen, device=dev).item())
        return B.make_gptq(k, n, gs, dev, gen, dtype, derive_from=torch.roll(base[key], 1 + salt % 97, 0) ^ salt)

    layers = []
    h = cfg["hidden"]
    for _ in range(cfg["layers"]):
        lins = [mk(k, n) for k, n in shapes]
        nw = [(1.0 + 0.1 * torch.randn(h, device=dev, generator=gen)).to(dtype) for _ in range(2)]
        layers.append((lins, nw))
    base.clear()
    return layers


def tp_decode_entry(args, rank, world, dev, dist, steps, warmup, log=None):
    """Time the Llama-3-70B decode step on the `world` ranks of the default process group (collective: every rank calls it).
    Returns the configs[] object (identical on every rank up to the timing all-reduce)."""
    import bench as B
    from gptqmodel_amd.utils.tp import _bounds
    cfg = B.LLAMA3_70B
    tp = world
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    gs = 128
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    h, inter, q, kv = cfg["hidden"], cfg["inter"], cfg["q"], cfg["kv"]
    for total, mult, what in ((q + 2 * kv, 8, "qkv out"), (2 * inter, 8, "gate_up out"), (q, gs, "o_proj in"), (inter, gs, "down in")):
        _bounds(total, 0, tp, mult, what)
    # shards generated directly at their sharded shapes: [K, N/tp] column-parallel, [K/tp, N] row-parallel
    shapes = [(h, (q + 2 * kv) // tp), (q // tp, h), (h, 2 * inter // tp), (inter // tp, h)]
    eps = 1e-5
    t_build = time.perf_counter()
    layers = _build_70b_layers(cfg, shapes, dev, gen, dtype, gs)
    x0 = (torch.randn(h, device=dev, generator=gen) * 0.5).to(dtype)
    if dist is not None and tp > 1:
        if dist.get_backend() == "gloo":
            xc = x0.cpu()
            dist.broadcast(xc, 0)
            x0 = xc.to(dev)
        else:
            dist.broadcast(x0, 0)
    qs, inter_s = q // tp, inter // tp
    comm, comm_note = None, "none (tp=1)"
    if tp > 1:
        comm_note = "RCCL all_reduce (GPTQHIP_BENCH_RCCL set)"
        if not os.environ.get("GPTQHIP_BENCH_RCCL"):
            try:
                from gptqmodel_amd.utils.xgmi_allreduce import OneShotAllReduce
                comm = OneShotAllReduce(h, dev)
                comm_note = "one-shot peer-to-peer kernel (gptqhip_allreduce_oneshot), self-test passed"
            except Exception as e:  # noqa: BLE001
                comm_note = f"RCCL all_reduce (one-shot all-reduce unavailable: {str(e)[:120]})"
                comm = None
            flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
            if dist.get_backend() != "gloo":
                flag = flag.to(dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)     # all ranks or none
            if int(flag.item()) == 0:
                if comm is not None:
                    comm_note = "RCCL all_reduce (one-shot all-reduce unavailable on another rank)"
                comm = None
            if comm is not None and not comm.self_test():
                # bit-for-bit comparison with the process group's own collective failed (or a wait timed out) on this machine
                comm_note = "RCCL all_reduce (one-shot all-reduce FAILED its self-test on this machine)"
                comm = None
    if log:
        log(f"bench_tp: tp={tp} shards built in {time.perf_counter() - t_build:.1f} s; all-reduce: {comm_note}")

    chain = None
    if tp == 1:
        from gptqmodel_amd.utils.decode_chain import DecodeLayer, DecodeStep
        dl = []
        for (qkv, o, gu, down), (w_in, w_post) in layers:
            gu.gate_up_interleaved = True
            dl.append(DecodeLayer(qkv, o, gu, down, w_in, w_post))
        chain = DecodeStep(dl, h, q, dtype, eps=eps)
        chain.x_in.copy_(x0)
    elif comm is not None and not os.environ.get("GPTQHIP_BENCH_TP_MODULES"):
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
        dist.all_reduce(part)                                      # RCCL over xGMI
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
        if dist is not None and tp > 1:
            dist.barrier()
        if not args.no_graph:
            try:   # RCCL collectives are capturable; fall back to eager launches if this build refuses
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    token_step()
                graph = g
            except Exception as e:  # noqa: BLE001
                if log:
                    log(f"bench_tp: graph capture failed ({str(e)[:120]}); eager launches")
                graph = None

    def run(n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                if graph is not None:
                    graph.replay()
                else:
                    token_step()

    run(warmup)
    torch.cuda.synchronize()
    if dist is not None and tp > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    if dist is not None and tp > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if dist is not None and tp > 1:
        tmax = torch.tensor([wall], dtype=torch.float64)
        if dist.get_backend() != "gloo":
            tmax = tmax.to(dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    with torch.cuda.stream(stream):
        finite = bool(torch.isfinite(token_step()).all())    # every rank takes part (the step contains all-reduces)
        stream.synchronize()
    status_ok = True
    if comm is not None:
        try:
            comm.check_status()
        except RuntimeError:
            status_ok = False
    ms = wall * 1e3 / steps
    step_bytes, step_flops = B.model_bytes_flops(cfg)
    n_launch = cfg["layers"] * 4
    gbs = step_bytes / tp / (ms * 1e-3) / 1e9
    entry = {
        "config": "C5", "tp": tp,
        "workload": f"Llama-3-70B GPTQ int4 g128 batch=1 decode: 560 quantised linears per token (x 80 layers), M=1, tensor parallel "
                    f"TP={tp} (column qkv/gate_up, row o/down + fp32 all-reduce), true data dependencies through the layer glue, "
                    "random packed weights",
        "value": steps / wall, "unit": "tokens/s", "ms_per_token": ms, "steps": steps, "warmup": warmup, "scaling": "strong",
        "path": ("DecodeStep (decode ops, glue fused)" if tp == 1 else
                 ("TPDecodeStep (decode ops + fused one-shot all-reduce)" if chain is not None else "modules + torch glue + all_reduce")),
        "graph": graph is not None, "allreduce": comm_note, "allreduce_per_token": 2 * cfg["layers"] if tp > 1 else 0,
        # which exchange actually ran: the hand-written one-shot peer-to-peer kernel over xGMI, RCCL's all_reduce, or none
        "exchange": "none" if tp == 1 else ("oneshot" if comm is not None else "rccl"),
        "finite": finite, "comm_status_ok": status_ok, "weight_bytes_per_token": step_bytes,
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / B.HBM_PEAK_GBS,
                     "traffic": None, "kernel": "gptqhip::skinny1_kernel / skinny_kernel (per rank)", "bytes_per_launch": step_bytes / tp / n_launch,
                     "avg_launch_us": ms * 1e3 / n_launch,
                     "note": "per-GPU HBM rate: each rank streams 1/tp of the packed weights per token"},
        "gemm_tflops_equiv": step_flops / (ms * 1e-3) / 1e12,
    }
    if comm is not None:
        try:
            comm.close()
        except Exception:  # noqa: BLE001
            pass
    del layers, chain, graph
    torch.cuda.empty_cache()
    return entry


def run_70b(args, rank, local_rank, world, dev, dist, ranks_seen=None):
    """bench.py --model llama3-70b: the TP decode step as the headline of its own JSON line."""
    import sys
    log = (lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None
    e = tp_decode_entry(args, rank, world, dev, dist, args.steps, args.warmup, log=log)
    if rank == 0:
        if not e["finite"] or not e["comm_status_ok"]:
            raise SystemExit("bench_tp: non-finite activations / a peer wait timed out; refusing to report a number")
        dtype = "f16" if args.dtype == "fp16" else "bf16"
        out = {
            "metric": "llama3_70b_gptq_int4_g128_decode_linear_stack_tokens_per_s", "value": e["value"], "unit": "tokens/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": e["ms_per_token"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": e["workload"], "parallelism": f"tp{world}", "launches_per_step": 320, "graph": e["graph"], "path": e["path"],
                       "weight_bytes_per_token": e["weight_bytes_per_token"], "allreduce_per_token": e["allreduce_per_token"],
                       "allreduce": e["allreduce"]},
            "roofline": e["roofline"], "gemm_tflops_equiv": e["gemm_tflops_equiv"],
        }
        import bench as B
        B.emit(out)
