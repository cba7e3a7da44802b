#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X GPTQ/AWQ dequant-matmul backend.

Workload (BASELINE.json configs[1]): Llama-3-8B GPTQ int4 group_size=128 desc_act=False, batch=1 decode.
One "step" = one pass of the hot path over one token: the 224 quantised linears of the model
(32 layers x {q,k,v,o,gate,up,down}), M=1, fp16, synthetic random packed weights of the real shapes
(3.63 GB of distinct packed weight + scale/zero bytes, already resident in HBM), executed through the
product path (HipGptqLinear.forward -> libgptqhip.so) and replayed as ONE captured HIP graph per token.
value = tokens/s of that path (whole job; replicas x N for --gpus N: the 8B model fits one GPU so ranks
are independent replicas, no data-path collective -- SURVEY.md §8e).

Extra objects on the JSON line (tier contract):
  roofline      dominant kernel = skinny fused dequant-GEMM; achieved = algorithmic bytes per launch
                (K*N/2 + G*N*2 + G*N/2 + M*(K+N)*2, averaged over the 224 launches) / average
                launch duration measured with HIP events on the launch stream over the timed region.
  prefill       the TFLOPS half of BASELINE.json's metric: one decoder layer's launches of the same modules at M=8192
                tokens through the MFMA-bound prefill kernel (outside the timed region), vs the dense fp16 MFMA peak.
  cpu_baseline  the oracle's torch-CPU port of BACKEND.TORCH (oracle/gptq_oracle.py:torch_cpu_forward_gptq)
                timed on this host's cores on a bounded sample (one decoder layer), rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16

LLAMA3_8B = dict(hidden=4096, inter=14336, q=4096, kv=1024, layers=32)
LLAMA3_70B = dict(hidden=8192, inter=28672, q=8192, kv=1024, layers=80)


def layer_shapes(cfg):
    h, i = cfg["hidden"], cfg["inter"]
    return [("q_proj", h, cfg["q"]), ("k_proj", h, cfg["kv"]), ("v_proj", h, cfg["kv"]), ("o_proj", cfg["q"], h),
            ("gate_proj", h, i), ("up_proj", h, i), ("down_proj", i, h)]


def launch_shapes(cfg, fuse):
    """Kernel launches per decoder layer.  With sibling fusion (gptqmodel_amd.utils.model.
    fuse_siblings) q/k/v and gate/up -- which read the same x -- are concatenated along N: 7 linears, 4 launches."""
    if not fuse:
        return [(k, n, 1) for _, k, n in layer_shapes(cfg)]
    h, i = cfg["hidden"], cfg["inter"]
    return [(h, cfg["q"] + 2 * cfg["kv"], 3), (cfg["q"], h, 1), (h, 2 * i, 2), (i, h, 1)]


def algorithmic_bytes(m, k, n, gs=128):
    g = k // gs
    return k * n // 2 + g * n * 2 + g * n // 2 + m * (k + n) * 2


def make_linear(k, n, gs, device, gen, dtype=torch.float16):
    """Synthetic GPTQ-v2 tensors: random int32 qweight, scales rand*0.01+0.005, sym zeros 0x88888888."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    lin = HipGptqLinear(bits=4, group_size=gs, sym=True, desc_act=False, in_features=k, out_features=n, bias=False,
                        register_buffers=False)
    lin.qweight = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=device, generator=gen)
    lin.qzeros = torch.full((k // gs, n // 8), -2004318072, dtype=torch.int32, device=device)  # 0x88888888
    lin.scales = (torch.rand((k // gs, n), device=device, generator=gen) * 0.01 + 0.005).to(dtype)
    lin.g_idx = (torch.arange(k, device=device, dtype=torch.int32) // gs)
    lin.bias = None
    lin.qzero_format(format=2)
    lin.eval()
    lin.post_init()
    return lin


def prefill_tflops(layer, dtype, dev, m=8192, iters=5):
    """The TFLOPS half of BASELINE.json's metric (methodology of scripts/benchmark_marlin_a100.py: 2*M*K*N / t): one
    decoder layer's launches of the SAME modules at M = 8192 tokens (4 x 2048-token sequences), MFMA-bound prefill
    kernel, reported next to the dense fp16/bf16 MFMA peak.  Not part of the timed decode region."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    xs = {}
    for lin, _ in layer:
        if lin.in_features not in xs:
            xs[lin.in_features] = (torch.randn((m, lin.in_features), device=dev, generator=gen) * 0.5).to(dtype)
    def run():
        for lin, _ in layer:
            lin(xs[lin.in_features])
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = sum(2.0 * m * lin.in_features * lin.out_features for lin, _ in layer)
    tf = flops / ms / 1e9
    return {"workload": f"one Llama-3-8B decoder layer's quantised linears ({len(layer)} launches) at M={m} tokens",
            "tflops": tf, "ms": ms, "bound": "mfma", "peak": MFMA_PEAK_TFLOPS, "frac": tf / MFMA_PEAK_TFLOPS,
            "kernel": "gptqhip::tiled_kernel<BITS=4,...,BM=256,D=2>"}


def cpu_baseline(cfg, gs=128, budget_s=20.0):
    """Oracle torch-CPU port of the reference BACKEND.TORCH forward on ONE decoder layer's 7 linears, M=1 fp16...
    timed on all host cores; extrapolated x layers to tokens/s."""
    from oracle.gptq_oracle import torch_cpu_forward_gptq
    torch.manual_seed(1234)
    mods = []
    for _, k, n in layer_shapes(cfg):
        qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32)
        qz = torch.full((k // gs, n // 8), -2004318072, dtype=torch.int32)
        sc = (torch.rand((k // gs, n)) * 0.01 + 0.005).to(torch.bfloat16)
        gi = (torch.arange(k, dtype=torch.int32) // gs)
        x = (torch.randn(1, k) * 0.5).to(torch.bfloat16)  # upstream's CPU test runs bf16 (tests/test_q4_torch.py:27,50)
        mods.append((x, qw, qz, sc, gi))
    def one_pass():
        for x, qw, qz, sc, gi in mods:
            torch_cpu_forward_gptq(x, qw, qz, sc, gi, 4)
    one_pass()
    iters, t0 = 0, time.perf_counter()
    while True:
        one_pass()
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 50:
            break
    per_layer = el / iters
    return {
        "value": 1.0 / (per_layer * cfg["layers"]), "unit": "tokens/s", "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"1 of {cfg['layers']} decoder layers (7 linears, M=1, bf16 like upstream's CPU test), {iters} passes "
                  f"in {el:.1f}s, extrapolated x{cfg['layers']}; host os.cpu_count()={os.cpu_count()}",
        "ms_per_layer": per_layer * 1e3,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b"],
                    help="llama3-8b: the headline config, ranks are independent replicas (weak scaling). "
                         "llama3-70b: BASELINE configs[4], tensor parallel over all ranks (strong scaling): column-parallel "
                         "qkv/gate_up, row-parallel o/down with one fp32 RCCL all-reduce each (gptqmodel_amd/utils/tp.py)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one HIP graph per token")
    ap.add_argument("--no-fuse", action="store_true", help="7 launches per layer instead of fused qkv / gate_up")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--exact-bf16", action="store_true",
                    help="opt in to GPTQHIP_GEMM_EXACT_BF16 (bf16 only; leaves the reference's per-weight rounding)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    cfg = LLAMA3_8B if args.model == "llama3-8b" else LLAMA3_70B
    tp = world if args.model == "llama3-70b" else 1   # 8B: replicas only (fits one GPU); 70B: TP over the node
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    gs = 128
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    fuse = not args.no_fuse
    if args.exact_bf16:
        from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
        HipGptqLinear.EXACT_BF16_DECODE = True
    lshapes = launch_shapes(cfg, fuse)
    if tp > 1:
        # Megatron split of every launch: column-parallel (N / tp) when K == hidden, row-parallel (K / tp) otherwise
        # (o_proj: K = q dim, down_proj: K = inter).  Shards are generated directly at their sharded shapes.
        from gptqmodel_amd.utils.tp import _bounds
        sharded = []
        for k, n, cnt in lshapes:
            row_parallel = (n == cfg["hidden"])
            if row_parallel:
                _bounds(k, 0, tp, gs, "in_features")
                sharded.append((k // tp, n, cnt, True))
            else:
                _bounds(n, 0, tp, 8, "out_features")
                sharded.append((k, n // tp, cnt, False))
    else:
        sharded = [(k, n, cnt, False) for k, n, cnt in lshapes]
    layers = []
    for _ in range(cfg["layers"]):
        layers.append([(make_linear(k, n, gs, dev, gen, dtype), rowp) for k, n, _, rowp in sharded])
    xs = {}
    for k, _, _, _ in sharded:
        if k not in xs:
            xs[k] = (torch.randn((1, k), device=dev, generator=gen) * 0.5).to(dtype)
    n_launch = cfg["layers"] * len(lshapes)
    n_linear = cfg["layers"] * sum(c for _, _, c in lshapes)
    step_bytes = cfg["layers"] * sum(algorithmic_bytes(1, k, n, gs) for _, k, n in layer_shapes(cfg))   # whole model
    step_flops = cfg["layers"] * sum(2 * k * n for _, k, n in layer_shapes(cfg))

    def token_step():
        for layer in layers:
            for lin, rowp in layer:
                if rowp and tp > 1:
                    part = lin.forward_partial(xs[lin.in_features])          # fp32 partial sums of this K-shard
                    dist.all_reduce(part, op=dist.ReduceOp.SUM)               # RCCL over xGMI, 2 per decoder layer
                    part.to(dtype)                                            # the reference's single rounding
                else:
                    lin(xs[lin.in_features])

    stream = torch.cuda.Stream(device=dev)
    graph = None
    with torch.cuda.stream(stream):
        token_step()  # allocates the workspace for this stream outside of capture
        stream.synchronize()
        if not args.no_graph and tp == 1:  # TP>1: eager (RCCL inside graph capture is untested on the 1-GPU dev box)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                token_step()

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
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
    run(args.steps)
    with torch.cuda.stream(stream):
        ev1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)

    tmax = torch.tensor([wall], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    ms_per_step = wall * 1e3 / args.steps
    value = (world if tp == 1 else 1) * args.steps / wall   # replicas add up; a TP group produces one token stream

    if rank == 0:
        # HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this same command
        # (profiles/*_pmc_summary.json; gfx950 FETCH_SIZE correction applied there).  PMC cannot be read live here.
        traffic = None
        try:
            import glob
            summ = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
            if summ and fuse:
                with open(summ[-1]) as f:
                    pm = json.load(f)
                traffic = pm["hbm_read_bytes_per_launch_corrected"] + 1024.0 * pm.get("WRITE_SIZE_KB_per_launch_raw", 0.0)
        except Exception:
            traffic = None
        launch_us = ev_ms * 1e3 / (args.steps * n_launch)  # average launch duration incl. inter-kernel gaps
        bytes_per_launch = step_bytes / n_launch / tp        # per rank
        achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
        out = {
            "metric": ("llama3_8b" if args.model == "llama3-8b" else "llama3_70b") + "_gptq_int4_g128_decode_tokens_per_s",
            "value": value, "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak" if tp == 1 else "strong", "vs_baseline": None, "dtype": "f16" if dtype == torch.float16 else "bf16",
            "data": "synthetic",
            "config": {"workload": ("Llama-3-8B GPTQ int4 g128 desc_act=False batch=1 decode: 224 quantised linears per token "
                                    "(q,k,v,o,gate,up,down x 32), M=1, random packed weights") if args.model == "llama3-8b" else
                                   ("Llama-3-70B GPTQ int4 g128 batch=1 decode: 560 quantised linears per token (x 80 layers), M=1, "
                                    f"tensor parallel TP={tp} (column qkv/gate_up, row o/down + fp32 all-reduce), random packed weights"),
                       "parallelism": f"replicas x{world}" if tp == 1 else f"tp{tp}",
                       "linears_per_step": n_linear, "launches_per_step": n_launch, "fused_siblings": fuse,
                       "graph": graph is not None, "replicas": world,
                       "weight_bytes_per_token": step_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "gptqhip::skinny_kernel<BITS=4,ACT,SCL,MT=1,GPC=1,AM_ROW1,D=4>",
                         "bytes_per_launch": bytes_per_launch, "avg_launch_us": launch_us,
                         "note": "event-timed average over the timed region incl. inter-kernel gaps of the graph"},
            "gemm_tflops_equiv": step_flops * value / world / 1e12,
        }
        if args.model != "llama3-8b":
            out["roofline"]["traffic"] = None
        if world == 1 and tp == 1:
            out["prefill"] = prefill_tflops(layers[0], dtype, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, gs)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
