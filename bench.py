#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X GPTQ/AWQ dequant-matmul backend.

Headline workload (BASELINE.json configs[1], "C2"): Llama-3-8B GPTQ int4 group_size=128 desc_act=False, batch=1 decode.
One "step" = one pass of the hot path over one token: the 224 quantised linears of the model (32 layers x
{q,k,v,o,gate,up,down}; q/k/v and gate/up fused along N -> 128 launches), M=1, synthetic random packed weights of the
real shapes (3.63 GB of distinct packed weight + scale/zero bytes resident in HBM), executed with TRUE DATA DEPENDENCIES:
every linear consumes the previous one's output through the decoder layer's elementwise glue (RMSNorm, SiLU*mul,
residual adds -- fused into the decode ops, gptqmodel_amd/utils/decode_chain.py; attention itself is not a quantised
linear and is replaced by the stand-in "attention output = q").  No attention / KV cache / lm_head / sampling: this
is the quantised-linear stack of a token, which is what the metric name says.  The step is captured once and replayed
as ONE HIP graph per token.  value = tokens/s (x N for --gpus N: the 8B model fits one GPU, ranks are independent
replicas, no data-path collective -- SURVEY.md 8e).

Launching: `python bench.py --gpus N ...` with no WORLD_SIZE in the environment SPAWNS the N ranks itself (it re-executes under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`), rank r bound to GPU r, RCCL process
group; launched under torchrun by somebody else it uses the environment it finds and insists that WORLD_SIZE == --gpus.  Either
way ONE JSON line comes out with n_gpus = N and ranks_seen = an all-reduce of ones over the process group.  For N > 1 the line
keeps the 8B replica headline (so N = 1 agrees with the single-GPU record) and carries configs[] = [{"config": "C5", "tp": N, ...}]:
the Llama-3-70B decode step tensor-parallel over the same N ranks (bench_tp.tp_decode_entry), i.e. one scaling sweep of this
command also yields the 70B strong-scaling curve BASELINE.json's metric names.

Output (round 5): stdout carries exactly ONE line -- a compact (< 4 KB) JSON object with the contract's keys, `roofline`, `cpu_baseline`
and `configs_summary` (one {id, config, value, unit, frac} record per entry, the shape of the reference's own benchmark records,
scripts/benchmark_marlin_a100.py:181-201).  The full record described below (per-config workload prose, the 28-case T1 table, e2e, the
CPU thread sweep; ~21 KB) goes to stderr as `BENCH_DETAIL {...}` and to gpurun_out/bench_detail.json.  (Round 4 printed the full record
on stdout and the driver could not parse it.)  Timed region: 40 clock spin-up steps (`spin_up_steps`), W warm-up steps, barrier +
synchronize, K timed steps, synchronize + barrier; `value` / `ms_per_step` are wall-clock over the bracket (MAX over ranks),
`gpu_ms_per_step` the same K steps between two HIP events on the launch stream.

The legs outside the timed region (configs[], T1, e2e, cpu_baseline) live in bench_legs.py; this file keeps the launch logic, the
headline's timed region, the roofline and the result line.

Extra objects of the full record (tier contract; the compact line keeps their numbers):
  roofline      dominant kernel = gptqhip::skinny1_kernel (round 6: the batch-1 fused dequant-GEMV in its preload form, decode-op
                instantiation with the layer glue fused; bf16 runs and GPTQHIP_DECODE_BITFAITHFUL=1 use its bit-faithful instantiation);
                achieved = algorithmic bytes per launch (K*N/2 + G*N*2 + G*N/2 + M*(K+N)*2, averaged over the launches) /
                average launch duration measured with HIP events on the launch stream over the timed region.  traffic = HBM bytes per
                launch measured LIVE by default: two separate `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, WRITE_SIZE) over a
                3-step eager run of this same command on this box (live_pmc below; --no-live-pmc falls back to the committed
                profiles/*_pmc_summary.json and says so in traffic_source).
  configs       one object per BASELINE.json config with its own workload / value / roofline (outside the timed region):
                C2 variant (per-module launches + torch glue kernels), C3 act-order prefill at M=65536, C4 AWQ decode +
                M=2048 prefill, C5 Llama-3-70B decode at TP=1.
  e2e           tokens/s of a whole HF LlamaForCausalLM with Llama-3-8B shapes (random init, real attention / KV cache / norms /
                lm_head) decoding through the plugin classes: eager generate() and one HIP graph per decode step.
  cpu_baseline  kind "reference": the reference's OWN TorchLinear / AwqTorchLinear modules (oracle/_ref snapshot through oracle/ref_import.py;
                bench_legs.cpu_baseline) on this host's cores, thread count swept, rank 0 at N=1 only: C1 (single 4096x4096 linear, M in {1,32,2048}, fp16 and
                bf16), the AWQ leg (C4: the AwqTorchLinear op sequence, torch_awq.py:157-195, bf16, M in {1,32}), one decoder layer
                at M=1 extrapolated to tokens/s, and the model-level figure from the per-shape timings; `measured` / `extrapolated`
                say which numbers are which.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import re
DECODE_KERNEL_RE = re.compile(r"gptqhip::(skinny1?p?_kernel|decode_stream_kernel)")   # the batch-1 decode kernels (forms 0 / 2, 3 / 4, 1)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/fp16

LLAMA3_8B = dict(hidden=4096, inter=14336, q=4096, kv=1024, layers=32)
LLAMA3_70B = dict(hidden=8192, inter=28672, q=8192, kv=1024, layers=80)


def layer_shapes(cfg):
    h, i = cfg["hidden"], cfg["inter"]
    return [("q_proj", h, cfg["q"]), ("k_proj", h, cfg["kv"]), ("v_proj", h, cfg["kv"]), ("o_proj", cfg["q"], h),
            ("gate_proj", h, i), ("up_proj", h, i), ("down_proj", i, h)]


def launch_shapes(cfg):
    """Kernel launches per decoder layer with sibling fusion: 7 linears, 4 launches."""
    h, i = cfg["hidden"], cfg["inter"]
    return [(h, cfg["q"] + 2 * cfg["kv"]), (cfg["q"], h), (h, 2 * i), (i, h)]


def algorithmic_bytes(m, k, n, gs=128):
    g = k // gs
    return k * n // 2 + g * n * 2 + g * n // 2 + m * (k + n) * 2


def model_bytes_flops(cfg, m=1, gs=128):
    b = cfg["layers"] * sum(algorithmic_bytes(m, k, n, gs) for _, k, n in layer_shapes(cfg))
    f = cfg["layers"] * sum(2 * m * k * n for _, k, n in layer_shapes(cfg))
    return b, f


# ---------------------------------------------------------------------------------------------------------------------
# synthetic modules: random int32 qweight, scales rand*0.01+0.005
# ---------------------------------------------------------------------------------------------------------------------
def make_gptq(k, n, gs, dev, gen, dtype, desc_act=False, derive_from=None):
    """GPTQ-v2 module: sym zeros 0x88888888; desc_act=True uses g_idx = randperm(K)//gs (test_torch_kernel_accuracy.py:55-56).
    derive_from: reuse another module's random words xor a constant (70B: 34 GB of Philox output would dominate the run)."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    lin = HipGptqLinear(bits=4, group_size=gs, sym=True, desc_act=desc_act, in_features=k, out_features=n, bias=False,
                        register_buffers=False)
    if derive_from is None:
        w = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev, generator=gen)
    else:
        w = derive_from
    # code 0 -> 8: the 16 code values become symmetric around the sym zero-point 8, i.e. ZERO-MEAN weights like a real
    # checkpoint.  With plain uniform codes E[q - 8] = -0.5, every linear gets a DC gain of -0.005 * K and the fp16 residual
    # stream of the dependent chain overflows within three layers (found by the chain parity test).
    lin.qweight = w | (((~(w | (w >> 1) | (w >> 2) | (w >> 3))) & 0x11111111) << 3)
    del w
    lin.qzeros = torch.full((k // gs, n // 8), -2004318072, dtype=torch.int32, device=dev)  # 0x88888888
    lin.scales = (torch.rand((k // gs, n), device=dev, generator=gen) * 0.01 + 0.005).to(dtype)
    if desc_act:
        lin.g_idx = (torch.randperm(k, device=dev, generator=gen) // gs).to(torch.int32)
    else:
        lin.g_idx = (torch.arange(k, device=dev, dtype=torch.int32) // gs)
    lin.bias = None
    lin.qzero_format(format=2)
    lin.eval()
    lin.post_init()
    return lin


def make_awq(k, n, gs, dev, gen, dtype):
    """AWQ GEMM-layout module: random int32 qweight [K,N/8] and qzeros [G,N/8] (asymmetric, all 16 zero values)."""
    from gptqmodel_amd.nn_modules.qlinear.hip_awq import HipAwqLinear
    lin = HipAwqLinear(bits=4, group_size=gs, sym=False, desc_act=False, in_features=k, out_features=n, bias=False,
                       register_buffers=False)
    lin.qweight = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int32, device=dev, generator=gen)
    lin.qzeros = torch.randint(-2**31, 2**31 - 1, (k // gs, n // 8), dtype=torch.int32, device=dev, generator=gen)
    lin.scales = (torch.rand((k // gs, n), device=dev, generator=gen) * 0.01 + 0.005).to(dtype)
    lin.bias = None
    lin.eval()
    lin.post_init()
    return lin


def build_stack(cfg, maker, dev, gen, dtype, n_layers=None):
    from gptqmodel_amd.utils.decode_chain import DecodeLayer
    layers = []
    for _ in range(n_layers or cfg["layers"]):
        qkv, o, gu, down = [maker(k, n) for k, n in launch_shapes(cfg)]
        # random columns: the fused gate|up module counts as interleaved in blocks of 8 (utils.model.fuse_gate_up_interleaved),
        # so the decode chain applies SiLU*mul in its epilogue; the "modules" mode de-interleaves its output accordingly
        gu.gate_up_interleaved = True
        nw = lambda: (1.0 + 0.1 * torch.randn(cfg["hidden"], device=dev, generator=gen)).to(dtype)
        layers.append(DecodeLayer(qkv, o, gu, down, nw(), nw()))
    return layers


def time_graph(fn, stream, steps, warmup):
    """Capture fn() on `stream`, replay: (ms per replay by HIP events on that stream, graph)."""
    with torch.cuda.stream(stream):
        fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        for _ in range(warmup):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(steps):
            g.replay()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) / steps, g


def live_pmc(dtype_flag, limit_s=100):
    """HBM bytes per decode launch measured NOW on this box (the default; --no-live-pmc skips it): two separate `rocprofv3 --kernel-trace --pmc <counter>`
    passes (MI355X_MICROARCH.md: counters in their own run, no other trace domains) over a 3-step eager run of this same script,
    FETCH_SIZE doubled (the guide's gfx950 correction for wide coalesced reads) + WRITE_SIZE.  (bytes | None, source string)."""
    import csv
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "live PMC unavailable (no rocprofv3)"
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="gptqhip_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", td, "-o", "bench", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-configs", "--no-graph", "--no-live-pmc", "--dtype", dtype_flag]
        try:
            pr = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp", start_new_session=True,
                                  env=dict(os.environ, TMPDIR="/tmp"))
            try:
                pr.wait(timeout=limit_s)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)        # the process group WE started, nothing else
                pr.wait()
                return None, f"live PMC pass {counter} exceeded {limit_s} s"
            rows = []
            for root, _, files in os.walk(td):
                for fn in files:
                    if fn.endswith("counter_collection.csv"):
                        with open(os.path.join(root, fn)) as f:
                            rows += [float(r["Counter_Value"]) for r in csv.DictReader(f)
                                     if DECODE_KERNEL_RE.search(r["Kernel_Name"]) and r["Counter_Name"] == counter]
            if not rows:
                return None, f"live PMC pass {counter} produced no decode-kernel rows"
            vals[counter] = sum(rows) / len(rows)
        except Exception as e:  # noqa: BLE001
            return None, f"live PMC failed: {str(e)[:120]}"
        finally:
            shutil.rmtree(td, ignore_errors=True)
    tr = 2.0 * 1024.0 * vals["FETCH_SIZE"] + 1024.0 * vals["WRITE_SIZE"]
    return tr, ("measured live in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 3 eager steps of this "
                "command on this box; FETCH_SIZE x2 = the guide's gfx950 correction, KB -> bytes)")


def latest_pmc():
    """(traffic bytes per launch | None, source string).  Read from the committed rocprofv3 --pmc summary of the SAME bench
    command (profiles/*_pmc_summary.json, gfx950 FETCH_SIZE correction applied there); not measured in this run."""
    try:
        summ = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
        if not summ:
            return None, "none (no committed PMC summary)"
        with open(summ[-1]) as f:
            pm = json.load(f)
        tr = pm["hbm_read_bytes_per_launch_corrected"] + 1024.0 * pm.get("WRITE_SIZE_KB_per_launch_raw", 0.0)
        return tr, f"committed file profiles/{os.path.basename(summ[-1])} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, earlier run of this command; not measured live)"
    except Exception as e:  # noqa: BLE001
        return None, f"unavailable ({e})"


RESULT_LINE_LIMIT = 4096     # bytes; the driver's parser lost round 4's 21 KB line (BENCH_r04.json: parsed null)
T1_GATE_CASES = ("mlp_up_m72", "mlp_up_m128", "mlp_up_m136", "mlp_down_m128", "attn_m128")


def _r(v, nd=4):
    """Round floats for the result line (4 significant-ish digits are what the measurement supports)."""
    if isinstance(v, float):
        return float(f"{v:.{nd}g}") if abs(v) < 1.0 else round(v, max(0, nd - 1 - len(str(int(abs(v)))) + 2))
    return v


def result_line(detail):
    """The ONE stdout line of a run: a compact (< RESULT_LINE_LIMIT bytes) JSON object with the contract's headline keys, `roofline`,
    `cpu_baseline` and a one-record-per-case `configs_summary` (the shape of the reference's own benchmark records,
    scripts/benchmark_marlin_a100.py:181-201).  Everything else of `detail` (per-config workload prose, the 28-case T1 table, the
    e2e leg, the CPU thread sweep) goes to stderr and gpurun_out/bench_detail.json, not here.  Pure function (tests/test_host_logic.py)."""
    keep = ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "gpu_ms_per_step", "spin_up_steps", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    line = {k: _r(detail[k]) for k in keep if k in detail}
    cfg = detail.get("config", {})
    line["config"] = {"workload": "Llama-3-8B GPTQ int4 g128 desc_act=False batch=1 decode: the 224 quantised linears of a token, "
                                  "chained through the layer glue, M=1, random packed weights (BASELINE configs[1])"}
    if "workload_short" in cfg:
        line["config"]["workload"] = cfg["workload_short"]
    for k in ("parallelism", "mode", "decode_form", "linears_per_step", "launches_per_step", "graph", "weight_bytes_per_token", "path", "allreduce"):
        if k in cfg:
            line["config"][k] = cfg[k]
    rf = detail.get("roofline", {})
    line["roofline"] = {k: _r(rf[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_us")
                        if k in rf}
    if "kernel" in rf:
        line["roofline"]["kernel"] = rf["kernel"].split("<")[0].split(" ")[0]
    if "traffic_source" in rf:
        line["roofline"]["traffic_live"] = rf["traffic_source"].startswith("measured live")
    cb = detail.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _r(cb[k]) for k in ("value", "unit", "cores", "kind", "cpu_model", "c1_ms", "model_tokens_per_s_from_per_shape")
                                if k in cb}
        line["cpu_baseline"]["sample"] = "1 of 32 decoder layers (7 linears, M=1, bf16), extrapolated x32; c1_ms = single 4096x4096 linear"
    summ = []
    for c in list(detail.get("configs", [])) + ([detail["prefill"]] if "prefill" in detail else []):
        if "error" in c or "skipped" in c:
            summ.append({"id": c.get("id", c.get("config")), "error": str(c.get("error", c.get("skipped")))[:80]})
            continue
        rec = {"id": c.get("id", c.get("config")), "config": c.get("config"), "value": _r(c.get("value")), "unit": c.get("unit"),
               "frac": _r(c.get("roofline", {}).get("frac"))}
        if c.get("config") == "_prefill_headline":
            rec["id"], rec["config"] = "c2_layer_m8192", "C2"
        if "tp" in c:
            rec["tp"] = c["tp"]
        if "exchange" in c:
            rec["exchange"] = c["exchange"]
        if "hbm_frac" in c.get("roofline", {}):
            rec["hbm_frac"] = _r(c["roofline"]["hbm_frac"])
        summ.append(rec)
        if c.get("config") == "T1":
            by_id = {k["case_id"]: k for k in c.get("cases", [])}
            for cid in T1_GATE_CASES:
                if cid in by_id:
                    k = by_id[cid]
                    summ.append({"id": "t1_" + cid, "config": "T1", "us": _r(k["us_graph"]), "value": _r(k["tflops_graph"]), "unit": "TFLOP/s",
                                 "frac": _r(k["roofline"]["mfma_frac"]), "hbm_frac": _r(k["roofline"]["hbm_frac"])})
    if summ:
        line["configs_summary"] = summ
    e2e = detail.get("e2e")
    if isinstance(e2e, dict):
        line["e2e"] = {k: _r(v) for k, v in e2e.items() if isinstance(v, (int, float)) and not isinstance(v, bool)} or {"error": str(e2e.get("error", ""))[:80]}
    if "note" in detail:
        line["note"] = detail["note"][:160]
    line["detail"] = "gpurun_out/bench_detail.json (also on stderr)"
    txt = json.dumps(line, separators=(",", ":"))
    if len(txt) >= RESULT_LINE_LIMIT:      # never lose the headline to an over-long line: drop the optional blocks, largest first
        for k in ("e2e", "configs_summary"):
            line.pop(k, None)
            txt = json.dumps(line, separators=(",", ":"))
            if len(txt) < RESULT_LINE_LIMIT:
                break
    return txt


def emit(detail):
    """stderr + gpurun_out/bench_detail.json get the full record; stdout gets the one compact result line, LAST."""
    full = json.dumps(detail)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_detail.json"), "w") as f:
            f.write(full + "\n")
    except OSError:
        pass
    print("BENCH_DETAIL " + full, file=sys.stderr, flush=True)
    print(result_line(detail), flush=True)


def spawn_command(gpus, environ, argv):
    """The command that launches `gpus` ranks of this script, or None when this process IS a rank already (WORLD_SIZE set by a
    launcher) or a single rank is wanted.  Pure function of its arguments (tests/test_host_logic.py)."""
    if gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def handshake(rank, local_rank, world):
    """--handshake-only: rendezvous + one all-reduce of ones (RCCL with one GPU per rank, else gloo); rank 0 prints the JSON line."""
    backend = "gloo"
    dev = "cpu"
    if torch.cuda.is_available() and torch.cuda.device_count() >= world and not os.environ.get("GPTQHIP_BENCH_SHARE_GPU"):
        backend, dev = "nccl", torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
    seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend) if backend == "gloo" else dist.init_process_group(backend, device_id=dev)
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        seen = int(ones.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "handshake", "n_gpus": world, "ranks_seen": seen, "backend": backend}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "llama3-70b"],
                    help="llama3-8b: the headline config, ranks are independent replicas (weak scaling). "
                         "llama3-70b: BASELINE configs[4], tensor parallel over all ranks (strong scaling): column-parallel "
                         "qkv/gate_up, row-parallel o/down with one fp32 all-reduce each (gptqmodel_amd/utils/tp.py)")
    ap.add_argument("--mode", default="chain", choices=["chain", "modules"],
                    help="chain: decode ops with the layer glue fused into the GEMV, 4 launches per layer (default); "
                         "modules: HipGptqLinear.forward per launch + the glue as separate torch kernels (what an HF model runs)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one HIP graph per token")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs[] array (C3/C4/C5 legs)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end HF Llama-3-8B-shaped decode leg (`e2e` key)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--tp-steps", type=int, default=30, help="timed steps of the C5 tp=N entry appended when N > 1")
    ap.add_argument("--live-pmc", action="store_true", help="(default since round 4; kept so that old command lines still parse)")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure roofline.traffic live (two short rocprofv3 --pmc passes of this script, ~20-40 s) and read the "
                         "committed profiles/*_pmc_summary.json instead")
    ap.add_argument("--handshake-only", action="store_true",
                    help="launch / rendezvous check only: spawn the ranks, all-reduce ones, print {n_gpus, ranks_seen}; needs no GPU "
                         "(gloo when CUDA is unavailable) -- tests/test_host_logic.py uses it to prove --gpus is honoured")
    args = ap.parse_args()

    cmd = spawn_command(args.gpus, os.environ, sys.argv[1:])
    if cmd is not None:
        # `python bench.py --gpus N` outside torchrun: become the launcher of N ranks (one process per GPU)
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL / peer mappings across processes need it here
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         "(or without torchrun: bench.py spawns the ranks itself)")
    if args.handshake_only:
        return handshake(rank, local_rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # GPTQHIP_BENCH_SHARE_GPU=1 (smoke-testing the N > 1 code path on a one-GPU box, tests/dev): every rank uses cuda:0 and
    # the process group runs on gloo (RCCL refuses two ranks on one device); the numbers of such a run mean nothing.
    share = bool(os.environ.get("GPTQHIP_BENCH_SHARE_GPU")) and world > 1
    if not share and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {world} but only {torch.cuda.device_count()} GPU(s) are visible "
                         "(GPTQHIP_BENCH_SHARE_GPU=1 runs the N > 1 code paths with all ranks on GPU 0: a smoke switch, not a measurement)")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    ranks_seen = 1
    if dist is not None:
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if share else dev)
        dist.all_reduce(ones)                                  # over RCCL unless the smoke switch put every rank on one GPU
        ranks_seen = int(ones.item())

    if args.model == "llama3-70b":
        from bench_tp import run_70b   # tensor-parallel 70B leg lives in its own file
        run_70b(args, rank, local_rank, world, dev, dist, ranks_seen=ranks_seen)
        if dist is not None:
            dist.destroy_process_group()
        return

    from gptqmodel_amd.utils.decode_chain import DecodeStep
    cfg = LLAMA3_8B
    dtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    gs = 128
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    layers = build_stack(cfg, lambda k, n: make_gptq(k, n, gs, dev, gen, dtype), dev, gen, dtype)
    n_launch = cfg["layers"] * 4
    n_linear = cfg["layers"] * 7
    step_bytes, step_flops = model_bytes_flops(cfg)
    x0 = (torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype)

    def make_step(mode):
        if mode == "modules":
            from bench_legs import ModulesStep
            return ModulesStep(layers, cfg, dtype, x0)
        st = DecodeStep(layers, cfg["hidden"], cfg["q"], dtype)
        st.x_in.copy_(x0)
        return st

    mode = args.mode
    step = make_step(mode)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        out0 = step.run()
        stream.synchronize()
    if not torch.isfinite(out0).all():
        raise SystemExit("bench.py: the decode chain produced non-finite activations; refusing to time garbage")
    del out0
    graph = None
    if not args.no_graph:
        with torch.cuda.stream(stream):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                step.run()

    def run(n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                if graph is not None:
                    graph.replay()
                else:
                    step.run()

    # clock / TLB spin-up before the W warm-up steps of the contract: with the driver's --warmup 5 (6 ms of GPU work) the timed steps ran
    # 2.5 % slower than after 20+ warm-up steps (8.85 vs 8.63 us per launch; the chip ramps its clocks over the first ~25 ms of a
    # burst).  Reported as `spin_up_steps`; the W warm-up steps and the K timed steps follow unchanged.
    SPIN_UP_STEPS = int(os.environ.get("GPTQHIP_BENCH_SPIN_UP", "40"))
    run(SPIN_UP_STEPS)
    run(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # The timed region is ~20 ms at the driver's --steps 20: host-side noise of that order must stay out of it.  The collector is
    # off while it runs and the host SPINS on the closing event (a sleeping synchronize() wakes up late by up to milliseconds on a busy
    # host; measured: 785 vs 896 tokens/s for the same 8.68 us/launch of GPU time) before the contract's synchronize + barrier.
    import gc
    gc.collect()
    gc.disable()
    try:
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            ev0.record(stream)
        run(args.steps)
        with torch.cuda.stream(stream):
            ev1.record(stream)
        while not ev1.query():
            pass
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    finally:
        gc.enable()
    ev_ms = ev0.elapsed_time(ev1)
    tmax = torch.tensor([wall], device="cpu" if share else dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall = float(tmax.item())
    ms_per_step = wall * 1e3 / args.steps
    value = world * args.steps / wall   # replicas add up

    used_graph = graph is not None
    tp_entry = None
    if world > 1 and not args.no_configs:
        # BASELINE configs[4] on the SAME ranks: the 70B decode step tensor-parallel over all of them (every rank takes part)
        del step, graph
        layers.clear()
        torch.cuda.empty_cache()
        step = graph = None
        try:
            from bench_tp import tp_decode_entry
            tp_entry = tp_decode_entry(args, rank, world, dev, dist, args.tp_steps, 5,
                                       log=(lambda m: print(m, file=sys.stderr, flush=True)) if rank == 0 else None)
        except Exception as e:  # noqa: BLE001
            tp_entry = {"config": "C5", "tp": world, "error": f"{type(e).__name__}: {str(e)[:300]}"}

    if rank == 0:
        traffic, traffic_source = (None, "")
        if not args.no_live_pmc and world == 1 and not os.environ.get("GPTQHIP_BENCH_SHARE_GPU"):
            traffic, traffic_source = live_pmc(args.dtype)
        if traffic is None:
            note = traffic_source
            traffic, traffic_source = latest_pmc()
            if note:
                traffic_source += f" [{note}]"
        launch_us = ev_ms * 1e3 / (args.steps * n_launch)   # average launch duration incl. whatever the ops do not overlap
        bytes_per_launch = step_bytes / n_launch
        achieved = bytes_per_launch / (launch_us * 1e-6) / 1e9
        kernel = {"chain": "gptqhip::skinny1_kernel<ACT,SCL,D=4,GLUE,ALG> (decode op, preload form, glue fused; ALG=2 raw 4-bit codes as fp16 denormals on the f16 matrix pipe -- bf16 activations converted to fp16 per wave, exactly; ALG=0 = the reference's per-weight rounding, GPTQHIP_DECODE_BITFAITHFUL=1)",
                  "modules": "gptqhip::skinny1_kernel<ACT,SCL,D=4,GLUE=0,ALG>"}[mode]
        out = {
            "metric": "llama3_8b_gptq_int4_g128_decode_linear_stack_tokens_per_s",
            "value": value, "unit": "tokens/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "gpu_ms_per_step": ev_ms / args.steps,      # the same K steps between two HIP events on the launch stream (rank 0)
            "spin_up_steps": SPIN_UP_STEPS,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if dtype == torch.float16 else "bf16", "data": "synthetic",
            "config": {"workload": "Llama-3-8B GPTQ int4 g128 desc_act=False batch=1 decode: the 224 quantised linears of a token "
                                   "(q,k,v,o,gate,up,down x 32) with true data dependencies through the layer glue (RMSNorm, "
                                   "SiLU*mul, residual adds fused into the ops; attention stand-in = q), M=1, random packed weights",
                       "parallelism": f"replicas x{world}", "mode": mode,
                       # include/gptqhip.h gptqhip_set_decode_form: the process default of this dtype (GPTQHIP_DECODE_BITFAITHFUL=1 -> 4)
                       "decode_form": (4 if os.environ.get("GPTQHIP_DECODE_BITFAITHFUL", "0") not in ("", "0") else 5),
                       "loader_path": "mode=chain is what gptqmodel_post_init yields on HF Llama-family layers by itself since round 6 "
                                      "(utils.hf_llama.auto_fuse; the reference's own post_init through integration/gptqmodel_overlay/utils/model.patch); "
                                      "mode=modules is the GPTQHIP_AUTO_FUSE=0 path: plugin forward() per linear + torch glue kernels",
                       "linears_per_step": n_linear, "launches_per_step": n_launch, "fused_siblings": True,
                       "graph": used_graph, "replicas": world, "weight_bytes_per_token": step_bytes},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": kernel, "bytes_per_launch": bytes_per_launch, "avg_launch_us": launch_us,
                         "note": "event-timed over the timed region: avg_launch_us = step time / launches (kernel + "
                                 "dependent-launch gap)"},
            "gemm_tflops_equiv": step_flops * value / world / 1e12,
        }
        if tp_entry is not None:
            out["configs"] = [tp_entry]
        if share:
            out["note"] = "GPTQHIP_BENCH_SHARE_GPU=1: every rank ran on GPU 0 over gloo -- a launch / code-path smoke run, NOT a measurement"
        if world == 1 and not args.no_configs:
            from bench_legs import extra_configs
            out["configs"] = extra_configs(args, cfg, layers, dtype, dev, gen, stream, make_step, mode, ms_per_step, n_launch)
            if out["configs"] and out["configs"][-1].get("config") == "_prefill_headline":
                out["prefill"] = out["configs"].pop()
        layers.clear()
        step = graph = None
        torch.cuda.empty_cache()
        if world == 1 and not args.no_e2e and not args.no_configs:
            # the whole model, not just the quantised linears: HF LlamaForCausalLM with Llama-3-8B shapes (random init), real
            # attention / KV cache / norms / lm_head, decode through make_quant -> fuse_siblings -> post_init -- eager
            # generate() and one captured graph per decode step (the reference's speed metric is generate() wall time,
            # tests/inference_speed.py:95-113).  Outside the timed region.
            try:
                sys.path.insert(0, os.path.join(ROOT, "examples"))
                from hf_llama_dropin import run as e2e_run
                out["e2e"] = e2e_run("8b", args.dtype, 64)
            except Exception as e:  # noqa: BLE001
                out["e2e"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            from bench_legs import cpu_baseline
            out["cpu_baseline"] = cpu_baseline(cfg, gs)
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
