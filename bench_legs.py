#!/usr/bin/env python
"""The legs of bench.py that run OUTSIDE its timed region (split out of bench.py in round 5; bench.py keeps the headline, the timed
region and the result line): one entry per BASELINE.json config (`extra_configs`: C2 variants, mid-M, C3 act-order prefill, C4 AWQ,
C5 70B at TP = 1), T1 = the reference's own dequant-GEMM TFLOPS benchmark replayed case by case (`t1_entry`), and the CPU baseline
(`cpu_baseline`: the reference's own TorchLinear / AwqTorchLinear on the host cores, or the oracle's torch port when no reference tree
is there).  Only `cpu_baseline` touches `oracle/` -- as the thing being timed beside the GPU path, never inside it."""
from __future__ import annotations

import os
import sys
import time

import torch

from bench import (HBM_PEAK_GBS, LLAMA3_70B, MFMA_PEAK_TFLOPS, ROOT, algorithmic_bytes, build_stack, layer_shapes, make_awq, make_gptq,
                   model_bytes_flops, time_graph)

def decode_entry(name, workload, cfg, ms, n_launch, tp=1, extra=None):
    b, f = model_bytes_flops(cfg)
    gbs = b / tp / (ms * 1e-3) / 1e9
    d = {"config": name, "workload": workload, "value": 1e3 / ms, "unit": "tokens/s", "ms_per_token": ms,
         "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                      "bytes_per_launch": b / tp / n_launch, "avg_launch_us": ms * 1e3 / n_launch},
         "gemm_tflops_equiv": f / (ms * 1e-3) / 1e12}
    if extra:
        d.update(extra)
    return d


def prefill_entry(name, workload, lins, m, dtype, dev, iters=3, kernel="gptqhip::tiled_kernel"):
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    xs = {}
    for lin in lins:
        if lin.in_features not in xs:
            xs[lin.in_features] = (torch.randn((m, lin.in_features), device=dev, generator=gen) * 0.5).to(dtype)

    def run():
        for lin in lins:
            lin(xs[lin.in_features])
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = sum(2.0 * m * lin.in_features * lin.out_features for lin in lins)
    tf = flops / ms / 1e9
    del xs
    torch.cuda.empty_cache()
    return {"config": name, "workload": workload, "value": tf, "unit": "TFLOP/s", "ms": ms,
            "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / MFMA_PEAK_TFLOPS, "kernel": kernel}}


# ---------------------------------------------------------------------------------------------------------------------
# T1: the reference's OWN dequant-GEMM TFLOPS benchmark, replayed (scripts/benchmark_marlin_a100.py:35-44 cases, :127-160 tensors,
# :163-201 timing: warmup 30, iters 80, eager module(x) calls bracketed by synchronize, tflops = 2 M K N / t)
# ---------------------------------------------------------------------------------------------------------------------
T1_CASES = ([("mlp_up", m, 4096, 11008) for m in (64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192)]
            + [("mlp_down", m, 11008, 4096) for m in (64, 80, 96, 112, 128, 160, 192)]
            + [("attn", m, 4096, 4096) for m in (64, 96, 128, 192)])


def t1_entry(dtype, dev, stream, warmup=30, iters=80, seed=1234):
    """28 cases of the reference's Marlin benchmark through the plugin class's forward(): the reference's eager wall-clock method
    (`tflops`) and, next to it, the same launch timed from a HIP graph (`tflops_graph`: the kernel without the Python call)."""
    from gptqmodel_amd.nn_modules.qlinear.hip_gptq import HipGptqLinear
    from gptqmodel_amd import ops as _ops
    gs = 128
    mods, cases = {}, []
    for idx, (tag, m, k, n) in enumerate(T1_CASES):
        if (k, n) not in mods:
            g = torch.Generator(device=dev)
            g.manual_seed(seed + len(mods))
            lin = HipGptqLinear(bits=4, group_size=gs, sym=True, desc_act=False, in_features=k, out_features=n, bias=False,
                                register_buffers=False)
            lin.qweight = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev, generator=g)
            lin.scales = (torch.rand((k // gs, n), device=dev, generator=g) * 0.5 + 0.5).to(dtype)
            lin.qzeros = torch.zeros((k // gs, n // 8), dtype=torch.int32, device=dev)      # the benchmark zeroes them (:157)
            lin.g_idx = (torch.arange(k, device=dev, dtype=torch.int32) // gs)
            lin.bias = None
            lin.qzero_format(format=2)
            lin.eval()
            lin.post_init()
            mods[(k, n)] = lin
        lin = mods[(k, n)]
        gx = torch.Generator(device=dev)
        gx.manual_seed(seed + idx)
        x = torch.rand((m, k), device=dev, generator=gx).to(dtype)
        with torch.inference_mode():
            for _ in range(warmup):
                lin(x)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(iters):
                y = lin(x)
            torch.cuda.synchronize(dev)
            mean_ms = (time.perf_counter() - t0) * 1e3 / iters
        with torch.no_grad():
            ms_g, g = time_graph(lambda: [lin(x) for _ in range(8)], stream, 20, 3)
        ms_g /= 8
        del g
        flops = 2.0 * m * k * n
        by = algorithmic_bytes(m, k, n, gs)
        cases.append({"case_id": f"{tag}_m{m}", "m": m, "in_features": k, "out_features": n, "shape": list(y.shape),
                      "mean_ms": mean_ms, "tflops": flops / (mean_ms * 1e9), "us_graph": ms_g * 1e3, "tflops_graph": flops / (ms_g * 1e9),
                      "kernel": _ops.plan_describe(m, k, n, gs).split(" ")[0],
                      "roofline": {"mfma_frac": flops / (ms_g * 1e9) / MFMA_PEAK_TFLOPS, "hbm_frac": by / (ms_g * 1e-3) / 1e9 / HBM_PEAK_GBS}})
    best = max(c["tflops_graph"] for c in cases)
    return {"config": "T1", "workload": "the reference's own dequant-GEMM TFLOPS benchmark replayed (scripts/benchmark_marlin_a100.py: 28 cases, "
                                        "M = 64..192 on 4096x11008 / 11008x4096 / 4096x4096, int4 g128 sym, warmup 30 / iters 80, eager module(x) "
                                        "wall clock, tflops = 2MKN/t); tflops_graph = the same launch replayed from a HIP graph",
            "dtype": str(dtype).replace("torch.", ""), "unit": "TFLOP/s", "value": best, "value_is": "best tflops_graph over the 28 cases",
            "warmup": warmup, "iters": iters, "cases": cases,
            "roofline": {"bound": "mfma", "achieved": best, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": best / MFMA_PEAK_TFLOPS,
                         "kernel": "gptqhip::tiled_kernel<BM=64> + splitk_reduce_kernel (gptqhip_gemm's choice at these sizes)"}}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle's torch-CPU port of BACKEND.TORCH (kind "port"), thread count swept
# ---------------------------------------------------------------------------------------------------------------------
def _cpu_tensors(k, n, gs, dtype):
    qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32)
    qz = torch.full((k // gs, n // 8), -2004318072, dtype=torch.int32)
    sc = (torch.rand((k // gs, n)) * 0.01 + 0.005).to(dtype)
    gi = (torch.arange(k, dtype=torch.int32) // gs)
    return qw, qz, sc, gi


def _time_cpu(fn, budget_s, max_iters):
    fn()
    iters, t0 = 0, time.perf_counter()
    while True:
        fn()
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= max_iters:
            return el / iters * 1e3, iters


def _reference_modules():
    """The REAL reference classes (TorchLinear / AwqTorchLinear) through the oracle's import shim: /root/reference where it is mounted
    (the build container), else the snapshot oracle/_ref that oracle/make_ref_snapshot.py ships with the push (the GPU box).  None
    when neither is there: cpu_baseline then falls back to the oracle's torch port and says so (kind "port")."""
    try:
        from oracle.ref_import import load_reference, reference_available
        if not reference_available():
            return None
        had = os.environ.get("CUDA_VISIBLE_DEVICES")
        import logging
        lg = logging.getLogger("refshim")      # the reference's module-level logger (oracle/ref_shim/logbar): keep it off this run's stderr
        lg.addHandler(logging.NullHandler())
        lg.propagate = False
        try:
            return load_reference()     # (the shim hides the GPUs from the reference's import-time probes: undo that for this process)
        finally:
            if had is None:
                os.environ.pop("CUDA_VISIBLE_DEVICES", None)
            else:
                os.environ["CUDA_VISIBLE_DEVICES"] = had
    except Exception:  # noqa: BLE001 -- a broken snapshot must not take the benchmark down
        return None


def _ref_gptq_module(ref, k, n, gs, dtype, compiled=False):
    """A reference TorchLinear holding synthetic C1-style tensors (sym, v2 zero-points), post_init()ed on the CPU.  compiled=False
    replaces optimize() by a no-op (the reference's own trick, tests/test_torch.py:417): eager dequant + matmul."""
    qw, qz, sc, gi = _cpu_tensors(k, n, gs, dtype)
    lin = ref.TorchLinear(bits=4, group_size=gs, desc_act=False, sym=True, in_features=k, out_features=n, bias=False,
                          pack_dtype=torch.int32, register_buffers=True)
    lin.qweight, lin.qzeros, lin.scales, lin.g_idx = qw, qz, sc, gi
    lin.qzero_format(format=2)
    if not compiled:
        lin.optimize = lambda *a, **kw: None
    lin = lin.eval()
    lin.post_init()
    return lin


def _ref_awq_module(ref, k, n, gs, dtype):
    lin = ref.AwqTorchLinear(bits=4, group_size=gs, desc_act=False, sym=False, in_features=k, out_features=n, bias=False,
                             pack_dtype=torch.int32, register_buffers=True)
    lin.qweight = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int32)
    lin.qzeros = torch.randint(-2**31, 2**31 - 1, (k // gs, n // 8), dtype=torch.int32)
    lin.scales = (torch.rand((k // gs, n)) * 0.01 + 0.005).to(dtype)
    lin.optimize = lambda *a, **kw: None
    lin = lin.eval()
    lin.post_init()
    return lin


def cpu_baseline(cfg, gs=128, budget_s=24.0):
    """Reference path on the host cores: upstream's CPU test runs bf16 (tests/test_q4_torch.py:27,50) and
    flags fp16 CPU matmul as slow (:52-53); both are timed.  The thread count is swept on the actual sample (one decoder
    layer's 7 linears at M=1) and the best is used.  The torch.compile'd dequant upstream enables in post_init
    (torch.py:215-216,259) is timed for C1 (bf16, M=1) in a subprocess with a hard time limit (inductor compiles for ~20-60 s);
    `c1_ms.bf16_m1_compiled_dequant` is null when that does not finish."""
    from oracle.gptq_oracle import torch_cpu_forward_gptq
    torch.manual_seed(1234)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    t_start = time.perf_counter()
    ref = _reference_modules()
    kind = "reference" if ref is not None else "port"

    def gptq_fn(k, n, dtype, m):
        """-> a callable running ONE forward of a [k, n] layer at m rows: the reference module's forward() (kind "reference"), else
        the oracle's torch port of the same op sequence (kind "port")."""
        x = (torch.randn(m, k) * 0.5).to(dtype)
        if ref is not None:
            lin = _ref_gptq_module(ref, k, n, gs, dtype)

            def run():
                with torch.inference_mode():
                    return lin(x)
            return run
        t = _cpu_tensors(k, n, gs, dtype)
        return lambda: torch_cpu_forward_gptq(x, *t, 4)
    mods = [gptq_fn(kk, nn, torch.bfloat16, 1) for _, kk, nn in layer_shapes(cfg)]

    def one_pass():
        for f in mods:
            f()
    sweep = sorted({c for c in (8, 16, 32, 64, 128) if c <= ncpu})
    per_thread = {}
    for th in sweep:   # one decoder layer per thread count (after a warm-up pass), the workload the tokens/s is extrapolated from
        torch.set_num_threads(th)
        ms, _ = _time_cpu(one_pass, budget_s * 0.05, 2)
        per_thread[str(th)] = round(ms, 2)
    best = int(min(per_thread, key=lambda s: per_thread[s]))
    torch.set_num_threads(best)
    per_layer, iters = _time_cpu(one_pass, budget_s * 0.3, 20)
    del mods
    # C1: single QuantLinear 4096x4096 g128 sym=True
    k = n = 4096
    c1 = {}
    for dt, tag in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        t = _cpu_tensors(k, n, gs, dt)
        for m in (1, 32, 2048):
            if dt == torch.float16 and m > 1:
                # aten's CPU fp16 matmul is pathologically slow (measured on a GPU box's host: 2.6 s at M=32, 183 s at
                # M=2048 per call; upstream flags it too, tests/test_q4_torch.py:52-53): not part of a bounded leg
                c1[f"{tag}_m{m}"] = None
                continue
            ms, _ = _time_cpu(gptq_fn(k, n, dt, m), budget_s * 0.05, 4)
            c1[f"{tag}_m{m}"] = round(ms, 3)
    # per-shape C1-style timings (each linear of the model on its own, warm, M=1, bf16): the model-level figure they add up to is
    # the trustworthy CPU number -- the 7-linear layer pass above streams 218 M codes through the caches per pass and reads ~2x
    # slower than the sum of its parts
    per_shape = {}
    for name, kk, nn in layer_shapes(cfg):
        key = f"{kk}x{nn}"
        if key in per_shape:
            continue
        ms, _ = _time_cpu(gptq_fn(kk, nn, torch.bfloat16, 1), budget_s * 0.04, 3)
        per_shape[key] = round(ms, 3)
    model_ms = cfg["layers"] * sum(per_shape[f"{kk}x{nn}"] for _, kk, nn in layer_shapes(cfg))
    # C4: the AWQ reference path (AwqTorchLinear.forward op sequence, torch_awq.py:157-195 + dequantize_gemm) on a 4096x4096 layer
    from oracle.gptq_oracle import torch_cpu_forward_awq
    c4 = {}
    qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int32)
    qz = torch.randint(-2**31, 2**31 - 1, (k // gs, n // 8), dtype=torch.int32)
    sc = (torch.rand((k // gs, n)) * 0.01 + 0.005).to(torch.bfloat16)
    awq_ref = _ref_awq_module(ref, k, n, gs, torch.bfloat16) if ref is not None else None
    for m in (1, 32):
        x = (torch.randn(m, k) * 0.5).to(torch.bfloat16)
        if awq_ref is not None:
            def run_awq():
                with torch.inference_mode():
                    return awq_ref(x)
            ms, _ = _time_cpu(run_awq, budget_s * 0.05, 4)
        else:
            ms, _ = _time_cpu(lambda: torch_cpu_forward_awq(x, qw, qz, sc, gs), budget_s * 0.05, 4)
        c4[f"bf16_m{m}"] = round(ms, 3)
    del qw, qz, sc, awq_ref
    torch.set_num_threads(default_threads)
    c1["bf16_m1_compiled_dequant"], compiled_note = _cpu_compiled_c1(best, gs, use_reference=ref is not None)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {
        "value": 1e3 / (per_layer * cfg["layers"]), "unit": "tokens/s", "cores": best, "kind": kind,
        "kind_note": ("the reference's own TorchLinear.forward / AwqTorchLinear.forward (gptqmodel/nn_modules/qlinear/torch.py:302, "
                      "torch_awq.py:157) imported through oracle/ref_import.py" if kind == "reference" else
                      "oracle/gptq_oracle.py torch port of BACKEND.TORCH (pinned bit for bit to the reference's outputs): no reference "
                      "tree or oracle/_ref snapshot on this box"),
        "cpu_model": cpu_model, "torch": torch.__version__,
        "measured": ["ms_per_layer (one decoder layer's 7 linears, M=1)", "threads_swept", "c1_ms", "c4_awq_ms", "per_shape_ms"],
        "extrapolated": ["value = 1000 / (ms_per_layer x layers)", "model_tokens_per_s_from_per_shape = 1000 / (layers x sum of per_shape_ms)"],
        "per_shape_ms": per_shape, "per_shape_workload": "each distinct linear shape of the model alone, warm, M=1, bf16, best_threads",
        "model_tokens_per_s_from_per_shape": 1e3 / model_ms,
        "c4_awq_ms": c4, "c4_workload": "AWQ reference path: the reference's own AwqTorchLinear.forward (torch_awq.py:157-195: column unpack, AWQ reverse "
                                         "order, (w - z) * s, matmul) when kind is 'reference', else the oracle's pinned torch port of it; 4096x4096 g128 "
                                         "asym, bf16, best_threads",
        "trust": "c1_ms / c4_awq_ms / per_shape_ms are warm single-layer timings and the figures to compare with; `value` (layer pass "
                 "x layers) includes the cache thrash of streaming 7 layers' codes per pass and reads ~2x lower",
        "c1_compiled_note": compiled_note,
        "sample": f"1 of {cfg['layers']} decoder layers (7 linears, M=1, bf16 like upstream's CPU test), {iters} passes, "
                  f"extrapolated x{cfg['layers']}; "
                  + ("the REFERENCE's TorchLinear modules, eager (optimize() replaced by a no-op like tests/test_torch.py:417)" if kind == "reference"
                     else "torch CPU port of BACKEND.TORCH (oracle/gptq_oracle.py), not the reference module itself")
                  + f"; host os.cpu_count()={ncpu}",
        "ms_per_layer": per_layer, "threads_swept": per_thread, "threads_swept_unit": "ms per decoder layer (7 linears, M=1, bf16)",
        "best_threads": best,
        "c1_ms": c1, "c1_workload": "single QuantLinear 4096x4096 int4 g128 sym=True, eager dequant + matmul, best_threads "
                                      "(null: fp16 CPU matmul at M>1 takes 2.6-183 s per call on such a host; not timed)",
        "leg_s": round(time.perf_counter() - t_start, 1),
    }


def _cpu_compiled_c1(threads, gs, limit_s=150, use_reference=False):
    """C1 (4096x4096, bf16, M=1) with the dequant under torch.compile like upstream's post_init: (ms | None, note).  With the
    reference available it is the reference's OWN post_init() (TorchLinear.optimize, torch.py:215-216,259) that compiles."""
    import subprocess
    if use_reference:
        code = f"""
import sys, time, torch
sys.path.insert(0, {ROOT!r})
import bench_legs as B
torch.set_num_threads({threads})
torch.manual_seed(1234)
ref = B._reference_modules()
t0 = time.perf_counter()
lin = B._ref_gptq_module(ref, 4096, 4096, {gs}, torch.bfloat16, compiled=True)
x = (torch.randn(1, 4096) * 0.5).to(torch.bfloat16)
with torch.inference_mode():
    lin(x)
    comp = time.perf_counter() - t0
    ms, _ = B._time_cpu(lambda: lin(x), 2.0, 4)
print("RESULT", ms, comp)
"""
    else:
        code = f"""
import sys, time, torch
sys.path.insert(0, {ROOT!r})
import bench_legs as B
from oracle.gptq_oracle import torch_cpu_dequant_gptq, torch_cpu_forward_gptq
torch.set_num_threads({threads})
torch.manual_seed(1234)
t = B._cpu_tensors(4096, 4096, {gs}, torch.bfloat16)
x = (torch.randn(1, 4096) * 0.5).to(torch.bfloat16)
deq = torch.compile(torch_cpu_dequant_gptq)
t0 = time.perf_counter()
torch_cpu_forward_gptq(x, *t, 4, dequant=deq)
comp = time.perf_counter() - t0
ms, _ = B._time_cpu(lambda: torch_cpu_forward_gptq(x, *t, 4, dequant=deq), 2.0, 4)
print("RESULT", ms, comp)
"""
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=limit_s, cwd="/tmp",
                           env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
        for line in r.stdout.splitlines():
            if line.startswith("RESULT"):
                _, ms, comp = line.split()
                return round(float(ms), 3), f"torch.compile'd dequant (inductor, compile {float(comp):.0f} s), {threads} threads"
        return None, f"torch.compile leg failed: {(r.stderr or r.stdout)[-200:]}"
    except subprocess.TimeoutExpired:
        return None, f"torch.compile leg exceeded {limit_s} s"
    except Exception as e:  # noqa: BLE001
        return None, f"torch.compile leg unavailable: {e}"



class ModulesStep:
    """The step through the plugin classes' forward() (one launch per fused linear) with the glue as separate torch
    kernels -- what an HF model with fuse_siblings runs; kept as a comparison leg."""

    def __init__(self, layers, cfg, dtype, x0):
        self.layers, self.cfg, self.dtype = layers, cfg, dtype
        self.x_in = x0.clone()
        self.out = None
        self.eps = 1e-5

    def _rms(self, v, w):
        v32 = v.float()
        return w * (v32 * torch.rsqrt(v32.pow(2).mean(-1, keepdim=True) + self.eps)).to(self.dtype)

    def run(self):
        from gptqmodel_amd.utils.model import deinterleave_gate_up
        q = self.cfg["q"]
        h = self.x_in[None]
        for L in self.layers:
            qkv = L.qkv(self._rms(h, L.input_norm))
            h = h + L.o(qkv[:, :q])
            g, u = deinterleave_gate_up(L.gate_up(self._rms(h, L.post_norm)))
            h = h + L.down(torch.nn.functional.silu(g) * u)
        self.out = h
        return h


def extra_configs(args, cfg, layers, dtype, dev, gen, stream, make_step, mode, ms_headline, n_launch):
    """One object per BASELINE.json config.  Everything here runs OUTSIDE the timed region of the headline."""
    res = []
    gs = 128
    t_start = time.perf_counter()
    res.append(decode_entry("C2", f"headline ({mode})", cfg, ms_headline, n_launch, extra={"mode": mode, "id": f"c2_decode_{mode}"}))
    for other in ("chain", "modules"):
        if other == mode:
            continue
        try:
            st = make_step(other)
            ms, g = time_graph(st.run, stream, 50, 5)
            res.append(decode_entry("C2", f"same token step, mode={other}", cfg, ms, n_launch, extra={"mode": other, "id": f"c2_decode_{other}"}))
            del g, st
        except Exception as e:  # noqa: BLE001
            res.append({"config": "C2", "mode": other, "error": str(e)[:300]})
    # the same chain with the reference's per-weight rounding kept (decode form 4 = what GPTQHIP_DECODE_BITFAITHFUL=1 selects): the price of
    # bit-faithfulness next to the default's raw-code dequant (form 5; fp16 and bf16 activations)
    if mode == "chain":
        try:
            from gptqmodel_amd import ops as _ops
            _ops.set_decode_form(4)
            try:
                st = make_step("chain")
                ms, g = time_graph(st.run, stream, 50, 5)
            finally:
                _ops.set_decode_form(-1)
            res.append(decode_entry("C2", "same token step, decode form 4 (bit-faithful per-weight rounding, preload kernel)", cfg, ms, n_launch,
                                    extra={"mode": "chain, bit-faithful", "id": "c2_decode_chain_bitfaithful"}))
            del g, st
        except Exception as e:  # noqa: BLE001
            res.append({"config": "C2", "mode": "chain, bit-faithful", "error": str(e)[:300]})
    # round 1's way of timing the same 128 launches (kept for continuity, NOT a decode step): constant input, no glue, no data
    # dependencies between the launches -- shows what the dependent chain + fused glue cost on the same kernels
    try:
        xs = {k: (torch.randn((1, k), device=dev, generator=gen) * 0.5).to(dtype) for k in (cfg["hidden"], cfg["inter"])}
        pairs = [(lin, xs[lin.in_features if lin.in_features in xs else cfg["hidden"]]) for L in layers for lin in (L.qkv, L.o, L.gate_up, L.down)]

        def indep():
            for lin, x in pairs:
                lin(x)
        ms, g = time_graph(indep, stream, 50, 5)
        res.append(decode_entry("C2", "round-1 method: the same 128 launches with a constant input, no glue, no data dependencies (not a "
                                "decode step)", cfg, ms, n_launch, extra={"mode": "independent launches", "id": "c2_decode_independent"}))
        del g
    except Exception as e:  # noqa: BLE001
        res.append({"config": "C2", "mode": "independent launches", "error": str(e)[:300]})
    # the rounds 2-5 exact-arithmetic flag (GPTQHIP_GEMM_EXACT_BF16; bf16 activations only: 128 + q offsets on the bf16 matrix pipe inside the
    # rounds 1-5 kernel).  Superseded by decode form 5, which bf16 activations take by default since round 6; kept as a legacy flag
    if dtype == torch.bfloat16:
        try:
            from gptqmodel_amd.utils.decode_chain import DecodeStep as _DSx
            st = _DSx(layers, cfg["hidden"], cfg["q"], dtype, exact=True)
            st.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
            ms, g = time_graph(st.run, stream, 100, 10)
            res.append(decode_entry("C2", "bf16 decode chain with the LEGACY exact-arithmetic flag of rounds 2-5 (GPTQHIP_GEMM_EXACT_BF16: the older "
                                    "kernel; the default -- decode form 5 -- is the faster exact form)", cfg, ms, n_launch,
                                    extra={"mode": "chain, exact-arithmetic opt-in", "id": "c2_decode_exact_optin"}))
            del g, st
        except Exception as e:  # noqa: BLE001
            res.append({"config": "C2", "mode": "chain, exact-arithmetic opt-in", "error": str(e)[:300]})
    # headline-model prefill (one decoder layer at M=8192, desc_act=False) -- the TFLOPS half of the metric
    L0 = layers[0]
    lins = [L0.qkv, L0.o, L0.gate_up, L0.down]
    pre = prefill_entry("_prefill_headline", "one Llama-3-8B decoder layer's quantised linears (4 launches) at M=8192 tokens",
                        lins, 8192, dtype, dev, iters=5, kernel="gptqhip::tiled_kernel<BITS=4,...,BM=256,D=2>")
    # serving-batch sizes (the backend's weakest regime): one decoder layer's four linears at M = 128 and 512 rows
    for m_mid in (128, 512):
        e = prefill_entry("C2", f"one Llama-3-8B decoder layer's quantised linears (4 launches) at M={m_mid} rows (batched decode / chunked "
                          "prefill: prefill kernel with split-K on the narrow layers)", lins, m_mid, dtype, dev, iters=20)
        e["tokens_per_s_linear_stack_equiv"] = m_mid / (e["ms"] * cfg["layers"] * 1e-3)
        # at these sizes the op is neither purely HBM- nor MFMA-bound: both fractions are given
        bytes_layer = sum(algorithmic_bytes(m_mid, lin.in_features, lin.out_features, gs) for lin in lins)
        e["roofline"]["hbm_frac"] = bytes_layer / (e["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        e["id"] = f"c2_layer_m{m_mid}"
        res.append(e)
    # T1: the reference's own TFLOPS benchmark (M = 64..192), case by case
    try:
        res.append(t1_entry(dtype, dev, stream))
        res[-1]["id"] = "t1_best"
    except Exception as e:  # noqa: BLE001
        res.append({"config": "T1", "error": str(e)[:300]})
    # C3: act-order prefill, batch 32 x 2048 ctx = 65536 tokens
    torch.cuda.empty_cache()
    try:
        a44 = make_gptq(4096, 4096, gs, dev, gen, dtype, desc_act=True)
        res.append(prefill_entry("C3", "GPTQ int4 g128 desc_act=True (act-order g_idx gather), M=65536 (batch 32 x 2048 ctx), "
                                 "4096x4096 (q/o_proj shape)", [a44], 65536, dtype, dev))
        res[-1]["id"] = "c3_4096x4096_actorder"
        del a44
        # the same layer WITHOUT act-order in the same run: what the x-gather pass (1 GiB of HBM traffic at M = 65536) costs
        p44 = make_gptq(4096, 4096, gs, dev, gen, dtype, desc_act=False)
        plain = prefill_entry("C3", "", [p44], 65536, dtype, dev)
        res[-1]["same_layer_desc_act_false"] = {"value": plain["value"], "unit": "TFLOP/s", "ms": plain["ms"]}
        del p44
        agu = make_gptq(4096, 2 * cfg["inter"], gs, dev, gen, dtype, desc_act=True)
        res.append(prefill_entry("C3", "GPTQ int4 g128 desc_act=True, M=65536, 4096x28672 (fused gate_up)", [agu], 65536, dtype, dev,
                                 iters=2))
        res[-1]["id"] = "c3_gate_up_actorder"
        del agu
        # C3 the way a MODEL runs it (utils/hf_llama prefill path): one decoder layer of the act-order checkpoint at M = 65536 -- the
        # RMSNorm in front of q|k|v and of gate|up is ONE HIP kernel that writes the normalised activations already in the linear's
        # row order (ops.rmsnorm_gather -> forward_pregathered: no x-gather pass for those two), o_proj keeps its gather pre-pass
        # (its input comes from attention), down_proj's permutation is folded into gate / up at load time (no gather)
        from gptqmodel_amd import ops as _ops
        m_c3 = 65536
        lq = make_gptq(4096, cfg["q"] + 2 * cfg["kv"], gs, dev, gen, dtype, desc_act=True)
        lo = make_gptq(cfg["q"], 4096, gs, dev, gen, dtype, desc_act=True)
        lg = make_gptq(4096, 2 * cfg["inter"], gs, dev, gen, dtype, desc_act=True)
        ld = make_gptq(cfg["inter"], 4096, gs, dev, gen, dtype, desc_act=False)      # (act-order folded into gate / up)
        hh = (torch.randn((m_c3, 4096), device=dev, generator=gen) * 0.5).to(dtype)
        a_in = (torch.randn((m_c3, cfg["q"]), device=dev, generator=gen) * 0.5).to(dtype)
        m_in = (torch.randn((m_c3, cfg["inter"]), device=dev, generator=gen) * 0.5).to(dtype)
        nw1 = (1.0 + 0.1 * torch.randn(4096, device=dev, generator=gen)).to(dtype)

        def c3_layer(fused):
            if fused:
                lq.forward_pregathered(_ops.rmsnorm_gather(hh, nw1, 1e-5, lq.perm))
            else:
                lq(_ops.rmsnorm_gather(hh, nw1, 1e-5))
            lo(a_in)
            if fused:
                lg.forward_pregathered(_ops.rmsnorm_gather(hh, nw1, 1e-5, lg.perm))
            else:
                lg(_ops.rmsnorm_gather(hh, nw1, 1e-5))
            ld(m_in)

        flops = sum(2.0 * m_c3 * l.in_features * l.out_features for l in (lq, lo, lg, ld))
        for fused, what in ((True, "RMSNorm fused with the act-order gather of q|k|v and gate|up (rmsnorm_gather -> forward_pregathered), "
                                   "o_proj with its gather pre-pass, down_proj folded"),
                            (False, "the same layer with a separate x-gather pass in front of every act-order linear (round-2 path; the "
                                    "two RMSNorm kernels included as well)")):
            c3_layer(fused)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(2):
                c3_layer(fused)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 2
            tf = flops / ms / 1e9
            res.append({"config": "C3", "id": "c3_layer_fused" if fused else "c3_layer_gather_passes",
                        "workload": f"one Llama-3-8B decoder layer of a desc_act=True checkpoint at M={m_c3} (4 quantised "
                                                    f"linears + 2 RMSNorm kernels in the timed region): {what}",
                        "value": tf, "unit": "TFLOP/s", "ms": ms, "flops_counted": "the 4 GEMMs only",
                        "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS,
                                     "kernel": "gptqhip::tiled_kernel (+ rmsnorm_gather_kernel, gather_cols)"}})
        del lq, lo, lg, ld, hh, a_in, m_in
        torch.cuda.empty_cache()
        # the same checkpoint kind at batch-1 decode: the act-order permutation is applied inside the decode op
        from gptqmodel_amd.utils.decode_chain import DecodeStep as _DS
        act_layers = build_stack(cfg, lambda k, n: make_gptq(k, n, gs, dev, gen, dtype, desc_act=True), dev, gen, dtype)
        st = _DS(act_layers, cfg["hidden"], cfg["q"], dtype)
        st.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
        ms, g = time_graph(st.run, stream, 100, 10)
        res.append(decode_entry("C3", "Llama-3-8B GPTQ int4 g128 desc_act=True batch=1 decode, decode chain (permutation applied in the "
                                "kernel on the glued input row)", cfg, ms, n_launch, extra={"id": "c3_decode_actorder"}))
        del g, st, act_layers
        # ... and the way a LOADED model runs it: utils/hf_llama folds down_proj's permutation into gate|up's column order at load time
        # (fold_act_order_into_producers: exact), so down_proj takes the plain kernel; with random weights that is a down_proj without g_idx
        fold_layers = build_stack(cfg, lambda k, n: make_gptq(k, n, gs, dev, gen, dtype, desc_act=(k != cfg["inter"])), dev, gen, dtype)
        st = _DS(fold_layers, cfg["hidden"], cfg["q"], dtype)
        st.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
        ms, g = time_graph(st.run, stream, 100, 10)
        res.append(decode_entry("C3", "the same with down_proj's permutation folded into gate|up's columns (what utils/hf_llama does at load time)",
                                cfg, ms, n_launch, extra={"id": "c3_decode_actorder_folded"}))
        del g, st, fold_layers
    except Exception as e:  # noqa: BLE001
        res.append({"config": "C3", "error": str(e)[:300]})
    torch.cuda.empty_cache()
    # C4: AWQ g128 asym -- full-model decode through the same chain + one layer's prefill at M=2048
    try:
        awq_layers = build_stack(cfg, lambda k, n: make_awq(k, n, gs, dev, gen, dtype), dev, gen, dtype)
        from gptqmodel_amd.utils.decode_chain import DecodeStep
        st = DecodeStep(awq_layers, cfg["hidden"], cfg["q"], dtype)
        st.x_in.copy_((torch.randn(cfg["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
        ms, g = time_graph(st.run, stream, 100, 10)
        res.append(decode_entry("C4", "Llama-3-8B AWQ int4 g128 sym=False (AWQ packing, asymmetric qzeros) batch=1 decode, decode chain",
                                cfg, ms, n_launch, extra={"id": "c4_awq_decode"}))
        del g, st
        A0 = awq_layers[0]
        res.append(prefill_entry("C4", "AWQ int4 g128 asym, one decoder layer's linears (4 launches) at M=2048",
                                 [A0.qkv, A0.o, A0.gate_up, A0.down], 2048, dtype, dev, iters=5))
        res[-1]["id"] = "c4_awq_layer_m2048"
        del awq_layers, A0
    except Exception as e:  # noqa: BLE001
        res.append({"config": "C4", "error": str(e)[:300]})
    torch.cuda.empty_cache()
    # C5 at TP=1: Llama-3-70B shapes, 35.6 GB of packed weights on the one GPU
    try:
        if time.perf_counter() - t_start < 150:
            c70 = LLAMA3_70B
            base = {}

            def mk70(k, n):
                # one Philox draw per distinct shape; later layers reuse the words rotated + xored (distinct bytes in HBM)
                key = (k, n)
                if key not in base:
                    base[key] = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev, generator=gen)
                    return make_gptq(k, n, gs, dev, gen, dtype, derive_from=base[key].clone())
                salt = int(torch.randint(1, 2**31 - 1, (1,), generator=gen, device=dev).item())
                return make_gptq(k, n, gs, dev, gen, dtype, derive_from=torch.roll(base[key], 1 + salt % 97, 0) ^ salt)
            l70 = build_stack(c70, mk70, dev, gen, dtype)
            base.clear()
            st = DecodeStep(l70, c70["hidden"], c70["q"], dtype)
            st.x_in.copy_((torch.randn(c70["hidden"], device=dev, generator=gen) * 0.5).to(dtype))
            ms, g = time_graph(st.run, stream, 20, 3)
            res.append(decode_entry("C5", "Llama-3-70B GPTQ int4 g128 batch=1 decode at TP=1 (560 linears / 320 launches per token), decode chain",
                                    c70, ms, c70["layers"] * 4, extra={"tp": 1, "id": "c5_70b_decode_tp1"}))
            del g, st, l70
        else:
            res.append({"config": "C5", "skipped": "time budget of the default run"})
    except Exception as e:  # noqa: BLE001
        res.append({"config": "C5", "error": str(e)[:300]})
    torch.cuda.empty_cache()
    res.append(pre)
    return res
