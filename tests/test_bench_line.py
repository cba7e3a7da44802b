"""bench.py's stdout contract: ONE compact JSON line the driver can parse (round 4's 21 KB line came back `parsed: null`).
The line builder is a pure function; it is fed a real kept record (profiles/r04_bench_final.json: 28 T1 cases, 15 config
entries, e2e, CPU thread sweep = 20.8 KB) and a padded worst case."""
import copy
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.fixture()
def kept_record():
    with open(os.path.join(ROOT, "profiles", "r04_bench_final.json")) as f:
        d = json.load(f)
    ids = ["c2_decode_chain", "c2_decode_modules", "c2_decode_independent", "c2_layer_m128", "c2_layer_m512", "t1_best",
           "c3_4096x4096_actorder", "c3_gate_up_actorder", "c3_layer_fused", "c3_layer_gather_passes", "c3_decode_actorder",
           "c4_awq_decode", "c4_awq_layer_m2048", "c5_70b_decode_tp1"]
    assert len(d["configs"]) == len(ids)
    for c, i in zip(d["configs"], ids):
        c["id"] = i
    return d


def _check(line, detail):
    assert "\n" not in line
    assert len(line.encode()) < bench.RESULT_LINE_LIMIT == 4096
    obj = json.loads(line)
    for k in REQUIRED:
        assert k in obj, k
    assert obj["metric"] == detail["metric"] and obj["steps"] == detail["steps"] and obj["warmup"] == detail["warmup"]
    assert abs(obj["value"] - detail["value"]) <= 1e-3 * detail["value"]
    assert abs(obj["ms_per_step"] - detail["ms_per_step"]) <= 1e-3 * detail["ms_per_step"]
    assert "workload" in obj["config"] and "model" not in obj["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in obj["roofline"], k
    assert abs(obj["roofline"]["frac"] - obj["roofline"]["achieved"] / obj["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample", "cpu_model", "c1_ms"):
        assert k in obj["cpu_baseline"], k
    return obj


def test_result_line_of_a_real_record_is_compact_and_complete(kept_record):
    obj = _check(bench.result_line(kept_record), kept_record)
    summ = {r["id"]: r for r in obj["configs_summary"]}
    # one record per config entry + the reference benchmark's gate cases + the prefill headline
    for i in ("c2_decode_chain", "c3_4096x4096_actorder", "c4_awq_decode", "c5_70b_decode_tp1", "t1_best", "t1_attn_m128", "t1_mlp_up_m136",
              "c2_layer_m8192"):
        assert i in summ, i
        assert {"value", "unit", "frac"} <= set(summ[i])
    assert summ["t1_attn_m128"]["us"] == pytest.approx(15.12, rel=1e-2)


def test_result_line_survives_a_padded_worst_case(kept_record):
    d = copy.deepcopy(kept_record)
    d["configs"] = d["configs"] * 6                  # 84 entries: the optional blocks must go, the headline must stay
    d["note"] = "x" * 1000
    obj = _check(bench.result_line(d), d)
    assert "configs_summary" not in obj


def test_result_line_reports_failed_legs_briefly(kept_record):
    d = copy.deepcopy(kept_record)
    d["configs"][3] = {"config": "C2", "id": "c2_layer_m128", "error": "RuntimeError: " + "y" * 500}
    obj = _check(bench.result_line(d), d)
    rec = next(r for r in obj["configs_summary"] if r["id"] == "c2_layer_m128")
    assert len(rec["error"]) <= 80


def test_emit_puts_exactly_one_line_on_stdout_and_the_detail_elsewhere(kept_record, tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(kept_record)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1
    _check(lines[0], kept_record)
    assert err.getvalue().startswith("BENCH_DETAIL {")
    with open(tmp_path / "gpurun_out" / "bench_detail.json") as f:
        assert json.load(f)["configs"][5]["cases"][0]["case_id"] == "mlp_up_m64"


def test_bench_legs_import_without_a_gpu_and_keep_the_oracle_out_of_the_timed_path():
    """bench_legs.py (the legs outside the timed region) imports on a GPU-less box and exposes what bench.py calls lazily; bench.py itself
    never imports the oracle -- only bench_legs.cpu_baseline does, as the thing timed beside the GPU path."""
    import bench_legs
    for name in ("extra_configs", "cpu_baseline", "t1_entry", "prefill_entry", "decode_entry", "ModulesStep", "T1_CASES"):
        assert hasattr(bench_legs, name), name
    assert len(bench_legs.T1_CASES) == 28           # the reference benchmark's grid (scripts/benchmark_marlin_a100.py:35-44)
    with open(os.path.join(ROOT, "bench.py")) as f:
        src = f.read()
    assert "from oracle" not in src and "import oracle" not in src
