"""bench.py's stdout line: compact, strict JSON, below the driver's tail limit (BENCH_r05.json: parsed = null on a 22-KB line).

The recorded input is round 5's full result object (profiles/r05_bench_headline.json: what bench.py used to print as its
line and now writes to the detail file); headline_line() is what reaches stdout."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "whole_run_iters_per_s")


def _recorded():
    with open(os.path.join(ROOT, "profiles", "r05_bench_headline.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def _no_constants(name):
    raise AssertionError("non-strict JSON constant " + name)


def test_line_is_compact_strict_and_complete():
    res = _recorded()
    res["value_incl_run_tail"] = 80.0
    res["config"]["regroup"] = {"own_order_after_warmup": False, "warmup_calls": 5, "warmup_ms": 300.0}
    s = bench.headline_line(res)
    assert "\n" not in s and len(s.encode()) < 8192, len(s)
    line = json.loads(s, parse_constant=_no_constants)
    for k in REQUIRED:
        assert k in line, k
    rl = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl, k
    assert "by_kernel" not in rl and "valu_floor" not in rl
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    # the section-8(d) quantity: B_iter / the full-work launch's duration / 8 TB/s
    assert abs(rl["achieved"] - rl["algorithmic_bytes_per_launch"] / (rl["kernel_ms"] * 1e-3) / 1e9) < 1.0
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["value"] == res["value"] and line["ms_per_step"] == res["ms_per_step"]


def test_nan_and_oversized_inputs_still_give_a_valid_line():
    res = _recorded()
    res["config"]["final_obj"] = float("nan")
    res["roofline"]["traffic"] = float("inf")
    res["config"]["workload"] = "w" * 20000                 # whatever grows: the line sheds optional parts, never the contract's keys
    res["whole_run_iters_per_s"] = {"block": float("nan"), "shuffled": 1.0}
    s = bench.headline_line(res)
    assert len(s.encode()) < 8192
    line = json.loads(s, parse_constant=_no_constants)
    assert line["roofline"]["traffic"] is None and line["whole_run_iters_per_s"]["block"] is None
    assert all(k in line for k in REQUIRED)


def test_emit_writes_the_detail_beside_the_line(tmp_path, capsys):
    res = _recorded()
    res["regimes"]["block"]["final_obj"] = float("nan")
    out = tmp_path / "detail.json"
    bench.emit(res, str(out))
    cap = capsys.readouterr()
    lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0].encode()) < 8192          # stdout: the one line, nothing else
    json.loads(lines[0], parse_constant=_no_constants)
    detail = json.loads(out.read_text(), parse_constant=_no_constants)
    assert "regimes" in detail and "per_iter_ms" in detail["regimes"]["block"]
    assert detail["regimes"]["block"]["final_obj"] is None and not math.isnan(detail["value"])
