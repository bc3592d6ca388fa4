"""bench.py's stdout line stays parsable: <= 4 kB, strict JSON, contract keys + roofline + cpu_baseline present.

The canned input is round 5's whole 20.7 kB result (tests/golden/bench_full_result_r05.json), the line that left
`BENCH_r05.parsed` null.
"""
import json
import math
import os

import bench_line

HERE = os.path.dirname(os.path.abspath(__file__))


def _full():
    with open(os.path.join(HERE, "golden", "bench_full_result_r05.json")) as f:
        return json.load(f)


def _no_nan(o):
    if isinstance(o, float):
        assert math.isfinite(o)
    elif isinstance(o, dict):
        for v in o.values():
            _no_nan(v)
    elif isinstance(o, list):
        for v in o:
            _no_nan(v)


def test_line_is_small_and_round_trips():
    full = _full()
    assert len(json.dumps(full)) > 16000  # (the canned result really is the oversized one)
    line = bench_line.dumps(full, "gpurun_out/bench_extras.json")
    assert "\n" not in line
    assert len(line.encode()) < bench_line.MAX_LINE_BYTES
    back = json.loads(line)
    _no_nan(back)
    for k in bench_line.CONTRACT_KEYS:
        assert k in back, k
    assert back["metric"].startswith("ref-sketch containment queries/sec")
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert back["steps"] == full["steps_timed"] and back["steps_requested"] == full["steps"]
    assert back["config"]["workload"] and len(back["config"]) <= 20
    assert all(not isinstance(v, (dict, list)) for v in back["config"].values())
    rl = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rl
    assert len(rl) <= 12 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    cb = back["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb
    assert back["extras"] == "gpurun_out/bench_extras.json"


def test_nan_and_long_strings_do_not_break_the_line():
    full = _full()
    full["ms_per_step"] = float("nan")
    full["config"]["workload"] = "w" * 5000
    full["roofline"]["traffic"] = float("inf")
    full["cpu_baseline"]["sample"] = "s" * 3000
    line = bench_line.dumps(full)
    assert len(line.encode()) < bench_line.MAX_LINE_BYTES
    back = json.loads(line)
    _no_nan(back)
    assert back["ms_per_step"] is None and back["roofline"]["traffic"] is None


def test_multi_gpu_line_without_cpu_baseline():
    full = _full()
    full["n_gpus"] = 8
    full["cpu_baseline"] = None
    full["train"] = None
    back = json.loads(bench_line.dumps(full))
    assert back["cpu_baseline"] is None and back["n_gpus"] == 8 and "train_cpu_baseline" not in back
