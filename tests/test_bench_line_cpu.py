"""CPU: the size / strictness guard of bench.py's stdout line (round 5's 22 KB line left the driver's
record with parsed = null).  compact_line() drops the optional tables, largest first, until the line fits 4,000 bytes, and
refuses NaN / Infinity; _clean() turns non-finite floats (and numpy scalars) into null for both the line and the detail file."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench as b
    finally:
        sys.argv = argv
    return b


def _line(n_others):
    return {"metric": "m", "value": 1.0, "unit": "Msamples/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 110}, "roofline": {"bound": "hbm", "frac": 0.6}, "cpu_baseline": {"value": 90.0},
            "metric_chain": {"value": 1.0}, "north_star_target": {"ratio": 1.0}, "parity": {"tol": 1e-5},
            "verified": {"ok": True}, "verified_worst": {"max_err": 1e-7},
            "others_brief": {f"workload_{i}": [1.0, 2.0, 0.5, 0.25] for i in range(n_others)}, "detail": "gpurun_out/bench_detail.json"}


def test_line_fits_and_is_strict(bench):
    s = bench.compact_line(_line(17))
    assert len(s.encode()) <= bench.LINE_LIMIT < 4096 and "others_brief" in json.loads(s)
    # far too many workloads: the optional table goes, the contract keys stay
    s = bench.compact_line(_line(400))
    d = json.loads(s)
    assert len(s.encode()) <= bench.LINE_LIMIT and "others_brief" not in d
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k


def test_non_finite_values_never_reach_the_line(bench):
    bad = _line(3)
    bad["roofline"]["frac"] = float("nan")
    with pytest.raises(ValueError):
        bench.compact_line(bad)                               # json.dumps(allow_nan=False)
    ok = bench._clean({"a": float("nan"), "b": [float("inf"), 1.5, np.float32(2.0), np.int64(3)], "c": {"d": -float("inf")}})
    assert ok == {"a": None, "b": [None, 1.5, 2.0, 3], "c": {"d": None}}
    json.dumps(ok, allow_nan=False)
    assert json.loads(bench.compact_line(bench._clean(bad)))["roofline"]["frac"] is None
