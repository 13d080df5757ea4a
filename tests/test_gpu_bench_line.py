"""GPU: bench.py itself, as the driver runs it — one short single-GPU line and one two-rank line (`--gpus 2`: the parent
starts its ranks before anything touches the GPU; on the one-GPU box they share the device and the fan-out runs over gloo).
Checks the contract's keys, that both ranks really ran, and the N > 1 extras (collective, resident-source reference)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                   # rank 0 prints ONE JSON line
    return json.loads(lines[0])


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def test_single_gpu_line():
    d = _run("--steps", "3", "--warmup", "1", "--no-cpu", "--no-others")
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["launches"] == 3 and 0 < r["frac"] < 1 and r["peak"] == 8000.0
    assert "configs[1]" in d["config"]["workload"]


def test_two_rank_line():
    d = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu")
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "configs[3]" in d["config"]["workload"]
    c = d["collective"]
    assert c["ranks"] == 2 and c["tile_steps"] == 4 and c["tile_bytes"] == 4 * 19_200_000 and c["broadcasts_timed"] >= 1
    assert c["algorithm"] in ("bcast", "scatter_allgather")
    res = d["resident_source"]
    assert res["value"] > 0 and d["value"] > 0
    assert 0 < d["fanout_efficiency"] < 1.5                       # (over gloo on one GPU the fan-out dominates; over RCCL it may not)
    if c["backend"] == "gloo":
        assert res["value"] > d["value"]
    # both shards ran: 2 ranks x 32 channels x 2.4e6 samples x 3 steps in the timed region
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 3 - 2 * 32 * 2_400_000 * 3) / (2 * 32 * 2_400_000 * 3) < 0.01
    for name in ("fm_multi_u8", "channelizer"):
        assert d["others"][name]["msamples_per_s"] > 0
    assert "translate" in d["others"]["channelizer"]["workload"]  # configs[4], N > 1: one channel offset per rank
