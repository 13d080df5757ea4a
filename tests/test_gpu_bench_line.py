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
    assert "ALGORITHMIC bytes" in r["frac_counts"]
    assert "configs[1]" in d["config"]["workload"]
    assert d["config"]["settle_steps"] >= 40 and d["ms_per_step_from_idle"] > 0 and d["value_from_idle"] > 0
    # the tolerance the parity tests hold the chains to travels with the line
    assert d["parity"]["tol"] == 1e-5 and d["parity"]["chain_bound"] == "propagated" and "above_plain_share" in d["parity"]


def test_default_line_carries_the_metric_chain_and_the_north_star_target():
    """the metric string names the four-block chain and the north star states ">= 100x" on the FIR + FftFilter
    pair: both are first-class objects of the driver's line, each with its GPU rate, its roofline fraction and its own
    single-thread CPU leg (the oracle chain on this host)."""
    d = _run("--steps", "3", "--warmup", "1", "--cpu-seconds", "4", "--no-dropin")
    for k in CONTRACT + ("cpu_baseline", "metric_chain", "north_star_target"):
        assert k in d, k
    m = d["metric_chain"]
    assert m["workload_key"] == "full_chain_fused" and "RationalResampler(1:4)" in m["workload"] and m["value"] > 0
    assert 0 < m["roofline"]["frac"] < 1 and m["roofline"]["peak"] == 8000.0 and m["roofline"]["alg_bytes_per_sample"] == 9.0
    assert m["cpu_baseline"]["cores"] == 1 and m["cpu_baseline"]["kind"] == "port" and m["cpu_baseline"]["value"] > 0
    assert abs(m["gpu_over_cpu_1thread_port"] - m["value"] / m["cpu_baseline"]["value"]) < 0.1 * m["gpu_over_cpu_1thread_port"]
    t = d["north_star_target"]
    assert t["workload_key"] == "fir_fft_chain" and t["gpu_msamples"] > 0 and t["cpu_msamples_1thread"] > 0
    assert abs(t["ratio"] - t["gpu_msamples"] / t["cpu_msamples_1thread"]) < 0.1 * t["ratio"]
    assert t["met"] == (t["ratio"] >= 100.0) and t["met"]
    assert "configs[1]" in d["config"]["workload"]              # `value` stays the configuration the metric is quoted on
    # the multi-GPU prediction is regenerated from THIS run's measured configs[3] step, not from a constant
    from rustradio_amd import multi
    pr = d["multi_gpu_prediction"]
    step = d["others"]["fm_multi"]["ms_per_step"]
    assert pr["measured_fm_multi_ms_per_step"] == step and pr["tile_steps"] == 4 and "never measured" in pr["what"]
    for key, tile in (("complex_f32_source", 4 * 19_200_000), ("u8_source", 4 * 4_800_000)):
        assert pr[key]["tile_bytes"] == tile
        for n in (2, 4, 8):
            assert pr[key][str(n)] == multi.predict_fanout(n, tile, 4 * step)
    # (structure only: the step of a 3-step run on a cold box can be 1.5x the sustained one, which moves the efficiencies)
    f8, u8 = pr["complex_f32_source"]["8"], pr["u8_source"]["8"]
    assert f8["bcast"]["efficiency"] < f8["scatter_allgather"]["efficiency"] <= 1.0      # one broadcast is single-link-bound
    assert u8["bcast"]["efficiency"] >= f8["bcast"]["efficiency"] and u8["bcast"]["fanout_ms_per_tile"] < f8["bcast"]["fanout_ms_per_tile"]


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
    for name in ("fm_multi_u8", "channelizer", "channelizer_model"):
        assert d["others"][name]["msamples_per_s"] > 0
    assert "translate" in d["others"]["channelizer"]["workload"]  # configs[4], N > 1: one channel offset per rank
    # ... measured on the ON-PARITY rotator by default, named in the workload string; the opt-in model beside it, labelled
    assert d["others"]["channelizer"]["rotator"] == "replay" and "rotator=replay" in d["others"]["channelizer"]["workload"]
    assert d["others"]["channelizer_model"]["rotator"] == "model" and "OFF parity" in d["others"]["channelizer_model"]["workload"]
    # the line anchors its own scaling curve: the same workload on one rank, source resident
    a = d["scale_anchor"]
    assert a["workload_key"] == d["config"]["workload_key"] == "fm_multi" and a["n1_value"] > 0
    assert abs(d["scaling_efficiency_vs_anchor"] - d["value"] / (2 * a["n1_value"])) < 1e-3
    # ... and carries the fan-out PREDICTION (no run on more than one GPU has happened yet) for bcast and mesh at 2 / 4 / 8
    pr = c["predicted"]
    assert set(pr["at_2_4_8_gpus"]) == {"2", "4", "8"}
    for n in ("2", "4", "8"):
        for algo in ("bcast", "scatter_allgather"):
            e = pr["at_2_4_8_gpus"][n][algo]
            assert e["fanout_ms_per_tile"] > 0 and 0 < e["efficiency"] <= 1
    assert pr["assumptions"]["xgmi_link_gbs"] == 153.0
    # the library's own fan-out (rr_fanout_*) is the measured one whenever every rank has a GPU
    assert c["fanout"] == ("abi" if c["backend"] == "rccl" else "torch")
