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


def _strict(text):
    def fail(c):
        raise AssertionError(f"not strict JSON: {c}")
    return json.loads(text, parse_constant=fail)                 # NaN / Infinity / -Infinity are not JSON


def _run(*args, tmp="bench_detail_test.json"):
    """-> (the ONE stdout line, parsed; the detail file, parsed).  The line is what the driver reads: strict JSON, under 4 KB
    (round 5's 22 KB line left BENCH_r05.json with parsed = null), and the LAST thing on stdout."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    detail_path = os.path.join(ROOT, "gpurun_out", tmp)
    if os.path.exists(detail_path):
        os.unlink(detail_path)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--detail-out", detail_path, *args], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                   # rank 0 prints ONE JSON line
    assert out.stdout.strip().split("\n")[-1] == lines[0]        # ... and nothing after it
    assert len(lines[0].encode()) < 4096, len(lines[0])
    d = _strict(lines[0])
    for v in _strings(d):
        assert len(v) <= 120, v                                  # (the driver's record keeps 120 characters per string)
    with open(detail_path) as f:
        detail = _strict(f.read())
    assert d["detail"].endswith(tmp)
    return d, detail


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def test_single_gpu_line():
    d, det = _run("--steps", "3", "--warmup", "1", "--no-cpu", "--no-others")
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["launches"] == 3 and 0 < r["frac"] < 1 and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["alg_bytes_per_launch"] == 1.6e9 and 0 < r["executed_fp32_frac"] < r["hbm_frac"] == r["frac"]
    assert abs(r["achieved"] - 1.6e9 / (r["avg_kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    assert "configs[1]" in d["config"]["workload"] and d["dtype"] == "f32" and d["config"]["samples_per_step_per_gpu"] == 100_000_000
    assert d["config"]["settle_steps"] >= 40 and d["value_from_idle"] > 0
    # the tolerance the parity tests hold the chains to travels with the line
    assert d["parity"]["tol"] == 1e-5 and d["parity"]["chain_bound"] == "propagated" and "above_plain_share_max" in d["parity"]
    # bench.py checked what it timed: 16 segments of the 1e8-sample output against an f64 evaluation of the filter
    assert d["verified"] == {"ok": True, "workloads": 1, "segments_each": 16, "failed": []}
    v = det["verified"]["fftfilter"]
    assert v["ok"] and v["segments"] == 16 and v["max_err"] <= 1e-5 and v["produced"] == (100_000_000 // 623) * 623
    assert "ALGORITHMIC bytes" in det["roofline"]["frac_counts"] and det["value"] == d["value"]


def test_a_wrong_output_fails_the_check():
    """bench_verify itself: the f64 evaluation against an output buffer that was tampered with after the step"""
    import torch
    sys.path.insert(0, ROOT)
    import bench_verify
    import bench_workloads
    import rustradio_amd as rr
    rr.set_device(0)
    dev = torch.device("cuda", 0)
    w = bench_workloads.make("fm_chain", dev, 0, 1, lambda gen, numel, dtype: gen())
    ok = bench_verify.verify(w, torch.cuda.current_stream())
    assert ok["ok"] and ok["segments"] == 16 and ok["max_err"] <= 1.0, ok
    real_step = w.step

    def tampering_step(stream, src_ptr=None):
        u = real_step(stream, src_ptr)
        torch.cuda.synchronize()
        w.bufs[-1][:int(w.last_p)] += 1e-3            # a kernel that is 1e-3 rad off everywhere
        return u
    w.step = tampering_step
    bad = bench_verify.verify(w, torch.cuda.current_stream())
    assert not bad["ok"] and bad["max_err"] > 1.0, bad


def test_default_line_carries_the_metric_chain_and_the_north_star_target():
    """The metric string names the four-block chain and the north star states ">= 100x" on the FIR + FftFilter pair: both are
    first-class objects of the driver's line (compact) and of the detail file (full), each with its GPU rate, its roofline
    fractions and its own single-thread CPU leg (the oracle chain on this host).  Every workload of the run is checked against
    f64 and none has a null roofline fraction."""
    d, det = _run("--steps", "3", "--warmup", "1", "--cpu-seconds", "4", "--no-dropin")
    for k in CONTRACT + ("cpu_baseline", "metric_chain", "north_star_target", "verified", "others_brief"):
        assert k in d, k
    cb = d["cpu_baseline"]
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0 and cb["host_cores"] >= 1 and cb["all_cores_value"] > 0
    m = d["metric_chain"]
    assert m["workload_key"] == "full_chain_fused" and m["value"] > 0 and 0 < m["hbm_frac"] < 1 and 0 < m["executed_fp32_frac"] < 1
    assert abs(m["x_cpu_1thread"] - m["value"] / m["cpu_1thread"]) < 0.1 * m["x_cpu_1thread"]
    mf = det["metric_chain"]
    assert "RationalResampler(1:4)" in mf["workload"] and mf["alg_bytes_per_sample"] == 9.0
    assert mf["cpu_baseline"]["cores"] == 1 and mf["cpu_baseline"]["kind"] == "port"
    t = d["north_star_target"]
    assert t["workload_key"] == "fir_fft_chain" and t["gpu_msamples"] > 0 and t["cpu_msamples_1thread"] > 0
    assert abs(t["ratio"] - t["gpu_msamples"] / t["cpu_msamples_1thread"]) < 0.1 * t["ratio"]
    assert t["met"] == (t["ratio"] >= 100.0) and t["met"]
    assert "configs[1]" in d["config"]["workload"]              # `value` stays the configuration the metric is quoted on
    # every workload was checked against f64 after its timed passes, and every one has both roofline fractions
    assert d["verified"]["ok"] and d["verified"]["workloads"] >= 18 and d["verified"]["failed"] == []
    for name, o in det["others"].items():
        assert o["verified"]["ok"], (name, o["verified"])
        assert o["dominant_kernel_hbm_frac"] is not None and o["dominant_kernel_executed_fp32_frac"] is not None, name
        assert o["bound"] in ("hbm", "vector_fp32", "sequential_rotator"), name
    for name in ("channelizer", "fir_fft_chain", "full_chain_fused", "rtl_fm_example", "fm_multi"):
        assert name in d["others_brief"] and d["others_brief"][name][0] > 0
    # the multi-GPU prediction is regenerated from THIS run's measured configs[3] step, not from a constant (detail file)
    from rustradio_amd import multi
    pr = det["multi_gpu_prediction"]
    step = det["others"]["fm_multi"]["ms_per_step"]
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
    d, det = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu")
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "configs[3]" in d["config"]["workload"]
    c = det["collective"]
    assert c["ranks"] == 2 and c["tile_steps"] == 4 and c["tile_bytes"] == 4 * 19_200_000 and c["broadcasts_timed"] >= 1
    assert c["algorithm"] in ("bcast", "scatter_allgather")
    assert d["collective"]["tile_bytes"] == c["tile_bytes"] and d["collective"]["algorithm"] == c["algorithm"]
    res = det["resident_source"]
    assert res["value"] > 0 and d["value"] > 0 and d["resident_source_value"] == res["value"]
    assert 0 < d["fanout_efficiency"] < 1.5                       # (over gloo on one GPU the fan-out dominates; over RCCL it may not)
    if c["backend"] == "gloo":
        assert res["value"] > d["value"]
    # both shards ran: 2 ranks x 32 channels x 2.4e6 samples x 3 steps in the timed region
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 3 - 2 * 32 * 2_400_000 * 3) / (2 * 32 * 2_400_000 * 3) < 0.01
    for name in ("fm_multi_u8", "channelizer", "channelizer_model"):
        assert det["others"][name]["msamples_per_s"] > 0
    assert "translate" in det["others"]["channelizer"]["workload"]  # configs[4], N > 1: one channel offset per rank
    # ... measured on the ON-PARITY rotator by default, named in the workload string; the opt-in model beside it, labelled
    assert det["others"]["channelizer"]["rotator"] == "replay" and "rotator=replay" in det["others"]["channelizer"]["workload"]
    assert det["others"]["channelizer_model"]["rotator"] == "model" and "OFF parity" in det["others"]["channelizer_model"]["workload"]
    # rank 0 checked its shard (and its translated channelizers) against f64
    assert d["verified"]["ok"] and det["verified"]["fm_multi"]["ok"] and det["verified"]["channelizer"]["ok"]
    # the line anchors its own scaling curve: the same workload on one rank, source resident
    a = det["scale_anchor"]
    assert a["workload_key"] == d["config"]["workload_key"] == "fm_multi" and a["n1_value"] > 0 and d["scale_anchor_n1_value"] == a["n1_value"]
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
