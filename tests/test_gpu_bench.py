"""bench.py contract: one JSON line with the driver's keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=600, cwd=ROOT, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _run("--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "96", "--width", "160")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "frames/s" and c["sample"]
    assert c["cpu_model"] and c["cnn_ms_per_frame"] > 0 and c["warp_ms_per_frame"] > 0
    s2 = d["secondary"]      # the f32s rate, beside the line of record and outside its timed region
    assert s2["precision"] == "f32s" and s2["dtype"] == "f32x2f16" and s2["value"] > 0
    x3 = d["f32x3"]          # the f32x3 rate (products from three bfloat16 pieces per operand), with its own roofline and latency
    assert "error" not in x3, x3
    assert x3["precision"] == "f32x3" and x3["value"] > 0 and x3["roofline"]["peak"] == pytest.approx(2500.0 / 6.0)
    assert 0 < x3["latency"]["1280x720"]["ms_per_frame"] < 2.4 and 0 < x3["latency"]["512x288"]["ms_per_frame"] < 1.25
    # the reference's own operating point -- eval.py's batch-1 autoregressive clip loop -- beside the line, with a bound
    # that catches a regression of the per-frame path: 1.25 x the measured values (MI355X, rounds 3 and 4: 2.14-2.15 ms at
    # 720p, 1.10-1.11 ms at 512x288)
    lat = d["latency"]
    assert "error" not in lat, lat
    assert 0 < lat["1280x720"]["ms_per_frame"] < 2.7 and 0 < lat["512x288"]["ms_per_frame"] < 1.39
    assert abs(lat["1280x720"]["achieved_tflops"] - 154.5 / lat["1280x720"]["ms_per_frame"]) < 1e-6


def test_bench_default_line_carries_the_other_single_gpu_configs():
    """The line the driver records (no workload flags) is BASELINE configs[1] and, beside it and outside its timed region,
    carries configs[2] (B=64 720p tf_warp) and configs[4] (B=32 3840x2160, float16 mode), each with its own roofline."""
    d = _run("--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--no-latency")
    assert d["config"]["workload"].startswith("configs[1]") and d["dtype"] == "f32" and d["config"]["batch_per_gpu"] == 16
    c2, c4 = d["cfg2_tf_warp"], d["cfg4_f16_4k"]
    assert "error" not in c2 and "skipped" not in c2, c2
    assert "error" not in c4 and "skipped" not in c4, c4
    assert c2["workload"].startswith("configs[2]") and c2["roofline"]["bound"] == "hbm" and "flow_warp_strip_kernel" in c2["roofline"]["kernel"]
    assert c2["value"] > 100000 and c2["roofline"]["traffic"] and abs(c2["value"] - 64 / (c2["ms_per_step"] * 1e-3)) / c2["value"] < 1e-6
    assert 0.2 < c2["roofline"]["frac"] < 1.0 and c2["roofline"]["launches"] == c2["steps"]
    assert c4["workload"].startswith("configs[4]") and c4["dtype"] == "f16" and c4["roofline"]["bound"] == "hbm"
    assert c4["value"] > 200 and abs(c4["value"] - 32 / (c4["ms_per_step"] * 1e-3)) / c4["value"] < 1e-6
    assert c4["roofline"]["launches"] == 32 * c4["steps"] and 0.2 < c4["roofline"]["frac"] < 1.0
    assert c4["value"] > 1.1 * c4["pairs_everywhere"]["value"] > 250     # the calibrated mode is the faster one


def test_bench_other_kernel_classes_and_precision():
    d = _run("--steps", "1", "--warmup", "1", "--batch", "2", "--height", "96", "--width", "160", "--prof-class", "6",
             "--no-cpu-baseline", "--precision", "f16")
    assert d["dtype"] == "f16" and d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s"
    assert "cpu_baseline" not in d


def test_bench_f32s_precision_is_separately_named():
    d = _run("--steps", "2", "--warmup", "1", "--batch", "2", "--height", "96", "--width", "160", "--precision", "f32s",
             "--no-cpu-baseline")
    assert d["dtype"] == "f32x2f16" and "secondary" not in d and d["roofline"]["peak"] == pytest.approx(2500.0 / 3.0)


def test_bench_f32x3_precision_is_separately_named():
    d = _run("--steps", "2", "--warmup", "1", "--batch", "2", "--height", "96", "--width", "160", "--precision", "f32x3",
             "--no-cpu-baseline")
    assert d["dtype"] == "f32x3bf16" and "f32x3" not in d and d["roofline"]["peak"] == pytest.approx(2500.0 / 6.0)


def test_bench_tf_warp_workload():
    d = _run("--workload", "tf_warp", "--steps", "2", "--warmup", "1", "--batch", "3", "--height", "96", "--width", "160")
    assert d["metric"].startswith("tf_warp") and d["dtype"] == "f32" and d["roofline"]["bound"] == "hbm"
    assert "stn_kernel" in d["roofline"]["kernel"] and d["roofline"]["launches"] == 2 and d["cpu_baseline"]["cores"] == 1


def test_bench_rccl_calls_of_the_sharded_path_run_on_one_rank():
    """The N > 1 path's RCCL calls (async gather of the shard's output, device barrier, max-reduce of the time) on a
    one-rank process group: what a one-GPU box can exercise of the backend the 8-GPU run uses."""
    d = _run("--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "4", "--call-batch", "2", "--height", "96",
             "--width", "160", "--no-cpu-baseline", "--no-secondary", env={"DVSG_BENCH_FORCE_DIST": "1"})
    c = d["config"]
    assert c["backend"] == "rccl" and c["gather"] is True and c["ranks_seen"] == 1
    assert c["windows_per_step"] == 4 and c["windows_per_call"] == 2 and d["value"] > 0
    assert c["rccl_version"] and c["rccl_version"][0].isdigit()
    r = c["ms_per_step_over_ranks"]       # slowest / fastest rank: a straggler shows as min << max
    assert 0 < r["min"] <= r["max"] and abs(r["max"] - d["ms_per_step"]) < 1e-9


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what the driver runs): the script spawns
    its two ranks itself, each stabilises its shard in calls of `windows_per_call`, the frames are
    gathered to rank 0 every step, and ONE JSON line comes back.  Rehearsal on the one-GPU box: both
    ranks share cuda:0 and the gather goes through gloo (the real run uses RCCL, one GPU per rank)."""
    env = {"DVSG_BENCH_BACKEND": "gloo", "DVSG_BENCH_SHARE_DEVICE": "1"}
    env_clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "6", "--call-batch", "4", "--height", "96", "--width", "160"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(env_clean, **env))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["gather"] is True and c["backend"] == "gloo"
    assert c["batch_per_gpu"] == 6 and c["windows_per_call"] == 4 and c["windows_per_step"] == 12
    assert abs(d["value"] - 12 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d and d["roofline"]["launches"] > 0


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that dies must fail the whole run (non-zero exit, no JSON line), not hang it."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env = dict(env_clean, DVSG_BENCH_BACKEND="gloo", DVSG_BENCH_SHARE_DEVICE="1", DVSG_BENCH_FAIL_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--batch", "2", "--height", "64", "--width", "96"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_rank_that_cannot_join_exits_with_its_identity():
    """A rank whose peers never arrive (or whose RCCL initialisation fails) must say who and where it is and exit
    non-zero instead of hanging: WORLD_SIZE=2 with only rank 0 started, 5 s rendezvous timeout."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               DVSG_BENCH_BACKEND="gloo", DVSG_BENCH_SHARE_DEVICE="1", DVSG_BENCH_INIT_TIMEOUT_S="5")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--batch", "2", "--height", "64", "--width", "96"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 3, (out.returncode, out.stderr[-1500:])
    assert "rank 0 of 2 could not join" in out.stderr and "MASTER_PORT" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
