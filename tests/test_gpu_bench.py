"""bench.py contract: one JSON line with the driver's keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
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


def test_bench_other_kernel_classes_and_precision():
    d = _run("--steps", "1", "--warmup", "1", "--batch", "2", "--height", "96", "--width", "160", "--prof-class", "6",
             "--no-cpu-baseline", "--precision", "f16")
    assert d["dtype"] == "f16" and d["roofline"]["bound"] == "hbm" and d["roofline"]["unit"] == "GB/s"
    assert "cpu_baseline" not in d


def test_bench_tf_warp_workload():
    d = _run("--workload", "tf_warp", "--steps", "2", "--warmup", "1", "--batch", "3", "--height", "96", "--width", "160")
    assert d["metric"].startswith("tf_warp") and d["dtype"] == "f32" and d["roofline"]["bound"] == "hbm"
    assert d["roofline"]["kernel"] == "stn_kernel" and d["roofline"]["launches"] == 2 and d["cpu_baseline"]["cores"] == 1
