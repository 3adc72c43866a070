"""bench.py prints ONE JSON line with the fields the driver and the judge read.  Checked here on a small
workload (GPU), plus the parts of the script that need no GPU (CPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_share_and_argument_defaults():
    sys.path.insert(0, ROOT)
    import bench
    n = bench.cpu_share()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert bench.CFG3["update_freq"] == 400.0 and bench.CFG3["measurement_freq"] == 30.0
    assert bench.HBM_PEAK_GBS == 8000.0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--workload", "cfg5"], ["--workload", "cfg3mr"], ["--workload", "cfg2"]])
def test_bench_json_contract(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "42", "--warmup", "14", "--batch-per-gpu", "4096",
           "--predict-only-steps", "60"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 42 and d["warmup"] == 14
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - d["config"]["global_batch"] * 42 / (d["ms_per_step"] * 1e-3 * 42)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) / rf["achieved"] < 1e-6
    assert d["nonfinite_filters"] == 0
    if not extra:
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
        assert d["dtype"] == "f32"
    if extra == ["--workload", "cfg2"]:
        assert d["dtype"] == "f64"


@pytest.mark.gpu
def test_bench_two_ranks_print_one_json_line():
    """The driver's N > 1 launch line, rehearsed with both ranks on the one GPU of the box: stdout of the whole job is
    exactly one JSON line (gloo's own chatter must not land there), value = units of all ranks / max-over-ranks time."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "42", "--warmup", "14",
           "--batch-per-gpu", "4096", "--predict-only-steps", "60"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["scaling"] == "weak"
    assert abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert "cpu_baseline" not in d          # rank 0 at N = 1 only
