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
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.HBM_COPY_GBS == 6290.0
    # the kernel named in the roofline record is the one the workload's dominant tick is launched on
    lanes = {"coop_ticks": 0}; coop = {"coop_ticks": 1}
    assert bench.kernel_name(lanes, "f32", False) == "k_predict<float>" and bench.kernel_name(lanes, "f64", True) == "k_step<double>"
    assert bench.kernel_name(coop, "f64", True) == "kw_tick<double,step>" and bench.kernel_name(coop, "f64", False) == "k_predict<double>"
    assert bench.kernel_name(lanes, "f32", False, mr=True) == "k_predict<float,MR>"
    # placement is best effort and never raises: without a KFD topology (this container) it says why nothing was pinned
    before = os.sched_getaffinity(0)
    pl = bench.pin_to_gpu_numa_node(1, 2)
    assert isinstance(pl, dict) and "pinned" in pl and (pl["pinned"] or pl["why"])
    os.sched_setaffinity(0, before)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--workload", "cfg5"], ["--workload", "cfg3mr"], ["--workload", "cfg2"], ["--workload", "rotors"],
                                   ["--workload", "hardware"], ["--workload", "hardware", "--dtype", "f64"]])
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
    assert rf["bound"] in ("hbm", "infinity_cache", "split") and rf["bound"] == rf["served_by"] and rf["bound_class"] == "memory"
    assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "hbm_frac" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) / rf["achieved"] < 1e-6
    assert d["nonfinite_filters"] == 0
    assert d["repeats"] >= 1 and abs(d["region_ms"]["median"] - d["ms_per_step"] * 42) < 1e-9
    assert rf["served_by"] in ("infinity_cache", "hbm", "split") and abs(rf["frac_of_measured_copy"] - rf["achieved"] / 6290.0) < 1e-12
    # 4 096 filters: in fp64 the workgroup-cooperative kernel takes every single-rate tick at this size (quarter-tile workgroups); in fp32
    # only the ticks with tag poses, and only on a cadence that has them at least every third tick (cfg 3 / cfg 5: every 14th -> lane kernels)
    want_kernel = {"": "k_predict<float>", "cfg5": "k_predict<float>+per-filter-params", "cfg3mr": "k_predict<float,MR>",
                   "cfg2": "kw_tick<double,step>", "rotors": "k_predict<float,MR>", "hardware": "k_step_mr<float>"}[extra[1] if extra else ""]
    if "f64" in extra:
        want_kernel = want_kernel.replace("float", "double")
    assert rf["kernel"] == want_kernel, rf["kernel"]
    if extra and extra[1] in ("rotors", "hardware"):   # the shipped parameter files: decisions on the device, every synthetic pose in view
        dd = d["device_decisions"]
        assert dd["upd_per_meas"] == (7 if extra[1] == "rotors" else 1) and dd["measurement_step_delay"] == (3 if extra[1] == "rotors" else 15)
        assert dd["n_tags"] == (1 if extra[1] == "rotors" else 13)
        assert dd["measurement_consumed_frac"] == 1.0 and dd["performed_correction_frac"] > 0.9
    assert sum(rf["mixed_kernels"].values()) == 42
    if not extra:
        for sub in ("hbm_resident", "f64_same_batch"):
            assert d[sub]["achieved"] > 0 and d[sub]["nonfinite_filters"] == 0 and d[sub]["served_by"] in ("infinity_cache", "hbm", "split")
        assert d["hbm_resident"]["batch"] == 2097152 and d["hbm_resident"]["served_by"] == "split"
        hr = d["hbm_resident"]
        assert 0 < hr["hbm_share"] < 1 and abs(hr["hbm_frac"] - hr["hbm_share"] * hr["achieved"] / 8000.0) < 1e-12
        assert rf["hbm_frac"] == hr["hbm_frac"] if rf["bound"] != "hbm" else rf["hbm_frac"] == rf["frac"]
        assert d["f64_same_batch"]["dtype"] == "f64" and d["f64_same_batch"]["batch"] == 4096
    if "rmse_vs_truth" in d:
        assert d["rmse_vs_truth"]["wraps_in_timed_regions"] >= 0 and d["rmse_vs_truth"]["ticks"] == d["config"]["ticks_resident_in_hbm"]
    if not extra:
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
        assert d["dtype"] == "f32"
    if extra == ["--workload", "cfg2"] or "f64" in extra:
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
    _check_per_rank(d, [4096, 4096], [0, 4096])


@pytest.mark.gpu
def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torch.distributed.run (the form of the driver's N = 1 command): the script starts its two
    ranks itself as fresh child processes, before its own process touches the GPU, and relays rank 0's one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "42", "--warmup", "14", "--batch-per-gpu", "4096",
           "--kernel-steps", "60"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8192 and d["scaling"] == "weak"
    _check_per_rank(d, [4096, 4096], [0, 4096])


def test_bench_plain_command_launch_line_without_a_gpu():
    """The same self-launch on a machine with no GPU (this container): both ranks must come up under the launcher, fail LOUDLY in
    qle_create (no CPU fallback), and the parent must hand the launcher's non-zero exit code on -- no JSON line, no hang."""
    sys.path.insert(0, ROOT)
    import quadrotor_landing_amd as qla
    import ctypes
    n = ctypes.c_int32(0)
    qla.lib().qle_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present: covered by test_bench_plain_command_starts_its_own_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch-per-gpu", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "no HIP device" in r.stderr or "QLE_ERR_NO_DEVICE" in r.stderr or "no CPU fallback" in r.stderr, r.stderr[-1500:]
    assert r.stderr.count("no CPU fallback") + r.stderr.count("no HIP device") >= 2      # both ranks got as far as the engine


def _check_per_rank(d, filters, offsets):
    """What every rank saw on its own survives next to the max-over-ranks figure: shard sizes, global filter offsets, per-rank
    wall and HIP-event time per step (the reported ms_per_step is their maximum, region by region), CPU placement."""
    pr = d["per_rank"]
    n = len(filters)
    assert pr["filters"] == filters and pr["filter_offset"] == offsets
    assert len(pr["ms_per_step"]) == n and len(pr["hip_event_ms_per_step"]) == n and len(pr["placement"]) == n
    assert all(0 < ev <= ms * 1.001 for ev, ms in zip(pr["hip_event_ms_per_step"], pr["ms_per_step"]))
    assert max(pr["ms_per_step"]) <= d["region_ms"]["max"] / d["steps"] * 1.001
    assert all(isinstance(p, dict) and "pinned" in p for p in pr["placement"])


@pytest.mark.gpu
@pytest.mark.parametrize("workload,glob", [("cfg4", 16384), ("cfg5", 8192)])
def test_bench_strong_scaling_workloads_split_a_fixed_global_batch(workload, glob):
    """BASELINE cfg 4 / cfg 5 as written: a FIXED population split over the ranks (here 2 ranks on the box's one GPU and a small
    population; the defaults are 1 048 576 and 262 144 filters)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "42", "--warmup", "14",
           "--workload", workload, "--global-batch", str(glob), "--kernel-steps", "60"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == glob and d["config"]["batch_per_gpu"] == glob // 2
    _check_per_rank(d, [glob // 2, glob // 2], [0, glob // 2])
    assert abs(d["value"] - glob / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["nonfinite_filters"] == 0
    if workload == "cfg5":
        assert d["rmse_vs_truth"]["filters"] == glob     # the three sums of both devices combined on the host
