#!/usr/bin/env python3
"""Benchmark of the batched EKF hot path (BASELINE.json metric).

A "step" is one filter tick over the whole batch: one predict launch, or on every 14th tick one fused predict+update
launch (BASELINE cfg 3: 65 536 fp32 filters, 400 Hz IMU predict interleaved with 30 Hz tag update).  Inputs (IMU and
tag-pose sequences for every tick) are generated on the device beforehand and are resident in HBM when the timed
region starts.

    python bench.py --gpus N --steps K --warmup W [--workload cfg3|cfg2|cfg4|cfg5|cfg3mr]

N > 1 is launched by torch.distributed.run, one rank per GPU.  Filters are sharded across ranks with NO data-path
collective (filters are independent); torch.distributed (gloo) is used only for the barrier and the max-over-ranks
timing.  Rank 0 prints one JSON line.

Workloads (BASELINE.json configs): cfg3 (default, the metric's configuration; weak scaling: 65 536 filters per GPU),
cfg2 (4 096 fp64 filters, update on every tick), cfg4 (1 048 576 fp32 filters in all, split over the ranks: strong
scaling), cfg5 (Monte-Carlo sweep, 262 144 filters in all with device-drawn per-filter parameters, split over the ranks,
per-device RMSE reduction combined on the host; --batch-per-gpu B runs one B-filter shard per rank instead),
cfg3mr (cfg3 with the multirate delayed-measurement replay).

The K-step timed region is bracketed by barrier + device synchronise on both sides; when it is shorter than ~20 ms it
is repeated (`repeats`) and the median region (max over ranks of each) is reported, so that `value` does not carry the
~45 us synchronisation tail of a single sub-millisecond region.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_COPY_GBS = 6290.0       # the same guide's measured float4 streaming-copy rate (79 % of the spec peak)
IC_GATHER_GBS = 8600.0      # the same guide's measured chip-wide rate of a 38 MB table gathered from the Infinity Cache (reads only)

CFG3 = dict(update_freq=400.0, measurement_freq=30.0, limit_measurement_freq=1, direct_orien_method=1,
            est_bias=1, corner_margin_enbl=1, multirate_ekf=0,
            # noise values of relative_pose_EKF_rotors.yaml:13-19
            Q_a=[0.0005] * 3, Q_w=[0.00005] * 3, Q_ab=[5e-5] * 3, Q_wb=[5e-6] * 3,
            R_r=[0.015, 0.015, 0.020], R_ang=[0.0015, 0.0015, 0.04])

WORKLOADS = {
    "cfg3": "cfg3: 65536 filters/GPU, 400 Hz IMU predict + 30 Hz tag-pose update (every 14th tick fused), ROTORS noise set",
    "cfg2": "cfg2: 4096 fp64 filters, predict + update on every tick (100 Hz), ROTORS noise set",
    "cfg4": "cfg4: 1048576 fp32 filters in all, sharded over the ranks, cfg3 schedule (400 Hz predict + 30 Hz update)",
    "cfg5": "cfg5: Monte-Carlo sweep, 262144 filters in all sharded over the ranks, per-filter Q scaled by 10^U(-0.5,0.5) and static biases, "
            "400 Hz predict + 30 Hz update, per-device RMSE reduction",
    "cfg3mr": "cfg3 with multirate_ekf: 30 Hz tag poses arrive 12 ticks late, corrected in the history ring and replayed",
    # the two parameter files the reference ships (quad_state_estimation/config), as filter_update runs them in the field
    "rotors": "relative_pose_EKF_rotors.yaml as shipped: 100 Hz IMU, 15 Hz tag poses (every 7th tick, rate limit on), multirate EKF with dynamic "
              "delay (3-tick camera latency), 1-tag corner gate, decisions on the device",
    "hardware": "relative_pose_EKF_hardware.yaml as shipped: 100 Hz IMU, 100 Hz tag poses (a correction + 15-tick replay on EVERY tick), multirate EKF "
                "with dynamic delay, 13-tag corner gate, static IMU biases, decisions on the device",
}
YAML = {"rotors": "ekf_sim_rotors.yaml", "hardware": "ekf_hardware.yaml"}


def cpu_share():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def pin_to_gpu_numa_node(local_rank, world):
    """Before the first GPU call: keep this rank's launch thread on the CPUs of its GPU's NUMA node (the HIP runtime's helper threads
    inherit the mask).  GPU order = KFD topology order (what HIP enumerates in), optionally filtered by ROCR/HIP_VISIBLE_DEVICES given as
    indices.  Ranks that share a node split its CPUs.  Best effort: any surprise leaves the affinity alone and says so."""
    try:
        top = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for n in sorted(os.listdir(top), key=int):
            props = dict(l.split() for l in open(f"{top}/{n}/properties") if len(l.split()) == 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props["drm_render_minor"]))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
            v = os.environ.get(var)
            if v and all(t.strip().isdigit() for t in v.split(",")):
                gpus = [gpus[int(t)] for t in v.split(",")]
        if not gpus:
            return {"pinned": False, "why": "no KFD GPU node visible"}
        ndev = len(gpus)
        nodes = [int(open(f"/sys/class/drm/renderD{m}/device/numa_node").read()) for m in gpus]
        node = nodes[local_rank % ndev]
        if node < 0:
            return {"pinned": False, "why": "GPU reports no NUMA node"}
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus = sorted(cpus & os.sched_getaffinity(0))
        sharers = [r for r in range(world) if nodes[r % ndev] == node]     # local ranks on the same node, in rank order
        k = sharers.index(local_rank) if local_rank in sharers else 0
        mine = cpus[k::max(len(sharers), 1)] if len(cpus) >= len(sharers) else cpus
        if not mine:
            return {"pinned": False, "why": f"no allowed CPU on NUMA node {node}"}
        os.sched_setaffinity(0, mine)
        return {"pinned": True, "numa_node": node, "cpus": len(mine), "first_cpu": mine[0]}
    except Exception as e:   # noqa: BLE001 -- never let placement break the run
        return {"pinned": False, "why": f"{type(e).__name__}: {e}"}


def cpu_baseline(seq, x0, P0, n_filters, n_ticks):
    """Oracle (reference-shaped dense fp64 C restatement) timed on this host's cores
    on a bounded sample of the same workload.  Reported baseline, not the target."""
    import oracle
    po = oracle.make_params(**CFG3)
    U = np.empty((n_ticks, n_filters, 6)); Z = np.zeros((n_ticks, n_filters, 7)); M = np.zeros((n_ticks, n_filters), np.uint8)
    for t in range(n_ticks):
        u, z, m = seq.download_tick(t)
        U[t], Z[t], M[t] = u[:n_filters], z[:n_filters], m[:n_filters]
    nthr = max(1, min(oracle.max_threads(), cpu_share()))
    t0 = time.perf_counter()
    oracle.run_batch(po, x0[:n_filters], P0[:n_filters], U, Z, M, n_threads=nthr)
    dt_all = time.perf_counter() - t0
    n1 = max(n_filters // 64, 64)
    t0 = time.perf_counter()
    oracle.run_batch(po, x0[:n1], P0[:n1], U[:, :n1], Z[:, :n1], M[:, :n1], n_threads=1)
    dt_one = time.perf_counter() - t0
    # second, stronger baseline: the engine's own block-structured arithmetic compiled for the host (fp32, all threads)
    oracle.structured_run_batch(po, x0[:64], P0[:64], U[:2, :64], Z[:2, :64], M[:2, :64], dtype="f32", levels=True, n_threads=1)   # loads (and, if
    t0 = time.perf_counter()                                                           # stale, rebuilds) the checker library outside the timed call
    oracle.structured_run_batch(po, x0[:n_filters], P0[:n_filters], U, Z, M, dtype="f32", levels=True, n_threads=nthr)
    dt_struct = time.perf_counter() - t0
    return {"value": n_filters * n_ticks / dt_all, "unit": "EKF ticks/s", "cores": nthr, "kind": "port",
            "structured": {"value": n_filters * n_ticks / dt_struct, "unit": "EKF ticks/s", "cores": nthr, "dtype": "f32",
                           "note": "structure-exploiting CPU variant: the engine's per-filter arithmetic (ekf_device.hpp) compiled for the host by ROCm's clang++ -O3 "
                                   "(oracle/ekf_structured_cpu.cpp), same sample, OpenMP static split"},
            "sample": f"{n_filters} filters x {n_ticks} ticks of the cfg3 sequence (fp64, dense reference-shaped arithmetic, OpenMP static split)",
            "single_thread_value": n1 * n_ticks / dt_one, "host_cpus": os.cpu_count()}


def head_sha():
    if os.environ.get("QLE_HEAD_SHA"):     # the GPU box's snapshot has no .git: the profiling script exports the head it was made from
        return os.environ["QLE_HEAD_SHA"]
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
    except Exception:
        return None


def kernel_name(pol, dtype, step, mr=False, pfp=False):
    """The kernel a tick of this kind is launched on (ekf_capi.hip: launch_predict / launch_step)."""
    t = "float" if dtype == "f32" else "double"
    if mr:
        return f"k_step_mr<{t}>" if step else f"k_predict<{t},MR>"
    if pol["coop_ticks"] & (1 if step else 2):
        return f"kw_tick<{t},{'step' if step else 'predict'}>"
    return (f"k_step<{t}>" if step else f"k_predict<{t}>") + ("+per-filter-params" if pfp else "")


def time_ticks(ekf, seq, t0, n):
    """n back-to-back ticks of `seq` between HIP events on the launch stream -> microseconds per tick."""
    ekf.timer_begin()
    ekf.run(seq, t0, n)
    return ekf.timer_end() / n * 1e3


def sub_record(qla, cfg, B, dtype, upd, n_pred, n_mixed, seed, device):
    """A second population timed in the same run (rank 0, N = 1): predict-only ticks and the cfg3 mix."""
    T = upd * 2
    thm = np.zeros(T, np.uint8); thm[upd - 1::upd] = 1
    ekf = qla.BatchedRelativePoseEKF(B, dtype, device=device, **cfg)
    pol = ekf.policy()
    seq = ekf.make_inputs(T, thm)
    ekf.synth_generate(seq, seed=seed)
    ekf.run(seq, 0, T); ekf.synchronize()
    us_mixed = time_ticks(ekf, seq, 0, n_mixed)
    n_upd = sum(int(thm[k % T]) for k in range(n_mixed))
    pseq = ekf.make_inputs(T, None)
    ekf.synth_generate(pseq, seed=seed + 1)
    ekf.run(pseq, 0, T); ekf.synchronize()
    us_pred = time_ticks(ekf, pseq, 0, n_pred)
    bad = ekf.count_nonfinite()
    b0, b1 = ekf.algorithmic_bytes(0), ekf.algorithmic_bytes(1)
    us_step = (us_mixed * n_mixed - us_pred * (n_mixed - n_upd)) / max(n_upd, 1)
    gbs = b0 / us_pred / 1e3
    out = {"batch": B, "dtype": dtype, "state_MiB": pol["state_bytes"] / 2 ** 20, "served_by": pol["served_by"],
           "state_policy": pol["state_policy"], "split_k64": pol.get("split_k64", 0), "kernel": kernel_name(pol, dtype, False),
           "predict_tick_us": us_pred, "achieved": gbs, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "frac_of_measured_copy": gbs / HBM_COPY_GBS,
           "algorithmic_bytes_per_launch": b0, "launches": n_pred,
           "mixed_ticks_per_s": B / (us_mixed * 1e-6), "mixed_tick_us": us_mixed, "mixed_ticks": n_mixed,
           "fused_tick_us_derived": us_step, "fused_tick_achieved": b1 / us_step / 1e3 if us_step > 0 else None,
           "fused_kernel": kernel_name(pol, dtype, True), "nonfinite_filters": bad}
    ekf.close()
    return out


def self_launch(n, stdout_fd):
    """Run this script as n ranks (one per GPU) under torch.distributed.run and return the launcher's exit code."""
    import socket
    with socket.socket() as s:      # a free rendezvous port on the loopback (the container hostname may not resolve)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=stdout_fd, stderr=2, env=env, cwd=os.getcwd())
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()       # the launcher we started (exact PID); it takes its ranks down with it
        return child.wait()


def main():
    # stdout carries exactly ONE JSON line (rank 0).  Libraries print there too (gloo announces its peers on
    # stdout), so fd 1 is pointed at stderr for the whole run and the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="filters per rank (weak scaling); default: the workload's own size")
    ap.add_argument("--global-batch", type=int, default=0, help="filters in all, split over the ranks (strong scaling); default for cfg4 / cfg5")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--seq-ticks", type=int, default=0, help="ticks of generated input kept in HBM (0 = steps+warmup, capped)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the hbm_resident / f64_same_batch sub-records and the on-chip-resident variant")
    ap.add_argument("--kernel-steps", "--predict-only-steps", type=int, default=2000, dest="kernel_steps",
                    help="back-to-back launches of the dominant kernel for the roofline record")
    ap.add_argument("--repeats", type=int, default=0, help="repetitions of the K-step timed region (0 = as many as make ~20 ms, at most 25)")
    ap.add_argument("--workload", default="cfg3", choices=list(WORKLOADS))
    args = ap.parse_args()
    cfg = dict(CFG3)
    upd = 14  # ceil(400/30), relative_pose_EKF.cpp:91
    perturb = False
    scaling = "weak"
    per_gpu_default, global_default = 65536, 0
    if args.workload == "cfg2":
        cfg.update(update_freq=100.0, measurement_freq=100.0, limit_measurement_freq=0)
        upd = 1
        args.dtype = "f64"
        per_gpu_default = 4096
        if args.steps == 4000:
            args.steps, args.warmup = 1000, 20
    elif args.workload == "cfg4":
        global_default = 1048576
    elif args.workload == "cfg5":
        perturb = True
        global_default = 262144
    mr_step = 0
    params = None
    if args.workload in YAML:
        # everything from the shipped file (relative_pose_EKF_{rotors,hardware}.yaml re-laid-out under quadrotor_landing_amd/config):
        # cadence, delays, noise, static biases, camera and tag bundle, flags
        from quadrotor_landing_amd import params as qparams
        params = qparams.load_yaml(os.path.join(ROOT, "quadrotor_landing_amd", "config", YAML[args.workload]))
        der = qparams.derive(params)
        cfg = {}
        upd = int(der.upd_per_meas)                                   # ceil(update_freq / measurement_freq), EKF.cpp:91
        mr_step = int(der.measurement_step_delay)                     # int(measurement_delay / dT + 0.5), EKF.cpp:93
        if args.steps == 4000:
            args.steps, args.warmup = 1400, 70
    if args.workload == "cfg3mr":
        # delays of relative_pose_EKF_rotors.yaml:5-7 at 400 Hz: step delay int(0.030/0.0025 + 0.5) = 12 ticks
        cfg.update(multirate_ekf=1, dynamic_meas_delay=1, measurement_delay=0.030, measurement_delay_max=0.200,
                   dyn_measurement_delay_offset=0.005)
        mr_step = 12

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes under torch.distributed.run, BEFORE
            # this process makes any GPU call (it never does: it only relays).  The children's rank 0 writes the one JSON line straight to
            # our stdout; we exit with the launcher's code (non-zero if any rank failed).  Nothing is exec-replaced.
            sys.exit(self_launch(args.gpus, real_stdout))
        args.gpus = world
    dist = None
    placement = pin_to_gpu_numa_node(local_rank, world) if world > 1 else None   # before the first GPU call of this process
    if world > 1:
        import torch.distributed as dist  # rendezvous + barrier + timing reduction only (gloo, CPU tensors)
        import torch
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    import ctypes
    import quadrotor_landing_amd as qla
    from quadrotor_landing_amd.sharding import shard_range

    ndev = ctypes.c_int32(0)
    qla.lib().qle_device_count(ctypes.byref(ndev))
    # one rank per GPU; a rehearsal with more ranks than devices (1-GPU box) wraps around and says so
    device = local_rank % max(ndev.value, 1)
    oversubscribed = world > max(ndev.value, 1)

    # which filters this rank owns
    if args.batch_per_gpu:
        B, offset, global_batch = args.batch_per_gpu, rank * args.batch_per_gpu, world * args.batch_per_gpu
    elif args.global_batch or global_default:
        global_batch = args.global_batch or global_default
        lo, hi = shard_range(global_batch, rank, world)
        B, offset = hi - lo, lo
        scaling = "strong"
    else:
        B, offset, global_batch = per_gpu_default, rank * per_gpu_default, world * per_gpu_default
    K, W = args.steps, args.warmup
    T = args.seq_ticks or min(K + W, max(140, int(4200 * 65536 / max(B, 1))))   # about 6 GB of fp32 inputs at most
    T = max(upd, (T // upd) * upd)  # whole measurement periods so the wrapped schedule stays periodic
    thm = np.zeros(T, np.uint8); thm[upd - 1::upd] = 1

    ekf = qla.BatchedRelativePoseEKF(B, args.dtype, device=device, params=params, **cfg)
    if params is not None:
        ekf.enable_gating(True)   # rate limit + corner gate per filter on the device (EKF.cpp:147-186), as the node runs them
    pol = ekf.policy()
    seq = ekf.make_inputs(T, thm)
    seed = {"cfg2": 0xE4F00002, "cfg3": 0xE4F00003, "cfg4": 0xE4F00003, "cfg5": 0xE4F00005, "cfg3mr": 0xE4F00003,
            "rotors": 0xE4F00006, "hardware": 0xE4F00007}[args.workload]
    if mr_step:
        ekf.set_uniform_measurement_age(mr_step / ekf.params.update_freq - ekf.params.dyn_measurement_delay_offset)
    # the shipped-file workloads fly a landing approach (tag bundle inside the image, so that the corner gate passes as it does while the
    # node has detections); the BASELINE configs keep the free flight they were specified with
    synth_kw = dict(filter_offset=offset, perturb_filter_params=perturb, meas_delay_ticks=mr_step, view_scale=0.2 if params is not None else 1.0)
    ekf.synth_generate(seq, seed=seed, **synth_kw)
    x0 = P0 = None
    if rank == 0 and not args.no_cpu_baseline and world == 1 and args.workload == "cfg3":
        x0, P0 = ekf.get_state()

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_region(t_start):
        """EXACTLY K steps between barrier + device synchronise on both sides -> (wall seconds, HIP-event ms) of this rank."""
        barrier()
        ekf.synchronize()
        t0 = time.perf_counter()
        ekf.timer_begin()
        ekf.run(seq, t_start, K)
        ev = ekf.timer_end()  # HIP events on the launch stream; synchronises the stream (the device is idle from here)
        w = time.perf_counter() - t0
        barrier()
        return w, ev

    ekf.run(seq, 0, W)
    ekf.synchronize()
    pos = W
    regions = [timed_region(pos)]
    pos += K
    R = args.repeats or int(min(25, max(1, np.ceil(0.020 / max(regions[0][0], 1e-6)))))
    if dist is not None:   # every rank must run the same number of regions
        rt = torch.tensor([R], dtype=torch.int64)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        R = int(rt[0])
    for _ in range(R - 1):
        regions.append(timed_region(pos))
        pos += K
    walls = np.array([r[0] for r in regions]); evs = np.array([r[1] for r in regions])
    per_rank = None
    if dist is not None:
        mine = torch.tensor([float(np.median(walls)) / K * 1e3, float(np.median(evs)) / K, float(B), float(offset), float(device)], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)        # what each rank saw on its own, before the max: a slow rank or device shows here
        per_rank = {"ms_per_step": [float(a[0]) for a in allr], "hip_event_ms_per_step": [float(a[1]) for a in allr],
                    "filters": [int(a[2]) for a in allr], "filter_offset": [int(a[3]) for a in allr], "device": [int(a[4]) for a in allr]}
        pl = [None] * world
        dist.all_gather_object(pl, placement)
        per_rank["placement"] = pl
        tt = torch.tensor(np.stack([walls, evs]), dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)      # MAX over ranks, region by region
        walls, evs = tt[0].numpy(), tt[1].numpy()
    wall = float(np.median(walls)); ev_ms = float(np.median(evs))
    n_upd = sum(int(thm[(W + k) % T]) for k in range(K))
    bytes_mixed = (K - n_upd) * ekf.algorithmic_bytes(0) + n_upd * ekf.algorithmic_bytes(1)
    wsz = 4 if args.dtype == "f32" else 8
    if mr_step:
        # multirate correction per filter and measurement cycle (k_step_mr, checkpointed history): read u6 + z8 + the record the replay
        # starts from (136: the extra checkpoint a predict tick wrote at the expected entry of this tag pose, so nothing is replayed in
        # front of the correction) + the stored IMU samples of the replayed ticks (8 words each); write the anchor (136), the state
        # (136), its IMU sample (8), the grid checkpoints passed on the way (136 x step / k, k = 32) -- and that extra checkpoint (136)
        # (a correction on every tick, upd == 1: the chain starts one tick before the entry, from the anchor, and no extra checkpoint exists)
        words = (6 + 8 + 136 + 8 * (mr_step + (upd == 1))) + (136 + 136 + 8 + 136 * mr_step // 32) + (136 if upd > 1 else 0)
        bytes_mixed = (K - n_upd) * ekf.algorithmic_bytes(0) + n_upd * words * wsz * B
    bad = ekf.count_nonfinite()
    decisions = None
    if params is not None:   # what the device-side decisions (EKF.cpp:147-186) did on the last timed tick that carried tag poses
        pc, co, up = ekf.tick_flags()
        decisions = {"performed_correction_frac": float(pc.mean()), "measurement_consumed_frac": float(co.mean()),
                     "upds_since_correction_max": int(up.max()), "upd_per_meas": upd, "measurement_step_delay": mr_step,
                     "n_tags": int(ekf.params.n_tags)}
    # per-device error sums vs the generator's truth (cfg 5 reduction).  The truth is the pose after ONE pass over the resident
    # sequence and the trajectory is not periodic in T, so the timed regions (which wrap `wraps_in_timed_regions` times) say nothing
    # about tracking: the filters are re-seeded and run through exactly one untimed pass 0..T first.
    wraps = (pos - 1) // T
    ekf.synth_generate(seq, seed=seed, **synth_kw)
    ekf.run(seq, 0, T)
    rm = ekf.synth_rmse(seq)
    if dist is not None:
        rt = torch.tensor(list(rm), dtype=torch.float64)
        dist.all_reduce(rt, op=dist.ReduceOp.SUM)   # 3 scalars per device, combined on the host
        rm = rt.numpy()

    # separately reported: the same K ticks in ONE launch with x and P held in
    # registers (qle_run_resident).  Not the streamed per-tick unit of work; never `value`, never the roofline.
    resident = None
    if args.workload in ("cfg3", "cfg5") and dist is None and not args.no_extras:
        ekf.run_resident(seq, 0, upd); ekf.synchronize()
        t1 = time.perf_counter()
        ekf.run_resident(seq, W, K)
        ekf.synchronize()
        dt_res = time.perf_counter() - t1
        resident = {"ticks_per_s": B * K / dt_res, "ms_total": dt_res * 1e3, "launches": 1,
                    "note": "on-chip resident: one launch, state in registers for all K ticks; HBM traffic = inputs only"}

    # The dominant kernel on its own: the tick kind that carries most of the workload's time (the predict-only tick: 13 of every
    # 14 launches; cfg2: the fused tick, its only kind), back to back over a long generated sequence with fresh inputs every
    # tick; HIP events on the stream the kernel is launched on.  Re-seeds the filters, so it runs after everything else.
    Kp = args.kernel_steps
    dom_step = upd == 1
    Tp = min(Kp, max(140, int(2000 * 65536 / max(B, 1))))
    pseq = ekf.make_inputs(Tp, np.ones(Tp, np.uint8) if dom_step else None)
    ekf.synth_generate(pseq, seed=seed + 1, **synth_kw)
    ekf.run(pseq, 0, 20)
    ekf.synchronize()
    p_us = time_ticks(ekf, pseq, 20, Kp)
    p_bytes = ekf.algorithmic_bytes(1 if dom_step else 0)
    p_gbs = p_bytes / p_us / 1e3
    traffic = traffic_src = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):   # offline PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE are separate runs), keyed by configuration
        try:
            tj = json.load(open(tpath))
            key = f"{args.workload if args.workload != 'cfg4' else 'cfg3'}:{B}:{args.dtype}:{'step' if dom_step else 'predict'}"
            if key in tj.get("per_launch", {}):
                traffic = tj["per_launch"][key].get("fabric_bytes", tj["per_launch"][key].get("hbm_bytes"))
                traffic_src = {"file": "profiles/traffic.json", "measured_at_sha": tj.get("sha"), "kernel": tj["per_launch"][key].get("kernel"),
                               "note": "L2<->fabric bytes: 2 x FETCH_SIZE + WRITE_SIZE per launch, separate rocprofv3 --pmc passes (MI355X_MICROARCH.md HBM "
                                       "section).  Infinity-Cache hits are included (no counter of this part separates them: the TCC_EA0_*_DRAM "
                                       "counters count the requests to the local memory's address space, in front of the Infinity Cache -- "
                                       "profiles/r04_pmc_dram_*.md), so this is HBM traffic only where the state cannot stay on die; it shows "
                                       "that nothing is fetched or written twice"}
        except Exception:
            traffic = traffic_src = None

    out = {
        "metric": "EKF predict+update steps/sec at batch=65536; achieved HBM GB/s vs roofline",
        "value": global_batch * K / wall,
        "unit": "EKF ticks/s",
        "n_gpus": world, "steps": K, "warmup": W, "repeats": R,
        "ms_per_step": wall / K * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload],
                   "batch_per_gpu": B, "global_batch": global_batch, "ticks_resident_in_hbm": T,
                   "parallelism": f"filters sharded x{world}, no collectives"
                                  + (f" (REHEARSAL: {world} ranks on {ndev.value} device(s))" if oversubscribed else "")},
        # bound = what serves the dominant kernel's bytes at THIS batch (from the handle's policy and state size): "hbm",
        # "infinity_cache" (the state never leaves the 256 MiB on-die cache: achieved / peak is then an on-die rate held against the
        # HBM peak, NOT an HBM fraction) or "split".  hbm_frac (filled in below from the hbm_resident sub-record) is the HBM-served figure.
        "roofline": {"bound": pol["served_by"], "bound_class": "memory", "kernel": kernel_name(pol, args.dtype, dom_step, mr=bool(mr_step), pfp=perturb),
                     "achieved": p_gbs, "peak": HBM_PEAK_GBS, "peak_is": "HBM3E spec peak (MI355X_MICROARCH.md)", "unit": "GB/s", "frac": p_gbs / HBM_PEAK_GBS,
                     "frac_is": "algorithmic bytes / kernel time / HBM spec peak" + ("" if pol["served_by"] == "hbm" else
                                "; bytes served " + ("by the Infinity Cache" if pol["served_by"] == "infinity_cache" else "partly by the Infinity Cache") + ": not an HBM fraction, see hbm_frac"),
                     "hbm_frac": p_gbs / HBM_PEAK_GBS if pol["served_by"] == "hbm" else None,
                     "frac_of_measured_copy": p_gbs / HBM_COPY_GBS, "measured_copy_GBs": HBM_COPY_GBS,
                     # an on-die rate has an on-die yardstick too: the guide's measured Infinity-Cache gather rate (a read-only pattern;
                     # this kernel reads AND writes its bytes, so this is context, not a bound)
                     "frac_of_infinity_cache_gather": (p_gbs / IC_GATHER_GBS) if pol["served_by"] == "infinity_cache" else None,
                     "infinity_cache_gather_GBs": IC_GATHER_GBS if pol["served_by"] == "infinity_cache" else None,
                     "served_by": pol["served_by"],
                     "served_by_note": {"infinity_cache": "the state (%.0f MiB) stays in the 256 MiB Infinity Cache from tick to tick: this rate is an on-die "
                                                          "rate and may exceed the HBM copy rate; see hbm_resident for the HBM-served figure" % (pol["ring_bytes"] / 2 ** 20),
                                        "split": "a fixed part of the state stays in the Infinity Cache, the rest streams from HBM",
                                        "hbm": "the state streams from HBM every tick"}[pol["served_by"]],
                     "state_policy": pol["state_policy"], "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_over_algorithmic": (traffic / p_bytes) if traffic else None,
                     "algorithmic_bytes_per_launch": p_bytes, "avg_launch_us": p_us, "launches": Kp,
                     "mixed_achieved": bytes_mixed / (ev_ms * 1e-3) / 1e9,
                     "mixed_kernels": {kernel_name(pol, args.dtype, False, mr=bool(mr_step), pfp=perturb): K - n_upd,
                                       kernel_name(pol, args.dtype, True, mr=bool(mr_step), pfp=perturb): n_upd},
                     "mixed_note": "all K timed launches of the median region, HIP-event time incl. inter-launch gaps"},
        "region_ms": {"median": wall * 1e3, "min": float(walls.min()) * 1e3, "max": float(walls.max()) * 1e3},
        "hip_event_ms_per_step": ev_ms / K,
        "nonfinite_filters": bad,
        "head": head_sha(),
    }
    if decisions is not None:
        out["device_decisions"] = decisions
    if rm is not None:
        from quadrotor_landing_amd.sharding import combine_rmse
        r_r, r_th, n = combine_rmse([rm])
        out["rmse_vs_truth"] = {"position_m": r_r, "attitude_rad": r_th, "filters": n, "ticks": T,
                                "note": "filters re-seeded, one untimed pass over the resident sequence, truth at its end",
                                "wraps_in_timed_regions": int(wraps)}
    if per_rank is not None:
        out["per_rank"] = per_rank
    if resident is not None:
        out["on_chip_resident"] = resident
    if x0 is not None:
        out["cpu_baseline"] = cpu_baseline(seq, x0, P0, min(B, 65536), min(140, T))
    ekf.close()
    if rank == 0 and world == 1 and args.workload == "cfg3" and not args.no_extras:
        # the same kernels where the state cannot stay on die, and the reference's own arithmetic type at the headline batch
        hr = out["hbm_resident"] = sub_record(qla, cfg, 2097152, "f32", upd, 200, 280, 0xE4F00013, device)
        out["f64_same_batch"] = sub_record(qla, cfg, B, "f64", upd, 400, 560, 0xE4F00023, device)
        # the HBM-served share of the same kernel where the state cannot stay on die: the split policy keeps split_k64/64 of the state
        # in the Infinity Cache, so only the rest of the algorithmic bytes comes from / goes to HBM in the measured time
        hr["hbm_share"] = 1.0 - hr["split_k64"] / 64.0
        hr["hbm_share_is"] = "MODELLED, not counted: the split policy keeps split_k64/64 of the state cached; assumes that share is served entirely on die"
        hr["hbm_achieved"] = hr["hbm_share"] * hr["achieved"]
        hr["hbm_frac"] = hr["hbm_achieved"] / HBM_PEAK_GBS
        hr["hbm_frac_of_measured_copy"] = hr["hbm_achieved"] / HBM_COPY_GBS
        if out["roofline"]["hbm_frac"] is None:
            out["roofline"]["hbm_frac"] = hr["hbm_frac"]
            out["roofline"]["hbm_frac_is"] = "modelled (hbm_share of the algorithmic bytes / time / 8 TB/s): an estimate, no counter separates Infinity-Cache hits from HBM"
            out["roofline"]["hbm_frac_source"] = ("hbm_resident: k_predict<float> on %d filters (%.0f MiB of state), (1 - split_k64/64) x algorithmic bytes / "
                                                  "HIP-event kernel time / 8 TB/s; rocprofv3 durations: profiles/r04_kernel_stats_b2097152.md" % (hr["batch"], hr["state_MiB"]))
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
