"""Replay of a RECORDED IMU + tag-pose event log through the engine: the ROS side of the reference's node, stubbed.

The reference's node (`quad_state_estimation/src/relative_pose_EKF_node.cpp`) is three callbacks on a single-threaded
spinner (`relative_pose_EKF_main.cpp:17`): `IMUSubCallback` (:144-151) latches the latest IMU sample,
`AprilTagSubCallback` (:153-176) latches the latest tag pose + header stamp, raises `measurement_ready` and seeds the
state on the first detection, and `FilterUpdateCallback` (:178-182) calls `filter_update(now)` every `1/update_freq` s.
An event log holds the two message streams in arrival order:

    imu,<t>,<ax>,<ay>,<az>,<wx>,<wy>,<wz>
    tag,<t_arrival>,<header stamp>,<px>,<py>,<pz>,<qx>,<qy>,<qz>,<qw>

(`#` starts a comment).  `replay()` drives a `BatchedRelativePoseEKF` from it exactly as `ekf_driver --sequence` does;
every filter of the batch sees the same stream (per-filter parameters make it a sweep over one recorded flight).
"""
import numpy as np


def read_event_log(path):
    """-> list of ("imu", t, v[6]) / ("tag", t_arrival, stamp, v[7]) in file (= arrival) order."""
    events = []
    with open(path) as fh:
        for ln, line in enumerate(fh, 1):
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            f = [s.strip() for s in line.split(",")]
            if f[0] == "imu" and len(f) == 8:
                events.append(("imu", float(f[1]), np.array([float(s) for s in f[2:]])))
            elif f[0] == "tag" and len(f) == 10:
                events.append(("tag", float(f[1]), float(f[2]), np.array([float(s) for s in f[3:]])))
            else:
                raise ValueError("%s:%d: expected `imu` + 7 numbers or `tag` + 9 numbers" % (path, ln))
            if len(events) > 1 and events[-1][1] < events[-2][1]:
                raise ValueError("%s:%d: events must be ordered by arrival time" % (path, ln))
    return events


def write_event_log(path, events, header=None):
    with open(path, "w") as fh:
        if header:
            for h in header.splitlines():
                fh.write("# %s\n" % h)
        for e in events:
            if e[0] == "imu":
                fh.write("imu,%.9f,%s\n" % (e[1], ",".join("%.17g" % v for v in e[2])))
            else:
                fh.write("tag,%.9f,%.9f,%s\n" % (e[1], e[2], ",".join("%.17g" % v for v in e[3])))


def tick_times(events, dT_nom):
    """Times at which the filter_update timer fires: every dT_nom from the first event to the last."""
    t_first, t_last = events[0][1], events[-1][1]
    out, k = [], 1
    while t_first + k * dT_nom <= t_last + 0.5 * dT_nom:
        out.append(t_first + k * dT_nom)
        k += 1
    return out


def replay(ekf, events, on_tick=None):
    """Drive `ekf` (gating enabled by this call) from the log.  `on_tick(t, ekf, performed, upds)` runs after every
    active tick.  Returns (ticks fired, ticks active, corrections performed by filter 0)."""
    B = ekf.batch
    ekf.enable_gating(True)
    dT = 1.0 / ekf.params.update_freq
    u = np.zeros((B, 6)); z = np.zeros((B, 7)); z[:, 6] = 1.0; stamp = np.zeros(B)
    state_initialized = measurement_ready = False
    k = n_active = n_corr = 0
    ticks = tick_times(events, dT)
    for t in ticks:
        while k < len(events) and events[k][1] <= t:
            e = events[k]; k += 1
            if e[0] == "imu":                                   # NODE.cpp:144-151
                u[:] = e[2]
            else:                                               # NODE.cpp:153-176
                z[:] = e[3]; stamp[:] = e[2]
                measurement_ready = True
                if not state_initialized:
                    ekf.initialize_state(z, reinit_bias=False)
                    state_initialized = True
        if not state_initialized:                               # EKF.cpp:129-130
            continue
        if measurement_ready:
            ekf.filter_update(u, z, np.ones(B, np.uint8), t_curr=t, apriltag_time=stamp)
        else:
            ekf.filter_update(u, None, None, t_curr=t)
        perf, cons, upds = ekf.tick_flags()
        if cons[0]:
            measurement_ready = False                           # EKF.cpp:152
        n_active += 1; n_corr += int(perf[0])
        if on_tick is not None:
            on_tick(t, ekf, perf, upds)
    return len(ticks), n_active, n_corr
