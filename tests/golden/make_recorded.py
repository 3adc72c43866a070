"""Generates tests/golden/recorded_cfg1.csv: the "recorded sequence" of BASELINE cfg 1 (the
reference ships no bag, so the recording is a seeded synthetic flight saved as a fixture; seed 0xE4F00001).

10 s of a 100 Hz IMU stream and a 15 Hz tag-pose stream as an event log in ARRIVAL order (format:
quadrotor_landing_amd/replay.py).  Tag detections arrive 30 +- 5 ms after their header stamp (the camera + detector
latency the multirate EKF compensates, EKF.cpp:196-236); IMU samples arrive with 0.5 ms jitter.  Truth: target-frame
position sinusoid above the tag, small body-rate sinusoid, measurement model = inverse of EKF.cpp:431-438 with the
ROTORS extrinsics.  Needs numpy only; run from the repo root:  python tests/golden/make_recorded.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from quadrotor_landing_amd.replay import write_event_log  # noqa: E402

SEED = 0xE4F00001
G = np.array([0.0, 0.0, -9.8])                 # EKF.cpp:81


def qmul(a, b):                                 # Hamilton, x,y,z,w
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def qexp(v):
    n = np.linalg.norm(v)
    if n < 1e-12:
        return np.array([0.5 * v[0], 0.5 * v[1], 0.5 * v[2], 1.0])
    return np.concatenate([v / n * np.sin(n / 2), [np.cos(n / 2)]])


def rot(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def main():
    rng = np.random.default_rng(SEED)
    # ROTORS extrinsics / noise (quadrotor_landing_amd/config/ekf_sim_rotors.yaml)
    import yaml
    with open(os.path.join(HERE, "..", "..", "quadrotor_landing_amd", "config", "ekf_sim_rotors.yaml")) as fh:
        cfg = yaml.safe_load(fh)
    r_v_cv = np.array(cfg["r_v_cv"], float); q_vc = np.array(cfg["q_vc"], float); q_vc /= np.linalg.norm(q_vc)
    C_vc = rot(q_vc)
    Q_a, Q_w = float(cfg["Q_a_diag"][0]), float(cfg["Q_w_diag"][0])
    R_r, R_ang = np.array(cfg["R_r_diag"], float), np.array(cfg["R_ang_diag"], float)
    ab_true = rng.normal(0, 0.05, 3); wb_true = rng.normal(0, 0.005, 3)
    r0 = np.array([0.15, -0.1, 2.0]); A = np.array([0.25, 0.2, 0.3]); om = np.array([0.7, 0.9, 0.5]); ph = rng.uniform(0, 2 * np.pi, 3)
    wa = np.array([0.08, 0.06, 0.1]); wo = np.array([0.8, 1.1, 0.6]); wp = rng.uniform(0, 2 * np.pi, 3)

    dt = 1e-3                                   # truth integration step
    T_end = 10.0
    n = int(round(T_end / dt)) + 1
    ts = np.arange(n) * dt
    q = np.array([0.0, 0.0, 0.0, 1.0])          # q_tv: vehicle attitude in the target frame
    qs = np.empty((n, 4))
    for i, t in enumerate(ts):
        qs[i] = q
        w = wa * np.sin(wo * t + wp)
        q = qmul(q, qexp(w * dt)); q /= np.linalg.norm(q)
    pos = lambda t: r0 + A * np.sin(om * t + ph)
    acc = lambda t: -A * om * om * np.sin(om * t + ph)
    q_at = lambda t: qs[min(int(round(t / dt)), n - 1)]

    events = []
    for k in range(int(T_end * 100)):           # IMU, 100 Hz
        t = 0.005 + k / 100.0
        C = rot(q_at(t))
        a = C.T @ (acc(t) - G) + ab_true + rng.normal(0, np.sqrt(Q_a), 3)
        w = wa * np.sin(wo * t + wp) + wb_true + rng.normal(0, np.sqrt(Q_w), 3)
        events.append(("imu", t + abs(rng.normal(0, 0.0005)), np.concatenate([a, w])))
    for k in range(int(T_end * 15)):            # tag poses, 15 Hz, 30 +- 5 ms late
        stamp = 0.02 + k / 15.0
        q_tv = q_at(stamp); r = pos(stamp)
        q_ct = qmul(qconj(q_vc), qconj(q_tv))   # inverse of EKF.cpp:431-432
        r_c = C_vc.T @ (-rot(q_tv).T @ r - r_v_cv)   # inverse of EKF.cpp:434-438
        r_c = r_c + rng.normal(0, np.sqrt(R_r))
        q_ct = qmul(q_ct, qexp(rng.normal(0, np.sqrt(R_ang) * 0.3)))
        arrival = stamp + 0.030 + rng.uniform(-0.005, 0.005)
        if arrival < T_end:
            events.append(("tag", arrival, stamp, np.concatenate([r_c, q_ct / np.linalg.norm(q_ct)])))
    events.sort(key=lambda e: e[1])
    out = os.path.join(HERE, "recorded_cfg1.csv")
    write_event_log(out, events, header="BASELINE cfg 1 recorded sequence (synthetic, seed 0xE4F00001): tests/golden/make_recorded.py\n"
                    "imu,t,ax,ay,az,wx,wy,wz | tag,t_arrival,stamp,px,py,pz,qx,qy,qz,qw")
    print("wrote", out, len(events), "events")


if __name__ == "__main__":
    main()
