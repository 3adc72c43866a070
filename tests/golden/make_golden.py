#!/usr/bin/env python3
"""Generate golden vectors from the reference's own Python twin EKF.

Run in the BUILD container only (needs /root/reference, which never travels):

    python -B tests/golden/make_golden.py

What it does (SURVEY.md section 8(c) recipe): imports the reference's
`quad_state_estimation/test/rel_pose_EKF_test_class.py` *unmodified* from
/root/reference with the ROS-only modules mocked in `sys.modules`, a textbook
`tf.transformations` stand-in for the four pure-math helpers the two step
methods call, and `numpy.math = math` (NumPy 2 removed the alias).  Then it
drives `RelativePoseEKF.prediction_step` / `correction_step` on seeded inputs
and stores inputs + outputs as data-only fixtures (`*.npz`, `kat.json`).

What the fixtures pin: predict == C++ prediction_step
with zero static biases; update == C++ correction_step with
direct_orien_method=true.  The C++-only branches (conventional method, static
bias, normalisation inside quaternion_exp, flip of delta_q) are NOT pinned by
these vectors; every update case below keeps delta_q.w > -0.75 so the flip
cannot fire and the two implementations agree to rounding.

The reference sources are never copied or modified; only numbers are saved.
"""
import json
import math
import os
import sys
import types
from unittest import mock

import numpy as np

REF_TEST_DIR = "/root/reference/quad_state_estimation/test"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------
# shims
# --------------------------------------------------------------------------
def tf_standins():
    """The five `tf.transformations` functions the reference's two step methods call, as textbook formulas in tf's conventions
    (quaternions x, y, z, w; Hamilton product; active rotations).  Registers nothing.  tests/test_oracle.py checks them against
    scipy.spatial.transform.Rotation, an implementation this repository did not write."""
    tft = types.ModuleType("tf.transformations")

    def quaternion_matrix(q):
        x, y, z, w = [float(v) for v in q]
        n = x * x + y * y + z * z + w * w
        s = 2.0 / n
        M = np.eye(4)
        M[0, 0] = 1 - s * (y * y + z * z)
        M[0, 1] = s * (x * y - w * z)
        M[0, 2] = s * (x * z + w * y)
        M[1, 0] = s * (x * y + w * z)
        M[1, 1] = 1 - s * (x * x + z * z)
        M[1, 2] = s * (y * z - w * x)
        M[2, 0] = s * (x * z - w * y)
        M[2, 1] = s * (y * z + w * x)
        M[2, 2] = 1 - s * (x * x + y * y)
        return M

    def quaternion_multiply(q1, q0):
        x1, y1, z1, w1 = q1
        x0, y0, z0, w0 = q0
        return np.array([
            x1 * w0 + y1 * z0 - z1 * y0 + w1 * x0,
            -x1 * z0 + y1 * w0 + z1 * x0 + w1 * y0,
            x1 * y0 - y1 * x0 + z1 * w0 + w1 * z0,
            -x1 * x0 - y1 * y0 - z1 * z0 + w1 * w0], dtype=np.float64)

    def quaternion_conjugate(q):
        return np.array([-q[0], -q[1], -q[2], q[3]], dtype=np.float64)

    def quaternion_about_axis(angle, axis):
        a = np.asarray(axis, dtype=np.float64).flatten()
        a = a / np.linalg.norm(a)
        return np.append(a * math.sin(angle / 2), math.cos(angle / 2))

    def rotation_matrix(angle, direction):
        d = np.asarray(direction, dtype=np.float64).flatten()[:3]
        d = d / np.linalg.norm(d)
        s, c = math.sin(angle), math.cos(angle)
        R = np.diag([c, c, c]) + np.outer(d, d) * (1.0 - c)
        d = d * s
        R += np.array([[0, -d[2], d[1]], [d[2], 0, -d[0]], [-d[1], d[0], 0]])
        M = np.eye(4)
        M[:3, :3] = R
        return M

    tft.quaternion_matrix = quaternion_matrix
    tft.quaternion_multiply = quaternion_multiply
    tft.quaternion_conjugate = quaternion_conjugate
    tft.quaternion_about_axis = quaternion_about_axis
    tft.rotation_matrix = rotation_matrix
    return tft


def _install_shims():
    np.math = math  # PYQH.py uses np.math.{cos,sin,atan2}
    for name in ("rospy", "geometry_msgs", "geometry_msgs.msg", "sensor_msgs",
                 "sensor_msgs.msg", "apriltag_ros", "apriltag_ros.msg",
                 "std_msgs", "std_msgs.msg"):
        sys.modules[name] = mock.MagicMock()
    tf_mod = mock.MagicMock()
    tft = tf_standins()
    tf_mod.transformations = tft
    sys.modules["tf"] = tf_mod
    sys.modules["tf.transformations"] = tft
    return tft


def _load_reference():
    tft = _install_shims()
    sys.path.insert(0, REF_TEST_DIR)
    sys.dont_write_bytecode = True
    import quaternion_helper as pyqh  # noqa: E402  (the reference's own)
    import rel_pose_EKF_test_class as pyekf  # noqa: E402
    # NumPy >= 1.24 rejects the ragged (3,1) argument at PYEKF.py:455
    _orig = pyekf.skew_symm
    pyekf.skew_symm = lambda v: _orig(np.asarray(v, dtype=np.float64).flatten())
    return pyqh, pyekf, tft


# --------------------------------------------------------------------------
# helpers for building inputs (ours, not the reference's)
# --------------------------------------------------------------------------
def rand_unit_quat(rng, max_angle=None):
    if max_angle is None:
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if q[3] < 0:
            q = -q
        return q
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    ang = rng.uniform(0, max_angle)
    return np.append(ax * math.sin(ang / 2), math.cos(ang / 2))


def rand_spd(rng, n, scale):
    A = rng.normal(size=(n, n)) * 0.3
    P = A @ A.T + np.diag(rng.uniform(0.5, 1.5, size=n))
    d = np.sqrt(np.asarray(scale, dtype=np.float64))
    P = P * np.outer(d, d)
    return 0.5 * (P + P.T)


PARAM_SETS = {
    # name: (update_freq, est_bias, Q diag 12, R diag 6, r_v_cv, q_vc xyzw)
    "pydefault": dict(update_freq=100.0, est_bias=True,
                      Q=[0.005] * 3 + [0.0005] * 3 + [5e-5] * 3 + [5e-6] * 3,
                      R=[0.005, 0.005, 0.015, 0.0025, 0.0025, 0.025],
                      r_v_cv=[0, 0, -0.073], q_vc=[0.70711, -0.70711, 0, 0]),
    "rotors400": dict(update_freq=400.0, est_bias=True,
                      Q=[0.0005] * 3 + [0.00005] * 3 + [5e-5] * 3 + [5e-6] * 3,
                      R=[0.015, 0.015, 0.020, 0.0015, 0.0015, 0.04],
                      r_v_cv=[0, 0, -0.073], q_vc=[0.70711, -0.70711, 0, 0]),
    "hardware": dict(update_freq=100.0, est_bias=True,
                     Q=[0.00025] * 3 + [0.00045] * 3 + [7e-6] * 3 + [4.4e-5] * 3,
                     R=[0.0015, 0.0015, 0.006, 0.0015, 0.0015, 0.04],
                     r_v_cv=[0.06036412, -0.00145196, -0.04439579],
                     q_vc=[-0.7035177, 0.7106742, 0.0014521, -0.0017207]),
    "nobias": dict(update_freq=100.0, est_bias=False,
                   Q=[0.005] * 3 + [0.0005] * 3,
                   R=[0.005, 0.005, 0.015, 0.0025, 0.0025, 0.025],
                   r_v_cv=[0, 0, -0.073], q_vc=[0.70711, -0.70711, 0, 0]),
}


def make_filter(pyekf, pyqh, tft, ps):
    """Instantiate the reference twin and overwrite its public parameters."""
    p = PARAM_SETS[ps]
    f = pyekf.RelativePoseEKF(p["update_freq"], 10.0)
    f.est_bias = p["est_bias"]
    f.num_states = 15 if p["est_bias"] else 9
    f.Q = np.diag(np.asarray(p["Q"], dtype=np.float64))
    f.R = np.diag(np.asarray(p["R"], dtype=np.float64))
    f.r_v_cv = np.asarray(p["r_v_cv"], dtype=np.float64).reshape(3, 1)
    f.q_vc = pyqh.quaternion_norm(np.asarray(p["q_vc"], dtype=np.float64))
    f.C_vc = tft.quaternion_matrix(f.q_vc)[0:3, 0:3]
    ci = [f.r_cov_init] * 3 + [f.v_cov_init] * 3 + [f.ang_cov_init] * 3 + [f.ab_cov_init] * 3 + [f.wb_cov_init] * 3
    f.cov_init = np.diag(np.asarray(ci[:f.num_states], dtype=np.float64))
    return f


def rand_state(rng, est_bias=True):
    x = np.zeros(16)
    x[0:3] = rng.uniform([-1, -1, 1], [1, 1, 4])
    x[3:6] = rng.normal(size=3) * 0.5
    x[6:10] = rand_unit_quat(rng)
    if est_bias:
        x[10:13] = rng.normal(size=3) * 0.1
        x[13:16] = rng.normal(size=3) * 0.01
    return x


def meas_from_pose(f, tft, r_t, q_tv, rng=None, sig_r=0.0, sig_a=0.0):
    """Invert the observation model (EKF.cpp:431-438): tag pose in camera frame."""
    q_vc = f.q_vc
    q_ct = tft.quaternion_multiply(tft.quaternion_conjugate(q_vc), tft.quaternion_conjugate(q_tv))
    C_tv = tft.quaternion_matrix(q_tv)[0:3, 0:3]
    r_c = f.C_vc.T @ (-C_tv.T @ r_t - f.r_v_cv.flatten())
    if rng is not None:
        r_c = r_c + rng.normal(size=3) * sig_r
        dq = np.append(0.5 * rng.normal(size=3) * sig_a, 1.0)
        q_ct = tft.quaternion_multiply(q_ct, dq / np.linalg.norm(dq))
    q_ct = q_ct / np.linalg.norm(q_ct)
    return r_c, q_ct


def main():
    pyqh, pyekf, tft = _load_reference()
    rng = np.random.default_rng(0xE4F0)

    # ---------------- helper KATs ----------------
    kat = {"source": "reference Python twin (quaternion_helper.py, rel_pose_EKF_test_class.py) via shims",
           "skew_symm": [], "quaternion_norm": [], "quaternion_exp": [], "quaternion_log": []}
    for v in ([1.0, 2.0, 3.0], [-0.5, 0.25, 4.0]):
        kat["skew_symm"].append({"in": v, "out": pyqh.skew_symm(np.array(v)).tolist()})
    for q in ([0, 0, 0.6, -0.8], [0, 0, 0.8, -0.6], [1.0, 2.0, -3.0, 4.0], [0.1, -0.2, 0.3, -2.0]):
        kat["quaternion_norm"].append({"in": q, "out": pyqh.quaternion_norm(np.array(q, dtype=np.float64)).tolist()})
    exp_in = [[0.1, 0.2, 0.3], [0, 0, 0], [1e-12, -2e-12, 5e-13], [1.5, -0.7, 2.0], [3e-4, 1e-5, -2e-4]]
    for _ in range(8):
        exp_in.append((rng.normal(size=3) * rng.choice([1e-3, 0.1, 1.0])).tolist())
    for v in exp_in:
        kat["quaternion_exp"].append({"in": v, "out": pyqh.quaternion_exp(np.array(v, dtype=np.float64)).tolist()})
    log_in = [(np.array([0.1, 0.2, 0.3, 0.9]) / np.linalg.norm([0.1, 0.2, 0.3, 0.9])).tolist(), [0, 0, 0, 1.0],
              [1e-12, 0, 0, 1.0]]
    for _ in range(8):
        log_in.append(rand_unit_quat(rng).tolist())
    for _ in range(4):
        log_in.append(rand_unit_quat(rng, 1e-3).tolist())
    for q in log_in:
        kat["quaternion_log"].append({"in": q, "out": pyqh.quaternion_log(np.array(q, dtype=np.float64)).tolist()})

    # Appendix-C style single-shot KAT (predict then update)
    f = make_filter(pyekf, pyqh, tft, "pydefault")
    x0 = np.array([0.3, -0.2, 2.0, 0.1, 0.0, -0.05, 0, 0, 0, 1] + [0.0] * 6)
    u0 = np.array([0.2, -0.1, 9.9, 0.05, -0.02, 0.3])
    xc, Pc, acc = f.prediction_step(x0.reshape(16, 1), u0.reshape(6, 1), f.cov_init.copy())
    q_ct = np.array([0.7, -0.71, 0.02, 0.01])
    q_ct = q_ct / np.linalg.norm(q_ct)
    r_c = np.array([0.21, 0.33, 1.93])
    xh, Ph = f.correction_step(xc, Pc, r_c.reshape(3, 1), q_ct)
    kat["appendix_c"] = {"param_set": "pydefault", "x": x0.tolist(), "P_diag": np.diag(f.cov_init).tolist(),
                         "u": u0.tolist(), "x_check": xc.flatten().tolist(), "P_check": Pc.tolist(),
                         "accel": acc.flatten().tolist(), "r_c_tc": r_c.tolist(), "q_ct_xyzw": q_ct.tolist(),
                         "x_hat": xh.flatten().tolist(), "P_hat": Ph.tolist()}
    with open(os.path.join(OUT_DIR, "kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1)

    # ---------------- predict cases ----------------
    out = {}
    for ps in PARAM_SETS:
        f = make_filter(pyekf, pyqh, tft, ps)
        n = f.num_states
        N = 48
        X = np.zeros((N, 16)); U = np.zeros((N, 6)); P = np.zeros((N, n, n))
        Xo = np.zeros((N, 16)); Po = np.zeros((N, n, n)); Ao = np.zeros((N, 3))
        scale = ([0.1] * 3 + [0.1] * 3 + [0.15] * 3 + [0.5] * 3 + [0.1] * 3)[:n]
        for i in range(N):
            X[i] = rand_state(rng, f.est_bias)
            U[i, 0:3] = rng.normal(size=3) * 1.5 + np.array([0, 0, 9.8])
            U[i, 3:6] = rng.normal(size=3) * 0.4
            if i == 0:  # exact zero rate -> small-angle branch (EKF.cpp:385-389)
                U[i, 3:6] = X[i, 13:16]
            if i == 1:  # tiny but non-zero rate below tolerance
                U[i, 3:6] = X[i, 13:16] + np.array([1e-9, -2e-9, 1e-9])
            if i == 2:  # large rate
                U[i, 3:6] = np.array([8.0, -6.0, 5.0])
            P[i] = rand_spd(rng, n, scale) if i % 4 else np.diag(scale)
            xc, Pc, acc = f.prediction_step(X[i].reshape(16, 1), U[i].reshape(6, 1), P[i].copy())
            Xo[i] = xc.flatten(); Po[i] = Pc; Ao[i] = acc.flatten()
        out[ps] = dict(x=X, u=U, P=P, x_check=Xo, P_check=Po, accel=Ao)
    np.savez_compressed(os.path.join(OUT_DIR, "predict_cases.npz"),
                        **{f"{ps}__{k}": v for ps, d in out.items() for k, v in d.items()})

    # ---------------- update cases ----------------
    out = {}
    for ps in PARAM_SETS:
        f = make_filter(pyekf, pyqh, tft, ps)
        n = f.num_states
        N = 48
        X = np.zeros((N, 16)); Z = np.zeros((N, 7)); P = np.zeros((N, n, n))
        Xo = np.zeros((N, 16)); Po = np.zeros((N, n, n))
        scale = ([0.02] * 3 + [0.1] * 3 + [0.01] * 3 + [0.5] * 3 + [0.1] * 3)[:n]
        for i in range(N):
            X[i] = rand_state(rng, f.est_bias)
            # truth near the predicted state so that delta_q.w stays > -0.75
            dq = rand_unit_quat(rng, 1.2 if i % 3 else 0.05)
            q_true = tft.quaternion_multiply(X[i, 6:10], dq)
            q_true /= np.linalg.norm(q_true)
            r_true = X[i, 0:3] + rng.normal(size=3) * 0.2
            if i == 0:  # zero innovation
                q_true = X[i, 6:10].copy(); r_true = X[i, 0:3].copy()
                r_c, q_ct = meas_from_pose(f, tft, r_true, q_true)
            else:
                r_c, q_ct = meas_from_pose(f, tft, r_true, q_true, rng, 0.05, 0.03)
            if i % 5 == 0:
                # double cover of the measurement itself -- kept only while delta_q.w stays
                # > -0.7, the domain where the Python twin (no flip of delta_q, PYEKF.py:449)
                # and the C++ (flip at EKF.cpp:449) agree
                q_obs_n = pyqh.quaternion_norm(tft.quaternion_conjugate(tft.quaternion_multiply(f.q_vc, -q_ct)))
                if tft.quaternion_multiply(tft.quaternion_conjugate(X[i, 6:10]), q_obs_n)[3] > -0.7:
                    q_ct = -q_ct
            Z[i, 0:3] = r_c; Z[i, 3:7] = q_ct
            P[i] = rand_spd(rng, n, scale) if i % 4 else np.diag(scale)
            xh, Ph = f.correction_step(X[i].reshape(16, 1), P[i].copy(), r_c.reshape(3, 1), q_ct.copy())
            Xo[i] = xh.flatten(); Po[i] = Ph
        out[ps] = dict(x=X, z=Z, P=P, x_hat=Xo, P_hat=Po)
    np.savez_compressed(os.path.join(OUT_DIR, "update_cases.npz"),
                        **{f"{ps}__{k}": v for ps, d in out.items() for k, v in d.items()})

    # ---------------- sequences (single-rate branch, EKF.cpp:238-249,265-301) ----------------
    out = {}
    for ps, T, every in (("pydefault", 350, 7), ("rotors400", 560, 14), ("hardware", 200, 1), ("nobias", 140, 7)):
        f = make_filter(pyekf, pyqh, tft, ps)
        n = f.num_states
        dT = f.dT
        # smooth truth: position sinusoid, attitude from integrated body rate
        r0 = rng.uniform([-1, -1, 1], [1, 1, 4]); A = rng.uniform(0, 0.5, size=3)
        om = rng.uniform(0.2, 1.5, size=3); ph = rng.uniform(0, 2 * math.pi, size=3)
        wa = rng.uniform(0, 0.3, size=3); wo = rng.uniform(0.2, 1.5, size=3); wp = rng.uniform(0, 2 * math.pi, size=3)
        ab_t = rng.normal(size=3) * 0.1 * (1 if f.est_bias else 0)
        wb_t = rng.normal(size=3) * 0.01 * (1 if f.est_bias else 0)
        q_t = rand_unit_quat(rng, 0.4)
        g = np.array([0, 0, -9.8])
        U = np.zeros((T, 6)); Z = np.zeros((T, 7)); M = np.zeros(T, dtype=np.uint8)
        Xs = np.zeros((T, 16)); Pd = np.zeros((T, n)); Pfull = {}
        # seed (EKF.cpp:310-313 semantic, here through the twin's own formula PYEKF.py:353-355)
        r_c, q_ct = meas_from_pose(f, tft, r0 + A * np.sin(ph), q_t, rng, 0.03, 0.02)
        q_nom = pyqh.quaternion_norm(tft.quaternion_conjugate(tft.quaternion_multiply(f.q_vc, q_ct)))
        C_nom = tft.quaternion_matrix(q_nom)[0:3, 0:3]
        r_nom = -C_nom @ (f.C_vc @ r_c + f.r_v_cv.flatten())
        x = np.zeros(16); x[0:3] = r_nom; x[6:10] = q_nom
        P = f.cov_init.copy()
        x_init = x.copy(); P_init = P.copy()
        for t in range(T):
            tt = t * dT
            acc_t = -A * om * om * np.sin(om * tt + ph)
            w_t = wa * np.sin(wo * tt + wp)
            C_t = tft.quaternion_matrix(q_t)[0:3, 0:3]
            U[t, 0:3] = C_t.T @ (acc_t - g) + ab_t + rng.normal(size=3) * math.sqrt(f.Q[0, 0])
            U[t, 3:6] = w_t + wb_t + rng.normal(size=3) * math.sqrt(f.Q[3, 3])
            xc, Pc, _ = f.prediction_step(x.reshape(16, 1), U[t].reshape(6, 1), P)
            # truth advances one tick
            q_t = tft.quaternion_multiply(q_t, pyqh.quaternion_exp(dT * w_t)); q_t /= np.linalg.norm(q_t)
            r_t = r0 + A * np.sin(om * (tt + dT) + ph)
            if (t + 1) % every == 0:
                r_c, q_ct = meas_from_pose(f, tft, r_t, q_t, rng, 0.03, 0.02)
                Z[t, 0:3] = r_c; Z[t, 3:7] = q_ct; M[t] = 1
                xh, Ph = f.correction_step(xc, Pc, r_c.reshape(3, 1), q_ct.copy())
                x = xh.flatten(); P = Ph
            else:
                x = xc.flatten(); P = Pc
            Xs[t] = x; Pd[t] = np.diag(P)
            if t in (0, every - 1, T // 2, T - 1):
                Pfull[t] = P.copy()
        ticks = sorted(Pfull)
        out[ps] = dict(x_init=x_init, P_init=P_init, u=U, z=Z, mask=M, x_seq=Xs, P_diag_seq=Pd,
                       P_full_ticks=np.array(ticks), P_full=np.stack([Pfull[t] for t in ticks]))
    np.savez_compressed(os.path.join(OUT_DIR, "sequence_cases.npz"),
                        **{f"{ps}__{k}": v for ps, d in out.items() for k, v in d.items()})
    # ---------------- full filter_update of the twin (PYEKF.py:172-337) ----------------
    # multirate replay with the twin's fixed step delay (measurement_delay 0.050 s -> 5 ticks, PYEKF.py:59-60),
    # rate limiting (upd_per_meas = ceil(100/10) = 10, PYEKF.py:58,184), single-tag corner gate
    # (PYEKF.py:199-211).  The twin reads ROS message objects; plain namespaces stand in for them.
    from types import SimpleNamespace as NS

    def imu_msg(u):
        return NS(linear_acceleration=NS(x=float(u[0]), y=float(u[1]), z=float(u[2])),
                  angular_velocity=NS(x=float(u[3]), y=float(u[4]), z=float(u[5])))

    def tag_msg(z):
        return NS(detections=[NS(pose=NS(pose=NS(pose=NS(position=NS(x=float(z[0]), y=float(z[1]), z=float(z[2])),
                                                         orientation=NS(x=float(z[3]), y=float(z[4]), z=float(z[5]), w=float(z[6]))))))])

    out = {}
    for name, mr in (("multirate", True), ("singlerate", False)):
        f = pyekf.RelativePoseEKF(100.0, 10.0)
        f.multirate_EKF = mr
        T = 160
        U = np.zeros((T, 6)); Z = np.zeros((T, 7)); NEW = np.zeros(T, dtype=np.uint8)
        Xs = np.zeros((T, 16)); Pd = np.zeros((T, 15)); UPD = np.zeros(T, dtype=np.int32); RDY = np.zeros(T, dtype=np.uint8)
        HL = np.zeros(T, dtype=np.int32)
        # truth: vehicle ~2 m above the tag, slow motion, camera looking down
        r0 = np.array([0.1, -0.05, 2.0]); A = np.array([0.15, 0.1, 0.2]); om = np.array([0.9, 0.7, 0.5]); ph = np.array([0.3, 1.1, 2.0])
        q_t = rand_unit_quat(rng, 0.15)
        g = np.array([0, 0, -9.8])
        P_last = None
        for t in range(T):
            tt = t * f.dT
            acc_t = -A * om * om * np.sin(om * tt + ph)
            w_t = np.array([0.05, -0.04, 0.08]) * np.sin(np.array([0.8, 1.1, 0.6]) * tt)
            C_t = tft.quaternion_matrix(q_t)[0:3, 0:3]
            U[t, 0:3] = C_t.T @ (acc_t - g) + rng.normal(size=3) * 0.05
            U[t, 3:6] = w_t + rng.normal(size=3) * 0.01
            f.IMU_msg = imu_msg(U[t])
            # a tag pose arrives on an irregular pattern; some of them far off-axis (rejected by the corner gate)
            if t == 0 or (t > 2 and rng.uniform() < 0.35):
                r_t = r0 + A * np.sin(om * tt + ph)
                r_c, q_ct = meas_from_pose(f, tft, r_t, q_t, rng, 0.02, 0.01)
                if t > 0 and rng.uniform() < 0.2:
                    r_c = r_c + np.array([3.5, 0.0, 0.0])
                Z[t, 0:3] = r_c; Z[t, 3:7] = q_ct; NEW[t] = 1
                f.apriltag_msg = tag_msg(Z[t])
                f.measurement_ready = True
                if not f.state_initialized:
                    f.initialize_state(False)
            f.filter_update()
            q_t = tft.quaternion_multiply(q_t, pyqh.quaternion_exp(f.dT * w_t)); q_t /= np.linalg.norm(q_t)
            Xs[t] = np.concatenate([f.r_nom.flatten(), f.v_nom.flatten(), f.q_nom.flatten(), f.ab_nom.flatten(), f.wb_nom.flatten()])
            Pd[t] = np.diag(f.cov_pert); UPD[t] = f.upds_since_correction; RDY[t] = 1 if f.measurement_ready else 0
            HL[t] = len(f.x_hist)
            P_last = f.cov_pert.copy()
        out[name] = dict(u=U, z=Z, new=NEW, x_seq=Xs, P_diag_seq=Pd, upds=UPD, ready=RDY, hist_len=HL, P_final=P_last)
        print(name, "corrections:", int((UPD == 0).sum()), "max history:", int(HL.max()))
    np.savez_compressed(os.path.join(OUT_DIR, "filter_update_cases.npz"),
                        **{f"{n}__{k}": v for n, d in out.items() for k, v in d.items()})

    with open(os.path.join(OUT_DIR, "param_sets.json"), "w") as fh:
        json.dump(PARAM_SETS, fh, indent=1)
    print("golden fixtures written to", OUT_DIR)


if __name__ == "__main__":
    main()
