"""Independent numpy restatement of the reference EKF steps.

TEST INFRASTRUCTURE ONLY (same rules as oracle/ekf_oracle.c).  Written
separately from the C restatement and in a different style (numpy linear
algebra, np.linalg.inv) so that an error in one is unlikely to be mirrored in
the other.  Follows the C++ text of the reference:
  quad_state_estimation/src/relative_pose_EKF.cpp:346-415 (prediction_step)
  quad_state_estimation/src/relative_pose_EKF.cpp:417-502 (correction_step)
  quad_state_estimation/src/relative_pose_EKF.cpp:305-344 (initialize_state)
  quad_state_estimation/src/relative_pose_EKF.cpp:156-186 (corner gate)
  quad_state_estimation/src/quaternion_helper.cpp:9-100
Quaternions are x,y,z,w.
"""
import math

import numpy as np


def quaternion_norm(q):  # QH.cpp:61-73
    q = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return -q if q[3] < -0.75 else q


def quaternion_exp(v):  # QH.cpp:9-33
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v)
    q = np.zeros(4)
    q[3] = math.cos(n / 2)
    q[:3] = v / 2 * (1 - n ** 2 / 24) if n < 1e-10 else v / n * math.sin(n / 2)
    return quaternion_norm(q)


def quaternion_log(q):  # QH.cpp:36-58
    q = np.asarray(q, dtype=np.float64)
    m = np.linalg.norm(q[:3])
    if m < 1e-10:
        return 2 / q[3] * (1 - (m / q[3]) ** 2 / 3) * q[:3]
    return 2 * math.atan2(m, q[3]) / m * q[:3]


def skew(v):  # QH.cpp:76-85
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def qmul(a, b):  # Hamilton product
    av, aw, bv, bw = a[:3], a[3], b[:3], b[3]
    return np.append(aw * bv + bw * av + np.cross(av, bv), aw * bw - av @ bv)


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def rot(q):  # Eigen toRotationMatrix for a unit quaternion
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def rodrigues(angle, axis):  # Eigen AngleAxis(angle, axis).toRotationMatrix()
    K = skew(axis)
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


class Params:
    """Plain attribute bag; `from_orc` copies an oracle.OrcParams."""

    @staticmethod
    def from_orc(p):
        s = Params()
        for name in ("dT_nom", "est_bias", "direct_orien_method", "num_states", "small_ang_tol",
                     "camera_width", "camera_height", "n_tags", "tag_in_view_margin"):
            setattr(s, name, getattr(p, name))
        for name in ("Q", "R", "ab_static", "wb_static", "r_v_cv", "q_vc", "g", "cov_init", "tag_widths", "tag_positions"):
            setattr(s, name, np.array(list(getattr(p, name)), dtype=np.float64))
        s.C_vc = np.array(list(p.C_vc)).reshape(3, 3)
        s.camera_K = np.array(list(p.camera_K)).reshape(3, 3)
        return s


def prediction_step(p, x, P, u):  # EKF.cpp:346-415
    x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
    n = p.num_states
    r, v, q, ab, wb = x[0:3], x[3:6], x[6:10], x[10:13], x[13:16]
    dT = p.dT_nom
    a = u[0:3] - ab - p.ab_static
    w = u[3:6] - wb - p.wb_static
    C = rot(q)
    accel = C @ a + p.g
    xo = np.zeros(16)
    xo[0:3] = r + dT * v
    xo[3:6] = v + dT * accel
    xo[6:10] = quaternion_norm(qmul(q, quaternion_exp(dT * w)))
    xo[10:13] = ab; xo[13:16] = wb
    F = np.eye(n)
    F[0:3, 3:6] = dT * np.eye(3)
    F[3:6, 6:9] = -dT * C @ skew(a)
    phi = dT * w
    ang = np.linalg.norm(phi)
    F[6:9, 6:9] = np.eye(3) - skew(phi) if ang < p.small_ang_tol else rodrigues(-ang, phi / ang)
    nq = 12 if p.est_bias else 6
    W = np.zeros((n, nq))
    W[3:6, 0:3] = -C
    W[6:n, 3:nq] = np.eye(n - 6)
    if p.est_bias:
        F[3:6, 9:12] = -dT * C
        F[6:9, 12:15] = -dT * np.eye(3)
    Po = F @ np.asarray(P, dtype=np.float64).reshape(n, n) @ F.T + W @ np.diag(p.Q[:nq]) @ W.T
    return xo, Po, accel


def observe(p, q_attitude_for_position, r_c_tc, q_ct):
    q_obs = quaternion_norm(qconj(qmul(p.q_vc, q_ct)))
    qa = q_obs if q_attitude_for_position is None else q_attitude_for_position
    r_obs = -(rot(qa) @ (p.C_vc @ r_c_tc + p.r_v_cv))
    return r_obs, q_obs


def correction_step(p, x, P, r_c_tc, q_ct):  # EKF.cpp:417-502
    x = np.asarray(x, dtype=np.float64)
    n = p.num_states
    P = np.asarray(P, dtype=np.float64).reshape(n, n)
    r, v, q, ab, wb = x[0:3], x[3:6], x[6:10], x[10:13], x[13:16]
    Cc = rot(q)
    r_obs, q_obs = observe(p, None if p.direct_orien_method else q, np.asarray(r_c_tc, float), np.asarray(q_ct, float))
    dy = np.zeros(6)
    dy[0:3] = r_obs - r
    dy[3:6] = quaternion_log(quaternion_norm(qmul(qconj(q), q_obs)))
    G = np.zeros((6, n))
    G[0:3, 0:3] = np.eye(3)
    G[3:6, 6:9] = np.eye(3)
    N = np.zeros((6, 6))
    N[0:3, 0:3] = -Cc @ p.C_vc
    N[3:6, 3:6] = p.C_vc
    if p.direct_orien_method:
        N[0:3, 3:6] = skew(r)
    else:
        G[0:3, 6:9] = Cc @ skew(Cc.T @ r)
    Rk = N @ np.diag(p.R) @ N.T
    K = P @ G.T @ np.linalg.inv(G @ P @ G.T + Rk)
    Po = (np.eye(n) - K @ G) @ P
    dx = K @ dy
    xo = np.zeros(16)
    xo[0:3] = r + dx[0:3]
    xo[3:6] = v + dx[3:6]
    xo[6:10] = quaternion_norm(qmul(q, quaternion_exp(dx[6:9])))
    if p.est_bias:
        xo[10:13] = ab + dx[9:12]; xo[13:16] = wb + dx[12:15]
    return xo, Po, r_obs, q_obs


def seed_pose(p, r_c_tc, q_ct):  # EKF.cpp:310-313
    r_obs, q_obs = observe(p, None, np.asarray(r_c_tc, float), np.asarray(q_ct, float))
    return r_obs, q_obs


def corner_gate(p, r_c_tc, q_ct):  # EKF.cpp:156-186
    C = rot(np.asarray(q_ct, float))
    t = np.asarray(r_c_tc, float)
    for i in range(p.n_tags):
        hw = p.tag_widths[i] / 2
        px, py = p.tag_positions[3 * i], p.tag_positions[3 * i + 1]
        corners = np.array([[hw + px, -hw + px, -hw + px, hw + px],
                            [hw + py, hw + py, -hw + py, -hw + py],
                            [0, 0, 0, 0]], dtype=np.float64)
        pc = C @ corners + t[:, None]
        pn = pc / pc[2:3, :]
        px_ = p.camera_K @ pn
        mn = px_.min(axis=1); mx = px_.max(axis=1)
        m = p.tag_in_view_margin
        if (mn[0] > p.camera_width * m and mn[1] > p.camera_height * m and
                mx[0] < p.camera_width * (1 - m) and mx[1] < p.camera_height * (1 - m)):
            return 1
    return 0
