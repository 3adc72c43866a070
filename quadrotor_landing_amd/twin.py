"""One-filter Python class with the interface of the reference's Python twin
(quad_state_estimation/test/rel_pose_EKF_test_class.py: `class RelativePoseEKF`), backed by the HIP engine.

Same constructor and method signatures, argument order and array shapes as the twin:

    RelativePoseEKF(update_freq, measurement_freq)                         PYEKF.py:29
    prediction_step(x_km1 (16,1), u (6,1), P_km1 (n,n)) -> x_check (16,1), P_check (n,n), accel_rel (3,1)   PYEKF.py:373-426
    correction_step(x_check (16,1), P_check (n,n), r_c_tc (3,1), q_ct (4,)) -> x_hat (16,1), P_hat (n,n)    PYEKF.py:428-487

and the twin's public attribute names for what the two steps read (dT, Q, R, est_bias, num_states, r_v_cv,
q_vc, g).  Like the twin it hard-codes the direct orientation method (PYEKF.py:436,455,458) and has no
static biases.  Changing Q / R / r_v_cv / q_vc / est_bias and calling `sync_params()` pushes them to the
engine (the twin reads its attributes on every call).  Every step runs on the GPU; nothing is computed here.
"""
import numpy as np

from .ekf import BatchedRelativePoseEKF


class RelativePoseEKF(object):
    def __init__(self, update_freq, measurement_freq, dtype="f64", device=0):
        self.update_freq = update_freq
        self.dT = 1 / update_freq                                   # PYEKF.py:56
        self.measurement_freq = measurement_freq
        self.est_bias = True                                        # PYEKF.py:61
        self.num_states = 15
        self.Q = np.diag([0.005] * 3 + [0.0005] * 3 + [5e-5] * 3 + [5e-6] * 3)   # PYEKF.py:94-104 (sim values)
        self.R = np.diag([0.005, 0.005, 0.015, 0.0025, 0.0025, 0.025])           # PYEKF.py:113-124
        self.r_v_cv = np.array([[0], [0], [-0.073]], dtype=np.float64)           # PYEKF.py:134
        self.q_vc = np.array([0.70711, -0.70711, 0, 0], dtype=np.float64)        # PYEKF.py:136 (normalised by the engine)
        self.g = np.array([[0], [0], [-9.8]], dtype=np.float64)                  # PYEKF.py:167
        self._e = BatchedRelativePoseEKF(1, dtype, device=device, direct_orien_method=1)
        self.sync_params()

    def sync_params(self):
        q = np.diag(np.asarray(self.Q, dtype=np.float64))
        r = np.diag(np.asarray(self.R, dtype=np.float64))
        kw = dict(update_freq=1.0 / self.dT, measurement_freq=self.measurement_freq, est_bias=int(bool(self.est_bias)),
                  Q_a=q[0:3], Q_w=q[3:6], R_r=r[0:3], R_ang=r[3:6], r_v_cv=np.asarray(self.r_v_cv).reshape(3),
                  q_vc=np.asarray(self.q_vc).reshape(4), g=np.asarray(self.g).reshape(3), direct_orien_method=1)
        if self.est_bias:
            kw.update(Q_ab=q[6:9], Q_wb=q[9:12])
        self._e.initialize_params(**kw)
        self.num_states = self._e.num_states

    def prediction_step(self, x_km1, u, P_km1):
        n = self.num_states
        x, P, acc = self._e.prediction_step(np.asarray(x_km1, dtype=np.float64).reshape(1, 16),
                                            np.asarray(P_km1, dtype=np.float64).reshape(1, n, n),
                                            np.asarray(u, dtype=np.float64).reshape(1, 6))
        return x.reshape(16, 1), P[0], acc.reshape(3, 1)

    def correction_step(self, x_check, P_check, r_c_tc, q_ct):
        n = self.num_states
        x, P = self._e.correction_step(np.asarray(x_check, dtype=np.float64).reshape(1, 16),
                                       np.asarray(P_check, dtype=np.float64).reshape(1, n, n),
                                       np.asarray(r_c_tc, dtype=np.float64).reshape(1, 3),
                                       np.asarray(q_ct, dtype=np.float64).reshape(1, 4))
        return x.reshape(16, 1), P[0]

    def close(self):
        self._e.close()
