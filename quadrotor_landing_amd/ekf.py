"""Host-side mirror of the reference filter interface over the C-ABI.

`BatchedRelativePoseEKF` keeps the names and argument meaning of the
reference's `class RelativePoseEKF` (quad_state_estimation/include/
relative_pose_EKF.hpp:20-141; Python twin test/rel_pose_EKF_test_class.py)
for a batch of independent filters resident on one MI355X:

    reference (one filter, Eigen)                 here (B filters, numpy in/out)
    prediction_step(x, P, u, &x', &P', &accel)    prediction_step(x, P, u) -> x', P', accel
    correction_step(x, P, r_c_tc, q_ct, &x, &P)   correction_step(x, P, r_c_tc, q_ct) -> x, P
    initialize_state(reinit_bias)                 initialize_state(z, reinit_bias)
    initialize_params()                           initialize_params()
    filter_update(t) single-rate tick             step(u, z, mask) / run(inputs, t0, n)

All compute happens in the HIP kernels behind libqle_ekf.so; nothing here
computes filter arithmetic.
"""
import ctypes as C
import weakref

import numpy as np

from . import params as _params
from ._lib import QLE_F32, QLE_F64, QlePolicy, QleSynthCfg, check, lib

_pd = C.POINTER(C.c_double)
_pu8 = C.POINTER(C.c_uint8)


def _dp(a):
    return None if a is None else a.ctypes.data_as(_pd)


def _f64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _u8(a, shape):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.shape != tuple(shape):
        raise ValueError(f"expected mask shape {tuple(shape)}, got {a.shape}")
    return a


class InputSequence:
    """Device-resident IMU / tag-pose sequence (`qle_inputs`)."""

    def __init__(self, ekf, n_ticks, tick_has_meas=None):
        self.ekf = ekf
        self.n_ticks = int(n_ticks)
        thm = np.zeros(self.n_ticks, dtype=np.uint8) if tick_has_meas is None else _u8(tick_has_meas, (self.n_ticks,))
        self.tick_has_meas = thm.copy()
        self._h = C.c_void_p()
        check(lib().qle_inputs_create(ekf._h, self.n_ticks, thm.ctypes.data_as(_pu8), C.byref(self._h)))
        ekf._sequences.append(weakref.ref(self))

    def upload_tick(self, t, u, z=None, mask=None):
        B = self.ekf.batch
        u = _f64(u, (B, 6))
        z = None if z is None else _f64(z, (B, 7))
        mask = _u8(mask, (B,))
        check(lib().qle_inputs_upload_tick(self._h, int(t), _dp(u), _dp(z), None if mask is None else mask.ctypes.data_as(_pu8)))

    def download_tick(self, t):
        B = self.ekf.batch
        u = np.empty((B, 6)); z = np.zeros((B, 7)); m = np.zeros(B, dtype=np.uint8)
        check(lib().qle_inputs_download_tick(self._h, int(t), _dp(u), _dp(z), m.ctypes.data_as(_pu8)))
        return u, z, m

    def close(self):
        if self._h:
            lib().qle_inputs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchedRelativePoseEKF:
    def __init__(self, batch, dtype="f32", device=0, params=None, **param_overrides):
        self.batch = int(batch)
        self.dtype = {"f32": QLE_F32, "fp32": QLE_F32, "float32": QLE_F32, "f64": QLE_F64, "fp64": QLE_F64,
                      "float64": QLE_F64}[str(dtype)]
        self.device = int(device)
        self.params = params if params is not None else _params.default_params()
        _params.set_fields(self.params, **param_overrides)
        self._h = C.c_void_p()
        self._sequences = []
        check(lib().qle_create(C.byref(self._h), self.batch, self.dtype, self.device, C.byref(self.params)))
        self.derived = _params.derive(self.params)

    # ---- lifetime
    def close(self):
        for ref in getattr(self, "_sequences", []):  # sequences live in this handle's device context
            seq = ref()
            if seq is not None:
                seq.close()
        self._sequences = []
        if self._h:
            lib().qle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters
    @property
    def num_states(self):
        return int(self.derived.num_states)

    def initialize_params(self, **param_overrides):
        """RelativePoseEKF::initialize_params (relative_pose_EKF.cpp:87-125) after changing public members."""
        _params.set_fields(self.params, **param_overrides)
        check(lib().qle_set_params(self._h, C.byref(self.params)))
        self.derived = _params.derive(self.params)

    def set_filter_params(self, pfp):
        """Per-filter [Q diag 12, ab_static 3, wb_static 3, R diag 6]; None = shared parameters."""
        check(lib().qle_set_filter_params(self._h, None if pfp is None else _dp(_f64(pfp, (self.batch, 24)))))

    def get_filter_params(self):
        """The per-filter [Q diag 12, ab_static 3, wb_static 3, R diag 6] records as the device holds them."""
        out = np.empty((self.batch, 24))
        check(lib().qle_get_filter_params(self._h, _dp(out)))
        return out

    # ---- state
    def set_state(self, x, P):
        n = self.num_states
        check(lib().qle_set_state(self._h, _dp(_f64(x, (self.batch, 16))), _dp(_f64(P, (self.batch, n, n)))))

    def get_state(self):
        n = self.num_states
        x = np.empty((self.batch, 16)); P = np.empty((self.batch, n, n))
        check(lib().qle_get_state(self._h, _dp(x), _dp(P)))
        return x, P

    def initialize_state(self, z, reinit_bias=False, mask=None):
        """RelativePoseEKF::initialize_state (relative_pose_EKF.cpp:305-344) from each filter's first tag pose.
        mask [batch]: seed only those filters (the node seeds a filter on its own first detection,
        relative_pose_EKF_node.cpp:169-174); filters never seeded are skipped by every tick (relative_pose_EKF.cpp:129-130)."""
        m = _u8(mask, (self.batch,))
        check(lib().qle_initialize_state_masked(self._h, _dp(_f64(z, (self.batch, 7))), None if m is None else m.ctypes.data_as(_pu8),
                                                int(bool(reinit_bias))))

    def state_initialized(self):
        out = np.zeros(self.batch, np.uint8)
        check(lib().qle_get_state_initialized(self._h, out.ctypes.data_as(_pu8)))
        return out

    def enable_aux(self, on=True):
        check(lib().qle_enable_aux(self._h, int(bool(on))))

    def get_aux(self):
        acc = np.empty((self.batch, 3)); obs = np.empty((self.batch, 7))
        check(lib().qle_get_aux(self._h, _dp(acc), _dp(obs)))
        return acc, obs

    # ---- hot path on the resident state
    def predict(self, u):
        check(lib().qle_predict(self._h, _dp(_f64(u, (self.batch, 6)))))

    def update(self, z, mask=None):
        m = _u8(mask, (self.batch,))
        check(lib().qle_update(self._h, _dp(_f64(z, (self.batch, 7))), None if m is None else m.ctypes.data_as(_pu8)))

    def step(self, u, z=None, mask=None):
        """One single-rate filter_update tick (relative_pose_EKF.cpp:238-249,265-290)."""
        m = _u8(mask, (self.batch,))
        zz = None if z is None else _f64(z, (self.batch, 7))
        check(lib().qle_step(self._h, _dp(_f64(u, (self.batch, 6))), _dp(zz), None if m is None else m.ctypes.data_as(_pu8)))

    # ---- filter_update with the decision logic on the device (relative_pose_EKF.cpp:147-186)
    def enable_gating(self, on=True):
        check(lib().qle_enable_gating(self._h, int(bool(on))))

    def filter_update(self, u, z=None, measurement_ready=None, t_curr=None, apriltag_time=None):
        """One filter_update tick: rate limit + corner gate + predict (+ correct) per filter; with
        multirate_ekf the correction goes to the delayed history entry and the predicts are replayed.
        t_curr / apriltag_time [batch] feed dynamic_meas_delay (relative_pose_EKF.cpp:199)."""
        m = _u8(measurement_ready, (self.batch,))
        zz = None if z is None else _f64(z, (self.batch, 7))
        mp = None if m is None else m.ctypes.data_as(_pu8)
        if t_curr is None:
            check(lib().qle_filter_update(self._h, _dp(_f64(u, (self.batch, 6))), _dp(zz), mp))
        else:
            st = None if apriltag_time is None else _f64(apriltag_time, (self.batch,))
            check(lib().qle_filter_update_stamped(self._h, _dp(_f64(u, (self.batch, 6))), _dp(zz), mp, float(t_curr), _dp(st)))

    def measurement_delay(self):
        out = np.zeros(self.batch)
        check(lib().qle_get_measurement_delay(self._h, _dp(out)))
        return out

    def set_uniform_measurement_age(self, seconds):
        check(lib().qle_set_uniform_measurement_age(self._h, float(seconds)))

    def tick_flags(self):
        """(performed_correction, consumed, upds_since_correction) after the last tick."""
        pc = np.zeros(self.batch, np.uint8); co = np.zeros(self.batch, np.uint8); up = np.zeros(self.batch, np.int32)
        check(lib().qle_get_tick_flags(self._h, pc.ctypes.data_as(_pu8), co.ctypes.data_as(_pu8), up.ctypes.data_as(C.POINTER(C.c_int32))))
        return pc, co, up

    # ---- reference-shaped value-in / value-out calls
    def prediction_step(self, x_km1, P_km1, u):
        """prediction_step(x_km1, P_km1, u) -> (x_check, P_check, pose_accel)   (relative_pose_EKF.cpp:346-415)."""
        self.set_state(x_km1, P_km1)
        self.enable_aux(True)
        self.predict(u)
        x, P = self.get_state()
        return x, P, self.get_aux()[0]

    def correction_step(self, x_check, P_check, r_c_tc, q_ct):
        """correction_step(x_check, P_check, r_c_tc, q_ct) -> (x_hat, P_hat)   (relative_pose_EKF.cpp:417-502)."""
        self.set_state(x_check, P_check)
        z = np.concatenate([_f64(r_c_tc, (self.batch, 3)), _f64(q_ct, (self.batch, 4))], axis=1)
        self.update(z)
        return self.get_state()

    # ---- device-resident sequences
    def make_inputs(self, n_ticks, tick_has_meas=None):
        return InputSequence(self, n_ticks, tick_has_meas)

    def run(self, inputs, t0, n):
        check(lib().qle_run(self._h, inputs._h, int(t0), int(n)))

    def run_resident(self, inputs, t0, n):
        """n ticks in one launch with the state held on-chip (reported separately from the per-tick metric)."""
        check(lib().qle_run_resident(self._h, inputs._h, int(t0), int(n)))

    def synth_generate(self, inputs, seed, filter_offset=0, perturb_filter_params=False, **kw):
        c = QleSynthCfg()
        check(lib().qle_synth_cfg_default(C.byref(c)))
        c.seed = int(seed); c.filter_offset = int(filter_offset); c.perturb_filter_params = int(bool(perturb_filter_params))
        for k, v in kw.items():
            setattr(c, k, v)
        check(lib().qle_synth_generate(self._h, inputs._h, C.byref(c)))

    def synth_truth(self, inputs):
        """(pose [batch,7], imu_bias [batch,6]) of the generator's truth at the end of the sequence."""
        pose = np.empty((self.batch, 7)); bias = np.empty((self.batch, 6))
        check(lib().qle_synth_get_truth(self._h, inputs._h, _dp(pose), _dp(bias)))
        return pose, bias

    def synth_rmse(self, inputs):
        out = np.zeros(3)
        check(lib().qle_synth_rmse(self._h, inputs._h, _dp(out)))
        return out

    # ---- reporting / control
    def report(self):
        """What the node publishes after a tick (relative_pose_EKF_node.cpp:192-220)."""
        B = self.batch
        pose = np.empty((B, 7)); cov = np.empty((B, 36)); vel = np.empty((B, 3)); bias = np.empty((B, 6))
        check(lib().qle_get_report(self._h, _dp(pose), _dp(cov), _dp(vel), _dp(bias)))
        return {"pose": pose, "pose_cov": cov.reshape(B, 6, 6), "vel": vel, "bias": bias}

    def node_report(self):
        """Everything the node publishes after a tick (relative_pose_EKF_node.cpp:192-281) in one call: a numpy record array
        with one entry per filter (fields of `struct qle_node_report`)."""
        from ._lib import QleNodeReport
        buf = (QleNodeReport * self.batch)()
        check(lib().qle_get_node_report(self._h, buf))
        return np.ctypeslib.as_array(buf).copy() if hasattr(np.ctypeslib, "as_array") else np.frombuffer(buf, dtype=np.dtype(QleNodeReport)).copy()

    def count_nonfinite(self):
        c = C.c_int64(0)
        check(lib().qle_count_nonfinite(self._h, C.byref(c)))
        return int(c.value)

    def synchronize(self):
        check(lib().qle_synchronize(self._h))

    def timer_begin(self):
        check(lib().qle_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float(0)
        check(lib().qle_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    def policy(self):
        """How this handle launches its ticks (dict of `struct qle_policy`) plus `served_by`: where the state lives between ticks."""
        pol = QlePolicy()
        check(lib().qle_get_policy(self._h, C.byref(pol)))
        d = {n: int(getattr(pol, n)) for n, _ in QlePolicy._fields_}
        # 256 MiB Infinity Cache (MI355X_MICROARCH.md): a state ring that fits stays resident from tick to tick under policies 0-2;
        # the split policy keeps a fixed part resident and streams the rest; anything larger streams from HBM
        # "split": part of what a tick touches is served on die, part by HBM -- the cached / streamed split policy, a state a
        # little larger than the cache under the cached policy, or the multirate filter (state on die, history streamed to HBM)
        cache = 250 * 2 ** 20
        if d["state_policy"] == 3 or (d["ring_bytes"] > cache and (d["state_bytes"] <= cache or d["state_policy"] == 0)):
            d["served_by"] = "split"
        elif d["ring_bytes"] <= cache:
            d["served_by"] = "infinity_cache"
        else:
            d["served_by"] = "hbm"
        return d

    def algorithmic_bytes(self, kind):
        return int(lib().qle_algorithmic_bytes(self._h, int(kind)))
