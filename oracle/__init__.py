"""ctypes loader for the CPU oracle (oracle/ekf_oracle.c).

TEST INFRASTRUCTURE ONLY.  Import this from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never from quadrotor_landing_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# QLE_ORACLE_SAN=1 (oracle/sanitize.sh): the AddressSanitizer + UBSan builds under oracle/_san/ instead (made by `make sanitize`)
_SAN = os.environ.get("QLE_ORACLE_SAN", "0") == "1"
_LIBDIR = os.path.join(_HERE, "_san") if _SAN else _HERE
_LIB = os.path.join(_LIBDIR, "libekf_oracle.so")
_LIB_STRUCT = os.path.join(_LIBDIR, "libekf_oracle_structured.so")
ORC_MAX_TAGS = 16

_d = C.c_double
_i = C.c_int


class OrcParams(C.Structure):
    """Mirror of `struct orc_params` (oracle/ekf_oracle.h)."""
    _fields_ = [
        ("update_freq", _d), ("measurement_freq", _d), ("measurement_delay", _d),
        ("measurement_delay_max", _d), ("dyn_measurement_delay_offset", _d),
        ("est_bias", _i), ("limit_measurement_freq", _i), ("corner_margin_enbl", _i),
        ("direct_orien_method", _i), ("multirate_ekf", _i), ("dynamic_meas_delay", _i),
        ("r_cov_init", _d), ("v_cov_init", _d), ("ang_cov_init", _d), ("ab_cov_init", _d), ("wb_cov_init", _d),
        ("Q_a", _d * 3), ("Q_w", _d * 3), ("Q_ab", _d * 3), ("Q_wb", _d * 3),
        ("R_r", _d * 3), ("R_ang", _d * 3),
        ("ab_static", _d * 3), ("wb_static", _d * 3),
        ("r_v_cv", _d * 3), ("q_vc", _d * 4),
        ("camera_K", _d * 9), ("camera_width", _i), ("camera_height", _i),
        ("n_tags", _i), ("tag_in_view_margin", _d),
        ("tag_widths", _d * ORC_MAX_TAGS), ("tag_positions", _d * (3 * ORC_MAX_TAGS)),
        ("small_ang_tol", _d), ("g", _d * 3),
        ("dT_nom", _d), ("upd_per_meas", _i), ("num_states", _i), ("measurement_step_delay", _i),
        ("Q", _d * 12), ("R", _d * 6), ("cov_init", _d * 15), ("C_vc", _d * 9),
    ]


class OrcFilter(C.Structure):
    """Mirror of `struct orc_filter`."""
    _fields_ = [
        ("p", OrcParams),
        ("IMU_accel", _d * 3), ("IMU_ang_vel", _d * 3),
        ("apriltag_pos", _d * 3), ("apriltag_orien", _d * 4), ("apriltag_time", _d),
        ("r_nom", _d * 3), ("v_nom", _d * 3), ("accel_rel", _d * 3), ("q_nom", _d * 4),
        ("ab_nom", _d * 3), ("wb_nom", _d * 3),
        ("cov_pert", _d * 225),
        ("r_t_vt_obs", _d * 3), ("q_tv_obs", _d * 4),
        ("hist_len", _i), ("hist_cap", _i),
        ("x_hist", C.POINTER(_d)), ("u_hist", C.POINTER(_d)), ("P_hist", C.POINTER(_d)),
        ("state_initialized", _i), ("measurement_ready", _i), ("performed_correction", _i), ("filter_active", _i),
        ("upds_since_correction", _i),
        ("measurement_delay_curr", _d),
    ]


def build(force=False):
    """Compile the C restatement (gcc, seconds).  Building the checker is not using it."""
    src = [os.path.join(_HERE, f) for f in ("ekf_oracle.c", "ekf_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src)
    if _SAN:
        return _LIB
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libekf_oracle.so"], check=True)
    return _LIB


def build_structured(force=False):
    csrc = os.path.join(_HERE, "..", "quadrotor_landing_amd", "csrc")
    src = [os.path.join(_HERE, "ekf_structured_cpu.cpp"), os.path.join(_HERE, "Makefile")] + \
          [os.path.join(csrc, h) for h in ("ekf_device.hpp", "ekf_quad.hpp", "ekf_fused.hpp", "ekf_packed.hpp", "ekf_split.hpp")]   # the Makefile rule's list
    stale = (not os.path.exists(_LIB_STRUCT)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_STRUCT) for s in src)
    if _SAN:
        return _LIB_STRUCT
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B", "libekf_oracle_structured.so"], check=True)
    return _LIB_STRUCT


_slib = None


def _structured_lib():
    global _slib
    if _slib is None:
        build_structured()
        _slib = C.CDLL(_LIB_STRUCT)
        _slib.orc_structured_run_batch.argtypes = [C.POINTER(OrcParams), C.c_int64, C.c_int64, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d),
                                                   C.POINTER(_d), C.POINTER(C.c_uint8), _i, _i, _i]
        _slib.orc_structured_run_batch.restype = C.c_int64
        _slib.orc_quad_run_batch.argtypes = [C.POINTER(OrcParams), C.c_int64, C.c_int64, C.POINTER(_d), C.POINTER(_d), C.POINTER(_d),
                                             C.POINTER(_d), C.POINTER(C.c_uint8), _i]
        _slib.orc_quad_run_batch.restype = C.c_int64
    return _slib


def quad_run_batch(p, x, P, u, z=None, mask=None, dtype="f64"):
    """The engine's four-lanes-per-filter arithmetic (quadrotor_landing_amd/csrc/ekf_quad.hpp) on an emulated quad:
    same contract as run_batch.  No-GPU algebra check of the quad kernels; never part of the product."""
    L = _structured_lib()
    n = p.num_states
    x = np.array(x, dtype=np.float64, order="C").reshape(-1, 16)
    B = x.shape[0]
    P = np.array(P, dtype=np.float64, order="C").reshape(B, n * n)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, B, 6)
    T = u.shape[0]
    zp = mp = None
    if mask is not None:
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(T, B, 7)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(T, B)
        zp = _p(z); mp = mask.ctypes.data_as(C.POINTER(C.c_uint8))
    L.orc_quad_run_batch(C.byref(p), B, T, _p(x), _p(P), _p(u), zp, mp, 0 if dtype == "f32" else 1)
    return x, P.reshape(B, n, n)


def structured_quat_exp(v, dtype="f64"):
    """The engine's quaternion_exp (ekf_device.hpp compiled for the host) for rotation vectors v [n, 3] -> [n, 4] (x, y, z, w)."""
    L = _structured_lib()
    v = np.ascontiguousarray(v, dtype=np.float64).reshape(-1, 3)
    q = np.empty((v.shape[0], 4))
    L.orc_structured_quat_exp.argtypes = [C.POINTER(_d), C.POINTER(_d), C.c_int64, _i]
    L.orc_structured_quat_exp.restype = None
    L.orc_structured_quat_exp(_p(v), _p(q), v.shape[0], 0 if dtype == "f32" else 1)
    return q


def structured_run_batch(p, x, P, u, z=None, mask=None, dtype="f64", levels=True, n_threads=0):
    """The engine's own per-filter arithmetic (quadrotor_landing_amd/csrc/ekf_device.hpp) compiled for the CPU:
    same contract as run_batch.  Second CPU baseline and no-GPU algebra check; never part of the product.
    levels: True = levelled predict + sequential update, False = in-place predict, "fused" = the fused tick (ekf_fused.hpp),
    "packed" = the register-block predict of the multirate replay loop (ekf_packed.hpp) + sequential update,
    "split" = predict and batch-form correction on the covariance split between LDS rows and registers (ekf_split.hpp, fp64 multirate replay)."""
    _slib = _structured_lib()
    n = p.num_states
    x = np.array(x, dtype=np.float64, order="C").reshape(-1, 16)
    B = x.shape[0]
    P = np.array(P, dtype=np.float64, order="C").reshape(B, n * n)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, B, 6)
    T = u.shape[0]
    zp = mp = None
    if mask is not None:
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(T, B, 7)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(T, B)
        zp = _p(z); mp = mask.ctypes.data_as(C.POINTER(C.c_uint8))
    _slib.orc_structured_run_batch(C.byref(p), B, T, _p(x), _p(P), _p(u), zp, mp, 0 if dtype == "f32" else 1, {"fused": 2, "packed": 3, "split": 4}.get(levels, int(bool(levels))), int(n_threads))
    return x, P.reshape(B, n, n)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB)
    pd = C.POINTER(_d)
    pp = C.POINTER(OrcParams)
    L.orc_params_default.argtypes = [pp]
    L.orc_initialize_params.argtypes = [pp]
    for name, args in (("orc_quaternion_exp", [pd, pd]), ("orc_quaternion_log", [pd, pd]),
                       ("orc_quaternion_norm", [pd]), ("orc_skew_symm", [pd, pd]),
                       ("orc_quat_mul", [pd, pd, pd]), ("orc_quat_to_rot", [pd, pd])):
        getattr(L, name).argtypes = args
        getattr(L, name).restype = None
    L.orc_angle_axis_to_rot.argtypes = [_d, pd, pd]
    L.orc_prediction_step.argtypes = [pp, pd, pd, pd, pd, pd, pd]
    L.orc_correction_step.argtypes = [pp, pd, pd, pd, pd, pd, pd, pd, pd]
    L.orc_seed_pose.argtypes = [pp, pd, pd, pd, pd]
    L.orc_corner_gate.argtypes = [pp, pd, pd]
    L.orc_corner_gate.restype = _i
    pf = C.POINTER(OrcFilter)
    L.orc_filter_init.argtypes = [pf, pp]
    L.orc_filter_free.argtypes = [pf]
    L.orc_filter_initialize_state.argtypes = [pf, _i]
    L.orc_filter_update.argtypes = [pf, _d]
    L.orc_run_batch.argtypes = [pp, C.c_int64, C.c_int64, pd, pd, pd, pd, C.POINTER(C.c_uint8), pd, _i]
    L.orc_run_batch.restype = C.c_int64
    L.orc_max_threads.restype = _i
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.POINTER(_d))


def _arr(v, n=None):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(-1))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


def default_params():
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    return p


_VEC_FIELDS = {"Q_a", "Q_w", "Q_ab", "Q_wb", "R_r", "R_ang", "ab_static", "wb_static", "r_v_cv", "q_vc",
               "camera_K", "tag_widths", "tag_positions", "g"}


def make_params(**kw):
    """Defaults (EKF.cpp ctor + node cov_init), overridden by keyword, then initialize_params."""
    p = default_params()
    for k, v in kw.items():
        if k in _VEC_FIELDS:
            a = _arr(v)
            dst = getattr(p, k)
            for i in range(a.size):
                dst[i] = a[i]
        else:
            cur = getattr(p, k)
            setattr(p, k, type(cur)(v))
    lib().orc_initialize_params(C.byref(p))
    return p


def quaternion_exp(v):
    o = np.zeros(4); lib().orc_quaternion_exp(_p(_arr(v, 3)), _p(o)); return o


def quaternion_log(q):
    o = np.zeros(3); lib().orc_quaternion_log(_p(_arr(q, 4)), _p(o)); return o


def quaternion_norm(q):
    o = _arr(q, 4).copy(); lib().orc_quaternion_norm(_p(o)); return o


def skew_symm(v):
    o = np.zeros(9); lib().orc_skew_symm(_p(_arr(v, 3)), _p(o)); return o.reshape(3, 3)


def quat_to_rot(q):
    o = np.zeros(9); lib().orc_quat_to_rot(_p(_arr(q, 4)), _p(o)); return o.reshape(3, 3)


def quat_mul(a, b):
    o = np.zeros(4); lib().orc_quat_mul(_p(_arr(a, 4)), _p(_arr(b, 4)), _p(o)); return o


def prediction_step(p, x, P, u):
    n = p.num_states
    xo = np.zeros(16); Po = np.zeros(n * n); acc = np.zeros(3)
    lib().orc_prediction_step(C.byref(p), _p(_arr(x, 16)), _p(_arr(P, n * n)), _p(_arr(u, 6)), _p(xo), _p(Po), _p(acc))
    return xo, Po.reshape(n, n), acc


def correction_step(p, x, P, r_c_tc, q_ct):
    n = p.num_states
    xo = np.zeros(16); Po = np.zeros(n * n); ro = np.zeros(3); qo = np.zeros(4)
    lib().orc_correction_step(C.byref(p), _p(_arr(x, 16)), _p(_arr(P, n * n)), _p(_arr(r_c_tc, 3)), _p(_arr(q_ct, 4)),
                              _p(xo), _p(Po), _p(ro), _p(qo))
    return xo, Po.reshape(n, n), ro, qo


def seed_pose(p, r_c_tc, q_ct):
    r = np.zeros(3); q = np.zeros(4)
    lib().orc_seed_pose(C.byref(p), _p(_arr(r_c_tc, 3)), _p(_arr(q_ct, 4)), _p(r), _p(q))
    return r, q


def corner_gate(p, r_c_tc, q_ct):
    return int(lib().orc_corner_gate(C.byref(p), _p(_arr(r_c_tc, 3)), _p(_arr(q_ct, 4))))


def run_batch(p, x, P, u, z=None, mask=None, per_filter_params=None, n_threads=0):
    """x [B,16], P [B,n,n] (copied), u [T,B,6], z [T,B,7], mask [T,B] -> (x, P) after T ticks."""
    n = p.num_states
    x = np.array(x, dtype=np.float64, order="C").reshape(-1, 16)
    B = x.shape[0]
    P = np.array(P, dtype=np.float64, order="C").reshape(B, n * n)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, B, 6)
    T = u.shape[0]
    zp = mp = fp = None
    if mask is not None:
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(T, B, 7)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(T, B)
        zp = _p(z); mp = mask.ctypes.data_as(C.POINTER(C.c_uint8))
    if per_filter_params is not None:
        per_filter_params = np.ascontiguousarray(per_filter_params, dtype=np.float64).reshape(B, 24)
        fp = _p(per_filter_params)
    lib().orc_run_batch(C.byref(p), B, T, _p(x), _p(P), _p(u), zp, mp, fp, int(n_threads))
    return x, P.reshape(B, n, n)


def max_threads():
    return int(lib().orc_max_threads())


class Filter:
    """Thin handle on `orc_filter` (the AoS, one-filter reference object)."""

    def __init__(self, p):
        self.f = OrcFilter()
        lib().orc_filter_init(C.byref(self.f), C.byref(p))

    def __del__(self):
        try:
            lib().orc_filter_free(C.byref(self.f))
        except Exception:
            pass

    def set_imu(self, accel, gyro):  # NODE.cpp:144-151
        for i in range(3):
            self.f.IMU_accel[i] = float(accel[i]); self.f.IMU_ang_vel[i] = float(gyro[i])

    def set_apriltag(self, pos, q_xyzw, stamp):  # NODE.cpp:153-176
        for i in range(3):
            self.f.apriltag_pos[i] = float(pos[i])
        for i in range(4):
            self.f.apriltag_orien[i] = float(q_xyzw[i])
        self.f.apriltag_time = float(stamp)
        self.f.measurement_ready = 1
        if not self.f.state_initialized:
            lib().orc_filter_initialize_state(C.byref(self.f), 0)

    def filter_update(self, t):
        lib().orc_filter_update(C.byref(self.f), float(t))

    def x(self):
        f = self.f
        return np.array(list(f.r_nom) + list(f.v_nom) + list(f.q_nom) + list(f.ab_nom) + list(f.wb_nom))

    def P(self):
        n = self.f.p.num_states
        return np.array(self.f.cov_pert[: n * n]).reshape(n, n)
