"""ctypes binding of the C-ABI in include/qle_ekf.h (libqle_ekf.so).

The shared library is built in-tree by `make -C quadrotor_landing_amd/csrc`
(or __graft_entry__.build()).  There is no fallback: if the library is missing
or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QLE_LIB selects an alternative in-tree build of the same engine (kernel tuning A/B runs)
LIB_PATH = os.environ.get("QLE_LIB") or os.path.join(_HERE, "libqle_ekf.so")

QLE_OK = 0
QLE_ERR_INVALID, QLE_ERR_HIP, QLE_ERR_NOMEM, QLE_ERR_STATE, QLE_ERR_NO_DEVICE = -1, -2, -3, -4, -5
QLE_F32, QLE_F64 = 0, 1
QLE_MAX_TAGS = 16

_d, _i32, _i64, _u64 = C.c_double, C.c_int32, C.c_int64, C.c_uint64


class QleParams(C.Structure):
    """`struct qle_params` == the reference's public members (relative_pose_EKF.hpp:66-133)."""
    _fields_ = [
        ("update_freq", _d), ("measurement_freq", _d), ("measurement_delay", _d),
        ("measurement_delay_max", _d), ("dyn_measurement_delay_offset", _d),
        ("est_bias", _i32), ("limit_measurement_freq", _i32), ("corner_margin_enbl", _i32),
        ("direct_orien_method", _i32), ("multirate_ekf", _i32), ("dynamic_meas_delay", _i32),
        ("r_cov_init", _d), ("v_cov_init", _d), ("ang_cov_init", _d), ("ab_cov_init", _d), ("wb_cov_init", _d),
        ("Q_a", _d * 3), ("Q_w", _d * 3), ("Q_ab", _d * 3), ("Q_wb", _d * 3),
        ("R_r", _d * 3), ("R_ang", _d * 3),
        ("ab_static", _d * 3), ("wb_static", _d * 3),
        ("r_v_cv", _d * 3), ("q_vc", _d * 4),
        ("camera_K", _d * 9), ("camera_width", _i32), ("camera_height", _i32),
        ("n_tags", _i32), ("_pad0", _i32),
        ("tag_in_view_margin", _d),
        ("tag_widths", _d * QLE_MAX_TAGS), ("tag_positions", _d * (3 * QLE_MAX_TAGS)),
        ("small_ang_tol", _d), ("g", _d * 3),
    ]


class QleDerived(C.Structure):
    """`struct qle_derived` == what initialize_params() computes (relative_pose_EKF.cpp:87-125)."""
    _fields_ = [
        ("dT_nom", _d), ("upd_per_meas", _i32), ("num_states", _i32), ("measurement_step_delay", _i32), ("_pad0", _i32),
        ("Q", _d * 12), ("R", _d * 6), ("cov_init", _d * 15), ("q_vc", _d * 4), ("C_vc", _d * 9),
    ]


class QleSynthCfg(C.Structure):
    _fields_ = [
        ("seed", _u64), ("filter_offset", _i64),
        ("ab_true_sigma", _d), ("wb_true_sigma", _d), ("meas_noise_scale", _d), ("imu_noise_scale", _d),
        ("perturb_filter_params", _i32), ("meas_delay_ticks", _i32), ("view_scale", _d),
    ]


class QleNodeReport(C.Structure):
    """`struct qle_node_report`: what the node publishes after a tick (NODE.cpp:192-281), one per filter."""
    _fields_ = [("pose", _d * 7), ("pose_cov", _d * 36), ("vel", _d * 3), ("accel", _d * 3), ("bias", _d * 6), ("obs", _d * 7),
                ("measurement_delay_curr", _d), ("upds_since_correction", _i32), ("performed_correction", C.c_uint8),
                ("measurement_consumed", C.c_uint8), ("state_initialized", C.c_uint8), ("reserved", C.c_uint8)]


class QlePolicy(C.Structure):
    """`struct qle_policy`: how a handle launches its ticks."""
    _fields_ = [("state_policy", _i32), ("refresh_period", _i32), ("split_k64", _i32), ("block", _i32), ("coop_ticks", _i32),
                ("ring_slots", _i32), ("state_bytes", _i64), ("ring_bytes", _i64), ("record_words", _i32), ("reserved", _i32)]


class QleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"qle error {code}: {msg}")
        self.code = code


# every symbol include/qle_ekf.h declares: name -> (restype, argtypes)
_pd = C.POINTER(_d)
_pu8 = C.POINTER(C.c_uint8)
_vp = C.c_void_p
SYMBOLS = {
    "qle_last_error": (C.c_char_p, []),
    "qle_version": (C.c_char_p, []),
    "qle_device_count": (C.c_int, [C.POINTER(_i32)]),
    "qle_params_default": (C.c_int, [C.POINTER(QleParams)]),
    "qle_params_derive": (C.c_int, [C.POINTER(QleParams), C.POINTER(QleDerived)]),
    "qle_create": (C.c_int, [C.POINTER(_vp), _i64, _i32, _i32, C.POINTER(QleParams)]),
    "qle_destroy": (C.c_int, [_vp]),
    "qle_set_params": (C.c_int, [_vp, C.POINTER(QleParams)]),
    "qle_set_filter_params": (C.c_int, [_vp, _pd]),
    "qle_get_filter_params": (C.c_int, [_vp, _pd]),
    "qle_batch_size": (_i64, [_vp]),
    "qle_dtype": (_i32, [_vp]),
    "qle_num_states": (_i32, [_vp]),
    "qle_set_state": (C.c_int, [_vp, _pd, _pd]),
    "qle_get_state": (C.c_int, [_vp, _pd, _pd]),
    "qle_initialize_state": (C.c_int, [_vp, _pd, _i32]),
    "qle_initialize_state_masked": (C.c_int, [_vp, _pd, _pu8, _i32]),
    "qle_get_state_initialized": (C.c_int, [_vp, _pu8]),
    "qle_enable_aux": (C.c_int, [_vp, _i32]),
    "qle_get_aux": (C.c_int, [_vp, _pd, _pd]),
    "qle_predict": (C.c_int, [_vp, _pd]),
    "qle_update": (C.c_int, [_vp, _pd, _pu8]),
    "qle_step": (C.c_int, [_vp, _pd, _pd, _pu8]),
    "qle_enable_gating": (C.c_int, [_vp, _i32]),
    "qle_filter_update": (C.c_int, [_vp, _pd, _pd, _pu8]),
    "qle_filter_update_stamped": (C.c_int, [_vp, _pd, _pd, _pu8, _d, _pd]),
    "qle_get_measurement_delay": (C.c_int, [_vp, _pd]),
    "qle_set_uniform_measurement_age": (C.c_int, [_vp, _d]),
    "qle_get_tick_flags": (C.c_int, [_vp, _pu8, _pu8, C.POINTER(_i32)]),
    "qle_inputs_create": (C.c_int, [_vp, _i64, _pu8, C.POINTER(_vp)]),
    "qle_inputs_destroy": (C.c_int, [_vp]),
    "qle_inputs_upload_tick": (C.c_int, [_vp, _i64, _pd, _pd, _pu8]),
    "qle_inputs_download_tick": (C.c_int, [_vp, _i64, _pd, _pd, _pu8]),
    "qle_run": (C.c_int, [_vp, _vp, _i64, _i64]),
    "qle_run_resident": (C.c_int, [_vp, _vp, _i64, _i64]),
    "qle_synth_cfg_default": (C.c_int, [C.POINTER(QleSynthCfg)]),
    "qle_synth_generate": (C.c_int, [_vp, _vp, C.POINTER(QleSynthCfg)]),
    "qle_synth_rmse": (C.c_int, [_vp, _vp, _pd]),
    "qle_synth_get_truth": (C.c_int, [_vp, _vp, _pd, _pd]),
    "qle_get_report": (C.c_int, [_vp, _pd, _pd, _pd, _pd]),
    "qle_get_node_report": (C.c_int, [_vp, C.POINTER(QleNodeReport)]),
    "qle_count_nonfinite": (C.c_int, [_vp, C.POINTER(_i64)]),
    "qle_synchronize": (C.c_int, [_vp]),
    "qle_timer_begin": (C.c_int, [_vp]),
    "qle_timer_end": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "qle_algorithmic_bytes": (_i64, [_vp, _i32]),
    "qle_get_policy": (C.c_int, [_vp, C.POINTER(QlePolicy)]),
}

_lib = None


def lib():
    """Load libqle_ekf.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `make -C quadrotor_landing_amd/csrc` "
                              "(hipcc, gfx950). There is no CPU fallback for the EKF engine.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != QLE_OK:
        raise QleError(rc, lib().qle_last_error().decode())
    return rc
