"""Parameter handling: the reference node's ROS-parameter loading without ROS.

`load_yaml` reads the reference's EKF parameter files
(quad_state_estimation/config/relative_pose_EKF_{rotors,hardware}.yaml) with
the key names, defaults and conversions of the node constructor
(quad_state_estimation/src/relative_pose_EKF_node.cpp:11-136).
"""
import ctypes as C

import numpy as np

from ._lib import QLE_MAX_TAGS, QleDerived, QleParams, check, lib

_VEC = {"Q_a": 3, "Q_w": 3, "Q_ab": 3, "Q_wb": 3, "R_r": 3, "R_ang": 3, "ab_static": 3, "wb_static": 3,
        "r_v_cv": 3, "q_vc": 4, "camera_K": 9, "tag_widths": QLE_MAX_TAGS, "tag_positions": 3 * QLE_MAX_TAGS, "g": 3}

# YAML key (NODE.cpp line) -> qle_params field
_YAML_SCALARS = {
    "update_freq": "update_freq",                                   # :31
    "measurement_freq": "measurement_freq",                         # :32
    "measurement_delay": "measurement_delay",                       # :33
    "measurement_delay_max": "measurement_delay_max",               # :34
    "dyn_measurement_delay_offset": "dyn_measurement_delay_offset", # :35
    "limit_measurement_freq": "limit_measurement_freq",             # :36
    "est_bias": "est_bias",                                         # :60
    "corner_margin_enbl": "corner_margin_enbl",                     # :61
    "direct_orien_method": "direct_orien_method",                   # :62
    "multirate_ekf": "multirate_ekf",                               # :63
    "dynamic_meas_delay": "dynamic_meas_delay",                     # :64
    "r_cov_init": "r_cov_init", "v_cov_init": "v_cov_init", "ang_cov_init": "ang_cov_init",  # :89-91
    "ab_cov_init": "ab_cov_init", "wb_cov_init": "wb_cov_init",     # :92-93
    "camera_width": "camera_width", "camera_height": "camera_height",  # :112-113
    "n_tags": "n_tags", "tag_in_view_margin": "tag_in_view_margin",    # :119-120
}
_YAML_VECTORS = {
    "Q_a_diag": "Q_a", "Q_w_diag": "Q_w", "Q_ab_diag": "Q_ab", "Q_wb_diag": "Q_wb",  # :71-79
    "R_r_diag": "R_r", "R_ang_diag": "R_ang",                                        # :84-87
    "accel_bias_static": "ab_static", "gyro_bias_static": "wb_static",               # :98-101
    "r_v_cv": "r_v_cv", "q_vc": "q_vc",                                              # :106-109 (q_vc is x,y,z,w)
    "camera_K": "camera_K",                                                          # :115-117 (row-major)
    "tag_widths": "tag_widths", "tag_positions": "tag_positions",                    # :125-136
}


def default_params():
    """Constructor defaults of the reference (relative_pose_EKF.cpp:8-85) + node cov_init defaults."""
    p = QleParams()
    check(lib().qle_params_default(C.byref(p)))
    return p


def set_fields(p, **kw):
    for k, v in kw.items():
        if k in _VEC:
            a = np.asarray(v, dtype=np.float64).reshape(-1)
            if a.size > _VEC[k]:
                raise ValueError(f"{k}: {a.size} values > capacity {_VEC[k]}")
            dst = getattr(p, k)
            for i in range(a.size):
                dst[i] = float(a[i])
        elif hasattr(p, k):
            cur = getattr(p, k)
            setattr(p, k, int(v) if isinstance(cur, int) else float(v))
        else:
            raise AttributeError(f"qle_params has no field {k!r}")
    return p


def make_params(**kw):
    return set_fields(default_params(), **kw)


def derive(p):
    """initialize_params() (relative_pose_EKF.cpp:87-125): host-only, needs no GPU."""
    d = QleDerived()
    check(lib().qle_params_derive(C.byref(p), C.byref(d)))
    return d


def load_yaml(path, base=None):
    """Read one of the reference's EKF YAML files into qle_params.

    Missing scalar keys keep the node's defaults (NODE.cpp `node.param`); the
    vector keys the node reads with `getParam` (no default, NODE.cpp:71-136)
    are required, as a missing one makes the reference dereference an empty
    vector.  camera_width/height accept 640 or 640.0 (hardware file).
    """
    import yaml

    with open(path) as fh:
        y = yaml.safe_load(fh)
    p = base if base is not None else default_params()
    missing = [k for k in _YAML_VECTORS if k not in y]
    if missing:
        raise KeyError(f"{path}: required parameter(s) missing: {missing}")
    kw = {}
    for yk, f in _YAML_SCALARS.items():
        if yk in y:
            v = y[yk]
            kw[f] = int(round(float(v))) if f in ("camera_width", "camera_height", "n_tags") else (int(bool(v)) if isinstance(v, bool) else v)
    for yk, f in _YAML_VECTORS.items():
        kw[f] = y[yk]
    n_tags = int(kw.get("n_tags", p.n_tags))
    if n_tags > QLE_MAX_TAGS:
        raise ValueError(f"n_tags={n_tags} exceeds QLE_MAX_TAGS={QLE_MAX_TAGS}")
    if len(kw["tag_widths"]) < n_tags or len(kw["tag_positions"]) < 3 * n_tags:
        raise ValueError("tag_widths/tag_positions shorter than n_tags")
    kw["tag_widths"] = list(kw["tag_widths"])[:n_tags]
    kw["tag_positions"] = list(kw["tag_positions"])[:3 * n_tags]
    return set_fields(p, **kw)
