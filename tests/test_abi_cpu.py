"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every
symbol include/qle_ekf.h declares, the host-only parameter logic matches the
oracle, and the engine fails loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
import quadrotor_landing_amd as qla
from quadrotor_landing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "qle_ekf.h")


def header_functions():
    txt = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(qle_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 30
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/qle_ekf.h but not exported"
    # and the Python binding covers exactly the header
    assert sorted(_lib.SYMBOLS) == names


def test_struct_layouts_match_header_sizes():
    # sizes follow from the header's field list: all members are 8-byte aligned groups
    assert C.sizeof(_lib.QleParams) == 8 * 5 + 4 * 6 + 8 * 5 + 8 * (12 + 6 + 6 + 3 + 4 + 9) + 4 * 4 + 8 * (1 + 16 + 48 + 1 + 3)
    assert C.sizeof(_lib.QleDerived) == 8 + 16 + 8 * (12 + 6 + 15 + 4 + 9)
    assert C.sizeof(_lib.QleSynthCfg) == 8 * 6 + 8 + 8   # ... two int32, view_scale
    assert C.sizeof(_lib.QleNodeReport) == 8 * (7 + 36 + 3 + 3 + 6 + 7 + 1) + 4 + 4


def test_params_default_and_derive_match_oracle():
    for kw in (dict(), dict(update_freq=400.0, measurement_freq=30.0, measurement_delay=0.03),
               dict(est_bias=0, measurement_freq=15.0),
               dict(q_vc=[-0.7035177, 0.7106742, 0.0014521, -0.0017207], r_v_cv=[0.06, -0.0014, -0.044]),
               dict(q_vc=[0.0, 0.0, 0.6, -0.8])):  # flip branch of quaternion_norm
        p = qla.make_params(**kw); d = qla.derive(p)
        o = oracle.make_params(**kw)
        assert d.dT_nom == o.dT_nom and d.upd_per_meas == o.upd_per_meas
        assert d.num_states == o.num_states and d.measurement_step_delay == o.measurement_step_delay
        np.testing.assert_array_equal(list(d.Q), list(o.Q))
        np.testing.assert_array_equal(list(d.R), list(o.R))
        np.testing.assert_array_equal(list(d.cov_init), list(o.cov_init))
        np.testing.assert_allclose(list(d.q_vc), list(o.q_vc), atol=1e-16)
        np.testing.assert_allclose(list(d.C_vc), list(o.C_vc), atol=4e-16)
    # public defaults field by field (EKF.cpp:8-85)
    p = qla.default_params(); o = oracle.default_params()
    for name, _ in _lib.QleParams._fields_:
        if name.startswith("_") or name == "q_vc":  # the oracle normalises q_vc in place (EKF.cpp:57); compared above
            continue
        a, b = getattr(p, name), getattr(o, name)
        if hasattr(a, "__len__"):
            np.testing.assert_array_equal(list(a), list(b)[:len(a)], err_msg=name)
        else:
            assert a == b, name


def test_params_validation_errors():
    p = qla.default_params()
    p.update_freq = 0.0
    with pytest.raises(qla.QleError) as e:
        qla.derive(p)
    assert e.value.code == _lib.QLE_ERR_INVALID
    p = qla.default_params(); p.n_tags = 99
    with pytest.raises(qla.QleError):
        qla.derive(p)
    # small_ang_tol (EKF.hpp:131): the engine evaluates the exact series at every angle, equal to the reference's small-angle forms at
    # its 1e-10; a tolerance at which the reference's first-order forms would differ is refused, not silently ignored
    p = qla.default_params(); p.small_ang_tol = 1e-3
    with pytest.raises(qla.QleError) as e:
        qla.derive(p)
    assert e.value.code == _lib.QLE_ERR_INVALID and "small_ang_tol" in str(e.value)
    p = qla.default_params(); p.small_ang_tol = 1e-9
    qla.derive(p)


def test_yaml_loader_reads_reference_key_names(tmp_path):
    y = tmp_path / "ekf.yaml"
    y.write_text("""
update_freq: 100.0
measurement_freq: 100.0
measurement_delay: 0.150
Q_a_diag: [0.00025,0.00025,0.00025]
Q_w_diag: [0.00045,0.00045,0.00045]
Q_ab_diag: [7.0E-6,7.0E-6,7.0E-6]
Q_wb_diag: [4.4E-5,4.4E-5,4.4E-5]
R_r_diag: [0.0015,0.0015,0.006]
R_ang_diag: [0.0015,0.0015,0.04]
accel_bias_static: [0.20,-0.09,-0.03]
gyro_bias_static: [-0.02,-0.01,0.0]
camera_K: [437.3,0,328.5,0,438.0,239.2,0,0,1]
camera_width: 640.0
camera_height: 480.0
r_v_cv: [0.06036412,-0.00145196,-0.04439579]
q_vc: [-0.7035177, 0.7106742, 0.0014521, -0.0017207]
n_tags: 2
tag_in_view_margin: 0.00
tag_widths: [0.08382,0.16764]
tag_positions: [0,0,0, 0,0.1571625,0]
limit_measurement_freq: False
direct_orien_method: True
multirate_ekf: True
""")
    p = qla.load_yaml(str(y))
    assert p.camera_width == 640 and p.camera_height == 480 and p.n_tags == 2
    assert p.direct_orien_method == 1 and p.multirate_ekf == 1 and p.limit_measurement_freq == 0
    assert p.est_bias == 1  # node default kept (NODE.cpp:60)
    assert list(p.ab_static) == [0.20, -0.09, -0.03]
    assert list(p.tag_positions)[:6] == [0, 0, 0, 0, 0.1571625, 0]
    assert qla.derive(p).measurement_step_delay == 15
    y.write_text("update_freq: 100.0\n")
    with pytest.raises(KeyError):
        qla.load_yaml(str(y))


def test_no_cpu_fallback_without_gpu():
    n = C.c_int32(-1)
    rc = _lib.lib().qle_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(qla.QleError) as e:
        qla.BatchedRelativePoseEKF(8)
    assert e.value.code == _lib.QLE_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "quadrotor_landing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "__never__", f"{f} mentions the oracle"


def test_shipped_parameter_files_load():
    cfg = os.path.join(ROOT, "quadrotor_landing_amd", "config")
    p = qla.load_yaml(os.path.join(cfg, "ekf_sim_rotors.yaml"))
    d = qla.derive(p)
    assert (p.update_freq, p.measurement_freq, d.upd_per_meas, d.measurement_step_delay) == (100.0, 15.0, 7, 3)
    assert p.limit_measurement_freq == 1 and p.multirate_ekf == 1 and p.n_tags == 1 and p.camera_width == 752
    assert list(p.Q_a) == [5e-4] * 3 and list(p.R_ang) == [0.0015, 0.0015, 0.04]
    p = qla.load_yaml(os.path.join(cfg, "ekf_hardware.yaml"))
    d = qla.derive(p)
    assert (d.upd_per_meas, d.measurement_step_delay, p.n_tags, p.camera_width, p.camera_height) == (1, 15, 13, 640, 480)
    assert list(p.ab_static) == [0.20, -0.09, -0.03] and p.tag_in_view_margin == 0.0
    assert list(p.tag_positions)[36:39] == [-0.314325, 0.0, 0.0] and list(p.tag_widths)[:2] == [0.08382, 0.16764]
    assert abs(np.linalg.norm(list(d.q_vc)) - 1) < 1e-15


def test_cpp_driver_parses_config_and_fails_loudly_without_gpu():
    import subprocess
    exe = os.path.join(ROOT, "quadrotor_landing_amd", "ekf_driver")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "quadrotor_landing_amd", "csrc")], check=True)
    n = C.c_int32(-1)
    if _lib.lib().qle_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "--config", os.path.join(ROOT, "quadrotor_landing_amd", "config", "ekf_hardware.yaml"), "--batch", "8", "--ticks", "10"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    r = subprocess.run([exe, "--config", "/nonexistent.yaml"], capture_output=True, text=True)
    assert r.returncode == 2


def test_generated_device_code_passes_the_stale_exec_audit(tmp_path):
    """The backend must not place register copies at a control-flow join in front of the EXEC restore (the cause of the GPU memory access
    faults of rounds 3 and 4, profiles/r04_tuning.md section 1).  `make audit` compiles every translation unit to device assembly with the
    flags the library is built with (cached under csrc/build/asm) and runs profiles/r04_scripts/exec_join_audit.py over it; the script itself is
    checked here on a reduced copy of the faulting pattern and of its repaired form."""
    audit = os.path.join(ROOT, "profiles", "r04_scripts", "exec_join_audit.py")
    bad = tmp_path / "bad.s"
    bad.write_text("""_Z6kernelv:
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB0_2
\tv_add_f64 v[2:3], v[2:3], v[4:5]
.LBB0_2:
\tv_accvgpr_write_b32 a4, v252
\tv_accvgpr_write_b32 a5, v253
\ts_or_b64 exec, exec, s[0:1]
\tv_accvgpr_read_b32 v252, a4
\ts_endpgm
""")
    good = tmp_path / "good.s"
    good.write_text(bad.read_text().replace("\tv_accvgpr_write_b32 a4, v252\n\tv_accvgpr_write_b32 a5, v253\n\ts_or_b64 exec, exec, s[0:1]\n",
                                            "\ts_or_b64 exec, exec, s[0:1]\n\tv_accvgpr_write_b32 a4, v252\n\tv_accvgpr_write_b32 a5, v253\n"))
    rb = subprocess.run([sys.executable, audit, str(bad)], capture_output=True, text=True)
    rg = subprocess.run([sys.executable, audit, str(good)], capture_output=True, text=True)
    assert rb.returncode == 1 and "2 register copies under a stale EXEC" in rb.stdout, rb.stdout
    assert rg.returncode == 0 and rg.stdout.strip() == "", rg.stdout
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "quadrotor_landing_amd", "csrc"), "audit"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "audit: no register copy under a stale EXEC in 11 translation units" in r.stdout
