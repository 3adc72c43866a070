"""CPU tests: the oracle against the reference-produced golden vectors
(tests/golden/, made by tests/golden/make_golden.py from the reference's own
Python twin), against the independent numpy restatement, and against analytic
known answers."""
import json
import math
import os

import numpy as np
import pytest

import oracle
from oracle import ekf_np

PSETS = ["pydefault", "rotors400", "hardware", "nobias"]


def load_param_sets(golden_dir):
    return json.load(open(os.path.join(golden_dir, "param_sets.json")))


def orc_params_for(ps, sets, **over):
    s = sets[ps]
    q = s["Q"]
    kw = dict(update_freq=s["update_freq"], est_bias=int(s["est_bias"]), direct_orien_method=1,
              Q_a=q[0:3], Q_w=q[3:6], R_r=s["R"][0:3], R_ang=s["R"][3:6], r_v_cv=s["r_v_cv"], q_vc=s["q_vc"])
    if s["est_bias"]:
        kw.update(Q_ab=q[6:9], Q_wb=q[9:12])
    kw.update(over)
    return oracle.make_params(**kw)


def qclose(a, b, tol):
    a = np.asarray(a); b = np.asarray(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


# ------------------------------------------------------------------ helpers
def test_helper_kats(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    for c in kat["skew_symm"]:
        np.testing.assert_array_equal(oracle.skew_symm(c["in"]), np.array(c["out"]))
    for c in kat["quaternion_norm"]:
        np.testing.assert_allclose(oracle.quaternion_norm(c["in"]), c["out"], rtol=0, atol=1e-15)
    for c in kat["quaternion_exp"]:
        # C++ normalises inside quaternion_exp (QH.cpp:30), the Python twin does not: rounding-level only
        np.testing.assert_allclose(oracle.quaternion_exp(c["in"]), c["out"], rtol=0, atol=1e-15)
        np.testing.assert_allclose(ekf_np.quaternion_exp(c["in"]), c["out"], rtol=0, atol=1e-15)
    for c in kat["quaternion_log"]:
        np.testing.assert_allclose(oracle.quaternion_log(c["in"]), c["out"], rtol=1e-14, atol=1e-16)
        np.testing.assert_allclose(ekf_np.quaternion_log(c["in"]), c["out"], rtol=1e-14, atol=1e-16)


def test_survey_appendix_c_literals():
    """Numbers captured in SURVEY.md Appendix C (reference Python twin)."""
    np.testing.assert_allclose(oracle.quaternion_exp([0.1, 0.2, 0.3]),
                               [0.04970884332485948, 0.09941768664971896, 0.14912652997457843, 0.9825509821552589], atol=1e-15)
    q = np.array([0.1, 0.2, 0.3, 0.9]); q /= np.linalg.norm(q)
    np.testing.assert_allclose(oracle.quaternion_log(q), [0.21060240739016323, 0.42120481478032645, 0.6318072221704897], atol=1e-15)
    np.testing.assert_array_equal(oracle.quaternion_norm([0, 0, 0.6, -0.8]), [-0.0, -0.0, -0.6, 0.8])
    np.testing.assert_array_equal(oracle.quaternion_norm([0, 0, 0.8, -0.6]), [0, 0, 0.8, -0.6])
    p = oracle.make_params(direct_orien_method=1)
    x = [0.3, -0.2, 2.0, 0.1, 0.0, -0.05, 0, 0, 0, 1] + [0.0] * 6
    P0 = np.diag([0.1] * 3 + [0.1] * 3 + [0.15] * 3 + [0.5] * 3 + [0.1] * 3)
    xc, Pc, acc = oracle.prediction_step(p, x, P0, [0.2, -0.1, 9.9, 0.05, -0.02, 0.3])
    np.testing.assert_allclose(xc[:10], [0.301, -0.2, 1.9995, 0.10200000000000001, -0.001, -0.04900000000000001,
                                         0.0002499999032291779, -9.999996129167116e-05, 0.0014999994193750673,
                                         0.9999988387502248], rtol=0, atol=2e-16)
    assert abs(np.linalg.norm(Pc) - 0.9554539418248377) < 1e-15
    assert abs(np.trace(Pc) - 2.8698167999999997) < 1e-14
    assert abs(Pc.sum() - 2.839957049839766) < 1e-14
    q_ct = np.array([0.7, -0.71, 0.02, 0.01]); q_ct /= np.linalg.norm(q_ct)
    xh, Ph, _, _ = oracle.correction_step(p, xc, Pc, [0.21, 0.33, 1.93], q_ct)
    np.testing.assert_allclose(xh, [0.2517276634781874, 0.17191972616426188, 2.0064899738410977, 0.09916861501951421,
                                    0.0038253092905989727, -0.04887168396119881, -0.0057461154909256565,
                                    -0.01176145238791947, -0.01680690870427726, 0.999773063357029, 0, 0, 0,
                                    7.993998189018213e-05, 0.00015491368551940671, 0.00024323407625875424], rtol=0, atol=5e-15)
    assert abs(np.linalg.norm(Ph) - 0.9025469084945956) < 1e-14
    assert abs(np.trace(Ph) - 2.1818996943402444) < 1e-14


# ----------------------------------------------------- golden step vectors
@pytest.mark.parametrize("ps", PSETS)
def test_predict_golden(golden_dir, ps):
    sets = load_param_sets(golden_dir)
    d = np.load(os.path.join(golden_dir, "predict_cases.npz"))
    p = orc_params_for(ps, sets)
    pn = ekf_np.Params.from_orc(p)
    for i in range(d[f"{ps}__x"].shape[0]):
        x, P, u = d[f"{ps}__x"][i], d[f"{ps}__P"][i], d[f"{ps}__u"][i]
        for impl in (lambda: oracle.prediction_step(p, x, P, u), lambda: ekf_np.prediction_step(pn, x, P, u)):
            xc, Pc, acc = impl()
            np.testing.assert_allclose(xc, d[f"{ps}__x_check"][i], rtol=1e-13, atol=1e-14)
            np.testing.assert_allclose(Pc, d[f"{ps}__P_check"][i], rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(acc, d[f"{ps}__accel"][i], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("ps", PSETS)
def test_update_golden(golden_dir, ps):
    sets = load_param_sets(golden_dir)
    d = np.load(os.path.join(golden_dir, "update_cases.npz"))
    p = orc_params_for(ps, sets)
    pn = ekf_np.Params.from_orc(p)
    for i in range(d[f"{ps}__x"].shape[0]):
        x, P, z = d[f"{ps}__x"][i], d[f"{ps}__P"][i], d[f"{ps}__z"][i]
        for impl in (lambda: oracle.correction_step(p, x, P, z[:3], z[3:]), lambda: ekf_np.correction_step(pn, x, P, z[:3], z[3:])):
            xh, Ph = impl()[:2]
            np.testing.assert_allclose(xh, d[f"{ps}__x_hat"][i], rtol=1e-11, atol=1e-12)
            np.testing.assert_allclose(Ph, d[f"{ps}__P_hat"][i], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("ps", PSETS)
def test_sequence_golden(golden_dir, ps):
    """Single-rate tick loop (EKF.cpp:238-249, 265-290) over hundreds of ticks."""
    sets = load_param_sets(golden_dir)
    d = np.load(os.path.join(golden_dir, "sequence_cases.npz"))
    p = orc_params_for(ps, sets)
    n = p.num_states
    U, Z, M = d[f"{ps}__u"], d[f"{ps}__z"], d[f"{ps}__mask"]
    x = d[f"{ps}__x_init"].copy(); P = d[f"{ps}__P_init"].copy()
    full = dict(zip(d[f"{ps}__P_full_ticks"].tolist(), d[f"{ps}__P_full"]))
    for t in range(U.shape[0]):
        x, P, _ = oracle.prediction_step(p, x, P, U[t])
        if M[t]:
            x, P = oracle.correction_step(p, x, P, Z[t, :3], Z[t, 3:])[:2]
        np.testing.assert_allclose(x, d[f"{ps}__x_seq"][t], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(np.diag(P), d[f"{ps}__P_diag_seq"][t], rtol=1e-9, atol=1e-13)
        if t in full:
            np.testing.assert_allclose(P, full[t], rtol=1e-8, atol=1e-12)
    # batch runner reproduces the same final state
    xb, Pb = oracle.run_batch(p, d[f"{ps}__x_init"][None], d[f"{ps}__P_init"][None], U[:, None, :], Z[:, None, :], M[:, None])
    np.testing.assert_allclose(xb[0], x, rtol=0, atol=0)
    np.testing.assert_allclose(Pb[0], P, rtol=0, atol=0)
    assert n == P.shape[0]


# --------------------------------------- C++-only branches: C vs numpy only
def rand_case(rng, n):
    x = np.zeros(16)
    x[0:3] = rng.uniform([-1, -1, 1], [1, 1, 4]); x[3:6] = rng.normal(size=3) * 0.5
    q = rng.normal(size=4); q /= np.linalg.norm(q); x[6:10] = q if q[3] > 0 else -q
    if n == 15:
        x[10:13] = rng.normal(size=3) * 0.1; x[13:16] = rng.normal(size=3) * 0.01
    A = rng.normal(size=(n, n)) * 0.1
    P = A @ A.T + np.diag(rng.uniform(0.01, 0.2, size=n))
    return x, 0.5 * (P + P.T)


@pytest.mark.parametrize("direct", [0, 1])
@pytest.mark.parametrize("est_bias", [0, 1])
def test_c_vs_numpy_all_branches(direct, est_bias):
    rng = np.random.default_rng(7 + 2 * direct + est_bias)
    p = oracle.make_params(direct_orien_method=direct, est_bias=est_bias, ab_static=[0.2, -0.09, -0.03],
                           wb_static=[-0.02, -0.01, 0.0], r_v_cv=[0.06036412, -0.00145196, -0.04439579],
                           q_vc=[-0.7035177, 0.7106742, 0.0014521, -0.0017207], update_freq=400.0)
    pn = ekf_np.Params.from_orc(p)
    n = p.num_states
    for i in range(40):
        x, P = rand_case(rng, n)
        u = np.append(rng.normal(size=3) * 1.5 + [0, 0, 9.8], rng.normal(size=3) * 0.4)
        a = oracle.prediction_step(p, x, P, u); b = ekf_np.prediction_step(pn, x, P, u)
        for s, t in zip(a, b):
            np.testing.assert_allclose(s, t, rtol=1e-12, atol=1e-14)
        # measurement near the state; includes cases where the delta_q flip fires (i % 7 == 0)
        ang = 3.0 if i % 7 == 0 else 0.8
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        dq = np.append(ax * math.sin(ang / 2), math.cos(ang / 2))
        q_true = ekf_np.qmul(x[6:10], dq)
        r_true = x[0:3] + rng.normal(size=3) * 0.1
        q_ct = ekf_np.qmul(ekf_np.qconj(pn.q_vc), ekf_np.qconj(q_true))
        r_c = pn.C_vc.T @ (-ekf_np.rot(q_true).T @ r_true - pn.r_v_cv)
        a = oracle.correction_step(p, x, P, r_c, q_ct); b = ekf_np.correction_step(pn, x, P, r_c, q_ct)
        assert qclose(a[0][6:10], b[0][6:10], 1e-11)
        np.testing.assert_allclose(np.delete(a[0], range(6, 10)), np.delete(b[0], range(6, 10)), rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(a[2], b[2], rtol=1e-12, atol=1e-13)
        # observation model inverts: r_obs is the true position, q_obs the true attitude (direct method)
        if direct:
            np.testing.assert_allclose(a[2], r_true, atol=1e-12)
        assert qclose(a[3], q_true, 1e-12)
        np.testing.assert_allclose(oracle.seed_pose(p, r_c, q_ct)[0], ekf_np.seed_pose(pn, r_c, q_ct)[0], atol=1e-13)


def test_corner_gate_c_vs_numpy():
    rng = np.random.default_rng(11)
    hw_w = [0.08382] + [0.16764] * 4 + [0.33528] * 4 + [0.16764] * 4
    hw_p = [0, 0, 0, 0, 0.1571625, 0, 0.1571625, 0, 0, 0, -0.1571625, 0, -0.1571625, 0, 0, -0.244475, 0.244475, 0,
            0.244475, 0.244475, 0, 0.244475, -0.244475, 0, -0.244475, -0.244475, 0, 0, 0.314325, 0, 0.314325, 0, 0,
            0, -0.314325, 0, -0.314325, 0, 0]
    sets = [oracle.make_params(),
            oracle.make_params(n_tags=13, tag_in_view_margin=0.0, tag_widths=hw_w, tag_positions=hw_p,
                               camera_K=[437.3412312213781, 0, 328.5442810236917, 0, 438.0867474272743, 239.2536470406629, 0, 0, 1],
                               camera_width=640, camera_height=480)]
    seen = set()
    for p in sets:
        pn = ekf_np.Params.from_orc(p)
        for _ in range(300):
            r = rng.uniform([-2.5, -1.5, 0.3], [2.5, 1.5, 4.0])
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax); ang = rng.uniform(0, 0.6)
            q = np.append(ax * math.sin(ang / 2), math.cos(ang / 2))
            a = oracle.corner_gate(p, r, q); b = ekf_np.corner_gate(pn, r, q)
            assert a == b
            seen.add(a)
    assert seen == {0, 1}


# ------------------------------------------------------------ analytic KATs
def test_hover_and_constant_yaw_rate():
    p = oracle.make_params()
    x = np.zeros(16); x[2] = 2.0; x[3:6] = [0.1, -0.2, 0.05]; x[9] = 1.0
    P = np.diag(list(p.cov_init))
    # hover: a_meas = C^T(-g), w = 0 -> v constant, q constant
    xc, Pc, acc = oracle.prediction_step(p, x, P, [0, 0, 9.8, 0, 0, 0])
    np.testing.assert_allclose(acc, 0, atol=1e-15)
    np.testing.assert_allclose(xc[3:6], x[3:6], atol=1e-16)
    np.testing.assert_array_equal(xc[6:10], x[6:10])
    np.testing.assert_allclose(xc[0:3], x[0:3] + 0.01 * x[3:6], atol=1e-16)
    # closed-form covariance from a diagonal P
    dT = p.dT_nom
    np.testing.assert_allclose(np.diag(Pc)[0:3], 0.1 + dT * dT * 0.1, rtol=1e-14)
    np.testing.assert_allclose(np.diag(Pc)[6:9], 0.15 + dT * dT * 0.1 + 0.0005, rtol=1e-14)
    np.testing.assert_allclose(np.diag(Pc)[9:12], 0.5 + 5e-5, rtol=1e-14)
    # constant yaw rate: q_N = exp(N dT w)
    w = np.array([0, 0, 0.7]); N = 200
    for _ in range(N):
        x, P, _ = oracle.prediction_step(p, x, P, [0, 0, 9.8, *w])
    assert qclose(x[6:10], oracle.quaternion_exp(N * dT * w), 1e-13)


def test_exp_log_roundtrip_and_zero_innovation():
    rng = np.random.default_rng(3)
    for _ in range(50):
        v = rng.normal(size=3) * rng.choice([1e-6, 0.1, 1.0])
        np.testing.assert_allclose(oracle.quaternion_log(oracle.quaternion_exp(v)), v, rtol=1e-12, atol=1e-18)
    for direct in (0, 1):
        p = oracle.make_params(direct_orien_method=direct)
        pn = ekf_np.Params.from_orc(p)
        x, P = rand_case(rng, 15)
        q_ct = ekf_np.qmul(ekf_np.qconj(pn.q_vc), ekf_np.qconj(x[6:10]))
        r_c = pn.C_vc.T @ (-ekf_np.rot(x[6:10]).T @ x[0:3] - pn.r_v_cv)
        xh, Ph, _, _ = oracle.correction_step(p, x, P, r_c, q_ct)
        np.testing.assert_allclose(xh, x, atol=1e-12)
        ev = np.linalg.eigvalsh(0.5 * (Ph + Ph.T))
        assert ev.min() > 0
        assert np.linalg.eigvalsh(P - 0.5 * (Ph + Ph.T)).min() > -1e-12  # P_hat <= P_check
        assert np.abs(Ph - Ph.T).max() < 1e-14


def test_initialize_params_derived():
    p = oracle.make_params(update_freq=400.0, measurement_freq=30.0, measurement_delay=0.030)
    assert p.upd_per_meas == 14 and p.num_states == 15 and abs(p.dT_nom - 0.0025) < 1e-18
    assert p.measurement_step_delay == 12
    p = oracle.make_params(update_freq=100.0, measurement_freq=15.0, est_bias=0)
    assert p.upd_per_meas == 7 and p.num_states == 9
    np.testing.assert_allclose(np.array(list(p.C_vc)).reshape(3, 3), [[0, -1, 0], [-1, 0, 0], [0, 0, -1]], atol=1e-15)


def test_filter_update_single_rate_matches_manual_loop(golden_dir):
    """orc_filter_update (EKF.cpp:127-303) with rate limiting reproduces the golden sequence."""
    sets = load_param_sets(golden_dir)
    d = np.load(os.path.join(golden_dir, "sequence_cases.npz"))
    ps = "pydefault"
    p = orc_params_for(ps, sets, measurement_freq=15.0, limit_measurement_freq=1, corner_margin_enbl=0)
    assert p.upd_per_meas == 7
    U, Z, M = d[f"{ps}__u"], d[f"{ps}__z"], d[f"{ps}__mask"]
    f = oracle.Filter(p)
    # seed directly from the golden initial state
    x0 = d[f"{ps}__x_init"]
    for i in range(3):
        f.f.r_nom[i] = x0[i]; f.f.v_nom[i] = x0[3 + i]; f.f.ab_nom[i] = x0[10 + i]; f.f.wb_nom[i] = x0[13 + i]
    for i in range(4):
        f.f.q_nom[i] = x0[6 + i]
    f.f.state_initialized = 1
    for t in range(U.shape[0]):
        f.set_imu(U[t, :3], U[t, 3:])
        if M[t]:
            f.set_apriltag(Z[t, :3], Z[t, 3:], t * 0.01)
        f.filter_update(t * 0.01)
        assert f.f.performed_correction == int(M[t])
        np.testing.assert_allclose(f.x(), d[f"{ps}__x_seq"][t], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(np.diag(f.P()), d[f"{ps}__P_diag_seq"][-1], rtol=1e-9)


TWIN_FU = dict(update_freq=100.0, measurement_freq=10.0, measurement_delay=0.050, dynamic_meas_delay=0,
               limit_measurement_freq=1, corner_margin_enbl=1, direct_orien_method=1, est_bias=1)


@pytest.mark.parametrize("mode", ["multirate", "singlerate"])
def test_full_filter_update_against_reference_twin(golden_dir, mode):
    """The oracle's filter object (EKF.cpp:127-303: rate limit, corner gate, multirate replay with the
    fixed step delay, counters) against the sequence the reference's own Python twin produced."""
    d = np.load(os.path.join(golden_dir, "filter_update_cases.npz"))
    U, Z, NEW = d[f"{mode}__u"], d[f"{mode}__z"], d[f"{mode}__new"]
    p = oracle.make_params(multirate_ekf=int(mode == "multirate"), **TWIN_FU)
    assert p.upd_per_meas == 10 and p.measurement_step_delay == 5
    f = oracle.Filter(p)
    n_corr = 0
    for t in range(U.shape[0]):
        f.set_imu(U[t, :3], U[t, 3:])
        if NEW[t]:
            f.set_apriltag(Z[t, :3], Z[t, 3:], 0.01 * t)
        f.filter_update(0.01 * t)
        assert f.f.upds_since_correction == d[f"{mode}__upds"][t], t
        assert f.f.measurement_ready == d[f"{mode}__ready"][t], t
        if mode == "multirate":
            assert f.f.hist_len == d[f"{mode}__hist_len"][t], t
        n_corr += f.f.performed_correction
        x = f.x()
        xr = d[f"{mode}__x_seq"][t]
        assert qclose(x[6:10], xr[6:10], 1e-9)
        np.testing.assert_allclose(np.delete(x, range(6, 10)), np.delete(xr, range(6, 10)), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(np.diag(f.P()), d[f"{mode}__P_diag_seq"][t], rtol=1e-8)
    assert n_corr >= 10
    np.testing.assert_allclose(f.P(), d[f"{mode}__P_final"], rtol=1e-7, atol=1e-11)


# ---------------- C++-only branches pinned through the twin by identity (tests/golden/make_golden_branches.py)
def hw_branch_kwargs(d, **over):
    """HW.yaml's values for the branches the `hw__*` / `gate__*` / `static__*` fixtures exercise (see make_golden_branches.py)."""
    kw = dict(update_freq=100.0, measurement_freq=100.0, limit_measurement_freq=0, corner_margin_enbl=1, direct_orien_method=1, est_bias=1,
              ab_static=d["static__ab_static"], wb_static=d["static__wb_static"],
              n_tags=13, tag_in_view_margin=float(d["gate__margin"][0]), tag_widths=d["gate__tag_widths"], tag_positions=d["gate__tag_positions"],
              camera_K=d["gate__camera_K"], camera_width=int(d["gate__camera_size"][0]), camera_height=int(d["gate__camera_size"][1]))
    kw.update(over)
    return kw


def test_static_bias_predict_against_twin_identity(golden_dir):
    """EKF.cpp:357-358 (static-bias subtraction) against twin outputs obtained with the statics inside the bias states."""
    sets = load_param_sets(golden_dir)
    d = np.load(os.path.join(golden_dir, "branch_cases.npz"))
    p = orc_params_for("hardware", sets, ab_static=d["static__ab_static"], wb_static=d["static__wb_static"])
    assert np.abs(np.array(list(p.ab_static))).max() > 0.1
    for i in range(d["static__x"].shape[0]):
        xo, Po, acc = oracle.prediction_step(p, d["static__x"][i], d["static__P"][i], d["static__u"][i])
        xr = d["static__x_check"][i]
        assert qclose(xo[6:10], xr[6:10], 1e-13)
        np.testing.assert_allclose(np.delete(xo, range(6, 10)), np.delete(xr, range(6, 10)), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(Po, d["static__P_check"][i], rtol=1e-11, atol=1e-15)
        np.testing.assert_allclose(acc, d["static__accel"][i], rtol=1e-12, atol=1e-13)
    # and the statics matter: without them the same call is far off
    p0 = orc_params_for("hardware", sets)
    assert np.abs(oracle.prediction_step(p0, d["static__x"][3], d["static__P"][3], d["static__u"][3])[0][3:6] - d["static__x_check"][3][3:6]).max() > 1e-4


def test_multi_tag_gate_against_composed_twin_decisions(golden_dir):
    """EKF.cpp:160-181: the loop over the bundle == `or` over the twin's single-tag decisions; each tag on its own == the twin's."""
    d = np.load(os.path.join(golden_dir, "branch_cases.npz"))
    kw = hw_branch_kwargs(d)
    p13 = oracle.make_params(**kw)
    Z, per_tag, composed = d["gate__z"], d["gate__per_tag"], d["gate__composed"]
    assert 0 < composed.sum() < len(composed) and (composed & (1 - per_tag[:, 0])).sum() > 20   # decided by a tag other than the first
    got = np.array([oracle.corner_gate(p13, z[:3], z[3:]) for z in Z], np.uint8)
    np.testing.assert_array_equal(got, composed)
    w, pos = d["gate__tag_widths"], d["gate__tag_positions"].reshape(13, 3)
    for k in range(13):
        p1 = oracle.make_params(**dict(kw, n_tags=1, tag_widths=[w[k]], tag_positions=list(pos[k])))
        got = np.array([oracle.corner_gate(p1, z[:3], z[3:]) for z in Z], np.uint8)
        np.testing.assert_array_equal(got, per_tag[:, k])
    pn = ekf_np.Params.from_orc(p13)
    np.testing.assert_array_equal(np.array([ekf_np.corner_gate(pn, z[:3], z[3:]) for z in Z], np.uint8), composed)


@pytest.mark.parametrize("mode", ["hw_multirate", "hw_singlerate"])
@pytest.mark.parametrize("delay", ["dynamic_exact", "dynamic_clamped", "fixed"])
def test_hardware_like_filter_update_against_twin(golden_dir, mode, delay):
    """The oracle's filter_update with everything HW.yaml switches on -- static biases, the 13-tag bundle, a correction on every
    tick, a 15-tick measurement delay taken from the stamps (EKF.cpp:199-200) -- against the twin's trajectory (identities in
    tests/golden/make_golden_branches.py).  dynamic_exact: age + offset = 150 ms; dynamic_clamped: age 300 ms, delay_max 150 ms."""
    d = np.load(os.path.join(golden_dir, "branch_cases.npz"))
    sets = load_param_sets(golden_dir)
    s = sets["hardware"]
    q = s["Q"]
    dyn = dict(fixed=dict(dynamic_meas_delay=0, measurement_delay=0.150),
               dynamic_exact=dict(dynamic_meas_delay=1, measurement_delay=0.010, measurement_delay_max=0.350, dyn_measurement_delay_offset=0.085),
               dynamic_clamped=dict(dynamic_meas_delay=1, measurement_delay=0.010, measurement_delay_max=0.150, dyn_measurement_delay_offset=0.085))[delay]
    age = float(d[f"{mode}__age_clamped"][0] if delay == "dynamic_clamped" else d[f"{mode}__age_exact"][0])
    p = oracle.make_params(**hw_branch_kwargs(d, multirate_ekf=int(mode == "hw_multirate"), Q_a=q[0:3], Q_w=q[3:6], Q_ab=q[6:9], Q_wb=q[9:12],
                                              R_r=s["R"][0:3], R_ang=s["R"][3:6], r_v_cv=s["r_v_cv"], q_vc=s["q_vc"], **dyn))
    assert p.upd_per_meas == 1
    U, Z, NEW = d[f"{mode}__u"], d[f"{mode}__z"], d[f"{mode}__new"]
    f = oracle.Filter(p)
    z0 = d[f"{mode}__z0"]
    f.set_apriltag(z0[:3], z0[3:], -1.0)      # seeds the filter (NODE.cpp:169-174)
    f.f.measurement_ready = 0
    for t in range(U.shape[0]):
        tc = 0.01 * t
        f.set_imu(U[t, :3], U[t, 3:])
        if NEW[t]:
            f.set_apriltag(Z[t, :3], Z[t, 3:], tc - age)
        f.filter_update(tc)
        assert f.f.performed_correction == d[f"{mode}__perf"][t], t
        assert f.f.upds_since_correction == d[f"{mode}__upds"][t], t
        assert f.f.measurement_ready == 0
        if mode == "hw_multirate":
            assert f.f.hist_len == d[f"{mode}__hist_len"][t], t
            if f.f.performed_correction and delay != "fixed":
                assert abs(f.f.measurement_delay_curr - 0.150) < 1e-12
        x = f.x()
        xr = d[f"{mode}__x_seq"][t]
        assert qclose(x[6:10], xr[6:10], 1e-9), t
        np.testing.assert_allclose(np.delete(x, range(6, 10)), np.delete(xr, range(6, 10)), rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(np.diag(f.P()), d[f"{mode}__P_diag_seq"][t], rtol=1e-8)
    np.testing.assert_allclose(f.P(), d[f"{mode}__P_full"][-1], rtol=1e-7, atol=1e-11)


# ---------------- the engine's block-structured algebra, compiled for the CPU, against the dense oracle
def engine_cpu_run(variant, p, x, P, U, Z, M, dtype):
    """The engine's per-filter arithmetic compiled for the host: one lane per filter (ekf_device.hpp: levelled or
    in-place predict, sequential decorrelated update; ekf_fused.hpp: the fused tick of k_step) or four lanes per filter on an
    emulated quad (ekf_quad.hpp)."""
    if variant == "quad":
        return oracle.quad_run_batch(p, x, P, U, Z, M, dtype=dtype)
    return oracle.structured_run_batch(p, x, P, U, Z, M, dtype=dtype, levels=variant if variant in ("fused", "packed", "split") else (variant == "levels"))


def test_engine_quaternion_exp_large_angles_halving_and_doubling():
    """quaternion_exp of the engine (ekf_device.hpp, host build): the half-angle series on |phi|/2 <= pi/4 and, beyond, halving until the
    series applies and doubling back (a wave-uniform count on the device; no branch to the library's sincos).  Against the closed form
    (QH.cpp:9-33 incl. the final quaternion_norm) for rotation vectors from 1e-12 rad to 40 rad."""
    rng = np.random.default_rng(7)
    n = 4000
    ax = rng.normal(size=(n, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = np.concatenate([10 ** rng.uniform(-12, 0.15, n // 2), rng.uniform(1.4, 40.0, n - n // 2)])
    v = ax * ang[:, None]
    for dtype, tol in (("f64", 2e-14), ("f32", 5e-6)):   # fp32: |v|^2 carries 1e-7 relative, i.e. 1.5e-6 rad at 30 rad, before any series
        vr = v.astype(np.float32).astype(np.float64) if dtype == "f32" else v     # the closed form on the inputs the engine sees
        a = np.linalg.norm(vr, axis=1)
        ref = np.concatenate([vr / a[:, None] * np.sin(a / 2)[:, None], np.cos(a / 2)[:, None]], axis=1)
        ref[ref[:, 3] < -0.75] *= -1                   # quaternion_norm's flip, QH.cpp:61-73
        q = oracle.structured_quat_exp(v, dtype)
        err = np.abs(q - ref).max(axis=1)
        small = ang < 1.5
        assert err[small].max() < (4e-16 if dtype == "f64" else 2e-7), (dtype, err[small].max())
        assert err.max() < tol, (dtype, err.max(), ang[err.argmax()])
        assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < (1e-15 if dtype == "f64" else 3e-7)


def test_packed_covariance_order_is_a_bijection_with_level_structure():
    """sidx (ekf_device.hpp) restated: 120 distinct words, block-rows in storage order r, v, th, ab, wb quad-wise,
    memory quad 3m + l = the m-th quad of quad-lane l, and lane l holds column l of every off-diagonal block."""
    base = [0, 14, 25, 33, 38]

    def sidx(i, k):
        i, k = min(i, k), max(i, k)
        b, c, ii, kk = i // 3, k // 3, i % 3, k % 3
        if b == c:
            lane, pos = (kk, base[b]) if ii == kk else ({(0, 1): 1, (1, 2): 2, (0, 2): 0}[(ii, kk)], base[b] + 1)
        else:
            lane, pos = kk, base[b] + 2 + 3 * (c - b - 1) + ii
        return 4 * (3 * (pos // 4) + lane) + pos % 4, lane, pos

    words = {}
    for i in range(15):
        for k in range(i, 15):
            w, lane, pos = sidx(i, k)
            assert w not in words
            words[w] = (i, k)
            assert (w // 4) % 3 == lane and 4 * ((w // 4) // 3) + w % 4 == pos
            if i // 3 != k // 3:
                assert lane == k % 3
    assert sorted(words) == list(range(120))
    rows = [words[w][0] // 3 for w in range(120)]
    for q in range(29):   # block-rows never decrease from one quad-row triple to the next
        assert min(rows[4 * q: 4 * q + 4]) <= min(rows[4 * q + 4: 4 * q + 8]) or q % 3 != 2


@pytest.mark.parametrize("variant", ["levels", "inplace", "quad", "fused", "packed", "split"])
@pytest.mark.parametrize("direct", [0, 1])
@pytest.mark.parametrize("est_bias", [0, 1])
def test_structured_cpu_build_of_engine_arithmetic_matches_dense_oracle(direct, est_bias, variant):
    """oracle/ekf_structured_cpu.cpp compiles quadrotor_landing_amd/csrc/ekf_device.hpp and ekf_quad.hpp for the host:
    the same F = L3 L2 L1 congruences / levelled predict and decorrelated sequential update (one lane per filter) and
    the column-distributed predict / L D L^T update (four lanes per filter, on an emulated quad) the HIP kernels run,
    checked here (no GPU) against the dense reference-shaped restatement over a 60-tick sequence."""
    rng = np.random.default_rng(40 + 2 * direct + est_bias)
    p = oracle.make_params(direct_orien_method=direct, est_bias=est_bias, update_freq=400.0, ab_static=[0.2, -0.09, -0.03],
                           wb_static=[-0.02, -0.01, 0.0], q_vc=[-0.7035177, 0.7106742, 0.0014521, -0.0017207],
                           r_v_cv=[0.06036412, -0.00145196, -0.04439579])
    pn = ekf_np.Params.from_orc(p)
    n = p.num_states
    B, T = 24, 60
    x = np.stack([rand_case(rng, n)[0] for _ in range(B)]); P = np.stack([rand_case(rng, n)[1] for _ in range(B)])
    U = rng.normal(size=(T, B, 6)) * np.array([1.5, 1.5, 1.5, 0.4, 0.4, 0.4]) + np.array([0, 0, 9.8, 0, 0, 0])
    M = np.zeros((T, B), np.uint8); M[4::5] = 1
    Z = np.zeros((T, B, 7)); Z[..., 6] = 1
    xr, Pr = x.copy(), P.copy()
    # build measurements near the dense oracle's own trajectory so the innovations stay moderate
    for t in range(T):
        xr, Pr = oracle.run_batch(p, xr, Pr, U[t][None])
        if M[t].any():
            for i in range(B):
                ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
                dq = np.append(ax * math.sin(0.15), math.cos(0.15))
                q_true = ekf_np.qmul(xr[i, 6:10], dq); r_true = xr[i, 0:3] + rng.normal(size=3) * 0.05
                Z[t, i, 3:] = ekf_np.qmul(ekf_np.qconj(pn.q_vc), ekf_np.qconj(q_true))
                Z[t, i, :3] = pn.C_vc.T @ (-ekf_np.rot(q_true).T @ r_true - pn.r_v_cv)
            xr2 = xr.copy(); Pr2 = Pr.copy()
            for i in range(B):
                xr2[i], Pr2[i] = oracle.correction_step(p, xr[i], Pr[i], Z[t, i, :3], Z[t, i, 3:])[:2]
            xr, Pr = xr2, Pr2
    xd, Pd = oracle.run_batch(p, x, P, U, Z, M)
    np.testing.assert_allclose(xd, xr, rtol=0, atol=0)  # the incremental construction above is the same computation
    for dtype, tol in (("f64", 1e-10), ("f32", 2e-3)):
        xs, Ps = engine_cpu_run(variant, p, x, P, U, Z, M, dtype)
        dqv = np.minimum(np.abs(xs[:, 6:10] - xd[:, 6:10]).max(1), np.abs(xs[:, 6:10] + xd[:, 6:10]).max(1)).max()
        assert dqv < tol, (dtype, dqv)
        keep = [i for i in range(16) if not 6 <= i < 10]
        np.testing.assert_allclose(xs[:, keep], xd[:, keep], rtol=tol, atol=tol)
        sc = np.sqrt(np.einsum("bii->bi", Pd)[:, :, None] * np.einsum("bii->bi", Pd)[:, None, :])
        assert (np.abs(Ps - Pd) / sc).max() < tol * 10, dtype


@pytest.mark.parametrize("ps", PSETS)
@pytest.mark.parametrize("variant", ["levels", "inplace", "quad", "fused", "packed", "split"])
def test_structured_cpu_build_against_reference_twin_sequences(golden_dir, ps, variant):
    """The engine's arithmetic (CPU build) against the trajectories the reference's Python twin produced."""
    sets = load_param_sets(golden_dir)
    d = np.load(os.path.join(golden_dir, "sequence_cases.npz"))
    p = orc_params_for(ps, sets)
    U, Z, M = d[f"{ps}__u"], d[f"{ps}__z"], d[f"{ps}__mask"]
    T = U.shape[0]
    for dtype, tol in (("f64", 1e-9), ("f32", 3e-3)):
        x, P = engine_cpu_run(variant, p, d[f"{ps}__x_init"][None], d[f"{ps}__P_init"][None], U[:, None, :], Z[:, None, :], M[:, None], dtype)
        xr = d[f"{ps}__x_seq"][T - 1]
        assert qclose(x[0, 6:10], xr[6:10], tol * 10)
        np.testing.assert_allclose(np.delete(x[0], range(6, 10)), np.delete(xr, range(6, 10)), rtol=tol, atol=tol)
        np.testing.assert_allclose(np.diag(P[0]), d[f"{ps}__P_diag_seq"][T - 1], rtol=tol * 10)


# ---------------- the host restatement of the synthetic source (oracle/synth_np.py)
def test_synth_restatement_counter_rng_and_sharding():
    """The counter-based generator of the synthetic source, restated on the host: splitmix64 finaliser against its published test
    vector, uniforms inside (0, 1) with the right moments, and the property the sharding rests on -- a filter's data depends on its
    GLOBAL index only, so any split of a population over shards reproduces the single-process sequence bit for bit."""
    import oracle.synth_np as sn
    # splitmix64 (Steele, Lea, Flood 2014): mix(gamma) is the generator's first output from state 0
    assert int(sn.mix64(np.uint64(0x9E3779B97F4A7C15))) == 0xE220A8397B1DCDAF
    gi = np.arange(20000)
    uu = sn.rng_uniform(7, gi, 3, 5)
    assert uu.min() > 0.0 and uu.max() < 1.0 and abs(uu.mean() - 0.5) < 0.01 and abs(uu.std() - 12 ** -0.5) < 0.01
    nn = sn.rng_normal(7, gi, 3, 5)
    assert abs(nn.mean()) < 0.03 and abs(nn.std() - 1.0) < 0.03
    assert not np.array_equal(sn.rng_uniform(7, gi, 3, 5), sn.rng_uniform(7, gi, 4, 5))
    p = oracle.make_params(update_freq=400.0, est_bias=1)
    thm = np.zeros(12, np.uint8); thm[3::4] = 1
    whole = sn.generate(p, 96, thm, seed=11, filter_offset=1000, perturb_filter_params=True, meas_delay_ticks=2)
    part = sn.generate(p, 32, thm, seed=11, filter_offset=1032, perturb_filter_params=True, meas_delay_ticks=2)
    for k in ("u", "z"):
        np.testing.assert_array_equal(part[k], whole[k][:, 32:64])
    for k in ("z0", "pfp", "truth", "truth_bias"):
        np.testing.assert_array_equal(part[k], whole[k][32:64])
    assert np.allclose(np.linalg.norm(whole["z"][..., 3:], axis=-1), 1.0) and np.allclose(np.linalg.norm(whole["truth"][:, 3:], axis=1), 1.0)
    assert abs(np.linalg.norm(whole["u"][..., :3], axis=-1).mean() - 9.81) < 0.5   # specific force ~ g


# ------------------------------------------------------------------ conventions against a third-party implementation
def test_rotation_conventions_against_scipy(golden_dir):
    """The fixtures under tests/golden were generated by the reference's Python twin with textbook stand-ins for the five
    `tf.transformations` functions it calls (make_golden.py: `tf` is not installed here), and the C++ text relies on Eigen's
    quaternion conventions (EKF.cpp:359,367,431,448; QH.cpp:9-58), which the oracle restates.  Neither library is in the image;
    scipy.spatial.transform.Rotation is, and documents the same conventions (scalar-last x, y, z, w; Hamilton product; active
    rotation matrices).  Both the stand-ins and the oracle's helpers are held against it here."""
    import importlib.util
    from scipy.spatial.transform import Rotation as R
    spec = importlib.util.spec_from_file_location("_make_golden_for_test", os.path.join(golden_dir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    tft = mg.tf_standins()
    ekf_np_mod = ekf_np
    rng = np.random.default_rng(20260704)
    tol = 4e-15
    for k in range(300):
        q1 = rng.normal(size=4); q1 /= np.linalg.norm(q1)
        q0 = rng.normal(size=4); q0 /= np.linalg.norm(q0)
        r1, r0 = R.from_quat(q1), R.from_quat(q0)
        # tf stand-ins
        assert np.abs(tft.quaternion_matrix(q1)[:3, :3] - r1.as_matrix()).max() < tol
        assert np.abs(tft.quaternion_matrix(3.0 * q1)[:3, :3] - r1.as_matrix()).max() < tol      # tf divides by |q|^2
        assert qclose(tft.quaternion_multiply(q1, q0), (r1 * r0).as_quat(), tol)
        assert qclose(tft.quaternion_conjugate(q1), r1.inv().as_quat(), tol)
        axis = rng.normal(size=3)
        ang = rng.uniform(-math.pi, math.pi) * (1e-9 if k % 10 == 0 else 1.0)
        rv = ang * axis / np.linalg.norm(axis)
        assert np.abs(tft.rotation_matrix(ang, axis)[:3, :3] - R.from_rotvec(rv).as_matrix()).max() < tol
        assert qclose(tft.quaternion_about_axis(ang, axis), R.from_rotvec(rv).as_quat(), tol)
        # the oracle's helpers (Eigen's toRotationMatrix / operator* / the reference's exp and log)
        assert np.abs(oracle.quat_to_rot(q1) - r1.as_matrix()).max() < tol
        assert qclose(oracle.quat_mul(q1, q0), (r1 * r0).as_quat(), tol)
        assert qclose(oracle.quaternion_exp(rv), R.from_rotvec(rv).as_quat(), tol)
        qw = q1 if q1[3] > 0 else -q1
        assert np.abs(oracle.quaternion_log(qw) - R.from_quat(qw).as_rotvec()).max() < 2e-14 * max(1.0, 1.0 / max(qw[3], 1e-3))
        # the numpy restatement says the same
        assert np.abs(ekf_np_mod.rot(q1) - r1.as_matrix()).max() < tol
        assert qclose(ekf_np_mod.qmul(q1, q0), (r1 * r0).as_quat(), tol)
        assert np.abs(ekf_np_mod.rodrigues(-ang, axis / np.linalg.norm(axis)) - R.from_rotvec(-rv).as_matrix()).max() < tol   # EKF.cpp:394
    # the observation model's frame chain (EKF.cpp:431-438): q_tv_obs = conj(q_vc * q_ct), r = -R(q_tv_obs) (C_vc r_c + r_v_cv)
    q_vc = np.array([0.70710678, -0.70710678, 0.0, 0.0]); q_vc /= np.linalg.norm(q_vc)
    q_ct = rng.normal(size=4); q_ct /= np.linalg.norm(q_ct)
    r_c = rng.normal(size=3); r_v_cv = np.array([0.1, 0.0, -0.05])
    q_tv = (R.from_quat(q_vc) * R.from_quat(q_ct)).inv()
    r_expect = -q_tv.apply(R.from_quat(q_vc).apply(r_c) + r_v_cv)
    p = oracle.make_params(q_vc=list(q_vc), r_v_cv=list(r_v_cv))
    r_seed, q_seed = oracle.seed_pose(p, r_c, q_ct)[:2]
    assert np.abs(np.asarray(r_seed) - r_expect).max() < 1e-14
    assert qclose(q_seed, q_tv.as_quat(), 1e-14)
