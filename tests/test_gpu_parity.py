"""GPU parity tests: the HIP engine (through the C-ABI) against the CPU oracle
on identical seeded inputs, against the committed golden fixtures (outputs of
the reference's Python twin), and -- at BASELINE's full sizes -- through
size-independent properties.

Tolerances (stated here, used below; `tests/tolerances.md` lists every one next to the deviation measured on an MI355X,
regenerated with QLE_TOL_RECORD -- none is more than ~10x its measurement):
  fp64 engine vs fp64 oracle, one step ("teacher-forced"): 1e-12 on the state, 5e-11 x sqrt(P_ii P_jj) on covariance
      entries, quaternion 1e-11 up to sign.  The engine exploits the block structure of F and an LDL^T update, the
      oracle multiplies dense matrices and inverts S by LU: same algebra, different rounding.
  fp64 free run, 1000 ticks: 1e-9.
  fp32 engine vs fp64 oracle, one PREDICT: 5e-7 (1 + |x|) on the state (measured 7e-8), 3e-6 x sqrt(P_ii P_jj) on covariance
      entries (3.4e-7); one UPDATE or fused tick: 5e-6 on the state (7e-7), 3e-4 x sqrt(P_ii P_jj) on covariance entries (3.3e-5:
      the correction divides by the pivots of S, cond(S) more digits go); free runs of 40-1000 ticks: 2e-5 ... 6e-5 (state), 5e-5
      ... 4e-4 (covariance), each about 10x what that run measures.
"""
import numpy as np
import pytest

import oracle
import quadrotor_landing_amd as qla
from util import (GOLDEN, assert_state_close, cov_dev, golden_kwargs, meas_near, note, oracle_predict_batch, oracle_update_batch,
                  quat_err, rand_imu, rand_states, state_dev)

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["default", "lanes-only", "coop-forced"])
def kernel_family(request, monkeypatch):
    """Two kernel families implement the single-rate tick: one lane per filter (ekf_kernels.hpp) and the workgroup-cooperative
    kernel (ekf_quad_kernels.hpp: one scalar wave + the covariance spread over quads of lanes; by default only for small fp64
    batches).  Every test runs with the default policy, with the cooperative kernel forced for every tick, and with it disabled,
    so both families are checked against the oracle in both dtypes (QLE_QUAD is read at handle creation)."""
    if request.param == "lanes-only":
        monkeypatch.setenv("QLE_QUAD", "0")
    elif request.param == "coop-forced":
        monkeypatch.setenv("QLE_QUAD", "3")
    else:
        monkeypatch.delenv("QLE_QUAD", raising=False)
    return request.param

F64 = dict(rtol=1e-12, atol=1e-14, qtol=1e-11, ptol=5e-11)
F32 = dict(rtol=5e-7, atol=5e-7, qtol=1e-6, ptol=3e-6)           # one predict: measured 7e-8 / 1.2e-7 / 3.4e-7
# one correction (stand-alone or inside the fused tick): fp64 20x the predict's, fp32 as measured (tests/tolerances.md)
F64U = dict(rtol=2e-11, atol=2e-13, qtol=2e-10, ptol=1e-9)
F32U = dict(rtol=5e-6, atol=5e-6, qtol=2e-6, ptol=3e-4)          # measured 7.3e-7 / 2.4e-7 / 3.3e-5 (fused tick; update alone 7e-6)
UPD = {"f64": F64U, "f32": F32U}


def free_run_close(xg, Pg, xr, Pr, tol, ftol=None, qtol=None):
    """Free runs: state within tol (1 + |x|), quaternion within qtol, covariance within ftol relative Frobenius norm."""
    ftol = ftol if ftol is not None else tol
    qtol = qtol if qtol is not None else tol
    assert note("quat", quat_err(xg[..., 6:10].reshape(-1, 4), xr[..., 6:10].reshape(-1, 4)), qtol) < qtol
    assert note("state", state_dev(xg, xr), tol) < tol
    relF = np.linalg.norm(Pg - Pr, axis=(-2, -1)) / np.linalg.norm(Pr, axis=(-2, -1))
    assert note("covF", relF.max(), ftol) < ftol, relF.max()

BRANCHES = [dict(direct_orien_method=d, est_bias=e) for d in (0, 1) for e in (0, 1)]
HW = dict(ab_static=[0.2, -0.09, -0.03], wb_static=[-0.02, -0.01, 0.0], r_v_cv=[0.06036412, -0.00145196, -0.04439579],
          q_vc=[-0.7035177, 0.7106742, 0.0014521, -0.0017207])


def both(**kw):
    return oracle.make_params(**kw), qla.make_params(**kw)


# ------------------------------------------------------------ single steps
@pytest.mark.parametrize("branch", BRANCHES)
@pytest.mark.parametrize("dtype,tol", [("f64", F64), ("f32", F32)])
def test_predict_teacher_forced(branch, dtype, tol):
    kw = dict(update_freq=400.0, **HW, **branch)
    po, pq = both(**kw)
    n = po.num_states
    rng = np.random.default_rng(100 + n + branch["direct_orien_method"])
    B = 333  # ragged: not a multiple of 64
    x, P = rand_states(rng, B, n)
    u = rand_imu(rng, B)
    u[0, 3:6] = x[0, 13:16] + np.array(kw["wb_static"])           # exactly zero rate: small-angle branch
    u[1, 3:6] = x[1, 13:16] + np.array(kw["wb_static"]) + 1e-9    # below small_ang_tol in dT*w
    u[2, 3:6] = [8.0, -6.0, 5.0]
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    xg, Pg, ag = ekf.prediction_step(x, P, u)
    xr, Pr, ar = oracle_predict_batch(po, x, P, u)
    assert_state_close(xg, Pg, xr, Pr, **tol)
    at = 1e-13 if dtype == "f64" else 3e-5       # |accel| ~ 10 m/s^2: a few fp32 ulp (measured 3.4e-6)
    note("accel", np.abs(ag - ar).max(), at)
    np.testing.assert_allclose(ag, ar, rtol=0, atol=at)
    assert np.abs(Pg - Pg.transpose(0, 2, 1)).max() == 0.0  # packed storage: exactly symmetric
    ekf.close()


@pytest.mark.parametrize("branch", BRANCHES)
@pytest.mark.parametrize("dtype,tol", [("f64", F64), ("f32", F32)])
def test_update_teacher_forced(branch, dtype, tol):
    kw = dict(update_freq=100.0, **HW, **branch)
    po, pq = both(**kw)
    n = po.num_states
    rng = np.random.default_rng(200 + n + branch["direct_orien_method"])
    B = 257
    x, P = rand_states(rng, B, n, cov_scale=0.3)
    z = meas_near(rng, po, x)
    z[5:40:7] = meas_near(rng, po, x[5:40:7], ang=3.0)   # large attitude innovation, delta_q flip region
    z[3, 3:] *= -1                                       # double cover of the measurement
    mask = (rng.uniform(size=B) < 0.7).astype(np.uint8)
    mask[:8] = 1
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    ekf.set_state(x, P)
    ekf.enable_aux(True)
    ekf.update(z, mask)
    xg, Pg = ekf.get_state()
    obs = ekf.get_aux()[1]
    xr, Pr, obr = oracle_update_batch(po, x, P, z, mask)
    # the update divides by the pivots of S: cond(S) more digits go than in a predict (UPD); on these states the stand-alone update
    # measures 6.8e-6 on the covariance in fp32 (the fused tick's 3.3e-5 comes from its predicted covariances)
    assert_state_close(xg, Pg, xr, Pr, **dict(UPD[dtype], ptol=UPD[dtype]["ptol"] if dtype == "f64" else 7e-5))
    m = mask.astype(bool)
    ot = 1e-12 if dtype == "f64" else 1e-5       # reported observation (measured 1.2e-6 / 1e-7 in fp32)
    note("obs", np.abs(obs[m, :3] - obr[m, :3]).max(), ot)
    np.testing.assert_allclose(obs[m, :3], obr[m, :3], rtol=0, atol=ot)
    assert note("obs_q", quat_err(obs[m, 3:], obr[m, 3:]), tol["qtol"]) < tol["qtol"]
    # masked-out filters are untouched bit for bit (fp64 storage round trip is exact)
    if dtype == "f64":
        np.testing.assert_array_equal(xg[~m], x[~m])
        np.testing.assert_array_equal(Pg[~m], P[~m])
    ekf.close()


@pytest.mark.parametrize("ps", ["pydefault", "rotors400", "hardware", "nobias"])
def test_golden_vectors_fp64(ps):
    """Engine against numbers produced by the reference's own Python twin."""
    kw = golden_kwargs(ps)
    pq = qla.make_params(**kw)
    d = np.load(f"{GOLDEN}/predict_cases.npz")
    x, P, u = d[f"{ps}__x"], d[f"{ps}__P"], d[f"{ps}__u"]
    ekf = qla.BatchedRelativePoseEKF(x.shape[0], "f64", params=pq)
    xg, Pg, ag = ekf.prediction_step(x, P, u)
    assert_state_close(xg, Pg, d[f"{ps}__x_check"], d[f"{ps}__P_check"], 1e-12, 1e-14, 1e-11)
    np.testing.assert_allclose(ag, d[f"{ps}__accel"], rtol=1e-12, atol=1e-12)
    d = np.load(f"{GOLDEN}/update_cases.npz")
    x, P, z = d[f"{ps}__x"], d[f"{ps}__P"], d[f"{ps}__z"]
    xg, Pg = ekf.correction_step(x, P, z[:, :3], z[:, 3:])
    assert_state_close(xg, Pg, d[f"{ps}__x_hat"], d[f"{ps}__P_hat"], 1e-10, 1e-12, 1e-10)
    ekf.close()


@pytest.mark.parametrize("ps", ["pydefault", "rotors400", "hardware", "nobias"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_golden_sequences(ps, dtype):
    """Hundreds of fused ticks against the reference twin's recorded trajectory."""
    kw = golden_kwargs(ps)
    pq = qla.make_params(**kw)
    d = np.load(f"{GOLDEN}/sequence_cases.npz")
    U, Z, M = d[f"{ps}__u"], d[f"{ps}__z"], d[f"{ps}__mask"]
    T = U.shape[0]
    B = 4  # the same sequence in every lane
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    ekf.set_state(np.repeat(d[f"{ps}__x_init"][None], B, 0), np.repeat(d[f"{ps}__P_init"][None], B, 0))
    seq = ekf.make_inputs(T, M)
    for t in range(T):
        seq.upload_tick(t, np.repeat(U[t][None], B, 0), np.repeat(Z[t][None], B, 0) if M[t] else None)
    full = dict(zip(d[f"{ps}__P_full_ticks"].tolist(), d[f"{ps}__P_full"]))
    xs = d[f"{ps}__x_seq"]
    rt, at = (1e-9, 1e-10) if dtype == "f64" else (3e-5, 1e-7)     # fp32 over 140-560 ticks: measured 3.2e-6 (state), 1e-7 (quaternion)
    t = 0
    for stop in sorted(full) + [T - 1]:
        ekf.run(seq, t, stop + 1 - t)
        t = stop + 1
        x, P = ekf.get_state()
        assert np.abs(x - x[0]).max() == 0.0  # all lanes identical
        assert note("quat", quat_err(x[:1, 6:10], xs[stop][None, 6:10]), at * 10) < at * 10
        note("state", state_dev(x[0], xs[stop]), rt)
        np.testing.assert_allclose(np.delete(x[0], range(6, 10)), np.delete(xs[stop], range(6, 10)), rtol=rt, atol=rt)
        pt = 1e-8 if dtype == "f64" else 2e-4                       # measured 1.8e-5
        note("Pdiag", np.abs(np.diag(P[0]) / d[f"{ps}__P_diag_seq"][stop] - 1).max(), pt)
        np.testing.assert_allclose(np.diag(P[0]), d[f"{ps}__P_diag_seq"][stop], rtol=pt)
        if stop in full:
            assert note("cov", cov_dev(P[0], full[stop]), pt) < pt
    ekf.close()


# ---------------------------------------------------------- fused / masked
@pytest.mark.parametrize("dtype,tol", [("f64", F64), ("f32", F32)])
def test_fused_step_equals_predict_then_update(dtype, tol):
    kw = dict(update_freq=400.0, direct_orien_method=1, **HW)
    po, pq = both(**kw)
    rng = np.random.default_rng(7)
    B = 1000
    x, P = rand_states(rng, B, 15, cov_scale=0.3)
    u = rand_imu(rng, B)
    xr, Pr, _ = oracle_predict_batch(po, x, P, u)
    z = meas_near(rng, po, xr)
    mask = (rng.uniform(size=B) < 0.5).astype(np.uint8)
    xr, Pr, _ = oracle_update_batch(po, xr, Pr, z, mask)
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    ekf.set_state(x, P)
    ekf.step(u, z, mask)
    xg, Pg = ekf.get_state()
    assert_state_close(xg, Pg, xr, Pr, **UPD[dtype])
    # predict-only tick through the same entry point
    ekf.set_state(x, P)
    ekf.step(u)
    xg, Pg = ekf.get_state()
    xr, Pr, _ = oracle_predict_batch(po, x, P, u)
    assert_state_close(xg, Pg, xr, Pr, **tol)
    ekf.close()


def test_per_filter_params_cfg5():
    kw = dict(update_freq=400.0, direct_orien_method=1)
    po, pq = both(**kw)
    rng = np.random.default_rng(9)
    B = 200
    x, P = rand_states(rng, B, 15, cov_scale=0.3)
    u = rand_imu(rng, B)
    pfp = np.zeros((B, 24))
    base_q = np.array(list(po.Q))
    pfp[:, 0:12] = base_q * 10 ** rng.uniform(-0.5, 0.5, size=(B, 4)).repeat(3, axis=1)
    pfp[:, 12:15] = rng.normal(size=(B, 3)) * 0.1
    pfp[:, 15:18] = rng.normal(size=(B, 3)) * 0.01
    pfp[:, 18:24] = np.array(list(po.R)) * rng.uniform(0.5, 2.0, size=(B, 6))
    xr, Pr = oracle.run_batch(po, x, P, u[None], per_filter_params=pfp)
    z = meas_near(rng, po, xr)
    xr2, Pr2 = oracle.run_batch(po, x, P, u[None], z[None], np.ones((1, B), np.uint8), per_filter_params=pfp)
    ekf = qla.BatchedRelativePoseEKF(B, "f64", params=pq)
    ekf.set_filter_params(pfp)
    ekf.set_state(x, P)
    ekf.step(u)
    xg, Pg = ekf.get_state()
    assert_state_close(xg, Pg, xr, Pr, 1e-12, 1e-14, 1e-11)
    ekf.set_state(x, P)
    ekf.step(u, z)
    xg, Pg = ekf.get_state()
    assert_state_close(xg, Pg, xr2, Pr2, 2e-11, 1e-13, 1e-10)
    rep = ekf.report()
    np.testing.assert_allclose(rep["bias"][:, :3], xg[:, 10:13] + pfp[:, 12:15], atol=1e-15)
    ekf.set_filter_params(None)
    ekf.set_state(x, P)
    ekf.step(u)
    xs, Ps = ekf.get_state()
    xr, Pr, _ = oracle_predict_batch(po, x, P, u)
    assert_state_close(xs, Ps, xr, Pr, 1e-12, 1e-14, 1e-11)
    ekf.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_randomized_configurations_short_runs_vs_oracle(dtype):
    """16 randomly drawn parameter sets -- both orientation methods, with and without bias states, update rates 50-800 Hz, noise
    levels over four decades, random camera mounting (q_vc, r_v_cv), static biases, gravity -- each on a 96-filter batch run for
    8 ticks (predict-only ticks, fully and partly masked corrections) with special inputs mixed in: exactly zero and
    sub-tolerance gyro rates (the small-angle branches of EKF.cpp:383-395 and QH.cpp:16-24), innovations of up to ~170 degrees (the
    quaternion flip of QH.cpp:70-72), identity attitude, large and tiny covariances.  Engine vs oracle after every run."""
    rng = np.random.default_rng(20261004)
    B, T = 96, 8
    for case in range(16):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        kw = dict(update_freq=float(rng.choice([50.0, 100.0, 400.0, 800.0])), direct_orien_method=int(case % 2), est_bias=int((case // 2) % 2),
                  Q_a=list(10 ** rng.uniform(-5, -1, size=3)), Q_w=list(10 ** rng.uniform(-6, -2, size=3)),
                  R_r=list(10 ** rng.uniform(-4, -1, size=3)), R_ang=list(10 ** rng.uniform(-4, -1, size=3)),
                  q_vc=list(q), r_v_cv=list(rng.normal(size=3) * 0.1), ab_static=list(rng.normal(size=3) * 0.2),
                  wb_static=list(rng.normal(size=3) * 0.02), g=[0.0, 0.0, float(rng.choice([-9.8, -9.81, -1.62]))])
        if kw["est_bias"]:
            kw.update(Q_ab=list(10 ** rng.uniform(-7, -3, size=3)), Q_wb=list(10 ** rng.uniform(-8, -4, size=3)))
        po, pq = both(**kw)
        n = po.num_states
        x, P = rand_states(rng, B, n, cov_scale=float(10 ** rng.uniform(-3, 0.5)))
        x[:8, 6:10] = np.array([0.0, 0.0, 0.0, 1.0])                     # identity attitude
        if not kw["est_bias"]:
            x[:, 10:16] = 0.0
        U = np.stack([rand_imu(rng, B) for _ in range(T)])
        U[:, 8:16, 3:6] = x[8:16, 13:16] + np.array(kw["wb_static"])                                       # w - wb - wb_static == 0
        U[:, 16:24, 3:6] = x[16:24, 13:16] + np.array(kw["wb_static"]) + 1e-11 * kw["update_freq"]            # |dT w| = 1e-11 < small_ang_tol
        M = (rng.uniform(size=(T, B)) < 0.5).astype(np.uint8); M[0] = 0; M[1] = 1
        Z = np.zeros((T, B, 7)); Z[..., 6] = 1.0
        if dtype == "f32":
            x, P, U = (a.astype(np.float32).astype(np.float64) for a in (x, P, U))
        ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
        ekf.set_state(x, P)
        xr, Pr = x.copy(), P.copy()
        for t in range(T):
            xp, Pp = oracle.run_batch(po, xr, Pr, U[t][None], n_threads=1)  # where the filters will be after this tick's predict
            z = meas_near(rng, po, xp, ang=0.4, pos=0.1)
            z[24:40] = meas_near(rng, po, xp[24:40], ang=2.9, pos=0.1)      # innovations into the flip region
            if dtype == "f32":
                z = z.astype(np.float32).astype(np.float64)
            Z[t] = z
            ekf.step(U[t], z, M[t])
            xr, Pr = oracle.run_batch(po, xr, Pr, U[t][None], z[None], M[t][None], n_threads=1)
        xg, Pg = ekf.get_state()
        assert ekf.count_nonfinite() == 0
        if dtype == "f64":
            assert_state_close(xg, Pg, xr, Pr, 1e-9, 1e-11, 1e-9)
        else:   # 8 ticks, 4-5 corrections with innovations of up to 170 degrees (the logarithm next to pi), noise levels over four
            #         decades (cond(S) up to 1e4): the one deliberately ill-conditioned fp32 comparison; measured 1.1e-3 / 1.1e-4 / 3.2e-4
            assert_state_close(xg, Pg, xr, Pr, 5e-3, 5e-3, 5e-4, ptol=2e-3)
        ekf.close()


# -------------------------------------------------------- cfg 2: free run
def test_cfg2_free_run_fp64_4096():
    """BASELINE cfg 2: B=4096 fp64, update on every tick; 1000-tick free run vs the oracle."""
    kw = golden_kwargs("rotors400", update_freq=100.0)
    po, pq = both(**kw)
    B, T = 4096, 1000
    ekf = qla.BatchedRelativePoseEKF(B, "f64", params=pq)
    seq = ekf.make_inputs(T, np.ones(T, np.uint8))
    ekf.synth_generate(seq, seed=0xE4F00002)
    x0, P0 = ekf.get_state()
    U = np.empty((T, B, 6)); Z = np.empty((T, B, 7)); M = np.empty((T, B), np.uint8)
    for t in range(T):
        U[t], Z[t], M[t] = seq.download_tick(t)
    assert M.all()
    ekf.run(seq, 0, T)
    xg, Pg = ekf.get_state()
    xr, Pr = oracle.run_batch(po, x0, P0, U, Z, M)
    assert ekf.count_nonfinite() == 0
    assert quat_err(xg[:, 6:10], xr[:, 6:10]) < 1e-9
    keep = [i for i in range(16) if not 6 <= i < 10]
    np.testing.assert_allclose(xg[:, keep], xr[:, keep], rtol=1e-9, atol=1e-10)
    sc = np.sqrt(np.einsum("bii->bi", Pr)[:, :, None] * np.einsum("bii->bi", Pr)[:, None, :])
    assert (np.abs(Pg - Pr) / sc).max() < 1e-9
    # the filters actually track the generator's truth
    er, eth, cnt = ekf.synth_rmse(seq)
    assert cnt == B and np.sqrt(er / cnt) < 0.1 and np.sqrt(eth / cnt) < 0.1
    ekf.close()


# ---------------------------------------------------- seeding and reporting
@pytest.mark.parametrize("dtype,tol", [("f64", 1e-13), ("f32", 2e-6)])
def test_initialize_state_and_report(dtype, tol):
    kw = dict(**HW)
    po, pq = both(**kw)
    rng = np.random.default_rng(21)
    B = 130
    x, P = rand_states(rng, B, 15)
    z = meas_near(rng, po, x, ang=1.0, pos=0.5)
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    with pytest.raises(qla.QleError) as e:  # filter_update before initialisation is refused (EKF.cpp:129-130)
        ekf.predict(np.zeros((B, 6)))
    assert e.value.code == -4
    ekf.set_state(x, P)
    ekf.initialize_state(z, reinit_bias=False)
    xg, Pg = ekf.get_state()
    for i in range(B):
        r, q = oracle.seed_pose(po, z[i, :3], z[i, 3:])
        np.testing.assert_allclose(xg[i, 0:3], r, atol=tol * 10)
        assert quat_err(xg[i:i + 1, 6:10], q[None]) < tol * 10
    np.testing.assert_array_equal(xg[:, 3:6], 0)
    np.testing.assert_allclose(xg[:, 10:16], x[:, 10:16], rtol=1e-6 if dtype == "f32" else 0)  # biases kept (reinit_bias=false)
    np.testing.assert_allclose(Pg, np.broadcast_to(np.diag(list(po.cov_init)), Pg.shape), rtol=1e-6 if dtype == "f32" else 0)
    ekf.initialize_state(z, reinit_bias=True)
    assert np.all(ekf.get_state()[0][:, 10:16] == 0)
    # report packing (NODE.cpp:192-220)
    ekf.set_state(x, P)
    rep = ekf.report()
    xs, Ps = ekf.get_state()
    np.testing.assert_array_equal(rep["pose"], np.concatenate([xs[:, 0:3], xs[:, 6:10]], 1))
    sel = [0, 1, 2, 6, 7, 8]
    np.testing.assert_array_equal(rep["pose_cov"], Ps[:, sel][:, :, sel])
    np.testing.assert_array_equal(rep["vel"], xs[:, 3:6])
    np.testing.assert_allclose(rep["bias"], xs[:, 10:16] + np.array(HW["ab_static"] + HW["wb_static"]), atol=1e-7)
    # the same and the rest of what the node puts on the wire (NODE.cpp:192-281) through the one-call struct, after a gated tick
    ekf.enable_gating(True)
    ekf.enable_aux(True)
    u = rng.normal(size=(B, 6)) * 0.1 + np.array([0, 0, 9.8, 0, 0, 0])
    ready = (np.arange(B) % 3 != 0).astype(np.uint8)
    ekf.filter_update(u, z, ready)
    nr = ekf.node_report()
    rep = ekf.report()
    pc, co, up = ekf.tick_flags()
    acc, obs = ekf.get_aux()
    np.testing.assert_array_equal(nr["pose"], rep["pose"])
    np.testing.assert_array_equal(nr["pose_cov"].reshape(B, 6, 6), rep["pose_cov"])
    np.testing.assert_array_equal(nr["vel"], rep["vel"])
    np.testing.assert_array_equal(nr["bias"], rep["bias"])
    np.testing.assert_array_equal(nr["accel"], acc)
    np.testing.assert_array_equal(nr["obs"], obs)
    np.testing.assert_array_equal(nr["performed_correction"], pc)
    np.testing.assert_array_equal(nr["measurement_consumed"], co)
    np.testing.assert_array_equal(nr["upds_since_correction"], up)
    assert nr["state_initialized"].all() and (nr["measurement_consumed"] == ready).all()
    assert (nr["measurement_delay_curr"] == 0).all()      # single-rate filter
    ekf.close()


def test_edge_batches_and_errors():
    pq = qla.make_params()
    for B in (1, 63, 64, 65):
        ekf = qla.BatchedRelativePoseEKF(B, "f64", params=pq)
        rng = np.random.default_rng(B)
        x, P = rand_states(rng, B, 15)
        u = rand_imu(rng, B)
        xg, Pg, _ = ekf.prediction_step(x, P, u)
        xr, Pr, _ = oracle_predict_batch(oracle.make_params(), x, P, u)
        assert_state_close(xg, Pg, xr, Pr, 1e-12, 1e-14, 1e-11)
        with pytest.raises(ValueError):
            ekf.predict(np.zeros((B + 1, 6)))
        ekf.close()
    with pytest.raises(qla.QleError):
        qla.BatchedRelativePoseEKF(0)
    with pytest.raises(qla.QleError):
        qla.BatchedRelativePoseEKF(8, device=99)
    # NaN input propagates silently in the reference (EKF.cpp:475); the engine can count it
    ekf = qla.BatchedRelativePoseEKF(8, "f32", params=pq)
    x, P = rand_states(np.random.default_rng(0), 8, 15)
    x[3, 0] = np.nan
    ekf.set_state(x, P)
    ekf.predict(np.zeros((8, 6)))
    assert ekf.count_nonfinite() == 1
    ekf.close()


# ------------------------------------------------------ synthetic generator
def test_synth_sharding_invariance_and_stats():
    kw = golden_kwargs("rotors400")
    pq = qla.make_params(**kw)
    B, T = 512, 42
    thm = np.zeros(T, np.uint8); thm[13::14] = 1
    full = qla.BatchedRelativePoseEKF(B, "f32", params=pq)
    sf = full.make_inputs(T, thm)
    full.synth_generate(sf, seed=0xE4F00003)
    xf, Pf = full.get_state()
    halves = []
    for k in range(2):
        h = qla.BatchedRelativePoseEKF(B // 2, "f32", params=pq)
        s = h.make_inputs(T, thm)
        h.synth_generate(s, seed=0xE4F00003, filter_offset=k * B // 2)
        halves.append((h, s))
    for t in (0, 13, 41):
        uf, zf, mf = sf.download_tick(t)
        for k, (h, s) in enumerate(halves):
            u, z, m = s.download_tick(t)
            sl = slice(k * B // 2, (k + 1) * B // 2)
            np.testing.assert_array_equal(u, uf[sl]); np.testing.assert_array_equal(z, zf[sl]); np.testing.assert_array_equal(m, mf[sl])
    for k, (h, s) in enumerate(halves):
        sl = slice(k * B // 2, (k + 1) * B // 2)
        np.testing.assert_array_equal(h.get_state()[0], xf[sl])
    # different seed -> different data; sane magnitudes (gravity on the accelerometer)
    u, z, m = sf.download_tick(0)
    assert 8.0 < np.linalg.norm(u[:, :3], axis=1).mean() < 11.5
    assert np.abs(np.linalg.norm(z[:, 3:] if thm[0] else sf.download_tick(13)[1][:, 3:], axis=1) - 1).max() < 1e-6
    other = qla.BatchedRelativePoseEKF(B, "f32", params=pq)
    so = other.make_inputs(T, thm)
    other.synth_generate(so, seed=1)
    assert np.abs(so.download_tick(0)[0] - u).max() > 1e-3
    for h, _ in halves:
        h.close()
    full.close(); other.close()


# --------------------------------------- full-size properties (cfg 3 sizes)
def test_cfg3_full_size_properties_fp32():
    """B=65536 fp32, 400 Hz predict / 30 Hz update: no oracle at this size, so check
    size-independent properties and a sampled slice against the oracle."""
    kw = golden_kwargs("rotors400")
    po, pq = both(**kw)
    B, T = 65536, 560
    thm = np.zeros(T, np.uint8); thm[13::14] = 1
    ekf = qla.BatchedRelativePoseEKF(B, "f32", params=pq)
    seq = ekf.make_inputs(T, thm)
    ekf.synth_generate(seq, seed=0xE4F00003)
    x0, P0 = ekf.get_state()
    ekf.run(seq, 0, T)
    xg, Pg = ekf.get_state()
    assert ekf.count_nonfinite() == 0
    assert np.abs(np.linalg.norm(xg[:, 6:10], axis=1) - 1).max() < 1e-5
    assert (xg[:, 9] >= -0.75 - 1e-6).all()                       # single-cover convention (QH.cpp:61-73)
    ev = np.linalg.eigvalsh(Pg[::64])
    assert ev.min() > 0                                           # P stays positive definite
    assert (np.einsum("bii->b", Pg) < np.einsum("bii->b", P0)).all()  # information gained
    er, eth, cnt = ekf.synth_rmse(seq)
    # the filtered pose beats the raw tag measurement noise (R_r, R_ang of the ROTORS set)
    assert cnt == B
    assert np.sqrt(er / cnt) < 0.75 * np.sqrt(0.015 + 0.015 + 0.020)
    assert np.sqrt(eth / cnt) < 0.75 * np.sqrt(0.0015 + 0.0015 + 0.04)
    # a strided sample of 256 filters against the fp64 oracle over the whole run
    idx = np.arange(0, B, 256)
    U = np.empty((T, idx.size, 6)); Z = np.zeros((T, idx.size, 7)); M = np.zeros((T, idx.size), np.uint8)
    for t in range(T):
        u, z, m = seq.download_tick(t)
        U[t], Z[t], M[t] = u[idx], z[idx], m[idx]
    xr, Pr = oracle.run_batch(po, x0[idx], P0[idx], U, Z, M)
    free_run_close(xg[idx], Pg[idx], xr, Pr, 5e-5, qtol=5e-6)      # 560 ticks, fp32 vs fp64: measured 6e-6 / 6e-7 / 4e-6
    ekf.close()


def test_launch_rules_do_not_change_results_large_ragged_batch(monkeypatch, kernel_family):
    """The launch rules chosen from the batch size -- XCD-chunked block map, workgroup size (64 threads from 262 144
    filters on), cache policy incl. the periodic cached-store tick of small states and the cached/streamed split of large ones, the
    order in which the predict tick requests its loads (every load first up to 65 536 filters) -- are invisible in the results:
    a 262 244-filter batch (ragged: not a multiple of 64, grid not a multiple of 8) and a 65 536-filter shard of the
    same global population agree BIT FOR BIT over 280 ticks, and so does the shard with every rule overridden.
    (The kernel FAMILY is the one rule that changes rounding -- the cooperative kernel fuses the measurement in batch form -- so the
    100-filter tail, which the default policy would give to it, is pinned to the one-lane kernels here.)"""
    kw = golden_kwargs("rotors400")
    pq = qla.make_params(**kw)
    T = 280
    thm = np.zeros(T, np.uint8); thm[13::14] = 1

    def run(B, offset):
        e = qla.BatchedRelativePoseEKF(B, "f32", params=pq)
        s = e.make_inputs(T, thm)
        e.synth_generate(s, seed=0xE4F00004, filter_offset=offset)
        e.run(s, 0, T)
        assert e.count_nonfinite() == 0
        x, P = e.get_state()
        e.close()
        return x, P

    Bbig, off, Bs = 262144 + 100, 131072, 65536
    xb, Pb = run(Bbig, 0)                                  # cached policy, 64-thread workgroups, ragged grid
    xs, Ps = run(Bs, off)                                  # non-temporal + refresh tick, 256-thread workgroups
    np.testing.assert_array_equal(xs, xb[off:off + Bs])
    np.testing.assert_array_equal(Ps, Pb[off:off + Bs])
    if kernel_family == "default":
        monkeypatch.setenv("QLE_QUAD", "0")
    np.testing.assert_array_equal(run(100, 262144)[0], xb[262144:])   # the ragged tail
    if kernel_family == "default":
        monkeypatch.delenv("QLE_QUAD")
    for env in (dict(QLE_NT="0", QLE_BLOCK="64"), dict(QLE_NT="2", QLE_BLOCK="128"), dict(QLE_NT="1", QLE_REFRESH="3"),
                dict(QLE_NT="3", QLE_SPLIT="-20", QLE_BLOCK="64"), dict(QLE_NT="3", QLE_SPLIT="100"),
                dict(QLE_LOADS_FIRST="0"), dict(QLE_LOADS_FIRST="1", QLE_NT="0", QLE_BLOCK="64")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        xo, Po = run(Bs, off)
        np.testing.assert_array_equal(xo, xs); np.testing.assert_array_equal(Po, Ps)
        for k in env:
            monkeypatch.delenv(k)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("direct", [0, 1])
def test_fused_tick_lanes_are_independent_and_predict_only_lanes_match_k_predict(monkeypatch, dtype, direct):
    """k_step runs predict and correction as ONE schedule with the scalar parts of the correction computed unconditionally
    (ekf_fused.hpp).  What a lane stores must not depend on that: a lane whose mask word is set ends up with exactly the bits it gets
    when every lane corrects and a lane whose mask word is clear with exactly the bits it gets when none does -- whatever its
    neighbours in the wave do, and whatever garbage the tag record of a non-correcting lane holds -- and the latter agree with the
    predict-only kernel to rounding (same expressions, compiled in another kernel: the contraction into FMAs may differ)."""
    monkeypatch.setenv("QLE_QUAD", "0")   # the one-lane kernels are what is under test
    kw = dict(golden_kwargs("rotors400"), direct_orien_method=direct)
    po, pq = both(**kw)
    rng = np.random.default_rng(911 + direct)
    B = 64 * 5 + 17
    x0, P0 = rand_states(rng, B, 15)
    u = rand_imu(rng, B)
    z = meas_near(rng, po, oracle_predict_batch(po, x0, P0, u)[0], ang=0.3, pos=0.1)
    mask = (rng.uniform(size=B) < 0.5).astype(np.uint8)
    mask[:64] = 1; mask[64:128] = 0                       # one wave that corrects throughout, one that does not at all
    zg = z.copy()
    zg[mask == 0] = 0.0                                   # an all-zero tag record (zero quaternion): NaN in the discarded scalar parts

    def tick(kind):
        e = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
        e.set_state(x0, P0)
        if kind == "predict":
            e.predict(u)
        elif kind == "all":
            e.step(u, z, np.ones(B, np.uint8))
        elif kind == "none":
            e.step(u, zg, np.zeros(B, np.uint8))
        else:
            e.step(u, zg, mask)
        assert e.count_nonfinite() == 0
        out = e.get_state()
        e.close()
        return out

    xp, Pp = tick("predict"); xa, Pa = tick("all"); xn, Pn = tick("none"); xm, Pm = tick("mixed")
    c = mask.astype(bool)
    np.testing.assert_array_equal(xm[~c], xn[~c]); np.testing.assert_array_equal(Pm[~c], Pn[~c])
    np.testing.assert_array_equal(xm[c], xa[c]); np.testing.assert_array_equal(Pm[c], Pa[c])
    tol = 1e-13 if dtype == "f64" else 3e-7      # same expressions in two kernels: measured 0 (state), 3e-8 (P) in fp32
    note("state", state_dev(xn, xp), tol); note("P", np.abs(Pn - Pp).max(), tol)
    np.testing.assert_allclose(xn, xp, rtol=tol, atol=tol); np.testing.assert_allclose(Pn, Pp, rtol=tol, atol=tol)
    assert np.abs(xa - xp).max() > 1e-3                   # the correction did something


# ------------------------------------------- filter_update decision logic
HW_TAGS = dict(
    n_tags=13, tag_in_view_margin=0.0,
    tag_widths=[0.08382] + [0.16764] * 4 + [0.33528] * 4 + [0.16764] * 4,
    tag_positions=[0, 0, 0, 0, 0.1571625, 0, 0.1571625, 0, 0, 0, -0.1571625, 0, -0.1571625, 0, 0, -0.244475, 0.244475, 0,
                   0.244475, 0.244475, 0, 0.244475, -0.244475, 0, -0.244475, -0.244475, 0, 0, 0.314325, 0, 0.314325, 0, 0,
                   0, -0.314325, 0, -0.314325, 0, 0],
    camera_K=[437.3412312213781, 0, 328.5442810236917, 0, 438.0867474272743, 239.2536470406629, 0, 0, 1],
    camera_width=640, camera_height=480)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("cfg", [dict(limit_measurement_freq=1, measurement_freq=30.0, corner_margin_enbl=1),
                                 dict(limit_measurement_freq=0, corner_margin_enbl=1, **HW_TAGS),
                                 dict(limit_measurement_freq=1, measurement_freq=15.0, corner_margin_enbl=0)])
def test_filter_update_gating_matches_reference_logic(cfg, dtype):
    """Device-side rate limit + corner gate + counters vs the oracle's full filter_update (EKF.cpp:127-303)."""
    kw = dict(update_freq=100.0, direct_orien_method=1, **cfg)
    po, pq = both(**kw)
    rng = np.random.default_rng(33)
    B, T = 96, 40
    x, P = rand_states(rng, B, 15, cov_scale=0.2)
    x[:, 0:3] = rng.uniform([-0.4, -0.4, 0.8], [0.4, 0.4, 2.5], size=(B, 3))
    x[:, 6:10] = np.array([0, 0, 0, 1.0])  # level, so that the tag is roughly in front of the camera
    if dtype == "f32":
        x = x.astype(np.float32).astype(np.float64); P = P.astype(np.float32).astype(np.float64)
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    ekf.enable_gating(True)
    ekf.set_state(x, P)
    filt = []
    for i in range(B):
        f = oracle.Filter(po)
        for k in range(3):
            f.f.r_nom[k] = x[i, k]; f.f.v_nom[k] = x[i, 3 + k]; f.f.ab_nom[k] = x[i, 10 + k]; f.f.wb_nom[k] = x[i, 13 + k]
        for k in range(4):
            f.f.q_nom[k] = x[i, 6 + k]
        for k in range(225):
            f.f.cov_pert[k] = P[i].reshape(-1)[k]
        f.f.state_initialized = 1
        filt.append(f)
    pending = np.zeros(B, np.uint8)
    zlast = np.zeros((B, 7)); zlast[:, 6] = 1
    n_perf = n_rej = 0
    for t in range(T):
        u = rand_imu(rng, B) * np.array([0.05, 0.05, 1, 0.2, 0.2, 0.2])
        new = rng.uniform(size=B) < 0.5
        xs = ekf.get_state()[0]
        znew = meas_near(rng, po, xs, ang=0.3, pos=0.05)
        znew[:, 0:2] += rng.choice([0.0, 0.0, 1.5], size=(B, 1)) * rng.normal(size=(B, 2))  # some tags out of view
        if dtype == "f32":
            u = u.astype(np.float32).astype(np.float64); znew = znew.astype(np.float32).astype(np.float64)
        zlast[new] = znew[new]
        pending |= new.astype(np.uint8)
        for i in range(B):
            filt[i].set_imu(u[i, :3], u[i, 3:])
            if new[i]:
                filt[i].set_apriltag(zlast[i, :3], zlast[i, 3:], 0.01 * t)
            filt[i].filter_update(0.01 * t)
        ekf.filter_update(u, zlast if pending.any() else None, pending if pending.any() else None)
        perf, cons, upds = ekf.tick_flags()
        ref_perf = np.array([f.f.performed_correction for f in filt], np.uint8)
        ref_ready = np.array([f.f.measurement_ready for f in filt], np.uint8)
        ref_upds = np.array([f.f.upds_since_correction for f in filt], np.int32)
        pending &= (1 - cons)
        np.testing.assert_array_equal(perf, ref_perf)
        np.testing.assert_array_equal(pending, ref_ready)   # consumed <=> the reference cleared measurement_ready
        np.testing.assert_array_equal(upds, ref_upds)
        n_perf += int(perf.sum()); n_rej += int((cons & (1 - perf)).sum())
    assert n_perf > B and (n_rej > 0 or not cfg["corner_margin_enbl"])
    xg, Pg = ekf.get_state()
    xr = np.stack([f.x() for f in filt]); Pr = np.stack([f.P() for f in filt])
    if dtype == "f64":
        assert_state_close(xg, Pg, xr, Pr, 1e-10, 1e-12, 1e-10)
    else:   # 40 ticks with 10-20 corrections each
        assert_state_close(xg, Pg, xr, Pr, 2e-5, 2e-5, 3e-6, ptol=3e-5)      # measured 1.7e-6 / 3.5e-7 / 2.4e-6
    ekf.close()


# -------------------------------------------------- multirate EKF (replay)
def _oracle_filters(po, x, P):
    filt = []
    for i in range(x.shape[0]):
        f = oracle.Filter(po)
        for k in range(3):
            f.f.apriltag_pos[k] = 0.0
        # seed through the reference's own path so that the history starts as initialize_state leaves it
        f.f.state_initialized = 0
        filt.append(f)
    return filt


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("cfg", [dict(dynamic_meas_delay=0, measurement_delay=0.030, limit_measurement_freq=1, measurement_freq=30.0),
                                 dict(dynamic_meas_delay=1, measurement_delay=0.030, measurement_delay_max=0.200,
                                      dyn_measurement_delay_offset=0.005, limit_measurement_freq=1, measurement_freq=15.0),
                                 dict(dynamic_meas_delay=1, measurement_delay=0.150, measurement_delay_max=0.350,
                                      dyn_measurement_delay_offset=0.085, limit_measurement_freq=0, **HW_TAGS)])
def test_multirate_replay_matches_reference_logic(cfg, dtype, T=60, loosen=1.0):
    """multirate_ekf = true (EKF.cpp:196-236, 251-264) against the oracle's full filter object: delayed
    correction index, history trimming, replay of the stored IMU samples, dynamic delay per filter."""
    kw = dict(update_freq=100.0, direct_orien_method=1, multirate_ekf=1, corner_margin_enbl=1, **cfg)
    po, pq = both(**kw)
    rng = np.random.default_rng(77)
    B = 64
    z0 = np.zeros((B, 7))
    z0[:, 0:2] = rng.normal(size=(B, 2)) * 0.1; z0[:, 2] = rng.uniform(0.8, 2.0, size=B)
    z0[:, 3:7] = np.array([0.7071067811865476, -0.7071067811865476, 0.0, 0.0])
    if dtype == "f32":
        z0 = z0.astype(np.float32).astype(np.float64)
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    ekf.enable_gating(True)
    ekf.initialize_state(z0, reinit_bias=True)
    filt = []
    for i in range(B):
        f = oracle.Filter(po)
        f.set_apriltag(z0[i, :3], z0[i, 3:], -1.0)   # initialises the state and the history (EKF.cpp:305-344)
        f.f.measurement_ready = 0
        filt.append(f)
    if dtype == "f32":  # start both from the engine's (fp32-rounded) seeded state
        xs, Ps = ekf.get_state()
        for i, f in enumerate(filt):
            for k in range(3):
                f.f.r_nom[k] = xs[i, k]
            for k in range(4):
                f.f.q_nom[k] = xs[i, 6 + k]
            for k in range(16):
                f.f.x_hist[k] = xs[i, k]
    pending = np.zeros(B, np.uint8)
    zlast = z0.copy(); stamp = np.zeros(B)
    n_perf = 0
    for t in range(T):
        tc = 0.01 * t
        u = rand_imu(rng, B) * np.array([0.05, 0.05, 1, 0.2, 0.2, 0.2])
        new = rng.uniform(size=B) < 0.45
        xs = ekf.get_state()[0]
        znew = meas_near(rng, po, xs, ang=0.2, pos=0.05)
        znew[:, 0:2] += rng.choice([0.0, 0.0, 0.0, 1.5], size=(B, 1)) * rng.normal(size=(B, 2))
        if dtype == "f32":
            u = u.astype(np.float32).astype(np.float64); znew = znew.astype(np.float32).astype(np.float64)
        zlast[new] = znew[new]
        stamp[new] = tc - rng.uniform(0.0, 0.3, size=int(new.sum()))  # camera latency 0..300 ms, per filter
        pending |= new.astype(np.uint8)
        for i in range(B):
            filt[i].set_imu(u[i, :3], u[i, 3:])
            if new[i]:
                filt[i].set_apriltag(zlast[i, :3], zlast[i, 3:], stamp[i])
            filt[i].filter_update(tc)
        ekf.filter_update(u, zlast if pending.any() else None, pending if pending.any() else None, t_curr=tc, apriltag_time=stamp)
        perf, cons, upds = ekf.tick_flags()
        ref_perf = np.array([f.f.performed_correction for f in filt], np.uint8)
        np.testing.assert_array_equal(perf, ref_perf)
        np.testing.assert_array_equal(upds, np.array([f.f.upds_since_correction for f in filt], np.int32))
        pending &= (1 - cons)
        np.testing.assert_array_equal(pending, np.array([f.f.measurement_ready for f in filt], np.uint8))
        if perf.any() and cfg["dynamic_meas_delay"]:
            d = ekf.measurement_delay()
            ref_d = np.array([f.f.measurement_delay_curr for f in filt])
            np.testing.assert_allclose(d[perf.astype(bool)], ref_d[perf.astype(bool)], atol=1e-12)
        n_perf += int(perf.sum())
        xg, Pg = ekf.get_state()
        xr = np.stack([f.x() for f in filt]); Pr = np.stack([f.P() for f in filt])
        if dtype == "f64":
            assert_state_close(xg, Pg, xr, Pr, 1e-10 * loosen, 1e-12 * loosen, 1e-10 * loosen)
        else:   # up to 60 ticks of a randomly driven free run, ~25 corrections per filter, each replaying up to 35 predictions
            assert_state_close(xg, Pg, xr, Pr, 2e-5 * loosen, 2e-5 * loosen, 5e-6 * loosen, ptol=5e-5 * loosen)   # measured 2e-6 / 5e-7 / 4.5e-6
    assert n_perf > B
    ekf.close()


def test_multirate_history_over_ring_wraps(dtype="f64"):
    """The engine keeps the state in place, an IMU ring, a state checkpoint every 16 ticks and a per-filter anchor at the last
    correction, and rebuilds the entry a measurement belongs to by replaying from the newest checkpoint or the anchor (k_step_mr).
    Measurements on ~45 % of the ticks with 0-300 ms of latency and no rate limit put corrections at every offset between
    checkpoints, before and after the previous correction's entry; 260 ticks cross the 64-slot IMU ring four times and every
    checkpoint slot several times.  Every tick must still match the reference logic (fp64: the randomly driven free run is too
    long for an fp32-vs-fp64 comparison tick by tick)."""
    test_multirate_replay_matches_reference_logic(
        dict(dynamic_meas_delay=1, measurement_delay=0.150, measurement_delay_max=0.350, dyn_measurement_delay_offset=0.085,
             limit_measurement_freq=0, **HW_TAGS), dtype, T=260, loosen=100.0)   # free run: rounding accumulates with the tick count


@pytest.mark.parametrize("mr", [1, 0])
@pytest.mark.parametrize("est_bias", [1, 0])
@pytest.mark.parametrize("direct", [0, 1])
def test_multirate_with_per_filter_parameters_and_conventional_method(direct, est_bias, mr):
    """The kernel instantiations the other multirate tests do not reach: k_step_mr / k_predict<MR> with the
    conventional orientation method (EKF.cpp:440-444,455-458), per-filter noise and static-bias records (cfg 5) and
    9 error states, fp64, every filter against its own oracle filter object on every tick.  mr = 0: the same through
    the single-rate fused tick with the decision logic on the device (k_step<..., PFP, GATE>)."""
    kw = dict(update_freq=100.0, direct_orien_method=direct, est_bias=est_bias, multirate_ekf=mr, dynamic_meas_delay=1,
              measurement_delay=0.030, measurement_delay_max=0.200, dyn_measurement_delay_offset=0.005,
              limit_measurement_freq=1, measurement_freq=30.0, corner_margin_enbl=0)
    po, pq = both(**kw)
    rng = np.random.default_rng(2025)
    B, T = 16, 90
    pfp = np.zeros((B, 24))
    pfp[:, 0:12] = np.array(list(po.Q)) * 10 ** rng.uniform(-0.5, 0.5, size=(B, 4)).repeat(3, axis=1)
    pfp[:, 12:15] = rng.normal(size=(B, 3)) * 0.1
    pfp[:, 15:18] = rng.normal(size=(B, 3)) * 0.01
    pfp[:, 18:24] = np.array(list(po.R)) * rng.uniform(0.5, 2.0, size=(B, 6))
    z0 = np.zeros((B, 7))
    z0[:, 0:2] = rng.normal(size=(B, 2)) * 0.1; z0[:, 2] = rng.uniform(0.8, 2.0, size=B)
    z0[:, 3:7] = np.array([0.7071067811865476, -0.7071067811865476, 0.0, 0.0])
    ekf = qla.BatchedRelativePoseEKF(B, "f64", params=pq)
    ekf.set_filter_params(pfp)
    ekf.enable_gating(True)
    ekf.initialize_state(z0, reinit_bias=True)
    filt = []
    for i in range(B):
        q = pfp[i, 0:12]
        pi = oracle.make_params(**dict(kw, Q_a=q[0:3], Q_w=q[3:6], Q_ab=q[6:9], Q_wb=q[9:12], ab_static=pfp[i, 12:15], wb_static=pfp[i, 15:18],
                                       R_r=pfp[i, 18:21], R_ang=pfp[i, 21:24]))
        f = oracle.Filter(pi)
        f.set_apriltag(z0[i, :3], z0[i, 3:], -1.0)
        f.f.measurement_ready = 0
        filt.append(f)
    pending = np.zeros(B, np.uint8)
    zlast = z0.copy(); stamp = np.zeros(B)
    n_perf = 0
    for t in range(T):
        tc = 0.01 * t
        u = rand_imu(rng, B) * np.array([0.05, 0.05, 1, 0.2, 0.2, 0.2])
        new = rng.uniform(size=B) < 0.35
        xs = ekf.get_state()[0]
        znew = meas_near(rng, po, xs, ang=0.2, pos=0.05)
        zlast[new] = znew[new]
        stamp[new] = tc - rng.uniform(0.0, 0.25, size=int(new.sum()))
        pending |= new.astype(np.uint8)
        for i in range(B):
            filt[i].set_imu(u[i, :3], u[i, 3:])
            if new[i]:
                filt[i].set_apriltag(zlast[i, :3], zlast[i, 3:], stamp[i])
            filt[i].filter_update(tc)
        ekf.filter_update(u, zlast if pending.any() else None, pending if pending.any() else None, t_curr=tc, apriltag_time=stamp)
        perf, cons, upds = ekf.tick_flags()
        np.testing.assert_array_equal(perf, np.array([f.f.performed_correction for f in filt], np.uint8))
        pending &= (1 - cons)
        n_perf += int(perf.sum())
        xg, Pg = ekf.get_state()
        xr = np.stack([f.x() for f in filt]); Pr = np.stack([f.P() for f in filt])
        assert_state_close(xg, Pg, xr, Pr, 1e-9, 1e-11, 1e-9)
    assert n_perf > B
    ekf.close()


# ------------------------------------------------------ API robustness
def test_api_round_trips_and_error_paths():
    pq = qla.make_params()
    rng = np.random.default_rng(5)
    B = 100
    a = qla.BatchedRelativePoseEKF(B, "f64", params=pq)
    b = qla.BatchedRelativePoseEKF(B, "f32", params=qla.make_params(update_freq=400.0))
    # state round trip (fp64 exact, fp32 to rounding), two handles alive on the same device
    x, P = rand_states(rng, B, 15)
    a.set_state(x, P); b.set_state(x, P)
    xa, Pa = a.get_state(); xb, Pb = b.get_state()
    np.testing.assert_array_equal(xa, x); np.testing.assert_array_equal(Pa, P)
    np.testing.assert_allclose(xb, x, rtol=1e-6, atol=1e-7); np.testing.assert_allclose(Pb, P, rtol=1e-6, atol=1e-8)
    # a non-symmetric P is symmetrised as (P + P^T)/2
    Pn = P.copy(); Pn[:, 0, 5] += 0.01
    a.set_state(x, Pn)
    np.testing.assert_allclose(a.get_state()[1], 0.5 * (Pn + Pn.transpose(0, 2, 1)), rtol=0, atol=1e-17)
    # input sequences: upload / download round trip, slot bookkeeping, bounds
    T = 9
    thm = np.zeros(T, np.uint8); thm[[2, 7]] = 1
    seq = a.make_inputs(T, thm)
    U = rng.normal(size=(T, B, 6)); Z = rng.normal(size=(T, B, 7)); M = (rng.uniform(size=(T, B)) < 0.5).astype(np.uint8)
    for t in range(T):
        seq.upload_tick(t, U[t], Z[t] if thm[t] else None, M[t] if thm[t] else None)
    for t in range(T):
        u, z, m = seq.download_tick(t)
        np.testing.assert_array_equal(u, U[t])
        if thm[t]:
            np.testing.assert_array_equal(z, Z[t]); np.testing.assert_array_equal(m, M[t])
        else:
            assert not m.any()
    with pytest.raises(qla.QleError):
        seq.upload_tick(T, U[0])                    # tick out of range
    with pytest.raises(qla.QleError):
        seq.upload_tick(0, U[0], Z[0])              # tick 0 has no measurement slot
    with pytest.raises(qla.QleError):
        seq.upload_tick(2, U[2])                    # tick 2 needs z
    with pytest.raises(qla.QleError):
        b.run(seq, 0, 1)                            # sequence belongs to another handle
    with pytest.raises(qla.QleError):
        a.synth_rmse(seq)                           # no generated truth in an uploaded sequence
    # parameters can be changed between ticks (NODE.cpp:138 re-runs initialize_params)
    u = rand_imu(rng, B)
    a.set_state(x, P)
    a.initialize_params(update_freq=400.0, Q_a=[1e-3, 2e-3, 3e-3])
    a.step(u)
    po = oracle.make_params(update_freq=400.0, Q_a=[1e-3, 2e-3, 3e-3])
    xr, Pr, _ = oracle_predict_batch(po, x, P, u)
    assert_state_close(*a.get_state(), xr, Pr, 1e-12, 1e-14, 1e-11)
    # an absurd batch fails with a clean out-of-memory error, not a crash
    with pytest.raises(qla.QleError) as e:
        qla.BatchedRelativePoseEKF(2 ** 34, "f64", params=pq)
    assert e.value.code in (-3, -2)
    a.close(); b.close()
    with pytest.raises(Exception):
        a.predict(u)                                # closed handle


# ------------------------------------ on-chip-resident multi-tick variant
@pytest.mark.parametrize("dtype,tol", [("f64", 1e-9), ("f32", 6e-5)])      # fp32, 84 ticks: measured 6.6e-6 / 7e-7 / 4.6e-6
@pytest.mark.parametrize("variant", ["direct", "conventional_per_filter_params"])
def test_run_resident_matches_per_tick_path_and_oracle(dtype, tol, variant):
    kw = golden_kwargs("rotors400")
    pfp = None
    if variant != "direct":
        kw = dict(kw, direct_orien_method=0)
    po, pq = both(**kw)
    B, T = 300, 84
    thm = np.zeros(T, np.uint8); thm[13::14] = 1
    a = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    if variant != "direct":   # k_step / k_predict / k_run_resident with the per-filter parameter record (cfg 5)
        r5 = np.random.default_rng(5)
        pfp = np.zeros((B, 24))
        pfp[:, 0:12] = np.array(list(po.Q)) * 10 ** r5.uniform(-0.5, 0.5, size=(B, 4)).repeat(3, axis=1)
        pfp[:, 12:15] = r5.normal(size=(B, 3)) * 0.1
        pfp[:, 15:18] = r5.normal(size=(B, 3)) * 0.01
        pfp[:, 18:24] = np.array(list(po.R)) * r5.uniform(0.5, 2.0, size=(B, 6))
        if dtype == "f32":
            pfp = pfp.astype(np.float32).astype(np.float64)
        a.set_filter_params(pfp)
    seq = a.make_inputs(T, thm)
    a.synth_generate(seq, seed=11)
    # give the masks some structure: a third of the filters skip each measurement
    rng = np.random.default_rng(2)
    for t in np.nonzero(thm)[0]:
        u, z, m = seq.download_tick(int(t))
        seq.upload_tick(int(t), u, z, (rng.uniform(size=B) < 0.67).astype(np.uint8))
    x0, P0 = a.get_state()
    a.run(seq, 0, T)
    xa, Pa = a.get_state()
    a.set_state(x0, P0)
    a.run_resident(seq, 0, 30)          # two launches, wrapping inside the sequence is allowed
    a.run_resident(seq, 30, T - 30)
    xb, Pb = a.get_state()
    U = np.empty((T, B, 6)); Z = np.zeros((T, B, 7)); M = np.zeros((T, B), np.uint8)
    for t in range(T):
        U[t], Z[t], M[t] = seq.download_tick(t)
    xr, Pr = oracle.run_batch(po, x0, P0, U, Z, M, per_filter_params=pfp)
    for xg, Pg in ((xa, Pa), (xb, Pb)):
        free_run_close(xg, Pg, xr, Pr, tol, qtol=tol / 8)
    a.enable_gating(True)
    with pytest.raises(qla.QleError):
        a.run_resident(seq, 0, 1)       # covers the single-rate filter with explicit masks only
    a.close()


@pytest.mark.parametrize("mode", ["multirate", "singlerate"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_full_filter_update_against_reference_twin_golden(mode, dtype):
    """Engine (device-side gating + multirate ring) directly against the reference twin's own
    filter_update trajectory (tests/golden/filter_update_cases.npz)."""
    d = np.load(f"{GOLDEN}/filter_update_cases.npz")
    U, Z, NEW = d[f"{mode}__u"], d[f"{mode}__z"], d[f"{mode}__new"]
    kw = dict(update_freq=100.0, measurement_freq=10.0, measurement_delay=0.050, dynamic_meas_delay=0,
              limit_measurement_freq=1, corner_margin_enbl=1, direct_orien_method=1, est_bias=1, multirate_ekf=int(mode == "multirate"))
    B = 3
    ekf = qla.BatchedRelativePoseEKF(B, dtype, **kw)
    ekf.enable_gating(True)
    rep = lambda a: np.repeat(np.asarray(a)[None], B, 0)
    ekf.initialize_state(rep(Z[0]), reinit_bias=True)   # the twin initialises on its first detection (tick 0)
    pending = np.ones(B, np.uint8); zlast = rep(Z[0])
    tol, qtol, ptol = (1e-8, 1e-8, 1e-7) if dtype == "f64" else (3e-5, 1e-6, 6e-5)     # fp32, 160 ticks: measured 2.8e-6 / 1e-7 / 6e-6
    for t in range(U.shape[0]):
        if NEW[t]:
            zlast = rep(Z[t]); pending[:] = 1
        ekf.filter_update(rep(U[t]), zlast if pending.any() else None, pending if pending.any() else None, t_curr=0.01 * t,
                          apriltag_time=np.full(B, 0.01 * t))
        perf, cons, upds = ekf.tick_flags()
        pending &= (1 - cons)
        assert (upds == d[f"{mode}__upds"][t]).all(), t
        assert (pending == d[f"{mode}__ready"][t]).all(), t
        x, P = ekf.get_state()
        xr = d[f"{mode}__x_seq"][t]
        assert note("quat", quat_err(x[:, 6:10], rep(xr[6:10])), qtol) < qtol, t
        keep = [i for i in range(16) if not 6 <= i < 10]
        note("state", state_dev(x, rep(xr)), tol)
        np.testing.assert_allclose(x[:, keep], rep(xr[keep]), rtol=tol, atol=tol)
        note("Pdiag", np.abs(np.einsum("bii->bi", P) / rep(d[f"{mode}__P_diag_seq"][t]) - 1).max(), ptol)
        np.testing.assert_allclose(np.einsum("bii->bi", P), rep(d[f"{mode}__P_diag_seq"][t]), rtol=ptol)
    ekf.close()


# ------------------ C++-only branches pinned through the twin by identity (tests/golden/make_golden_branches.py)
def _hw_branch_kwargs(d, **over):
    ps = golden_kwargs("hardware")
    kw = dict(ps, update_freq=100.0, measurement_freq=100.0, limit_measurement_freq=0, corner_margin_enbl=1, direct_orien_method=1, est_bias=1,
              ab_static=list(d["static__ab_static"]), wb_static=list(d["static__wb_static"]),
              n_tags=13, tag_in_view_margin=float(d["gate__margin"][0]), tag_widths=list(d["gate__tag_widths"]),
              tag_positions=list(d["gate__tag_positions"]), camera_K=list(d["gate__camera_K"]),
              camera_width=int(d["gate__camera_size"][0]), camera_height=int(d["gate__camera_size"][1]))
    kw.update(over)
    return kw


@pytest.mark.parametrize("dtype,tol", [("f64", F64), ("f32", F32)])
def test_static_bias_predict_against_twin_identity(dtype, tol):
    """EKF.cpp:357-358: the twin run with the statics inside its bias states produced these numbers."""
    d = np.load(f"{GOLDEN}/branch_cases.npz")
    pq = qla.make_params(**golden_kwargs("hardware", ab_static=list(d["static__ab_static"]), wb_static=list(d["static__wb_static"])))
    x, P, u = d["static__x"], d["static__P"], d["static__u"]
    ekf = qla.BatchedRelativePoseEKF(x.shape[0], dtype, params=pq)
    xg, Pg, ag = ekf.prediction_step(x, P, u)
    assert_state_close(xg, Pg, d["static__x_check"], d["static__P_check"], **tol)
    np.testing.assert_allclose(ag, d["static__accel"], rtol=0, atol=1e-12 if dtype == "f64" else 3e-5)
    ekf.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("mr", [0, 1])
def test_multi_tag_gate_against_composed_twin_decisions(dtype, mr):
    """EKF.cpp:160-181 on the device (corner_gate, ekf_kernels.hpp): the 13-tag bundle of HW.yaml == `or` over the twin's
    single-tag decisions, and each tag alone == the twin's decision for it, on 320 tag poses."""
    d = np.load(f"{GOLDEN}/branch_cases.npz")
    Z, per_tag, composed = d["gate__z"], d["gate__per_tag"], d["gate__composed"]
    if dtype == "f32":   # the decision is taken on the dtype-rounded pose: drop the poses whose margin is below fp32 resolution
        p13 = oracle.make_params(**_hw_branch_kwargs(d))
        Zr = Z.astype(np.float32).astype(np.float64)
        keep = np.array([oracle.corner_gate(p13, z[:3], z[3:]) for z in Zr], np.uint8) == composed
        assert keep.sum() >= len(Z) - 3
    else:
        keep = np.ones(len(Z), bool)
    B = len(Z)
    rng = np.random.default_rng(4)
    x, P = rand_states(rng, B, 15, cov_scale=0.2)
    w, pos = d["gate__tag_widths"], d["gate__tag_positions"].reshape(13, 3)
    cases = [(None, composed)] + [(k, per_tag[:, k]) for k in (0, 3, 7, 12)]
    for k, want in cases:
        kw = _hw_branch_kwargs(d, multirate_ekf=mr, measurement_delay=0.010)
        if k is not None:
            kw.update(n_tags=1, tag_widths=[float(w[k])], tag_positions=list(pos[k]))
        ekf = qla.BatchedRelativePoseEKF(B, dtype, **kw)
        ekf.enable_gating(True)
        ekf.set_state(x, P)
        if mr:
            ekf.initialize_state(Z, reinit_bias=True)
            ekf.filter_update(rand_imu(rng, B), None, None, t_curr=0.0, apriltag_time=np.zeros(B))
        ekf.filter_update(rand_imu(rng, B), Z, np.ones(B, np.uint8), t_curr=0.01, apriltag_time=np.zeros(B))
        perf, cons, _ = ekf.tick_flags()
        assert cons.all()
        np.testing.assert_array_equal(perf[keep], want[keep])
        ekf.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("delay", ["dynamic_exact", "dynamic_clamped", "fixed"])
@pytest.mark.parametrize("mode", ["hw_multirate", "hw_singlerate"])
def test_hardware_like_filter_update_against_twin_golden(mode, delay, dtype):
    """Everything HW.yaml switches on at once -- static biases (EKF.cpp:357-358), the 13-tag bundle (EKF.cpp:160-181), a
    correction on every tick, a 15-tick measurement delay taken from the stamps (EKF.cpp:199-200) -- against the twin's own
    filter_update trajectory (240 ticks, ~125 corrections, each replaying 15 predictions)."""
    d = np.load(f"{GOLDEN}/branch_cases.npz")
    dyn = dict(fixed=dict(dynamic_meas_delay=0, measurement_delay=0.150),
               dynamic_exact=dict(dynamic_meas_delay=1, measurement_delay=0.010, measurement_delay_max=0.350, dyn_measurement_delay_offset=0.085),
               dynamic_clamped=dict(dynamic_meas_delay=1, measurement_delay=0.010, measurement_delay_max=0.150, dyn_measurement_delay_offset=0.085))[delay]
    age = float(d[f"{mode}__age_clamped"][0] if delay == "dynamic_clamped" else d[f"{mode}__age_exact"][0])
    U, Z, NEW = d[f"{mode}__u"], d[f"{mode}__z"], d[f"{mode}__new"]
    B = 3
    ekf = qla.BatchedRelativePoseEKF(B, dtype, **_hw_branch_kwargs(d, multirate_ekf=int(mode == "hw_multirate"), **dyn))
    ekf.enable_gating(True)
    rep = lambda a: np.repeat(np.asarray(a)[None], B, 0)
    ekf.initialize_state(rep(d[f"{mode}__z0"]), reinit_bias=True)
    pending = np.zeros(B, np.uint8); zlast = rep(d[f"{mode}__z0"]); stamp = 0.0
    tol, qtol, ptol = (1e-8, 1e-8, 1e-7) if dtype == "f64" else (2e-5, 1e-6, 4e-4)     # fp32, 240 ticks: measured 2e-6 / 1e-7 / 4.1e-5 (diagonal of P, single-rate run)
    for t in range(U.shape[0]):
        tc = 0.01 * t
        if NEW[t]:
            zlast = rep(Z[t]); pending[:] = 1; stamp = tc - age
        ekf.filter_update(rep(U[t]), zlast if pending.any() else None, pending if pending.any() else None, t_curr=tc,
                          apriltag_time=np.full(B, stamp))
        perf, cons, upds = ekf.tick_flags()
        pending &= (1 - cons)
        assert (perf == d[f"{mode}__perf"][t]).all(), t
        assert (upds == d[f"{mode}__upds"][t]).all(), t
        assert not pending.any()
        if mode == "hw_multirate" and delay != "fixed" and perf.any():
            np.testing.assert_allclose(ekf.measurement_delay(), 0.150, atol=1e-12)
        x, P = ekf.get_state()
        xr = d[f"{mode}__x_seq"][t]
        assert note("quat", quat_err(x[:, 6:10], rep(xr[6:10])), qtol) < qtol, t
        keep = [i for i in range(16) if not 6 <= i < 10]
        note("state", state_dev(x, rep(xr)), tol)
        np.testing.assert_allclose(x[:, keep], rep(xr[keep]), rtol=tol, atol=tol)
        note("Pdiag", np.abs(np.einsum("bii->bi", P) / rep(d[f"{mode}__P_diag_seq"][t]) - 1).max(), ptol)
        np.testing.assert_allclose(np.einsum("bii->bi", P), rep(d[f"{mode}__P_diag_seq"][t]), rtol=ptol)
        if t in d[f"{mode}__P_full_ticks"]:
            Pr = d[f"{mode}__P_full"][list(d[f"{mode}__P_full_ticks"]).index(t)]
            assert note("cov", cov_dev(P, rep(Pr)), ptol) < ptol
    ekf.close()


# ------------------------------ the twin's own interface, the twin's own numbers
def test_twin_interface_reproduces_twin_goldens():
    """quadrotor_landing_amd.RelativePoseEKF has the Python twin's constructor and step signatures
    (PYEKF.py:29,373,428); driven exactly as tests/golden/make_golden.py drives the reference twin, it
    returns the numbers the reference twin returned."""
    import json
    kat = json.load(open(f"{GOLDEN}/kat.json"))["appendix_c"]
    f = qla.RelativePoseEKF(100, 10)
    x0 = np.array(kat["x"]).reshape(16, 1); u0 = np.array(kat["u"]).reshape(6, 1); P0 = np.diag(kat["P_diag"])
    xc, Pc, acc = f.prediction_step(x0, u0, P0)
    assert xc.shape == (16, 1) and Pc.shape == (15, 15) and acc.shape == (3, 1)
    np.testing.assert_allclose(xc.flatten(), kat["x_check"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(Pc, kat["P_check"], rtol=1e-12, atol=1e-17)
    np.testing.assert_allclose(acc.flatten(), kat["accel"], atol=1e-14)
    xh, Ph = f.correction_step(xc, Pc, np.array(kat["r_c_tc"]).reshape(3, 1), np.array(kat["q_ct_xyzw"]))
    np.testing.assert_allclose(xh.flatten(), kat["x_hat"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(Ph, kat["P_hat"], rtol=1e-10, atol=1e-15)
    # the 9-state variant through the same attribute the twin uses
    d = np.load(f"{GOLDEN}/predict_cases.npz")
    f.est_bias = False
    f.Q = np.diag([0.005] * 3 + [0.0005] * 3)
    f.sync_params()
    assert f.num_states == 9
    xc, Pc, _ = f.prediction_step(d["nobias__x"][5].reshape(16, 1), d["nobias__u"][5].reshape(6, 1), d["nobias__P"][5])
    np.testing.assert_allclose(xc.flatten(), d["nobias__x_check"][5], atol=1e-13)
    np.testing.assert_allclose(Pc, d["nobias__P_check"][5], rtol=1e-11, atol=1e-15)
    f.close()


def test_tick_origin_shift_is_invisible(monkeypatch):
    """Tick indices are 32-bit on the device (last_corr, hist_first, history slots = tick modulo the ring sizes); the host shifts the
    origin long before they could wrap.  With the threshold lowered to 48 ticks the gating and multirate parity
    tests cross several shifts and must still match the oracle tick for tick."""
    monkeypatch.setenv("QLE_TICK_REBASE", "48")
    test_filter_update_gating_matches_reference_logic(dict(limit_measurement_freq=1, measurement_freq=30.0, corner_margin_enbl=1), "f64")
    test_multirate_replay_matches_reference_logic(dict(dynamic_meas_delay=1, measurement_delay=0.030, measurement_delay_max=0.200,
                                                       dyn_measurement_delay_offset=0.005, limit_measurement_freq=1, measurement_freq=15.0), "f64")
    test_full_filter_update_against_reference_twin_golden("multirate", "f64")
