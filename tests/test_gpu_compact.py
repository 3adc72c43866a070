"""est_bias = false (relative_pose_EKF.cpp:92, num_states = 9; :405-409): COMPACT state records.

A filter that does not estimate the biases has identically zero bias blocks in P; on the batch sizes the lane-per-filter kernels
serve, its record then keeps only the 45 words of the 9 x 9 pose block (ekf_kernels.hpp, `load_P_compact`) and a tick moves
16 + 48 words per direction instead of 136.  The arithmetic is the same register image either way, so a compact handle must agree
with a full-record handle BIT FOR BIT on the tick kernels (the separately compiled on-chip-resident kernel: to rounding in fp32), and with
the reference twin's `nobias` goldens within the usual tolerances.
QLE_COMPACT=1 forces the layout at the small batch sizes of these tests (by default it is chosen wherever the lane-per-filter kernels serve every tick: above 4 096 filters).
"""
import numpy as np
import pytest

import oracle
import quadrotor_landing_amd as qla
import test_gpu_parity as tp
from util import assert_state_close, meas_near, oracle_predict_batch, oracle_update_batch, rand_imu, rand_states

pytestmark = pytest.mark.gpu

NOBIAS = dict(update_freq=400.0, direct_orien_method=1, est_bias=0, measurement_freq=30.0)


def _handle(B, dtype, monkeypatch, compact, **kw):
    monkeypatch.setenv("QLE_COMPACT", "1" if compact else "0")
    monkeypatch.setenv("QLE_QUAD", "0")     # both handles on the lane-per-filter kernels: the comparison is bit for bit
    pq = qla.make_params(**dict(NOBIAS, **kw))
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    assert ekf.policy()["record_words"] == (64 if compact else 136)
    return ekf


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("direct", [1, 0])
def test_compact_records_agree_with_full_records_bit_for_bit(dtype, direct, monkeypatch):
    """The same 60-tick sequence (predict ticks, masked fused ticks, one stand-alone update, one bare predict, a state round trip) on a
    compact and on a full-record handle: identical states and 9 x 9 covariances, identical published report; the compact handle
    reports the smaller algorithmic byte count."""
    rng = np.random.default_rng(321)
    B = 1000   # ragged: 15 whole tiles + 40 filters
    x, P = rand_states(rng, B, 9, cov_scale=0.3)
    x[:, 10:16] = 0.0
    out = []
    for compact in (False, True):
        ekf = _handle(B, dtype, monkeypatch, compact, direct_orien_method=direct)
        ekf.set_state(x, P)
        r2 = np.random.default_rng(5)
        for t in range(60):
            u = rand_imu(r2, B)
            if t % 4 == 3:
                xs = ekf.get_state()[0]
                z = meas_near(r2, oracle.make_params(**dict(NOBIAS, direct_orien_method=direct)), xs)
                mask = (r2.uniform(size=B) < 0.5).astype(np.uint8)
                ekf.step(u, z, mask)
            else:
                ekf.step(u, None, None)
            if t == 20:
                xs, Ps = ekf.get_state()
                assert Ps.shape == (B, 9, 9)
                ekf.set_state(xs, Ps)                       # AoS fp64 round trip through either layout
                ekf.predict(rand_imu(r2, B))                # bare prediction_step
                z = meas_near(r2, oracle.make_params(**dict(NOBIAS, direct_orien_method=direct)), ekf.get_state()[0])
                ekf.update(z, (r2.uniform(size=B) < 0.3).astype(np.uint8))   # stand-alone correction_step
        xs, Ps = ekf.get_state()
        rep = ekf.report()
        out.append((xs, Ps, rep, ekf.algorithmic_bytes(0), ekf.algorithmic_bytes(1)))
        assert np.isfinite(xs).all() and np.isfinite(Ps).all()
        ekf.close()
    (xf, Pf, rf, b0f, b1f), (xc, Pc, rc, b0c, b1c) = out
    np.testing.assert_array_equal(xc, xf)
    np.testing.assert_array_equal(Pc, Pf)
    for k in rf:
        np.testing.assert_array_equal(rc[k], rf[k])
    wsz = 8 if dtype == "f64" else 4
    assert b0f == (136 + 6 + 136) * wsz * B and b0c == (64 + 6 + 64) * wsz * B
    assert b1f == (136 + 13 + 136) * wsz * B and b1c == (64 + 13 + 64) * wsz * B


@pytest.mark.parametrize("dtype,tol", [("f64", tp.F64U), ("f32", tp.F32U)])
def test_compact_records_against_the_oracle(dtype, tol, monkeypatch):
    """One teacher-forced fused tick on compact records against the dense 9-state oracle (the reference's own num_states = 9 algebra)."""
    rng = np.random.default_rng(11)
    B = 777
    po = oracle.make_params(**NOBIAS)
    x, P = rand_states(rng, B, 9, cov_scale=0.3)
    x[:, 10:16] = 0.0
    u = rand_imu(rng, B)
    xr, Pr, _ = oracle_predict_batch(po, x, P, u)
    z = meas_near(rng, po, xr)
    mask = (rng.uniform(size=B) < 0.6).astype(np.uint8)
    xr, Pr, _ = oracle_update_batch(po, xr, Pr, z, mask)
    ekf = _handle(B, dtype, monkeypatch, True)
    ekf.set_state(x, P)
    ekf.step(u, z, mask)
    xg, Pg = ekf.get_state()
    assert_state_close(xg, Pg, xr, Pr, **tol)
    ekf.close()


def test_nobias_goldens_on_compact_records(monkeypatch):
    """The reference twin's `nobias` fixtures (one-step vectors and the 140-560-tick sequences) through the compact layout."""
    monkeypatch.setenv("QLE_COMPACT", "1")
    monkeypatch.setenv("QLE_QUAD", "0")
    tp.test_golden_vectors_fp64("nobias")
    tp.test_golden_sequences("nobias", "f64")
    tp.test_golden_sequences("nobias", "f32")


def test_changing_est_bias_on_a_live_handle_converts_the_records(monkeypatch):
    """qle_set_params with the other est_bias on a handle that holds a state: the covariance part of every record changes layout
    in place (k_relayout_P); the pose block survives exactly, the bias blocks come back as zeros."""
    monkeypatch.setenv("QLE_QUAD", "0")
    monkeypatch.setenv("QLE_COMPACT", "1")
    rng = np.random.default_rng(2)
    B = 130
    x, P = rand_states(rng, B, 9, cov_scale=0.3)
    x[:, 10:16] = 0.0
    ekf = qla.BatchedRelativePoseEKF(B, "f64", params=qla.make_params(**NOBIAS))
    assert ekf.policy()["record_words"] == 64
    ekf.set_state(x, P)
    ekf.initialize_params(est_bias=1)                                  # qle_set_params: compact -> full
    assert ekf.policy()["record_words"] == 136
    x15, P15 = ekf.get_state()
    assert P15.shape == (B, 15, 15)
    np.testing.assert_array_equal(P15[:, :9, :9], 0.5 * (P + P.transpose(0, 2, 1)))
    assert np.abs(P15[:, 9:, :]).max() == 0.0 and np.abs(P15[:, :, 9:]).max() == 0.0
    ekf.initialize_params(est_bias=0)                                  # full -> compact
    assert ekf.policy()["record_words"] == 64
    x9, P9 = ekf.get_state()
    np.testing.assert_array_equal(P9, P15[:, :9, :9])
    np.testing.assert_array_equal(x9, x15)
    ekf.close()


def test_default_rule_picks_compact_records_above_4096_filters(monkeypatch):
    """Compact records wherever the lane-per-filter kernels serve every tick: above 4 096 filters, and since round 4 also on the small fp32
    batches whose cadence has a tag pose less often than every third tick (the cooperative kernel keeps only the frequent-correction
    cadences there, and every small fp64 batch)."""
    monkeypatch.delenv("QLE_COMPACT", raising=False)
    monkeypatch.delenv("QLE_QUAD", raising=False)
    for B, dtype, est_bias, mr, mfreq, want in ((32768, "f32", 0, 0, 30.0, 64), (8192, "f32", 0, 0, 30.0, 64), (32768, "f32", 1, 0, 30.0, 136),
                                                (32768, "f32", 0, 1, 30.0, 136), (4096, "f32", 0, 0, 30.0, 64), (4096, "f32", 0, 0, 400.0, 136),
                                                (4096, "f64", 0, 0, 30.0, 136), (4096, "f32", 1, 0, 30.0, 136)):
        ekf = qla.BatchedRelativePoseEKF(B, dtype, params=qla.make_params(**dict(NOBIAS, est_bias=est_bias, multirate_ekf=mr, measurement_freq=mfreq)))
        assert ekf.policy()["record_words"] == want, (B, dtype, est_bias, mr, mfreq)
        assert ekf.policy()["coop_ticks"] == (0 if B > 4096 or mr else (3 if dtype == "f64" else (1 if mfreq == 400.0 else 0))), (B, dtype, mfreq)
        ekf.close()


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_compact_records_seeding_gating_and_resident_runs_bit_for_bit(dtype, monkeypatch):
    """The paths the first test does not reach: `initialize_state` (k_seed), the gated `filter_update` tick, device-resident sequences
    (`run`) and the on-chip-resident variant (`run_resident`) -- compact against full records, bit for bit."""
    B = 300
    out = []
    for compact in (False, True):
        ekf = _handle(B, dtype, monkeypatch, compact, limit_measurement_freq=1, corner_margin_enbl=1)
        rng = np.random.default_rng(9)
        z0 = np.zeros((B, 7)); z0[:, 0:2] = rng.normal(size=(B, 2)) * 0.1; z0[:, 2] = rng.uniform(0.8, 2.0, size=B)
        z0[:, 3:7] = np.array([0.7071067811865476, -0.7071067811865476, 0.0, 0.0])
        mask = (np.arange(B) % 5 != 0).astype(np.uint8)        # every fifth filter stays uninitialised
        ekf.enable_gating(True)
        ekf.initialize_state(z0, reinit_bias=True, mask=mask)
        pending = np.zeros(B, np.uint8); zlast = z0.copy()
        for t in range(30):
            u = rand_imu(rng, B) * np.array([0.05, 0.05, 1, 0.2, 0.2, 0.2])
            if t % 3 == 0:
                xs = ekf.get_state()[0]
                xs[mask == 0, 9] = 1.0
                zlast = meas_near(rng, oracle.make_params(**NOBIAS), xs, ang=0.2, pos=0.05)
                pending[:] = mask
            ekf.filter_update(u, zlast if pending.any() else None, pending if pending.any() else None)
            perf, cons, _ = ekf.tick_flags()
            pending &= (1 - cons)
        T = 28
        thm = np.zeros(T, np.uint8); thm[6::7] = 1
        ekf.enable_gating(False)
        seq = ekf.make_inputs(T, thm)
        ekf.synth_generate(seq, seed=4)
        ekf.initialize_state(z0, reinit_bias=True, mask=mask)   # synth_generate seeds all filters: back to the masked population
        ekf.run(seq, 0, T)
        a = ekf.get_state()
        ekf.run_resident(seq, 0, T)
        b = ekf.get_state()
        out.append((a, b))
        assert np.isfinite(b[0]).all() and np.isfinite(b[1]).all()
        ekf.close()
    # The tick kernels (seeding, gated ticks, device-resident sequences): bit for bit in both dtypes.  The on-chip-resident kernel in fp32
    # agrees to rounding only (1.3e-7 absolute after these 28 ticks): the two instantiations are compiled separately, and once the dead bias
    # arithmetic is gone the backend fuses some multiply-adds of the in-place predict differently (a product whose second use died becomes
    # fusable).
    for which, ((xf, Pf), (xc, Pc)) in zip(("run", "run_resident"), zip(out[0], out[1])):
        if dtype == "f64" or which == "run":
            np.testing.assert_array_equal(xc, xf, err_msg=which)
            np.testing.assert_array_equal(Pc, Pf, err_msg=which)
        else:
            np.testing.assert_allclose(xc, xf, rtol=0, atol=2e-6, err_msg=which)
            np.testing.assert_allclose(Pc, Pf, rtol=0, atol=2e-6, err_msg=which)
