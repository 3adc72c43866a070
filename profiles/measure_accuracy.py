#!/usr/bin/env python3
"""Measured engine-vs-oracle deviations (fp64 oracle on identical inputs), default kernel policy and the workgroup-cooperative
kernel forced; prints the table (profiles/r02_accuracy.md)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import quadrotor_landing_amd as qla  # noqa: E402
from util import golden_kwargs, meas_near, oracle_predict_batch, oracle_update_batch, quat_err, rand_imu, rand_states  # noqa: E402

rows = []


def dev(xg, Pg, xr, Pr):
    keep = [i for i in range(16) if not 6 <= i < 10]
    dx = np.abs(xg[:, keep] - xr[:, keep]).max()
    dq = quat_err(xg[:, 6:10], xr[:, 6:10])
    sc = np.sqrt(np.einsum("bii->bi", Pr)[:, :, None] * np.einsum("bii->bi", Pr)[:, None, :])
    dP = (np.abs(Pg - Pr) / sc).max()
    dF = (np.linalg.norm(Pg - Pr, axis=(1, 2)) / np.linalg.norm(Pr, axis=(1, 2))).max()
    return dx, dq, dP, dF


kw = golden_kwargs("rotors400")
po = oracle.make_params(**kw); pq = qla.make_params(**kw)
rng = np.random.default_rng(1)
B = 4096
x, P = rand_states(rng, B, 15, cov_scale=0.3)
u = rand_imu(rng, B)
xr, Pr, _ = oracle_predict_batch(po, x, P, u)
z = meas_near(rng, po, xr)
xr2, Pr2, _ = oracle_update_batch(po, xr, Pr, z)
xs, Ps = oracle.run_batch(po, x, P, u[None], z[None], np.ones((1, B), np.uint8))
for fam, env in (("one lane per filter", "0"), ("workgroup-cooperative kernel", "3")):
    os.environ["QLE_QUAD"] = env
    for dtype in ("f64", "f32"):
        ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
        xg, Pg, _ = ekf.prediction_step(x, P, u)
        rows.append((f"{dtype} one predict, {B} random states, {fam}",) + dev(xg, Pg, xr, Pr))
        if env == "0":
            xg2, Pg2 = ekf.correction_step(xr, Pr, z[:, :3], z[:, 3:])
            rows.append((f"{dtype} one update, {B} random states, {fam}",) + dev(xg2, Pg2, xr2, Pr2))
        ekf.set_state(x, P)
        ekf.step(u, z, None)
        xg3, Pg3 = ekf.get_state()
        rows.append((f"{dtype} one fused tick (predict + update), {B} random states, {fam}",) + dev(xg3, Pg3, xs, Ps))
        ekf.close()
os.environ.pop("QLE_QUAD")
for dtype, T in (("f64", 1400), ("f32", 1400), ("f32", 4060)):
    B = 2048
    thm = np.zeros(T, np.uint8); thm[13::14] = 1
    ekf = qla.BatchedRelativePoseEKF(B, dtype, params=pq)
    seq = ekf.make_inputs(T, thm)
    ekf.synth_generate(seq, seed=0xE4F00003)
    x0, P0 = ekf.get_state()
    U = np.empty((T, B, 6)); Z = np.zeros((T, B, 7)); M = np.zeros((T, B), np.uint8)
    for t in range(T):
        U[t], Z[t], M[t] = seq.download_tick(t)
    ekf.run(seq, 0, T)
    xg, Pg = ekf.get_state()
    xr, Pr = oracle.run_batch(po, x0, P0, U, Z, M)
    rows.append((f"{dtype} free run, {T} ticks of cfg3 (400 Hz predict, 30 Hz update), {B} filters",) + dev(xg, Pg, xr, Pr))
    er = ekf.synth_rmse(seq)
    rows.append((f"   (filter error vs truth for scale: position RMSE {np.sqrt(er[0]/er[2]):.3f} m, attitude RMSE {np.sqrt(er[1]/er[2]):.3f} rad)", "", "", "", ""))
    ekf.close()
out = ["# Measured engine-vs-oracle deviations, round 3 (MI355X), head " + os.environ.get("QLE_HEAD_SHA", "unknown"), "",
       "`python profiles/measure_accuracy.py` — engine through the C-ABI vs the fp64 CPU oracle on identical inputs.", "",
       "| case | max abs dev, state (r, v, biases) | quaternion (sign-insensitive) | max |dP_ij| / sqrt(P_ii P_jj) | max rel Frobenius dP |",
       "|---|---|---|---|---|"]
for r in rows:
    out.append("| " + r[0] + " | " + " | ".join(f"{v:.2e}" if v != "" else "" for v in r[1:]) + " |")
out += ["", "Stated test tolerances (tests/test_gpu_parity.py, each next to its measured deviation in tests/tolerances.md): fp64 per step 1e-12 / free run 1e-9;",
        "fp32 per predict 5e-7 (state) / 3e-6 (covariance), per update or fused tick 5e-6 / 3e-4, free runs 2e-5 ... 6e-5 -- every one within ~10x of its",
        "measurement.  The fp32 free-run deviation is four orders of magnitude below the filter's own estimation error.", ""]
print("\n".join(out))
