#!/usr/bin/env python3
"""Golden vectors for the C++-only branches that the reference's Python twin reaches BY IDENTITY.

Run in the BUILD container only (needs /root/reference, which never travels):

    python -B tests/golden/make_golden_branches.py        # writes tests/golden/branch_cases.npz

The twin (quad_state_estimation/test/rel_pose_EKF_test_class.py, imported unmodified through the shims of
make_golden.py) has no static biases, no dynamic measurement delay and a single-tag corner gate (SURVEY.md
Appendix B #2, #6, #8).  Three of the C++-only branches can nevertheless be pinned by numbers the twin produces:

  static biases      EKF.cpp:357-358 subtracts ab_static / wb_static next to the bias states.  The twin run with
                     ab + ab_static, wb + wb_static IN its bias states computes the same a_nom / w_nom, hence the same
                     x_check (bias words shifted by the statics), P_check and accel; the correction adds the same
                     delta to the bias states, so the identity holds along whole trajectories.  Stored: the twin's
                     outputs with the statics subtracted from the bias words again.
  dynamic delay      EKF.cpp:199-200: delay = min(t_curr - apriltag_time + offset, delay_max), step = int(delay/dT + 0.5).
                     With a uniform measurement age chosen so that the delay is the twin's fixed one (the twin's public
                     attribute measurement_step_delay, PYEKF.py:59-60,231) the C++ replays exactly the twin's history entry.
                     Stored: the stamps that give that age (and a second set that runs into the delay_max clamp).
  multi-tag gate     EKF.cpp:160-181 accepts a detection when ANY tag of the bundle projects inside the margins.  The twin
                     decides for ONE tag (its public attribute tag_corners, PYEKF.py:153-156,199-211); its decision is
                     taken per tag of the bundle and composed with `or` (first passing tag wins, as the C++ loop breaks).
                     Stored: the twin's per-tag decisions and the composition.

`hw__*` is one trajectory of the twin's full filter_update that needs all three at once: the parameter values of
config/relative_pose_EKF_hardware.yaml (100 Hz, a correction on every tick, 150 ms delay = 15 ticks, static biases,
the 13-tag bundle, camera_K / 640x480 / margin 0), multirate on and off.

What stays unpinned (no twin-side identity exists): the conventional linearisation (EKF.cpp:440-444,455-458) -- the twin
hard-codes the direct method -- and the flip of delta_q when it fires (EKF.cpp:449) -- the twin never normalises delta_q.

Only numbers are saved; the reference sources are never copied or modified.
"""
import math
import os
import sys
from types import SimpleNamespace as NS

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (shims + input helpers; importing it generates nothing)

OUT_DIR = os.path.dirname(os.path.abspath(__file__))

HW = dict(ab_static=[0.20, -0.09, -0.03], wb_static=[-0.02, -0.01, 0.0])
HW_TAG_WIDTHS = [0.08382] + [0.16764] * 4 + [0.33528] * 4 + [0.16764] * 4
HW_TAG_POS = [[0, 0, 0], [0, 0.1571625, 0], [0.1571625, 0, 0], [0, -0.1571625, 0], [-0.1571625, 0, 0],
              [-0.244475, 0.244475, 0], [0.244475, 0.244475, 0], [0.244475, -0.244475, 0], [-0.244475, -0.244475, 0],
              [0, 0.314325, 0], [0.314325, 0, 0], [0, -0.314325, 0], [-0.314325, 0, 0]]
HW_K = [[437.3412312213781, 0, 328.5442810236917], [0, 438.0867474272743, 239.2536470406629], [0, 0, 1]]
HW_W, HW_H, HW_MARGIN = 640, 480, 0.0
HW_STEP_DELAY = 15          # 0.150 s at 100 Hz (HW.yaml:3,5)
HW_DELAY, HW_OFFSET, HW_DELAY_MAX = 0.150, 0.085, 0.350


def imu_msg(u):
    return NS(linear_acceleration=NS(x=float(u[0]), y=float(u[1]), z=float(u[2])),
              angular_velocity=NS(x=float(u[3]), y=float(u[4]), z=float(u[5])))


def tag_msg(z):
    return NS(detections=[NS(pose=NS(pose=NS(pose=NS(position=NS(x=float(z[0]), y=float(z[1]), z=float(z[2])),
                                                     orientation=NS(x=float(z[3]), y=float(z[4]), z=float(z[5]), w=float(z[6]))))))])


def tag_corners(width, pos):
    """The twin's tag_corners attribute (PYEKF.py:153-156) for a tag of the bundle: same corner order as EKF.cpp:163-166."""
    h = width / 2
    return np.array([[h + pos[0], -h + pos[0], -h + pos[0], h + pos[0]],
                     [h + pos[1], h + pos[1], -h + pos[1], -h + pos[1]],
                     [0, 0, 0, 0], [1, 1, 1, 1]], dtype=np.float64)


def hw_twin(pyekf, pyqh, tft, measurement_freq=100.0):
    """The twin with the public parameters of HW.yaml (those it has)."""
    p = mg.PARAM_SETS["hardware"]
    f = pyekf.RelativePoseEKF(p["update_freq"], measurement_freq)
    f.Q = np.diag(np.asarray(p["Q"], dtype=np.float64))
    f.R = np.diag(np.asarray(p["R"], dtype=np.float64))
    f.r_v_cv = np.asarray(p["r_v_cv"], dtype=np.float64).reshape(3, 1)
    f.q_vc = pyqh.quaternion_norm(np.asarray(p["q_vc"], dtype=np.float64))
    f.C_vc = tft.quaternion_matrix(f.q_vc)[0:3, 0:3]
    f.camera_K = np.array(HW_K)
    f.camera_width, f.camera_height, f.tag_in_view_margin = HW_W, HW_H, HW_MARGIN
    f.measurement_step_delay = HW_STEP_DELAY
    return f


class GateProbe:
    """The twin's own corner-gate decision for ONE tag: a second twin instance whose filter_update is driven up to the
    decision; correction_step is replaced by a recorder, so `decided` says whether the twin chose to correct."""

    def __init__(self, pyekf, pyqh, tft):
        self.f = hw_twin(pyekf, pyqh, tft)
        self.f.multirate_EKF = False
        self.calls = 0
        self.f.correction_step = self._record

    def _record(self, x_check, P_check, r_c_tc, q_ct):
        self.calls += 1
        return x_check, P_check

    def decide(self, z, corners):
        f = self.f
        f.tag_corners = corners
        f.state_initialized = True
        f.filter_run_once = True
        f.upds_since_correction = 1000
        f.measurement_ready = True
        f.apriltag_msg = tag_msg(z)
        f.IMU_msg = imu_msg(np.zeros(6))
        f.r_nom = np.zeros((3, 1)); f.v_nom = np.zeros((3, 1)); f.q_nom = np.array([0.0, 0.0, 0.0, 1.0])
        f.ab_nom = np.zeros((3, 1)); f.wb_nom = np.zeros((3, 1)); f.cov_pert = f.cov_init.copy()
        n0 = self.calls
        f.filter_update()
        assert not f.measurement_ready
        return self.calls > n0


def main():
    pyqh, pyekf, tft = mg._load_reference()
    rng = np.random.default_rng(0xE4F0B)
    out = {}
    ab_s, wb_s = np.array(HW["ab_static"]), np.array(HW["wb_static"])

    # ---------------- static biases: single predicts (EKF.cpp:357-358) ----------------
    f = mg.make_filter(pyekf, pyqh, tft, "hardware")
    N = 48
    X = np.zeros((N, 16)); U = np.zeros((N, 6)); P = np.zeros((N, 15, 15))
    Xo = np.zeros((N, 16)); Po = np.zeros((N, 15, 15)); Ao = np.zeros((N, 3))
    scale = [0.1] * 3 + [0.1] * 3 + [0.15] * 3 + [0.5] * 3 + [0.1] * 3
    for i in range(N):
        X[i] = mg.rand_state(rng, True)
        U[i, 0:3] = rng.normal(size=3) * 1.5 + np.array([0, 0, 9.8])
        U[i, 3:6] = rng.normal(size=3) * 0.4
        if i == 0:   # w - wb - wb_static == 0 exactly: the small-angle branch with statics in play
            U[i, 3:6] = X[i, 13:16] + wb_s
        P[i] = mg.rand_spd(rng, 15, scale) if i % 4 else np.diag(scale)
        xt = X[i].copy(); xt[10:13] += ab_s; xt[13:16] += wb_s          # the identity: statics inside the twin's bias states
        xc, Pc, acc = f.prediction_step(xt.reshape(16, 1), U[i].reshape(6, 1), P[i].copy())
        xc = xc.flatten(); xc[10:13] -= ab_s; xc[13:16] -= wb_s
        Xo[i] = xc; Po[i] = Pc; Ao[i] = acc.flatten()
    out["static"] = dict(x=X, u=U, P=P, x_check=Xo, P_check=Po, accel=Ao, ab_static=ab_s, wb_static=wb_s)

    # ---------------- multi-tag gate: the twin's decision per tag, composed (EKF.cpp:160-181) ----------------
    probe = GateProbe(pyekf, pyqh, tft)
    corners = [tag_corners(w, p) for w, p in zip(HW_TAG_WIDTHS, HW_TAG_POS)]
    N = 320
    Zg = np.zeros((N, 7)); per_tag = np.zeros((N, 13), dtype=np.uint8)
    for i in range(N):
        # vehicle above the bundle at 0.25 .. 3 m, laterally up to ~1.2 heights off-axis, tilted up to 0.5 rad
        h = 10 ** rng.uniform(math.log10(0.25), math.log10(3.0))
        r_t = np.array([rng.normal() * 0.45 * h, rng.normal() * 0.45 * h, h])
        q_t = mg.rand_unit_quat(rng, 0.5)
        r_c, q_ct = mg.meas_from_pose(probe.f, tft, r_t, q_t, rng, 0.01, 0.01)
        Zg[i, 0:3] = r_c; Zg[i, 3:7] = q_ct
        for k in range(13):
            per_tag[i, k] = probe.decide(Zg[i], corners[k])
    composed = per_tag.any(axis=1).astype(np.uint8)
    print("gate: accepted", int(composed.sum()), "of", N, "| accepted by tag 0 alone", int(per_tag[:, 0].sum()),
          "| by some other tag only", int((composed & (1 - per_tag[:, 0])).sum()))
    out["gate"] = dict(z=Zg, per_tag=per_tag, composed=composed, tag_widths=np.array(HW_TAG_WIDTHS),
                       tag_positions=np.array(HW_TAG_POS).reshape(-1), camera_K=np.array(HW_K).reshape(-1),
                       camera_size=np.array([HW_W, HW_H]), margin=np.array([HW_MARGIN]))

    # ---------------- HW.yaml-like full filter_update: statics + 15-tick delay + 13-tag bundle, no rate limit ----------------
    for name, mr in (("hw_multirate", True), ("hw_singlerate", False)):
        f = hw_twin(pyekf, pyqh, tft)
        assert f.upd_per_meas == 1
        f.multirate_EKF = mr
        T = 240
        U = np.zeros((T, 6)); Z = np.zeros((T, 7)); NEW = np.zeros(T, dtype=np.uint8)
        Xs = np.zeros((T, 16)); Pd = np.zeros((T, 15)); UPD = np.zeros(T, dtype=np.int32); RDY = np.zeros(T, dtype=np.uint8)
        HL = np.zeros(T, dtype=np.int32); PERF = np.zeros(T, dtype=np.uint8); TAG = np.full(T, -1, dtype=np.int32)
        r0 = np.array([0.05, -0.03, 1.1]); A = np.array([0.35, 0.3, 0.45]); om = np.array([0.9, 0.7, 0.5]); ph = np.array([0.3, 1.1, 2.0])
        q_t = mg.rand_unit_quat(rng, 0.1)
        g = np.array([0, 0, -9.8])
        ab_t = rng.normal(size=3) * 0.05; wb_t = rng.normal(size=3) * 0.005    # estimated biases (on top of the statics)
        # seed: the node initialises on the first detection (NODE.cpp:169-174), reinit_bias -> zero bias states
        z0 = np.concatenate(mg.meas_from_pose(f, tft, r0 + A * np.sin(ph), q_t, rng, 0.01, 0.005))
        f.apriltag_msg = tag_msg(z0)
        f.initialize_state(True)
        f.measurement_ready = False          # consumed by the seeding; the first tick is a predict-only tick in both implementations
        f.ab_nom = ab_s.reshape(3, 1).copy(); f.wb_nom = wb_s.reshape(3, 1).copy()      # the identity (statics in the bias states)
        f.x_hist[0][10:13, 0] = ab_s; f.x_hist[0][13:16, 0] = wb_s
        P_full = {}
        q_hist = []                          # truth attitude per tick (the camera saw the tag HW_STEP_DELAY ticks ago)
        for t in range(T):
            tt = t * f.dT
            q_hist.append(q_t.copy())
            acc_t = -A * om * om * np.sin(om * tt + ph)
            w_t = np.array([0.08, -0.06, 0.1]) * np.sin(np.array([0.8, 1.1, 0.6]) * tt)
            C_t = tft.quaternion_matrix(q_t)[0:3, 0:3]
            U[t, 0:3] = C_t.T @ (acc_t - g) + ab_s + ab_t + rng.normal(size=3) * 0.02
            U[t, 3:6] = w_t + wb_s + wb_t + rng.normal(size=3) * 0.01
            f.IMU_msg = imu_msg(U[t])
            if t >= 2 and rng.uniform() < 0.6:
                # the camera saw the tag HW_STEP_DELAY ticks ago (truth of that time), the detection arrives now
                td = max(tt - HW_DELAY, 0.0)
                r_c, q_ct = mg.meas_from_pose(f, tft, r0 + A * np.sin(om * td + ph), q_hist[max(t - HW_STEP_DELAY, 0)], rng, 0.01, 0.005)
                if rng.uniform() < 0.15:
                    r_c = r_c + np.array([2.5, 0.0, 0.0])      # far off-axis: no tag of the bundle is in view
                Z[t, 0:3] = r_c; Z[t, 3:7] = q_ct; NEW[t] = 1
                f.apriltag_msg = tag_msg(Z[t])
                f.measurement_ready = True
            if f.measurement_ready:
                # compose the bundle's decision from the twin's per-tag decisions; the filter then decides by its own code
                # on the first passing tag (or on tag 0, which fails, when none passes)
                zc = np.array([f.apriltag_msg.detections[0].pose.pose.pose.position.x, f.apriltag_msg.detections[0].pose.pose.pose.position.y,
                               f.apriltag_msg.detections[0].pose.pose.pose.position.z, f.apriltag_msg.detections[0].pose.pose.pose.orientation.x,
                               f.apriltag_msg.detections[0].pose.pose.pose.orientation.y, f.apriltag_msg.detections[0].pose.pose.pose.orientation.z,
                               f.apriltag_msg.detections[0].pose.pose.pose.orientation.w])
                dec = [probe.decide(zc, c) for c in corners]
                first = dec.index(True) if any(dec) else 0
                f.tag_corners = corners[first]
                TAG[t] = first if any(dec) else -1
            f.filter_update()
            q_t = tft.quaternion_multiply(q_t, pyqh.quaternion_exp(f.dT * w_t)); q_t /= np.linalg.norm(q_t)
            x = np.concatenate([f.r_nom.flatten(), f.v_nom.flatten(), f.q_nom.flatten(), f.ab_nom.flatten(), f.wb_nom.flatten()])
            x[10:13] -= ab_s; x[13:16] -= wb_s
            Xs[t] = x
            Pd[t] = np.diag(f.cov_pert); UPD[t] = f.upds_since_correction; RDY[t] = 1 if f.measurement_ready else 0
            HL[t] = len(f.x_hist)
            if t in (20, T // 2, T - 1):
                P_full[t] = f.cov_pert.copy()
        PERF = (UPD == 0).astype(np.uint8)
        assert (TAG[PERF == 1] >= 0).all() and (RDY == 0).all()
        ticks = sorted(P_full)
        out[name] = dict(z0=z0, u=U, z=Z, new=NEW, x_seq=Xs, P_diag_seq=Pd, upds=UPD, perf=PERF, hist_len=HL, tag=TAG,
                         P_full_ticks=np.array(ticks), P_full=np.stack([P_full[t] for t in ticks]),
                         # measurement stamps as a function of the tick's t_curr: age such that the dynamic delay is the twin's fixed one
                         age_exact=np.array([HW_DELAY - HW_OFFSET]), age_clamped=np.array([0.30]))
        print(name, "corrections:", int(PERF.sum()), "rejected by the gate:", int(((NEW == 1) & (PERF == 0)).sum()),
              "max history:", int(HL.max()), "corrections decided by a tag other than tag 0:", int((TAG > 0).sum()))
    np.savez_compressed(os.path.join(OUT_DIR, "branch_cases.npz"), **{f"{n}__{k}": v for n, d in out.items() for k, v in d.items()})
    print("branch fixtures written to", os.path.join(OUT_DIR, "branch_cases.npz"))


if __name__ == "__main__":
    main()
