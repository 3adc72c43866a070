"""numpy restatement of the engine's synthetic source (quadrotor_landing_amd/csrc/synth_kernels.hpp: k_synth) -- TEST INFRASTRUCTURE.

The reference has no synthetic source (its inputs are the ROS topics /drone/imu and /tag_detections, NODE.cpp:144-176, and
Gazebo's ground truth, test/tf_extractor_node.py:26-63); the engine replaces them with a seeded, counter-based generator
that runs on the device.  This file restates that generator in vectorised numpy so that the test suite can check what the
device produced -- the IMU and tag-pose sequences, the seeding pose, the per-filter parameters of BASELINE cfg 5 and the
truth kept for the RMSE -- against a host computation (tests/test_gpu_population.py) instead of by properties only.
Same formulas in the same order, fp64; the device's sin / cos / log / sqrt / pow are not correctly rounded, so agreement is
to a few ulp, not to the bit.  Never imported by the product.
"""
import numpy as np

U64 = np.uint64
_M1, _M2 = U64(0xBF58476D1CE4E5B9), U64(0x94D049BB133111EB)
_K_F, _K_T, _K_C = U64(0x9E3779B97F4A7C15), U64(0xD1B54A32D192ED03), U64(0x8CB92BA72F3D8DD7)
TICK_STATIC = 0xFFFFFFFFFFFFFFF0
TICK_SEED_MEAS = 0xFFFFFFFFFFFFFFF1
MAX_DELAY = 40


def mix64(z):
    z = np.asarray(z, dtype=U64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> U64(30))) * _M1
        z = (z ^ (z >> U64(27))) * _M2
    return z ^ (z >> U64(31))


def rng_uniform(seed, filt, tick, channel):
    """Counter-based uniform in (0, 1): key = (seed, filter, tick, channel); filt is an array of global filter indices."""
    with np.errstate(over="ignore"):
        h = mix64(U64(seed) + _K_F * (np.asarray(filt, dtype=U64) + U64(1)))
        h = mix64(h ^ (_K_T * (U64(tick) + U64(1))))
        h = mix64(h ^ (_K_C * (U64(channel) + U64(1))))
    return ((h >> U64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def rng_normal(seed, filt, tick, channel):
    u1 = rng_uniform(seed, filt, tick, 2 * channel)
    u2 = rng_uniform(seed, filt, tick, 2 * channel + 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586 * u2)


# ---- quaternion helpers on [B, 4] arrays (x, y, z, w), as ekf_device.hpp has them
def _qnorm(q):  # QH.cpp:61-73
    n = np.sqrt((q * q).sum(1))
    inv = 1.0 / n
    s = np.where(q[:, 3] * inv < -0.75, -inv, inv)
    return q * s[:, None]


def _qexp(v):  # QH.cpp:9-33
    n = np.sqrt((v * v).sum(1))
    small = n < 1e-10
    k = np.where(small, 0.5 * (1.0 - n * n / 24.0), np.sin(0.5 * n) / np.where(small, 1.0, n))
    return _qnorm(np.concatenate([v * k[:, None], np.cos(0.5 * n)[:, None]], axis=1))


def _qmul(a, b):
    ax, ay, az, aw = a.T
    bx, by, bz, bw = b.T
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=1)


def _rot(q):  # Eigen toRotationMatrix, [B, 3, 3]
    x, y, z, w = q.T
    tx, ty, tz = x + x, y + y, z + z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    C = np.empty((q.shape[0], 3, 3))
    C[:, 0, 0] = 1.0 - (tyy + tzz); C[:, 0, 1] = txy - twz; C[:, 0, 2] = txz + twy
    C[:, 1, 0] = txy + twz; C[:, 1, 1] = 1.0 - (txx + tzz); C[:, 1, 2] = tyz - twx
    C[:, 2, 0] = txz - twy; C[:, 2, 1] = tyz + twx; C[:, 2, 2] = 1.0 - (txx + tyy)
    return C


def _measure(cfg, gi, tick, r, q, R):
    """Tag pose in the camera frame from the truth pose (synth_measure): inverse of the observation model of EKF.cpp:431-438
    plus N(0, R) noise in the measurement frame."""
    C = _rot(q)
    t = -np.einsum("bji,bj->bi", C, r) - cfg["r_v_cv"]                       # -(C^T r) - r_v_cv
    z = np.empty((r.shape[0], 7))
    Cvc = cfg["C_vc"]
    for k in range(3):
        z[:, k] = (Cvc[0, k] * t[:, 0] + Cvc[1, k] * t[:, 1] + Cvc[2, k] * t[:, 2]) + \
            cfg["meas_scale"] * np.sqrt(R[k]) * rng_normal(cfg["seed"], gi, tick, 50 + k)
    qvc_c = np.tile(np.array([-cfg["q_vc"][0], -cfg["q_vc"][1], -cfg["q_vc"][2], cfg["q_vc"][3]]), (r.shape[0], 1))
    q_c = q * np.array([-1.0, -1.0, -1.0, 1.0])
    qct = _qmul(qvc_c, q_c)
    nv = np.stack([cfg["meas_scale"] * np.sqrt(R[3 + k]) * rng_normal(cfg["seed"], gi, tick, 53 + k) for k in range(3)], axis=1)
    qn = _qmul(qct, _qexp(nv))
    z[:, 3:] = qn / np.sqrt((qn * qn).sum(1))[:, None]
    return z


def generate(p, B, tick_has_meas, seed, filter_offset=0, perturb_filter_params=False, ab_true_sigma=0.1, wb_true_sigma=0.01,
             meas_noise_scale=1.0, imu_noise_scale=1.0, meas_delay_ticks=0, view_scale=1.0):
    """What qle_synth_generate leaves on the device, for the orc_params `p` (oracle.make_params): a dict with
    u [T, B, 6], z [slots, B, 7] (one entry per tick whose tick_has_meas is set, in order), z0 [B, 7] (seeding tag pose),
    pfp [B, 24] (per-filter Q, static biases, R) or None, truth [B, 7], truth_bias [B, 6]."""
    thm = np.asarray(tick_has_meas, dtype=bool)
    T = thm.shape[0]
    gi = np.arange(B, dtype=np.int64) + int(filter_offset)
    cfg = dict(seed=int(seed), meas_scale=float(meas_noise_scale), r_v_cv=np.array(p.r_v_cv[:3]), q_vc=np.array(p.q_vc[:4]),
               C_vc=np.array(p.C_vc[:9]).reshape(3, 3))
    Q = np.array(p.Q[:12]); R = np.array(p.R[:6]); g = np.array(p.g[:3]); dT = float(p.dT_nom)
    delay = min(max(int(meas_delay_ticks), 0), MAX_DELAY)

    def U(ch, lo, hi):
        return lo + (hi - lo) * rng_uniform(seed, gi, TICK_STATIC, ch)

    def Nrm(ch):
        return rng_normal(seed, gi, TICK_STATIC, 100 + ch)

    vs = float(view_scale) if 0.0 < view_scale <= 1.0 else 1.0   # < 1: lateral offsets / amplitudes and attitude excursions shrunk
    r0 = np.stack([vs * U(0, -1, 1), vs * U(1, -1, 1), U(2, 1, 4)], axis=1)
    A = np.stack([(vs if k < 2 else 1.0) * U(3 + k, 0, 0.5) for k in range(3)], axis=1)
    om = np.stack([U(6 + k, 0.2, 1.5) for k in range(3)], axis=1)
    ph = np.stack([U(9 + k, 0, 6.283185307179586) for k in range(3)], axis=1)
    wa = np.stack([vs * U(12 + k, 0, 0.3) for k in range(3)], axis=1)
    wo = np.stack([U(15 + k, 0.2, 1.5) for k in range(3)], axis=1)
    wp = np.stack([U(18 + k, 0, 6.283185307179586) for k in range(3)], axis=1)
    if p.est_bias:
        ab = np.stack([ab_true_sigma * Nrm(k) for k in range(3)], axis=1)
        wb = np.stack([wb_true_sigma * Nrm(3 + k) for k in range(3)], axis=1)
    else:
        ab = np.zeros((B, 3)); wb = np.zeros((B, 3))
    q = _qexp(np.stack([vs * 0.2 * Nrm(6), vs * 0.2 * Nrm(7), vs * 0.2 * Nrm(8)], axis=1))

    # per-filter filter parameters (BASELINE cfg 5): Q groups scaled by 10^U(-0.5, 0.5), static biases ~ N(0, 0.1^2), N(0, 0.01^2)
    Qf = np.tile(Q, (B, 1)); abs_ = np.tile(np.array(p.ab_static[:3]), (B, 1)); wbs_ = np.tile(np.array(p.wb_static[:3]), (B, 1))
    pfp = None
    if perturb_filter_params:
        for grp in range(4):
            sc = np.power(10.0, rng_uniform(seed, gi, TICK_STATIC, 200 + grp) - 0.5)
            Qf[:, 3 * grp:3 * grp + 3] *= sc[:, None]
        abs_ = np.stack([0.1 * rng_normal(seed, gi, TICK_STATIC, 210 + k) for k in range(3)], axis=1)
        wbs_ = np.stack([0.01 * rng_normal(seed, gi, TICK_STATIC, 213 + k) for k in range(3)], axis=1)
        pfp = np.concatenate([Qf, abs_, wbs_, np.tile(R, (B, 1))], axis=1)

    r = r0 + A * np.sin(ph)
    z0 = _measure(cfg, gi, TICK_SEED_MEAS, r, q, R)
    N = MAX_DELAY + 1
    rh = np.tile(r[None], (N, 1, 1)); qh = np.tile(q[None], (N, 1, 1))   # entry (t+1) % N = pose after tick t
    u = np.empty((T, B, 6)); zs = []
    for t in range(T):
        tt = float(t) * dT
        acc = -A * om * om * np.sin(om * tt + ph) - g
        w = wa * np.sin(wo * tt + wp)
        C = _rot(q)
        for k in range(3):
            u[t, :, k] = (C[:, 0, k] * acc[:, 0] + C[:, 1, k] * acc[:, 1] + C[:, 2, k] * acc[:, 2]) + ab[:, k] + abs_[:, k] + \
                imu_noise_scale * np.sqrt(Q[k]) * rng_normal(seed, gi, t, k)
            u[t, :, 3 + k] = w[:, k] + wb[:, k] + wbs_[:, k] + imu_noise_scale * np.sqrt(Q[3 + k]) * rng_normal(seed, gi, t, 3 + k)
        q = _qnorm(_qmul(q, _qexp(dT * w)))        # exact exponential map with the rate held over the tick
        r = r0 + A * np.sin(om * (tt + dT) + ph)
        e = (t + 1) % N
        rh[e] = r; qh[e] = q
        if thm[t]:
            tm = max(t - delay, -1)
            e = (tm + 1) % N
            zs.append(_measure(cfg, gi, t, rh[e], qh[e], R))
    return dict(u=u, z=np.stack(zs) if zs else np.zeros((0, B, 7)), z0=z0, pfp=pfp, truth=np.concatenate([r, q], axis=1),
                truth_bias=np.concatenate([ab, wb], axis=1))
