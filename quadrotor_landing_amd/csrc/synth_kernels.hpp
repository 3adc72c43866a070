// synth_kernels.hpp -- device-side synthetic truth + IMU + tag-pose generator
// and the per-device error reduction.  Replaces the ROS inputs of the
// reference node (IMU on /drone/imu, NODE.cpp:144-151; AprilTag detections on
// /tag_detections, NODE.cpp:153-176) and the Gazebo ground truth of
// test/tf_extractor_node.py:26-63 with a seeded, sharding-invariant source
// (SURVEY.md section 8(d) "synthetic inputs").
//
// Every random value is a pure function of (seed, GLOBAL filter index, tick,
// channel) through a counter-based integer hash, so a batch split over any
// number of devices sees identical data.  Generation runs in fp64 and is cast
// to the compute dtype on store.  Not on the timed hot path.
#pragma once

#include "ekf_kernels.hpp"

namespace qle {

struct SynthArgs {
    uint64_t seed;
    int64_t filter_offset;
    double ab_sigma, wb_sigma, meas_scale, imu_scale;
    int32_t perturb, est_bias;
    int32_t meas_delay_ticks, _pad;   // tag pose delivered at tick t was taken meas_delay_ticks-1 ticks earlier (multirate runs)
    double dT, view_scale;
    double Q[12], R[6], g[3], r_v_cv[3], q_vc[4], C_vc[9], ab_static[3], wb_static[3];
    int64_t T, B, pitch_u_words, pitch_z_words;
};

// splitmix64 finaliser
__host__ __device__ inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
// counter-based uniform in (0,1): key = (seed, filter, tick, channel)
__host__ __device__ inline double rng_uniform(uint64_t seed, uint64_t filter, uint64_t tick, uint64_t channel)
{
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ULL * (filter + 1));
    h = mix64(h ^ (0xD1B54A32D192ED03ULL * (tick + 1)));
    h = mix64(h ^ (0x8CB92BA72F3D8DD7ULL * (channel + 1)));
    return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
// standard normal pair -> one value (Box-Muller on channels c, c+1)
__device__ inline double rng_normal(uint64_t seed, uint64_t filter, uint64_t tick, uint64_t channel)
{
    double u1 = rng_uniform(seed, filter, tick, 2 * channel);
    double u2 = rng_uniform(seed, filter, tick, 2 * channel + 1);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

constexpr int kSynthMaxDelay = 40;
constexpr uint64_t kTickStatic = 0xFFFFFFFFFFFFFFF0ULL;  // per-filter constants
constexpr uint64_t kTickSeedMeas = 0xFFFFFFFFFFFFFFF1ULL; // noise of the seeding tag pose

struct SynthTruth {
    double r0[3], A[3], om[3], ph[3];   // position r(t) = r0 + A sin(om t + ph)
    double wa[3], wo[3], wp[3];         // body rate w(t) = wa sin(wo t + wp)
    double ab[3], wb[3];                // true (unknown) IMU biases
    double q[4];                        // attitude q_tv at the current time
};

__device__ inline void synth_static(const SynthArgs& a, uint64_t gi, SynthTruth& s)
{
    auto U = [&](uint64_t ch, double lo, double hi) { return lo + (hi - lo) * rng_uniform(a.seed, gi, kTickStatic, ch); };
    auto Nrm = [&](uint64_t ch) { return rng_normal(a.seed, gi, kTickStatic, 100 + ch); };
    const double vs = a.view_scale;   // < 1: lateral offsets / amplitudes and attitude excursions shrunk (the tag stays in the image)
    s.r0[0] = vs * U(0, -1, 1); s.r0[1] = vs * U(1, -1, 1); s.r0[2] = U(2, 1, 4);
    for (int k = 0; k < 3; ++k) {
        s.A[k] = (k < 2 ? vs : 1.0) * U(3 + k, 0, 0.5);
        s.om[k] = U(6 + k, 0.2, 1.5);
        s.ph[k] = U(9 + k, 0, 6.283185307179586);
        s.wa[k] = vs * U(12 + k, 0, 0.3);
        s.wo[k] = U(15 + k, 0.2, 1.5);
        s.wp[k] = U(18 + k, 0, 6.283185307179586);
        s.ab[k] = a.est_bias ? a.ab_sigma * Nrm(k) : 0.0;
        s.wb[k] = a.est_bias ? a.wb_sigma * Nrm(3 + k) : 0.0;
    }
    double v[3] = {vs * 0.2 * Nrm(6), vs * 0.2 * Nrm(7), vs * 0.2 * Nrm(8)};
    quat_exp<double>(v, s.q);
}

// Tag pose in the camera frame from the truth pose: inverse of the observation
// model of EKF.cpp:431-438, q_ct = conj(q_vc) (x) conj(q_tv),
// r_c_tc = C_vc^T (-C_tv^T r - r_v_cv), plus N(0,R) noise in the measurement frame.
__device__ inline void synth_measure(const SynthArgs& a, uint64_t gi, uint64_t tick, const double (&r)[3], const double (&q)[4],
                                     const double (&Rm)[6], double (&z)[7])
{
    double C[9];
    quat_to_rot<double>(q, C);
    double t[3];
    for (int k = 0; k < 3; ++k) t[k] = -(C[k] * r[0] + C[3 + k] * r[1] + C[6 + k] * r[2]) - a.r_v_cv[k];
    for (int k = 0; k < 3; ++k)
        z[k] = (a.C_vc[k] * t[0] + a.C_vc[3 + k] * t[1] + a.C_vc[6 + k] * t[2]) +
               a.meas_scale * sqrt(Rm[k]) * rng_normal(a.seed, gi, tick, 50 + k);
    double qvc_c[4] = {-a.q_vc[0], -a.q_vc[1], -a.q_vc[2], a.q_vc[3]};
    double q_c[4] = {-q[0], -q[1], -q[2], q[3]};
    double qct[4], dq[4], qn[4];
    quat_mul<double>(qvc_c, q_c, qct);
    double nv[3];
    for (int k = 0; k < 3; ++k) nv[k] = a.meas_scale * sqrt(Rm[3 + k]) * rng_normal(a.seed, gi, tick, 53 + k);
    quat_exp<double>(nv, dq);
    quat_mul<double>(qct, dq, qn);
    double n = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int k = 0; k < 4; ++k) z[3 + k] = qn[k] / n;
}

// One lane per filter, sequential over the T ticks of the sequence.
//   us  : [T] ticks of 6-word IMU records (pitch_u_words between ticks)
//   zs  : measurement slots of 8-word records (7 words + mask word = 1)
//   z0  : seeding tag pose at time 0 (8-word record) for initialize_state
//   pfp : per-filter parameter record (cfg 5) or nullptr
//   truth / truth_bias : AoS fp64 [B][7] / [B][6] at the end of the sequence
template <typename T>
__global__ void k_synth(SynthArgs a, const int32_t* __restrict__ slot, T* __restrict__ us, T* __restrict__ zs, T* __restrict__ z0,
                        T* __restrict__ pfp, double* __restrict__ truth, double* __restrict__ truth_bias)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B) return;
    const uint64_t gi = (uint64_t)(a.filter_offset + i);
    SynthTruth s;
    synth_static(a, gi, s);

    // per-filter filter parameters (BASELINE cfg 5): Q groups scaled by 10^U(-0.5,0.5),
    // static (known) IMU biases ~ N(0,0.1^2), N(0,0.01^2)
    double Qf[12], Rf[6], abs_[3], wbs_[3];
    for (int k = 0; k < 12; ++k) Qf[k] = a.Q[k];
    for (int k = 0; k < 6; ++k) Rf[k] = a.R[k];
    for (int k = 0; k < 3; ++k) { abs_[k] = a.ab_static[k]; wbs_[k] = a.wb_static[k]; }
    if (a.perturb) {
        for (int grp = 0; grp < 4; ++grp) {
            double sc = pow(10.0, rng_uniform(a.seed, gi, kTickStatic, 200 + grp) - 0.5);
            for (int k = 0; k < 3; ++k) Qf[3 * grp + k] *= sc;
        }
        for (int k = 0; k < 3; ++k) {
            abs_[k] = 0.1 * rng_normal(a.seed, gi, kTickStatic, 210 + k);
            wbs_[k] = 0.01 * rng_normal(a.seed, gi, kTickStatic, 213 + k);
        }
    }
    if (pfp) {
        for (int k = 0; k < 12; ++k) pfp[word_off<T>(k, i, kFW)] = (T)Qf[k];
        for (int k = 0; k < 3; ++k) {
            pfp[word_off<T>(12 + k, i, kFW)] = (T)abs_[k];
            pfp[word_off<T>(15 + k, i, kFW)] = (T)wbs_[k];
        }
        for (int k = 0; k < 6; ++k) pfp[word_off<T>(18 + k, i, kFW)] = (T)Rf[k];
    }

    double r[3];
    for (int k = 0; k < 3; ++k) r[k] = s.r0[k] + s.A[k] * sin(s.ph[k]);
    {   // seeding pose at time 0
        double z[7];
        synth_measure(a, gi, kTickSeedMeas, r, s.q, a.R, z);
        for (int k = 0; k < 7; ++k) z0[word_off<T>(k, i, kZW)] = (T)z[k];
        z0[word_off<T>(7, i, kZW)] = T(1);
    }
    // truth poses of the last ticks, for delayed tag poses: entry (t+1) % N = pose after tick t
    constexpr int N = kSynthMaxDelay + 1;
    double rh[N][3], qh[N][4];
    for (int k = 0; k < N; ++k) {
        for (int c = 0; c < 3; ++c) rh[k][c] = r[c];
        for (int c = 0; c < 4; ++c) qh[k][c] = s.q[c];
    }
    for (int64_t t = 0; t < a.T; ++t) {
        const double tt = (double)t * a.dT;
        double acc[3], w[3], C[9];
        for (int k = 0; k < 3; ++k) {
            acc[k] = -s.A[k] * s.om[k] * s.om[k] * sin(s.om[k] * tt + s.ph[k]) - a.g[k];
            w[k] = s.wa[k] * sin(s.wo[k] * tt + s.wp[k]);
        }
        quat_to_rot<double>(s.q, C);
        T* ut = us + t * a.pitch_u_words;
        for (int k = 0; k < 3; ++k) {
            // a_meas = C^T (r'' - g) + ab_true + ab_static + N(0,Q_a); w_meas = w + wb_true + wb_static + N(0,Q_w)
            double am = (C[k] * acc[0] + C[3 + k] * acc[1] + C[6 + k] * acc[2]) + s.ab[k] + abs_[k] +
                        a.imu_scale * sqrt(a.Q[k]) * rng_normal(a.seed, gi, (uint64_t)t, k);
            double wm = w[k] + s.wb[k] + wbs_[k] + a.imu_scale * sqrt(a.Q[3 + k]) * rng_normal(a.seed, gi, (uint64_t)t, 3 + k);
            ut[word_off<T>(k, i, kUW)] = (T)am;
            ut[word_off<T>(3 + k, i, kUW)] = (T)wm;
        }
        // truth advances one tick: exact exponential map with the rate held over the tick
        double dw[3] = {a.dT * w[0], a.dT * w[1], a.dT * w[2]}, qe[4], qn[4];
        quat_exp<double>(dw, qe);
        quat_mul<double>(s.q, qe, qn);
        quat_norm<double>(qn);
        for (int k = 0; k < 4; ++k) s.q[k] = qn[k];
        for (int k = 0; k < 3; ++k) r[k] = s.r0[k] + s.A[k] * sin(s.om[k] * (tt + a.dT) + s.ph[k]);
        {
            const int e = (int)((t + 1) % N);
            for (int c = 0; c < 3; ++c) rh[e][c] = r[c];
            for (int c = 0; c < 4; ++c) qh[e][c] = s.q[c];
        }
        const int32_t sl = slot[t];
        if (sl >= 0) {
            // single-rate (delay 0): the pose the filter holds after this tick's predict.
            // multirate with step delay L: the filter fuses it into history entry "tick t-L",
            // i.e. the pose after tick t-L (EKF.cpp:201-209).
            int64_t tm = t - a.meas_delay_ticks;
            if (tm < -1) tm = -1;
            const int e = (int)((tm + 1) % N);
            const double rr[3] = {rh[e][0], rh[e][1], rh[e][2]};
            const double qq[4] = {qh[e][0], qh[e][1], qh[e][2], qh[e][3]};
            double z[7];
            synth_measure(a, gi, (uint64_t)t, rr, qq, a.R, z);
            T* zt = zs + (int64_t)sl * a.pitch_z_words;
            for (int k = 0; k < 7; ++k) zt[word_off<T>(k, i, kZW)] = (T)z[k];
            zt[word_off<T>(7, i, kZW)] = T(1);
        }
    }
    for (int k = 0; k < 3; ++k) truth[i * 7 + k] = r[k];
    for (int k = 0; k < 4; ++k) truth[i * 7 + 3 + k] = s.q[k];
    for (int k = 0; k < 3; ++k) { truth_bias[i * 6 + k] = s.ab[k]; truth_bias[i * 6 + 3 + k] = s.wb[k]; }
}

// Per-device error sums vs the generator's truth (cfg 5):
// out[0] += |r - r_true|^2, out[1] += |log(q_true^-1 (x) q)|^2, out[2] += 1.
template <typename T>
__global__ void k_rmse(const T* __restrict__ xs, const double* __restrict__ truth, double* __restrict__ out, int64_t B)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double er = 0.0, eth = 0.0, cnt = 0.0;
    if (i < B) {
        double q[4], qt_c[4], dq[4], th[3];
        for (int k = 0; k < 3; ++k) {
            double d = (double)xs[word_off<T>(k, i, kSW)] - truth[i * 7 + k];
            er += d * d;
        }
        for (int k = 0; k < 4; ++k) q[k] = (double)xs[word_off<T>(6 + k, i, kSW)];
        qt_c[0] = -truth[i * 7 + 3]; qt_c[1] = -truth[i * 7 + 4]; qt_c[2] = -truth[i * 7 + 5]; qt_c[3] = truth[i * 7 + 6];
        quat_mul<double>(qt_c, q, dq);
        if (dq[3] < 0) { dq[0] = -dq[0]; dq[1] = -dq[1]; dq[2] = -dq[2]; dq[3] = -dq[3]; }
        quat_norm<double>(dq);
        quat_log<double>(dq, th);
        eth = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
        cnt = 1.0;
    }
    // wave reduction (64 lanes), then one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        er += __shfl_down(er, off, 64);
        eth += __shfl_down(eth, off, 64);
        cnt += __shfl_down(cnt, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], er);
        atomicAdd(&out[1], eth);
        atomicAdd(&out[2], cnt);
    }
}

}  // namespace qle
