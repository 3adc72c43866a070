// ekf_rows.hpp -- "rows across lanes" variant of the filter tick for SMALL batches (gfx950).
//
// With one lane per filter a wave needs 64 filters, so a batch of 4 096 filters occupies 64 of the
// chip's 1 024 SIMDs and the tick is bound by the latency of one wave (and, in fp64, by register
// spills: P alone is 240 VGPRs).  Here a filter is spread over 16 lanes: lane k of the group holds
// ROW k of the full 15x15 covariance in registers (15 values; lane 15 mirrors row 14 and stores
// nothing), so a wave carries 4 filters, the same batch yields 16x more waves, and nothing spills.
//
//   F P     : row operations  -> rows travel between lanes with ds_bpermute (__shfl), 7 x 15 values
//   (FP)F^T : column operations, local to each lane
//   update  : h = P g'^T is one value per lane; s, the innovation and the gain vector travel by shuffle;
//             the rank-1 downdate of row k needs h_k (local) and the whole gain vector (15 shuffles)
//
// Same arithmetic as ekf_device.hpp (prediction_step EKF.cpp:346-415, correction_step EKF.cpp:417-502,
// decorrelated sequential fusion); the per-filter quantities (C, A, B, Rt, R_k, ...) are computed
// redundantly by the 16 lanes.  The state stays in the engine's one HBM layout (wave tiles, packed
// upper triangle): lane k gathers row k and writes back the entries (k, j >= k).
#pragma once

#include "ekf_kernels.hpp"

namespace qle {

constexpr int kRowLanes = 16;

template <typename T>
__device__ __forceinline__ T group_shfl(T v, int src_lane) { return __shfl(v, src_lane, 64); }

// One tick of the single-rate filter for filter f = (global lane) / 16: predict, then correct if the
// tag record's mask word is set (zs == nullptr: predict-only tick).  In place on `st`.
template <typename T, bool DIRECT, bool PFP>
__global__ __launch_bounds__(kBlock) void k_rows(DevParams<T> p, T* st, const T* __restrict__ us, const T* __restrict__ zs,
                                                 const T* __restrict__ pfp, int64_t B)
{
    const int64_t gl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t f = gl >> 4;                 // filter
    if (f >= B) return;
    const int lane = (int)(threadIdx.x & 63);
    const int kk_ = lane & 15;                 // lane within the group
    const int k = kk_ < 15 ? kk_ : 14;         // row held by this lane (lane 15 mirrors row 14)
    const int gb = lane & ~15;                 // first lane of the group inside the wave
    const bool writer = kk_ < 15;

    // ---- loads: x, u (replicated in the group), row k of P (gather from the packed triangle)
    T x[kXW], u[kUW], pr[15];
#pragma unroll
    for (int w = 0; w < kXW; ++w) x[w] = st[word_off<T>(w, f, kSW)];
#pragma unroll
    for (int w = 0; w < kUW; ++w) u[w] = us[word_off<T>(w, f, kUW)];
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        // sidx(k, j) with k a run-time value: offset of (min, max) in the row-major upper triangle
        const int a = k < j ? k : j, b = k < j ? j : k;
        const int w = a * 15 - (a * (a - 1)) / 2 + (b - a);
        pr[j] = st[word_off<T>(kXW + w, f, kSW)];
    }
    Noise<T> nz;
    if (PFP) {
#pragma unroll
        for (int w = 0; w < 12; ++w) nz.Q[w] = pfp[word_off<T>(w, f, kFW)];
#pragma unroll
        for (int w = 0; w < 3; ++w) { nz.ab_static[w] = pfp[word_off<T>(12 + w, f, kFW)]; nz.wb_static[w] = pfp[word_off<T>(15 + w, f, kFW)]; }
#pragma unroll
        for (int w = 0; w < 6; ++w) nz.R[w] = pfp[word_off<T>(18 + w, f, kFW)];
    } else {
        load_noise<T, false>(p, pfp, f, nz);
    }
    T zr[kZW];
    bool corr = false;
    if (zs) {
#pragma unroll
        for (int w = 0; w < kZW; ++w) zr[w] = zs[word_off<T>(w, f, kZW)];
        corr = zr[7] != T(0);
    }

    // ---- prologue (every lane): nominal state and the blocks of F  (EKF.cpp:350-400)
    const T dT = p.dT, dTw = p.dTw;
    T a[3], wv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = u[i] - x[10 + i] - nz.ab_static[i];
        wv[i] = u[3 + i] - x[13 + i] - nz.wb_static[i];
    }
    T q[4] = {x[6], x[7], x[8], x[9]};
    T C[9];
    quat_to_rot(q, C);
    T accel[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) accel[i] = (C[3 * i] * a[0] + C[3 * i + 1] * a[1] + C[3 * i + 2] * a[2]) + p.g[i];
    T dw[3] = {dT * wv[0], dT * wv[1], dT * wv[2]};
    {
        T qe[4], qn[4];
        quat_exp(dw, qe);
        quat_mul(q, qe, qn);
        quat_norm(qn);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            x[i] += dT * x[3 + i];
            x[3 + i] += dT * accel[i];
        }
        x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
    }
    T X[3][6];
    const T mdT = -dT, mdTb = -dT * p.bias_on;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T c0 = C[3 * i], c1 = C[3 * i + 1], c2 = C[3 * i + 2];
        X[i][0] = mdT * (c1 * a[2] - c2 * a[1]);
        X[i][1] = mdT * (c2 * a[0] - c0 * a[2]);
        X[i][2] = mdT * (c0 * a[1] - c1 * a[0]);
        X[i][3] = mdTb * c0; X[i][4] = mdTb * c1; X[i][5] = mdTb * c2;
    }
    T Rt[3][3];
    {
        T ang = t_sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
        bool small = ang < p.small_ang_tol;
        T inv = T(1) / (small ? T(1) : ang);
        T ax[3] = {dw[0] * inv, dw[1] * inv, dw[2] * inv};
        T sn, cs;
        t_sincos(-ang, &sn, &cs);
        T sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
        T ca[3] = {(T(1) - cs) * ax[0], (T(1) - cs) * ax[1], (T(1) - cs) * ax[2]};
        T t01 = ca[0] * ax[1], t02 = ca[0] * ax[2], t12 = ca[1] * ax[2];
        Rt[0][0] = small ? T(1) : ca[0] * ax[0] + cs;
        Rt[1][1] = small ? T(1) : ca[1] * ax[1] + cs;
        Rt[2][2] = small ? T(1) : ca[2] * ax[2] + cs;
        Rt[0][1] = small ? dw[2] : t01 - sa[2];
        Rt[1][0] = small ? -dw[2] : t01 + sa[2];
        Rt[0][2] = small ? -dw[1] : t02 + sa[1];
        Rt[2][0] = small ? dw[1] : t02 - sa[1];
        Rt[1][2] = small ? dw[0] : t12 - sa[0];
        Rt[2][1] = small ? -dw[0] : t12 + sa[0];
    }

    // ---- M = F P: row k of M from the rows of P held by other lanes
    //   r rows (k<3):      row_k + dT row_{k+3}
    //   v rows (3<=k<6):   row_k + sum_m X[k-3][m] row_{6+m}
    //   th rows (6<=k<9):  sum_m Rt[k-6][m] row_{6+m} - dTw row_{k+6}
    //   bias rows:         row_k
    const bool is_r = k < 3, is_v = k >= 3 && k < 6, is_t = k >= 6 && k < 9;
    T coef[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const T xv = k == 3 ? X[0][m] : (k == 4 ? X[1][m] : X[2][m]);
        const T rv = m < 3 ? (k == 6 ? Rt[0][m] : (k == 7 ? Rt[1][m] : Rt[2][m])) : T(0);
        coef[m] = is_v ? xv : (is_t ? rv : T(0));
    }
    const int s_src = gb + (is_r ? k + 3 : (is_t ? k + 6 : k));
    const T s_coef = is_r ? dT : (is_t ? -dTw : T(0));
    T mrow[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        T acc = is_t ? T(0) : pr[j];
#pragma unroll
        for (int m = 0; m < 6; ++m) acc += coef[m] * group_shfl(pr[j], gb + 6 + m);
        acc += s_coef * group_shfl(pr[j], s_src);
        mrow[j] = acc;
    }
    // ---- P' = M F^T: column operations inside the lane (all from the old columns of M)
    T pn[15];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        pn[j] = mrow[j] + dT * mrow[3 + j];
        T av = mrow[3 + j];
#pragma unroll
        for (int m = 0; m < 6; ++m) av += X[j][m] * mrow[6 + m];
        pn[3 + j] = av;
        pn[6 + j] = (Rt[j][0] * mrow[6] + Rt[j][1] * mrow[7] + Rt[j][2] * mrow[8]) - dTw * mrow[12 + j];
        pn[9 + j] = mrow[9 + j];
        pn[12 + j] = mrow[12 + j];
    }
    // ---- + W Q W^T (EKF.cpp:402-414): own row of blockdiag(0, C Qa C^T, Qw, Qab, Qwb)
    if (is_v) {
        const int i = k - 3;
        const T cq0 = (i == 0 ? C[0] : (i == 1 ? C[3] : C[6])) * nz.Q[0];
        const T cq1 = (i == 0 ? C[1] : (i == 1 ? C[4] : C[7])) * nz.Q[1];
        const T cq2 = (i == 0 ? C[2] : (i == 1 ? C[5] : C[8])) * nz.Q[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) pn[3 + j] += cq0 * C[3 * j] + cq1 * C[3 * j + 1] + cq2 * C[3 * j + 2];
    }
#pragma unroll
    for (int j = 6; j < 15; ++j)
        if (k == j) pn[j] += nz.Q[j - 3];

    // ---- correction (EKF.cpp:417-502), decorrelated sequential fusion, one value of h per lane
    if (corr) {  // uniform over the 16 lanes of a filter
        T r[3] = {x[0], x[1], x[2]};
        T qc_[4] = {x[6], x[7], x[8], x[9]};
        T Cc[9];
        quat_to_rot(qc_, Cc);
        T qo[4];
        {
            T qct[4] = {zr[3], zr[4], zr[5], zr[6]}, t[4];
            quat_mul(p.q_vc, qct, t);
            qo[0] = -t[0]; qo[1] = -t[1]; qo[2] = -t[2]; qo[3] = t[3];
            quat_norm(qo);
        }
        T dy[6];
        {
            T Cq[9];
            if (DIRECT) quat_to_rot(qo, Cq);
            else {
#pragma unroll
                for (int i = 0; i < 9; ++i) Cq[i] = Cc[i];
            }
            T pv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) pv[i] = (p.C_vc[3 * i] * zr[0] + p.C_vc[3 * i + 1] * zr[1] + p.C_vc[3 * i + 2] * zr[2]) + p.r_v_cv[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) dy[i] = -(Cq[3 * i] * pv[0] + Cq[3 * i + 1] * pv[1] + Cq[3 * i + 2] * pv[2]) - r[i];
            T qcj[4] = {-qc_[0], -qc_[1], -qc_[2], qc_[3]}, dq[4], dth[3];
            quat_mul(qcj, qo, dq);
            quat_norm(dq);
            quat_log(dq, dth);
            dy[3] = dth[0]; dy[4] = dth[1]; dy[5] = dth[2];
        }
        constexpr int J[6] = {0, 1, 2, 6, 7, 8};
        T Gm[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j < 6; ++j) Gm[i][j] = (i == j) ? T(1) : T(0);
        }
        if (!DIRECT) {
            T b[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) b[i] = Cc[i] * r[0] + Cc[3 + i] * r[1] + Cc[6 + i] * r[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                T c0 = Cc[3 * i], c1 = Cc[3 * i + 1], c2 = Cc[3 * i + 2];
                Gm[i][3] = c1 * b[2] - c2 * b[1];
                Gm[i][4] = c2 * b[0] - c0 * b[2];
                Gm[i][5] = c0 * b[1] - c1 * b[0];
            }
        }
        T Rk[6][6];
        {
            T N00[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) N00[i][j] = -(Cc[3 * i] * p.C_vc[j] + Cc[3 * i + 1] * p.C_vc[3 + j] + Cc[3 * i + 2] * p.C_vc[6 + j]);
            }
            T Sr[3][3] = {{T(0), -r[2], r[1]}, {r[2], T(0), -r[0]}, {-r[1], r[0], T(0)}};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = i; j < 3; ++j) {
                    T s = N00[i][0] * nz.R[0] * N00[j][0] + N00[i][1] * nz.R[1] * N00[j][1] + N00[i][2] * nz.R[2] * N00[j][2];
                    if (DIRECT) s += Sr[i][0] * nz.R[3] * Sr[j][0] + Sr[i][1] * nz.R[4] * Sr[j][1] + Sr[i][2] * nz.R[5] * Sr[j][2];
                    Rk[i][j] = s;
                    Rk[3 + i][3 + j] = p.C_vc[3 * i] * nz.R[3] * p.C_vc[3 * j] + p.C_vc[3 * i + 1] * nz.R[4] * p.C_vc[3 * j + 1] +
                                       p.C_vc[3 * i + 2] * nz.R[5] * p.C_vc[3 * j + 2];
                }
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Rk[i][3 + j] = DIRECT ? Sr[i][0] * nz.R[3] * p.C_vc[3 * j] + Sr[i][1] * nz.R[4] * p.C_vc[3 * j + 1] + Sr[i][2] * nz.R[5] * p.C_vc[3 * j + 2]
                                          : T(0);
            }
        }
        T d[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            d[c] = Rk[c][c];
            const T invd = T(1) / d[c];
#pragma unroll
            for (int j = c + 1; j < 6; ++j) {
                const T l = Rk[c][j] * invd;
#pragma unroll
                for (int j2 = j; j2 < 6; ++j2) Rk[j][j2] -= l * Rk[c][j2];
#pragma unroll
                for (int m = 0; m < 6; ++m)
                    if (!DIRECT || m <= c) Gm[j][m] -= l * Gm[c][m];
                dy[j] -= l * dy[c];
            }
        }
        T dxk = T(0);  // this lane's component of the error state
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            T h = T(0);
#pragma unroll
            for (int m = 0; m < 6; ++m)
                if (!DIRECT || m <= c) h += Gm[c][m] * pn[J[m]];
            T s = d[c], nu = dy[c];
#pragma unroll
            for (int m = 0; m < 6; ++m)
                if (!DIRECT || m <= c) {
                    s += Gm[c][m] * group_shfl(h, gb + J[m]);
                    nu -= Gm[c][m] * group_shfl(dxk, gb + J[m]);
                }
            const T g = h / s;   // gain component k
            dxk += g * nu;
#pragma unroll
            for (int j = 0; j < 15; ++j) pn[j] -= h * group_shfl(g, gb + j);
        }
        // inject (EKF.cpp:486-501): every lane needs the whole error state
        T dx[15];
#pragma unroll
        for (int j = 0; j < 15; ++j) dx[j] = group_shfl(dxk, gb + j);
        T dth[3] = {dx[6], dx[7], dx[8]}, qe[4], qn[4];
        quat_exp(dth, qe);
        quat_mul(qc_, qe, qn);
        quat_norm(qn);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            x[i] += dx[i];
            x[3 + i] += dx[3 + i];
            x[10 + i] = p.bias_on * (x[10 + i] + dx[9 + i]);
            x[13 + i] = p.bias_on * (x[13 + i] + dx[12 + i]);
        }
        x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
    }

    // ---- stores: lane k writes entries (k, j >= k) of the packed triangle; lane 0 writes x
    if (writer) {
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            if (j >= k) {
                const int w = k * 15 - (k * (k - 1)) / 2 + (j - k);
                st[word_off<T>(kXW + w, f, kSW)] = pn[j];
            }
        }
    }
    if (kk_ == 0) {
#pragma unroll
        for (int w = 0; w < kXW; ++w) st[word_off<T>(w, f, kSW)] = x[w];
    }
}

}  // namespace qle
