// ekf_quad.hpp -- per-filter arithmetic of the batched relative-pose EKF with FOUR lanes per filter (gfx950).
//
// Why: with one lane per filter a wave needs all of P in its registers (120 words: 240 VGPRs in fp64, which
// spills) and a batch of 65 536 filters is exactly one wave per SIMD, so nothing hides the dependent-issue
// latency of the ~3 600-instruction fused tick.  Here the covariance of a filter is spread over a QUAD of
// adjacent lanes: lane j in {0,1,2} owns COLUMN j of every 3x3 block (b, c), b <= c, of the 15x15 covariance
// (5x5 blocks r, v, th, ab, wb): 40 stored words + 5 symmetric duplicates per lane, 4x the waves, and every
// cross-lane operand is a quad_perm DPP read (full rate, no LDS).  Lane 3 carries no covariance.
//
// The per-filter SCALAR work (nominal state, quaternion exp / log, the small matrices A, Bm, Rt, C Qa C^T,
// the 6x6 innovation covariance and its L D L^T factor, the injection) is about a thousand instructions;
// replicated in the four lanes of a quad it cancels the gain (measured: profiles/r02_tuning.md).  So the
// arithmetic is cut in two: *_scalar functions run once per filter (one lane per filter, in one wave of the
// workgroup) and hand their results -- a few dozen words -- to the *_P functions, which run on the quads
// (ekf_quad_kernels.hpp passes them through LDS).
//
//   left-multiplication  (M B)[:, j]   = M B[:, j]                      lane-local
//   right-multiplication (B M^T)[:, j] = sum_k B[:, k] M[j][k]          B[:, k] read from lane k (DPP)
//   transposed block     (B^T)[:, j]   : element i is lane i's B[j]     3 DPP reads + a select per element
//
// Reference behaviour reproduced (mbrymer/quadrotor_landing, quad_state_estimation/):
//   src/relative_pose_EKF.cpp:346-415  prediction_step   -> quad::predict  (same levelled algebra as
//                                                            ekf_predict_levels in ekf_device.hpp)
//   src/relative_pose_EKF.cpp:417-502  correction_step   -> quad::update   (S = G P G^T + R_k = L D L^T,
//                                                            V = (P G^T) L^-T, P -= V D^-1 V^T, dx = V D^-1 L^-1 dy:
//                                                            K = P G^T S^-1, P = (I - K G) P of EKF.cpp:475-481
//                                                            in exact arithmetic)
//   src/quaternion_helper.cpp:9-100    quaternion_exp / log / norm
//
// The arithmetic is written over a lane context Q (value type Q::V): ScalarQ<T> for the per-filter parts, DevQ<T> for one
// value per lane with DPP cross-lane reads.  The same header compiles for the host with a four-value emulation of a quad,
// which is how the algebra is checked without a GPU (see tests/).
#pragma once

#include "ekf_device.hpp"

namespace qle {
namespace quad {

constexpr int kList = 40;   // words of P a lane loads / stores (ten 16-byte quads in fp32)

// ------------------------------------------------------------ lane contexts
// ScalarQ<T>: one value, no neighbours (the per-filter parts).  DevQ<T>: one value per lane with quad_perm reads.
template <typename T>
struct ScalarQ {
    using V = T;
    using M = bool;
    static __host__ __device__ __forceinline__ V sel(M m, V a, V b) { return m ? a : b; }
    static __host__ __device__ __forceinline__ M lt(V a, V b) { return a < b; }
    static __host__ __device__ __forceinline__ V sqrt(V v) { return t_sqrt(v); }
    static __host__ __device__ __forceinline__ V atan2(V a, V b) { return t_atan2(a, b); }
    static __host__ __device__ __forceinline__ void sincos(V v, V* s, V* c) { t_sincos(v, s, c); }
    // one plain value per instance: the polynomial half-angle pair of ekf_device.hpp applies (no libm call on the chain)
    static constexpr bool kLean = true;
    static __host__ __device__ __forceinline__ void half_angle(V h2, V& k, V& ch) { half_angle_sinc_cos(h2, k, ch); }
};
// `Q::kLean` for contexts that do not say: the general (library call + selects) forms below
template <class Q, typename = void> struct QLean { static constexpr bool value = false; };
template <class Q> struct QLean<Q, decltype((void)Q::kLean)> { static constexpr bool value = Q::kLean; };

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ float dpp_read(float v)
{
    // mov_dpp leaves `old` undefined: every lane of a quad_perm read is written, so no initialising move is needed and
    // the read can be folded into the consuming VOP2 instruction (v_fmac_f32_dpp, v_mul_f32_dpp, v_cndmask_b32_dpp)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_read(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true));
}

template <typename T>
struct DevQ : ScalarQ<T> {
    using V = T;
    using M = bool;
    static __device__ __forceinline__ int lane() { return (int)(__builtin_amdgcn_workitem_id_x() & 3u); }
    template <int K> static __device__ __forceinline__ V bc(V v) { return dpp_read<K * 0x55>(v); }   // quad_perm:[K,K,K,K]
    static __device__ __forceinline__ V rot1(V v) { return dpp_read<0xC9>(v); }   // quad_perm:[1,2,0,3]: lane j reads lane (j+1)%3
    static __device__ __forceinline__ V rot2(V v) { return dpp_read<0xD2>(v); }   // quad_perm:[2,0,1,3]: lane j reads lane (j+2)%3
    static __device__ __forceinline__ V pick3(V a, V b, V c) { const int j = lane(); return j == 1 ? b : (j == 2 ? c : a); }
};
#endif

// List position of the stored words (ekf_device.hpp, sidx): per block-row b the diagonal block's D0, D1, then
// column j of the blocks (b, c), c > b.
#define QLE_QD0(b) (::qle::quad_group_base(b))
#define QLE_QD1(b) (::qle::quad_group_base(b) + 1)
#define QLE_QO(b, c, i) (::qle::quad_group_base(b) + 2 + 3 * ((c) - (b)-1) + (i))

// Column j of a diagonal block in absolute row order from its stored part: D0 = row j, D1 = row (j+2)%3, and
// row (j+1)%3 is lane (j+1)%3's D1 (symmetry).
template <class Q>
__host__ __device__ __forceinline__ void diag_expand(const typename Q::V d0, const typename Q::V d1, typename Q::V (&col)[3])
{
    const typename Q::V d2 = Q::rot1(d1);
    col[0] = Q::pick3(d0, d1, d2);
    col[1] = Q::pick3(d2, d0, d1);
    col[2] = Q::pick3(d1, d2, d0);
}
template <class Q>
__host__ __device__ __forceinline__ void diag_compact(const typename Q::V (&col)[3], typename Q::V& d0, typename Q::V& d1)
{
    d0 = Q::pick3(col[0], col[1], col[2]);
    d1 = Q::pick3(col[2], col[0], col[1]);
}
// Column j of B^T from the column-distributed block B: element i is lane i's B[j].
template <class Q>
__host__ __device__ __forceinline__ void tr3(const typename Q::V& b0, const typename Q::V& b1, const typename Q::V& b2, typename Q::V (&out)[3])
{
    out[0] = Q::pick3(Q::template bc<0>(b0), Q::template bc<0>(b1), Q::template bc<0>(b2));
    out[1] = Q::pick3(Q::template bc<1>(b0), Q::template bc<1>(b1), Q::template bc<1>(b2));
    out[2] = Q::pick3(Q::template bc<2>(b0), Q::template bc<2>(b1), Q::template bc<2>(b2));
}
// sum_m B[:, m](i) * r[m]: element i of (B M^T)[:, j] with r = row j of M.
template <class Q>
__host__ __device__ __forceinline__ typename Q::V rdot(const typename Q::V& bi, const typename Q::V (&r)[3])
{
    return Q::template bc<0>(bi) * r[0] + Q::template bc<1>(bi) * r[1] + Q::template bc<2>(bi) * r[2];
}

// ------------------------------------------------------ quaternion helpers
template <class Q>
__host__ __device__ __forceinline__ void q_norm(typename Q::V (&q)[4])   // QH.cpp:61-73
{
    using V = typename Q::V;
    V n = Q::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    V inv = V(1) / n;
    V s = Q::sel(Q::lt(q[3] * inv, V(-0.75)), -inv, inv);
    q[0] = q[0] * s; q[1] = q[1] * s; q[2] = q[2] * s; q[3] = q[3] * s;
}
template <class Q>
__host__ __device__ __forceinline__ void q_exp(const typename Q::V (&v)[3], typename Q::V (&q)[4])   // QH.cpp:9-33
{
    using V = typename Q::V;
    if constexpr (QLean<Q>::value) {   // the same series the small-angle branch truncates, valid up to |v| = pi/2 (ekf_device.hpp)
        V k, ch;
        Q::half_angle(V(0.25) * (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), k, ch);
        q[0] = v[0] * k; q[1] = v[1] * k; q[2] = v[2] * k; q[3] = ch;
        q_norm<Q>(q);
        return;
    }
    V n = Q::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    V sh, ch;
    Q::sincos(n * V(0.5), &sh, &ch);
    const typename Q::M small = Q::lt(n, V(1E-10));
    V k = Q::sel(small, V(0.5) * (V(1) - n * n * (V(1) / V(24))), sh / Q::sel(small, V(1), n));
    q[0] = v[0] * k; q[1] = v[1] * k; q[2] = v[2] * k; q[3] = ch;
    q_norm<Q>(q);
}
template <class Q>
__host__ __device__ __forceinline__ void q_log(const typename Q::V (&q)[4], typename Q::V (&v)[3])   // QH.cpp:36-58
{
    using V = typename Q::V;
    V m = Q::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const typename Q::M small = Q::lt(m, V(1E-10));
    V mw = m / q[3];
    V k_small = V(2) / q[3] * (V(1) - mw * mw * (V(1) / V(3)));
    V k_full = V(2) * Q::atan2(m, q[3]) / Q::sel(small, V(1), m);
    V k = Q::sel(small, k_small, k_full);
    v[0] = k * q[0]; v[1] = k * q[1]; v[2] = k * q[2];
}
template <class V>
__host__ __device__ __forceinline__ void q_mul(const V (&a)[4], const V (&b)[4], V (&o)[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
template <class V>
__host__ __device__ __forceinline__ void q_to_rot(const V (&q)[4], V (&C)[9])
{
    V x = q[0], y = q[1], z = q[2], w = q[3];
    V tx = x + x, ty = y + y, tz = z + z;
    V twx = tx * w, twy = ty * w, twz = tz * w;
    V txx = tx * x, txy = ty * x, txz = tz * x;
    V tyy = ty * y, tyz = tz * y, tzz = tz * z;
    C[0] = V(1) - (tyy + tzz); C[1] = txy - twz;           C[2] = txz + twy;
    C[3] = txy + twz;           C[4] = V(1) - (txx + tzz); C[5] = tyz - twx;
    C[6] = txz - twy;           C[7] = tyz + twx;           C[8] = V(1) - (txx + tyy);
}

// Noise / static-bias values in the value type of the scalar part (shared parameters or a per-filter record).
template <class V>
struct NoiseV {
    V Q[12];
    V ab_static[3];
    V wb_static[3];
    V R[6];
};

// What the scalar part of the predict hands to the quads.
template <class V>
struct PredU {
    V A[9];    // -dT C [a]x, row-major                      (F[v,th], EKF.cpp:381)
    V Bm[9];   // -dT C with est_bias, else 0                (F[v,ab], EKF.cpp:399)
    V Rt[9];   // F[th,th]                                   (EKF.cpp:383-395)
    V CQ[9];   // C Qa C^T, full symmetric 3x3               (EKF.cpp:402-414)
    V Qd[9];   // diag of Qw, Qab, Qwb
};
constexpr int kPredUWords = 45;
// The quad's view of it: the full A, Bm, Rt (left-multiplications) and row j of A, Bm, Rt, column j of C Qa C^T and the
// j-th diagonal noise terms (right-multiplications / own column).
// predict_P / update_P read them through accessors (g.A(k), g.AR(m), ...) so that the device can fetch each value from
// LDS where it is used instead of holding them all in registers; PredQ / UpdQ are the plain value-holding forms.
template <class V>
struct PredQ {
    V a_[9], bm_[9], rt_[9];
    V ar_[3], br_[3], rtr_[3], cqc_[3];
    V qw_, qab_, qwb_;
    __host__ __device__ V A(int k) const { return a_[k]; }
    __host__ __device__ V Bm(int k) const { return bm_[k]; }
    __host__ __device__ V Rt(int k) const { return rt_[k]; }
    __host__ __device__ V AR(int m) const { return ar_[m]; }
    __host__ __device__ V BR(int m) const { return br_[m]; }
    __host__ __device__ V RtR(int m) const { return rtr_[m]; }
    __host__ __device__ V CQc(int m) const { return cqc_[m]; }
    __host__ __device__ V qw() const { return qw_; }
    __host__ __device__ V qab() const { return qab_; }
    __host__ __device__ V qwb() const { return qwb_; }
};
template <class V>
struct UpdU {
    V Lm[15];   // unit lower L of S = L D L^T, rows 1..5 packed: L[m][m2] at m (m-1)/2 + m2
    V invd[6];  // 1 / D
    V yd[6];    // D^-1 L^-1 dy
    V Gx[9];    // Cc [Cc^T r]x (conventional method), row-major
};
constexpr int kUpdUWords = 36;
template <class V>
struct UpdQ {
    UpdU<V> u;
    __host__ __device__ V Lm(int k) const { return u.Lm[k]; }
    __host__ __device__ V invd(int k) const { return u.invd[k]; }
    __host__ __device__ V yd(int k) const { return u.yd[k]; }
    __host__ __device__ V Gx(int k) const { return u.Gx[k]; }
};
__host__ __device__ constexpr int lm_idx(int m, int m2) { return m * (m - 1) / 2 + m2; }

// ---------------------------------------------------------- predict, scalar
// Nominal-state propagation and the blocks of F (EKF.cpp:350-400) for one filter.  x is updated in place.
template <class Q, typename T>
__host__ __device__ __forceinline__ void predict_scalar(const DevParams<T>& p, const NoiseV<typename Q::V>& nz, typename Q::V (&x)[16],
                                                        const typename Q::V (&u)[6], typename Q::V (&accel)[3], PredU<typename Q::V>& o)
{
    using V = typename Q::V;
    const V dT = V(p.dT);
    V a[3], w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = u[i] - x[10 + i] - nz.ab_static[i];      // EKF.cpp:357
        w[i] = u[3 + i] - x[13 + i] - nz.wb_static[i];  // EKF.cpp:358
    }
    V q[4] = {x[6], x[7], x[8], x[9]};
    V C[9];
    q_to_rot(q, C);                                     // EKF.cpp:359
#pragma unroll
    for (int i = 0; i < 3; ++i) accel[i] = (C[3 * i] * a[0] + C[3 * i + 1] * a[1] + C[3 * i + 2] * a[2]) + V(p.g[i]);  // EKF.cpp:362
    V dw[3] = {dT * w[0], dT * w[1], dT * w[2]};
    V qe[4];
    {
        V qn[4];
        q_exp<Q>(dw, qe);
        q_mul(q, qe, qn);
        q_norm<Q>(qn);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            x[i] = x[i] + dT * x[3 + i];
            x[3 + i] = x[3 + i] + dT * accel[i];
        }
        x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
    }
    const V mdT = -dT, mdTb = -dT * V(p.bias_on);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        V c0 = C[3 * i], c1 = C[3 * i + 1], c2 = C[3 * i + 2];
        o.A[3 * i] = mdT * (c1 * a[2] - c2 * a[1]);
        o.A[3 * i + 1] = mdT * (c2 * a[0] - c0 * a[2]);
        o.A[3 * i + 2] = mdT * (c0 * a[1] - c1 * a[0]);
        o.Bm[3 * i] = mdTb * c0; o.Bm[3 * i + 1] = mdTb * c1; o.Bm[3 * i + 2] = mdTb * c2;
    }
    if constexpr (QLean<Q>::value) {   // F[th,th] = R(exp(phi))^T (EKF.cpp:383-395): no second sine / cosine, no axis, no selects
        V Re[9];
        q_to_rot(qe, Re);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int m = 0; m < 3; ++m) o.Rt[3 * i + m] = Re[3 * m + i];
        }
    } else {
        V ang = Q::sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
        const typename Q::M small = Q::lt(ang, V(p.small_ang_tol));
        V inv = V(1) / Q::sel(small, V(1), ang);
        V ax[3] = {dw[0] * inv, dw[1] * inv, dw[2] * inv};
        V sn, cs;
        Q::sincos(-ang, &sn, &cs);
        V sa[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
        V ca[3] = {(V(1) - cs) * ax[0], (V(1) - cs) * ax[1], (V(1) - cs) * ax[2]};
        V t01 = ca[0] * ax[1], t02 = ca[0] * ax[2], t12 = ca[1] * ax[2];
        o.Rt[0] = Q::sel(small, V(1), ca[0] * ax[0] + cs);
        o.Rt[4] = Q::sel(small, V(1), ca[1] * ax[1] + cs);
        o.Rt[8] = Q::sel(small, V(1), ca[2] * ax[2] + cs);
        o.Rt[1] = Q::sel(small, dw[2], t01 - sa[2]);
        o.Rt[3] = Q::sel(small, -dw[2], t01 + sa[2]);
        o.Rt[2] = Q::sel(small, -dw[1], t02 + sa[1]);
        o.Rt[6] = Q::sel(small, dw[1], t02 - sa[1]);
        o.Rt[5] = Q::sel(small, dw[0], t12 - sa[0]);
        o.Rt[7] = Q::sel(small, -dw[0], t12 + sa[0]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) o.CQ[3 * i + k] = C[3 * i] * nz.Q[0] * C[3 * k] + C[3 * i + 1] * nz.Q[1] * C[3 * k + 1] + C[3 * i + 2] * nz.Q[2] * C[3 * k + 2];
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) o.Qd[k] = nz.Q[3 + k];
}

// --------------------------------------------------------- predict, on quads
// P <- F P F^T + W Q W^T (EKF.cpp:376-414) on the lane's 40 stored words: L = old, Ln = new.  Same levelled algebra
// as ekf_predict_levels (ekf_device.hpp).  Prr / Ptt receive column j of the new diagonal blocks (r, r) and (th, th)
// in absolute row order (the correction needs them).  done(level) is called when the words of block-rows >=
// {ab (0), th (1), v (2), r (3)} of Ln are final.
template <class Q, typename T, class G, typename Done>
__host__ __device__ __forceinline__ void predict_P(const DevParams<T>& p, const G& g, const typename Q::V (&L)[kList],
                                                   typename Q::V (&Ln)[kList], typename Q::V (&Prr)[3], typename Q::V (&Ptt)[3], Done done)
{
    using V = typename Q::V;
    const V dT = V(p.dT), dTw = V(p.dTw);
    // old diagonal blocks, column j in absolute row order
    V Ovv[3], Ott[3], Oaa[3], Oww[3];
    diag_expand<Q>(L[QLE_QD0(1)], L[QLE_QD1(1)], Ovv);
    diag_expand<Q>(L[QLE_QD0(2)], L[QLE_QD1(2)], Ott);
    diag_expand<Q>(L[QLE_QD0(3)], L[QLE_QD1(3)], Oaa);
    diag_expand<Q>(L[QLE_QD0(4)], L[QLE_QD1(4)], Oww);

    // ---- level 0: rows ab, wb ---------------------------------------------
    Ln[QLE_QD0(3)] = L[QLE_QD0(3)] + g.qab();
    Ln[QLE_QD1(3)] = L[QLE_QD1(3)];
    Ln[QLE_QD0(4)] = L[QLE_QD0(4)] + g.qwb();
    Ln[QLE_QD1(4)] = L[QLE_QD1(4)];
#pragma unroll
    for (int i = 0; i < 3; ++i) Ln[QLE_QO(3, 4, i)] = L[QLE_QO(3, 4, i)];
    done(0);

    const V RtR[3] = {g.RtR(0), g.RtR(1), g.RtR(2)};
    // ---- level 1: rows th ----------------------------------------------------
    {
        V Owa[3], Owt[3];
        tr3<Q>(L[QLE_QO(3, 4, 0)], L[QLE_QO(3, 4, 1)], L[QLE_QO(3, 4, 2)], Owa);
        tr3<Q>(L[QLE_QO(2, 4, 0)], L[QLE_QO(2, 4, 1)], L[QLE_QO(2, 4, 2)], Owt);
        V Ntw[3], Mtt[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Ntw[i] = (g.Rt(3 * i) * L[QLE_QO(2, 4, 0)] + g.Rt(3 * i + 1) * L[QLE_QO(2, 4, 1)] + g.Rt(3 * i + 2) * L[QLE_QO(2, 4, 2)]) - dTw * Oww[i];
            Ln[QLE_QO(2, 4, i)] = Ntw[i];
            Ln[QLE_QO(2, 3, i)] = (g.Rt(3 * i) * L[QLE_QO(2, 3, 0)] + g.Rt(3 * i + 1) * L[QLE_QO(2, 3, 1)] + g.Rt(3 * i + 2) * L[QLE_QO(2, 3, 2)]) - dTw * Owa[i];
            Mtt[i] = (g.Rt(3 * i) * Ott[0] + g.Rt(3 * i + 1) * Ott[1] + g.Rt(3 * i + 2) * Ott[2]) - dTw * Owt[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) Ptt[i] = rdot<Q>(Mtt[i], RtR) - dTw * Ntw[i];
        diag_compact<Q>(Ptt, Ln[QLE_QD0(2)], Ln[QLE_QD1(2)]);
        Ln[QLE_QD0(2)] = Ln[QLE_QD0(2)] + g.qw();                  // + Qw on the diagonal (D0 is element (j, j))
        Ptt[0] = Q::pick3(Ln[QLE_QD0(2)], Ptt[0], Ptt[0]);
        Ptt[1] = Q::pick3(Ptt[1], Ln[QLE_QD0(2)], Ptt[1]);
        Ptt[2] = Q::pick3(Ptt[2], Ptt[2], Ln[QLE_QD0(2)]);
    }
    done(1);

    const V AR[3] = {g.AR(0), g.AR(1), g.AR(2)}, BR[3] = {g.BR(0), g.BR(1), g.BR(2)};
    // ---- level 2: rows v -----------------------------------------------------
    {
        V Oat[3], Otv[3], Oav[3];
        tr3<Q>(L[QLE_QO(2, 3, 0)], L[QLE_QO(2, 3, 1)], L[QLE_QO(2, 3, 2)], Oat);
        tr3<Q>(L[QLE_QO(1, 2, 0)], L[QLE_QO(1, 2, 1)], L[QLE_QO(1, 2, 2)], Otv);
        tr3<Q>(L[QLE_QO(1, 3, 0)], L[QLE_QO(1, 3, 1)], L[QLE_QO(1, 3, 2)], Oav);
        V Mvt[3], Mva[3], Mvw[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            V t = L[QLE_QO(1, 2, i)], s = L[QLE_QO(1, 3, i)], r = L[QLE_QO(1, 4, i)];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                t = t + g.A(3 * i + m) * Ott[m];
                s = s + g.A(3 * i + m) * L[QLE_QO(2, 3, m)];
                r = r + g.A(3 * i + m) * L[QLE_QO(2, 4, m)];
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                t = t + g.Bm(3 * i + m) * Oat[m];
                s = s + g.Bm(3 * i + m) * Oaa[m];
                r = r + g.Bm(3 * i + m) * L[QLE_QO(3, 4, m)];
            }
            Mvt[i] = t; Mva[i] = s; Mvw[i] = r;
        }
        V Nvv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Ln[QLE_QO(1, 4, i)] = Mvw[i];
            Ln[QLE_QO(1, 3, i)] = Mva[i];
            Ln[QLE_QO(1, 2, i)] = rdot<Q>(Mvt[i], RtR) - dTw * Mvw[i];
            V acc = Ovv[i];
#pragma unroll
            for (int m = 0; m < 3; ++m) acc = acc + g.A(3 * i + m) * Otv[m];
#pragma unroll
            for (int m = 0; m < 3; ++m) acc = acc + g.Bm(3 * i + m) * Oav[m];
            acc = acc + rdot<Q>(Mvt[i], AR);
            acc = acc + rdot<Q>(Mva[i], BR);
            Nvv[i] = acc + g.CQc(i);
        }
        diag_compact<Q>(Nvv, Ln[QLE_QD0(1)], Ln[QLE_QD1(1)]);
    }
    done(2);

    // ---- level 3: rows r -----------------------------------------------------
    {
        V Orr[3], Ovr[3];
        diag_expand<Q>(L[QLE_QD0(0)], L[QLE_QD1(0)], Orr);
        tr3<Q>(L[QLE_QO(0, 1, 0)], L[QLE_QO(0, 1, 1)], L[QLE_QO(0, 1, 2)], Ovr);
        V M1v[3], M1t[3], M1a[3], M1w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            M1v[i] = L[QLE_QO(0, 1, i)] + dT * Ovv[i];
            M1t[i] = L[QLE_QO(0, 2, i)] + dT * L[QLE_QO(1, 2, i)];
            M1a[i] = L[QLE_QO(0, 3, i)] + dT * L[QLE_QO(1, 3, i)];
            M1w[i] = L[QLE_QO(0, 4, i)] + dT * L[QLE_QO(1, 4, i)];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            Ln[QLE_QO(0, 4, i)] = M1w[i];
            Ln[QLE_QO(0, 3, i)] = M1a[i];
            Ln[QLE_QO(0, 2, i)] = rdot<Q>(M1t[i], RtR) - dTw * M1w[i];
            Ln[QLE_QO(0, 1, i)] = M1v[i] + rdot<Q>(M1t[i], AR) + rdot<Q>(M1a[i], BR);
            Prr[i] = Orr[i] + dT * (Ovr[i] + M1v[i]);
        }
        diag_compact<Q>(Prr, Ln[QLE_QD0(0)], Ln[QLE_QD1(0)]);
    }
    done(3);
}

// ----------------------------------------------------------- update, scalar
// correction_step up to the gain (EKF.cpp:417-475) for one filter, in three parts so that the kernel can run the first two on
// other waves while wave 0 is still busy with the predict's scalar part (they need the predicted nominal state only):
//   update_innovation  dy = [r_obs - r; log(conj(q) (x) q_obs)]                     (EKF.cpp:429-450)
//   update_noise       Gx = Cc [Cc^T r]x (conventional method), R_k = N R N^T       (EKF.cpp:453-472)
//   update_factor      S = G P G^T + R_k = L D L^T, yd = D^-1 L^-1 dy               (EKF.cpp:475 in factored form)
// x is the PREDICTED nominal state and is not changed (update_inject applies the error state).
constexpr int kRkWords = 21;   // upper triangle of the 6x6 R_k, row-major
__host__ __device__ constexpr int rk_idx(int i, int j) { return i * 6 - i * (i - 1) / 2 + (j - i); }   // i <= j

// The nominal-state propagation alone (EKF.cpp:356-371): what the other waves need of the predict.
template <class Q, typename T>
__host__ __device__ __forceinline__ void predict_nominal(const DevParams<T>& p, const NoiseV<typename Q::V>& nz, typename Q::V (&x)[16],
                                                         const typename Q::V (&u)[6])
{
    using V = typename Q::V;
    const V dT = V(p.dT);
    V a[3], w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = u[i] - x[10 + i] - nz.ab_static[i];
        w[i] = u[3 + i] - x[13 + i] - nz.wb_static[i];
    }
    V q[4] = {x[6], x[7], x[8], x[9]};
    V C[9];
    q_to_rot(q, C);
    V dw[3] = {dT * w[0], dT * w[1], dT * w[2]}, qe[4], qn[4];
    q_exp<Q>(dw, qe);
    q_mul(q, qe, qn);
    q_norm<Q>(qn);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const V acc = (C[3 * i] * a[0] + C[3 * i + 1] * a[1] + C[3 * i + 2] * a[2]) + V(p.g[i]);
        x[i] = x[i] + dT * x[3 + i];
        x[3 + i] = x[3 + i] + dT * acc;
    }
    x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
}

template <class Q, typename T, bool DIRECT, typename EmitObs>
__host__ __device__ __forceinline__ void update_innovation(const DevParams<T>& p, const typename Q::V (&x)[16], const typename Q::V (&z)[7],
                                                           typename Q::V (&dy)[6], EmitObs&& emit_obs)
{
    using V = typename Q::V;
    V q[4] = {x[6], x[7], x[8], x[9]};
    V r[3] = {x[0], x[1], x[2]};
    V qo[4];
    {
        V qvc[4] = {V(p.q_vc[0]), V(p.q_vc[1]), V(p.q_vc[2]), V(p.q_vc[3])};
        V qct[4] = {z[3], z[4], z[5], z[6]}, t[4];
        q_mul(qvc, qct, t);                                  // EKF.cpp:431
        qo[0] = -t[0]; qo[1] = -t[1]; qo[2] = -t[2]; qo[3] = t[3];
        q_norm<Q>(qo);                                       // EKF.cpp:432
    }
    V Cq[9];
    if (DIRECT) q_to_rot(qo, Cq);                            // EKF.cpp:434-444
    else q_to_rot(q, Cq);
    V pv[3], obs[7];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        pv[i] = (V(p.C_vc[3 * i]) * z[0] + V(p.C_vc[3 * i + 1]) * z[1] + V(p.C_vc[3 * i + 2]) * z[2]) + V(p.r_v_cv[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        V ro = -(Cq[3 * i] * pv[0] + Cq[3 * i + 1] * pv[1] + Cq[3 * i + 2] * pv[2]);
        obs[i] = ro;
        dy[i] = ro - r[i];                                   // EKF.cpp:447
    }
    obs[3] = qo[0]; obs[4] = qo[1]; obs[5] = qo[2]; obs[6] = qo[3];
    emit_obs(obs);
    V qc[4] = {-q[0], -q[1], -q[2], q[3]}, dq[4], dth[3];
    q_mul(qc, qo, dq);                                       // EKF.cpp:448
    q_norm<Q>(dq);                                           // EKF.cpp:449
    q_log<Q>(dq, dth);                                       // EKF.cpp:450
    dy[3] = dth[0]; dy[4] = dth[1]; dy[5] = dth[2];
}

template <class Q, typename T, bool DIRECT>
__host__ __device__ __forceinline__ void update_noise(const DevParams<T>& p, const NoiseV<typename Q::V>& nz, const typename Q::V (&x)[16],
                                                      typename Q::V (&Gx)[9], typename Q::V (&Rk)[kRkWords])
{
    using V = typename Q::V;
    V q[4] = {x[6], x[7], x[8], x[9]};
    V r[3] = {x[0], x[1], x[2]};
    V Cc[9];
    q_to_rot(q, Cc);                                         // EKF.cpp:429
    // G = [I Gx; 0 I] over the columns {r, th}; Gx = Cc [Cc^T r]x unless direct (EKF.cpp:453-459)
#pragma unroll
    for (int k = 0; k < 9; ++k) Gx[k] = V(0);
    if (!DIRECT) {
        V b[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = Cc[i] * r[0] + Cc[3 + i] * r[1] + Cc[6 + i] * r[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            V c0 = Cc[3 * i], c1 = Cc[3 * i + 1], c2 = Cc[3 * i + 2];
            Gx[3 * i] = c1 * b[2] - c2 * b[1];
            Gx[3 * i + 1] = c2 * b[0] - c0 * b[2];
            Gx[3 * i + 2] = c0 * b[1] - c1 * b[0];
        }
    }
    // R_k = N R N^T (upper triangle), N = [-Cc C_vc, [r]x (direct); 0, C_vc] (EKF.cpp:462-472)
    V N00[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            N00[i][j] = -(Cc[3 * i] * V(p.C_vc[j]) + Cc[3 * i + 1] * V(p.C_vc[3 + j]) + Cc[3 * i + 2] * V(p.C_vc[6 + j]));
    }
    V Sr[3][3] = {{V(0), -r[2], r[1]}, {r[2], V(0), -r[0]}, {-r[1], r[0], V(0)}};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = i; j < 3; ++j) {
            V s = N00[i][0] * nz.R[0] * N00[j][0] + N00[i][1] * nz.R[1] * N00[j][1] + N00[i][2] * nz.R[2] * N00[j][2];
            if (DIRECT) s = s + (Sr[i][0] * nz.R[3] * Sr[j][0] + Sr[i][1] * nz.R[4] * Sr[j][1] + Sr[i][2] * nz.R[5] * Sr[j][2]);
            Rk[rk_idx(i, j)] = s;
            Rk[rk_idx(3 + i, 3 + j)] = V(p.C_vc[3 * i]) * nz.R[3] * V(p.C_vc[3 * j]) + V(p.C_vc[3 * i + 1]) * nz.R[4] * V(p.C_vc[3 * j + 1]) +
                                       V(p.C_vc[3 * i + 2]) * nz.R[5] * V(p.C_vc[3 * j + 2]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Rk[rk_idx(i, 3 + j)] = DIRECT ? Sr[i][0] * nz.R[3] * V(p.C_vc[3 * j]) + Sr[i][1] * nz.R[4] * V(p.C_vc[3 * j + 1]) +
                                                Sr[i][2] * nz.R[5] * V(p.C_vc[3 * j + 2])
                                          : V(0);
    }
}

// `in` hands over, value by value (in.Frr(k), in.Frt(k), in.Ftt(k): the predicted covariance blocks (r,r), (r,th), (th,th), full 3x3
// row-major; in.dy(k), in.Gx(k), in.Rk(k)): the device reads each from LDS where it is used, FactorIn holds plain values.
template <class V>
struct FactorIn {
    V frr[9], frt[9], ftt[9], dy_[6], gx[9], rk[kRkWords];
    __host__ __device__ V Frr(int k) const { return frr[k]; }
    __host__ __device__ V Frt(int k) const { return frt[k]; }
    __host__ __device__ V Ftt(int k) const { return ftt[k]; }
    __host__ __device__ V dy(int k) const { return dy_[k]; }
    __host__ __device__ V Gx(int k) const { return gx[k]; }
    __host__ __device__ V Rk(int k) const { return rk[k]; }
};
template <class Q, bool DIRECT, class In>
__host__ __device__ __forceinline__ void update_factor(const In& in, UpdU<typename Q::V>& o)
{
    using V = typename Q::V;
    V S[6][6];
    if (DIRECT) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= i) { S[i][k] = in.Rk(rk_idx(i, k)) + in.Frr(3 * i + k); S[3 + i][3 + k] = in.Rk(rk_idx(3 + i, 3 + k)) + in.Ftt(3 * i + k); }
                S[i][3 + k] = in.Rk(rk_idx(i, 3 + k)) + in.Frt(3 * i + k);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) o.Gx[k] = V(0);
    } else {
        V Gx[9], Frt[9], Ftt[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) { Gx[k] = in.Gx(k); Frt[k] = in.Frt(k); Ftt[k] = in.Ftt(k); o.Gx[k] = Gx[k]; }
        V E[3][3];   // P_rt + Gx P_tt
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) E[i][k] = Frt[3 * i + k] + (Gx[3 * i] * Ftt[k] + Gx[3 * i + 1] * Ftt[3 + k] + Gx[3 * i + 2] * Ftt[6 + k]);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k >= i) {
                    S[i][k] = in.Rk(rk_idx(i, k)) + in.Frr(3 * i + k) + (Gx[3 * i] * Frt[3 * k] + Gx[3 * i + 1] * Frt[3 * k + 1] + Gx[3 * i + 2] * Frt[3 * k + 2]) +
                              (E[i][0] * Gx[3 * k] + E[i][1] * Gx[3 * k + 1] + E[i][2] * Gx[3 * k + 2]);
                    S[3 + i][3 + k] = in.Rk(rk_idx(3 + i, 3 + k)) + Ftt[3 * i + k];
                }
                S[i][3 + k] = in.Rk(rk_idx(i, 3 + k)) + E[i][k];
            }
        }
    }
    // S = L D L^T (unit lower L); y' = L^-1 dy; yd = D^-1 y'
    V dy[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dy[k] = in.dy(k);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        o.invd[c] = V(1) / S[c][c];
#pragma unroll
        for (int j = c + 1; j < 6; ++j) {
            const V l = S[c][j] * o.invd[c];
            o.Lm[lm_idx(j, c)] = l;
#pragma unroll
            for (int j2 = j; j2 < 6; ++j2) S[j][j2] = S[j][j2] - l * S[c][j2];
            dy[j] = dy[j] - l * dy[c];
        }
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) o.yd[m] = dy[m] * o.invd[m];
}

// The three parts in one call (host check, and callers that have nothing to overlap).
template <class Q, typename T, bool DIRECT, typename EmitObs>
__host__ __device__ __forceinline__ void update_scalar(const DevParams<T>& p, const NoiseV<typename Q::V>& nz, const typename Q::V (&x)[16],
                                                       const typename Q::V (&z)[7], const typename Q::V (&Frr)[9], const typename Q::V (&Frt)[9],
                                                       const typename Q::V (&Ftt)[9], UpdU<typename Q::V>& o, EmitObs&& emit_obs)
{
    using V = typename Q::V;
    FactorIn<V> in;
    update_innovation<Q, T, DIRECT>(p, x, z, in.dy_, emit_obs);
    update_noise<Q, T, DIRECT>(p, nz, x, in.gx, in.rk);
#pragma unroll
    for (int k = 0; k < 9; ++k) { in.frr[k] = Frr[k]; in.frt[k] = Frt[k]; in.ftt[k] = Ftt[k]; }
    update_factor<Q, DIRECT>(in, o);
}

// --------------------------------------------------------- update, on quads
// P <- P - V D^-1 V^T with V = (P G^T) L^-T (EKF.cpp:475-481 in factored form) on the lane's 40 stored words, and the
// lane's rows of the error state: dxo[b] = dx(3b + j).  Prr / Ptt: column j of the diagonal blocks (r,r), (th,th).
template <class Q, bool DIRECT, class G>
__host__ __device__ __forceinline__ void update_P(const G& g, typename Q::V (&Ln)[kList], const typename Q::V (&Prr)[3],
                                                  const typename Q::V (&Ptt)[3], typename Q::V (&dxo)[5])
{
    using V = typename Q::V;
    // Rows 3b+j of W = P G^T, then of V = W L^-T
    V v[5][6];
    {
        V trt[3], tvt[3];
        tr3<Q>(Ln[QLE_QO(0, 2, 0)], Ln[QLE_QO(0, 2, 1)], Ln[QLE_QO(0, 2, 2)], trt);   // P(r_j, th_k)
        tr3<Q>(Ln[QLE_QO(1, 2, 0)], Ln[QLE_QO(1, 2, 1)], Ln[QLE_QO(1, 2, 2)], tvt);   // P(v_j, th_k)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[0][k] = Prr[k];                v[0][3 + k] = trt[k];
            v[1][k] = Ln[QLE_QO(0, 1, k)];   v[1][3 + k] = tvt[k];
            v[2][k] = Ln[QLE_QO(0, 2, k)];   v[2][3 + k] = Ptt[k];
            v[3][k] = Ln[QLE_QO(0, 3, k)];   v[3][3 + k] = Ln[QLE_QO(2, 3, k)];
            v[4][k] = Ln[QLE_QO(0, 4, k)];   v[4][3 + k] = Ln[QLE_QO(2, 4, k)];
        }
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            if (!DIRECT) {
#pragma unroll
                for (int k = 0; k < 3; ++k) v[b][k] = v[b][k] + (v[b][3] * g.Gx(3 * k) + v[b][4] * g.Gx(3 * k + 1) + v[b][5] * g.Gx(3 * k + 2));
            }
#pragma unroll
            for (int m = 1; m < 6; ++m) {
#pragma unroll
                for (int m2 = 0; m2 < m; ++m2) v[b][m] = v[b][m] - g.Lm(lm_idx(m, m2)) * v[b][m2];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        V s = v[b][0] * g.yd(0);
#pragma unroll
        for (int m = 1; m < 6; ++m) s = s + v[b][m] * g.yd(m);
        dxo[b] = s;
    }
    // Written as accumulations with the sign in the scaled factor (P += v_i (-v_k / d)): `acc + dpp_read(a) * b` is one
    // v_fmac_f32 with the cross-lane read folded into its first operand; `acc - dpp_read(a) * b` would need the 3-operand form, which
    // cannot take a DPP operand, so every read would become a move of its own and a register (+55 VGPRs, profiles/r02_tuning.md).
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        V nvs[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) nvs[c] = v[c][m] * (-g.invd(m));
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            Ln[QLE_QD0(b)] = Ln[QLE_QD0(b)] + v[b][m] * nvs[b];
            Ln[QLE_QD1(b)] = Ln[QLE_QD1(b)] + Q::rot2(v[b][m]) * nvs[b];
#pragma unroll
            for (int c = b + 1; c < 5; ++c) {
                Ln[QLE_QO(b, c, 0)] = Ln[QLE_QO(b, c, 0)] + Q::template bc<0>(v[b][m]) * nvs[c];
                Ln[QLE_QO(b, c, 1)] = Ln[QLE_QO(b, c, 1)] + Q::template bc<1>(v[b][m]) * nvs[c];
                Ln[QLE_QO(b, c, 2)] = Ln[QLE_QO(b, c, 2)] + Q::template bc<2>(v[b][m]) * nvs[c];
            }
        }
    }
}

// ------------------------------------------------------------ inject, scalar
// EKF.cpp:486-501: apply the error state dx (15) to the nominal state.
template <class Q, typename T>
__host__ __device__ __forceinline__ void update_inject(const DevParams<T>& p, typename Q::V (&x)[16], const typename Q::V (&dx)[15])
{
    using V = typename Q::V;
    V q[4] = {x[6], x[7], x[8], x[9]};
    V dth[3] = {dx[6], dx[7], dx[8]}, qe[4], qn[4];
    q_exp<Q>(dth, qe);
    q_mul(q, qe, qn);
    q_norm<Q>(qn);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x[i] = x[i] + dx[i];
        x[3 + i] = x[3 + i] + dx[3 + i];
        x[10 + i] = V(p.bias_on) * (x[10 + i] + dx[9 + i]);
        x[13 + i] = V(p.bias_on) * (x[13 + i] + dx[12 + i]);
    }
    x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
}

}  // namespace quad
}  // namespace qle
