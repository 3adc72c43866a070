// ekf_device.hpp -- per-filter arithmetic of the batched relative-pose EKF,
// written for one filter per lane with the whole 15x15 covariance (packed
// symmetric, 120 words) held in VGPRs.  gfx950 (CDNA4) only.
//
// Reference behaviour being reproduced (mbrymer/quadrotor_landing,
// quad_state_estimation/):
//   src/relative_pose_EKF.cpp:346-415  prediction_step
//   src/relative_pose_EKF.cpp:417-502  correction_step
//   src/quaternion_helper.cpp:9-100    quaternion_exp / log / norm, skew_symm
// The reference forms dense 15x15 F and W and multiplies them out.  Here the
// block structure of F is used directly: F = L3 * L2 * L1 with
//   L1: r <- r + dT v                       (F[r,v]      EKF.cpp:380)
//   L2: v <- v + A th + Bm ab               (F[v,th], F[v,ab]  EKF.cpp:381,399)
//   L3: th <- Rt th - dT wb                 (F[th,th], F[th,wb] EKF.cpp:383-395,400)
// and P <- L3 (L2 (L1 P L1^T) L2^T) L3^T + W Q W^T is applied as three in-place
// symmetric congruences on the packed upper triangle (~620 FMA instead of
// ~7000).  The correction decorrelates the 6-D measurement (LDL^T of
// R_k = N R N^T) and then fuses its six components one scalar at a time,
// P <- P - h h^T / s with h read from the live covariance -- algebraically
// identical to K = P G^T S^-1, P <- (I - K G) P of EKF.cpp:475-481, different
// rounding, and no 15x6 copy of P G^T in registers.  Every array index below is a compile-time constant after
// unrolling, so nothing lives in scratch memory.
#pragma once
#include <cmath>

#include <hip/hip_runtime.h>

namespace qle {

// ------------------------------------------------------------------ scalars
__device__ __forceinline__ float t_sqrt(float v) { return sqrtf(v); }
__device__ __forceinline__ double t_sqrt(double v) { return sqrt(v); }
__device__ __forceinline__ float t_atan2(float a, float b) { return atan2f(a, b); }
__device__ __forceinline__ double t_atan2(double a, double b) { return atan2(a, b); }
__device__ __forceinline__ void t_sincos(float v, float* s, float* c) { sincosf(v, s, c); }
__device__ __forceinline__ void t_sincos(double v, double* s, double* c) { sincos(v, s, c); }

// An fma the backend cannot re-associate: where one expression is instantiated in several kernels whose results must agree bit for bit,
// `a*b + c*d` must not be left to contract as fma(a, b, c*d) in one and fma(c, d, a*b) in the other.
template <typename T> __host__ __device__ __forceinline__ T fused_fma(T a, T b, T c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}
__host__ __device__ __forceinline__ float fused_fma(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}
// Packed storage order of the symmetric 15x15 covariance (120 words).
//
// P is 5x5 blocks of 3x3 (r, v, th, ab, wb).  The order serves two kernel families at once:
//   * one lane per filter (k_predict / k_step): the words of block-row r come first, then v, th, ab, wb, so
//     that the levelled predict can store a block-row as soon as it is final (qle::level_first_word);
//   * four lanes per filter (ekf_quad.hpp): lane j in {0,1,2} of a quad owns COLUMN j of every 3x3 block
//     (b, c), b <= c, and loads 40 of the 120 words as ten 16-byte quads: memory quad 3m + l is the m-th
//     quad of lane l.  Every lane's 40-word list has the same shape: per block-row b the stored part of
//     the diagonal block (D0 = element (3b+l, 3b+l), D1 = element (3b+(l+2)%3, 3b+l)) followed by the
//     three words of column l of each block (b, c), c > b.  The symmetric duplicate a lane does not store,
//     element (3b+(l+1)%3, 3b+l), is lane (l+1)%3's D1.
// Element (i, k), i <= k, block (b, c) = (i/3, k/3), position (ii, kk) = (i%3, k%3):
//   b <  c : lane kk, list position base(b) + 2 + 3 (c-b-1) + ii
//   b == c : ii == kk -> lane kk, position base(b);  (0,1) -> lane 1, (1,2) -> lane 2, (0,2) -> lane 0, position base(b)+1
// with base = {0, 14, 25, 33, 38}; word = 4 (3 (pos/4) + lane) + pos%4.
__host__ __device__ constexpr int quad_group_base(int b) { return b == 0 ? 0 : b == 1 ? 14 : b == 2 ? 25 : b == 3 ? 33 : 38; }
__host__ __device__ constexpr int quad_word(int lane, int pos) { return 4 * (3 * (pos / 4) + lane) + pos % 4; }
__host__ __device__ constexpr int sidx_formula(int i, int j)
{
    const int lo = i <= j ? i : j, hi = i <= j ? j : i;
    const int b = lo / 3, c = hi / 3, ii = lo % 3, kk = hi % 3;
    if (b == c) {
        if (ii == kk) return quad_word(kk, quad_group_base(b));
        return quad_word((ii == 0 && kk == 1) ? 1 : (ii == 1 && kk == 2) ? 2 : 0, quad_group_base(b) + 1);
    }
    return quad_word(kk, quad_group_base(b) + 2 + 3 * (c - b - 1) + ii);
}
// Device code folds the formula (every index is a compile-time constant after unrolling); host builds of this header (the
// test-only CPU build of the engine's arithmetic) look it up in a table instead of evaluating the divisions at run time.
struct SidxTable { unsigned char v[15][15]; };
constexpr SidxTable make_sidx_table()
{
    SidxTable t{};
    for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) t.v[i][j] = (unsigned char)sidx_formula(i, j);
    return t;
}
template <int Dummy = 0> struct SidxHolder { static constexpr SidxTable table = make_sidx_table(); };
template <int Dummy> constexpr SidxTable SidxHolder<Dummy>::table;
__host__ __device__ constexpr int sidx(int i, int j)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return sidx_formula(i, j);
#else
    return SidxHolder<>::table.v[i][j];
#endif
}
// Block-row (0 = r ... 4 = wb) of the element stored in word w.
__host__ __device__ constexpr int word_block_row(int w)
{
    const int pos = 4 * ((w / 4) / 3) + w % 4;
    return pos < 14 ? 0 : pos < 25 ? 1 : pos < 33 ? 2 : pos < 38 ? 3 : 4;
}
// Inverse of sidx: the element (row <= col) stored in word w.
__host__ __device__ constexpr int word_lane(int w) { return (w / 4) % 3; }
__host__ __device__ constexpr int word_pos(int w) { return 4 * ((w / 4) / 3) + w % 4; }
__host__ __device__ constexpr int word_row(int w)
{
    const int l = word_lane(w), b = word_block_row(w), off = word_pos(w) - quad_group_base(b);
    if (off == 0) return 3 * b + l;
    if (off == 1) return 3 * b + (l == 2 ? 1 : 0);              // lane 0: (0,2), lane 1: (0,1), lane 2: (1,2)
    return 3 * b + (off - 2) % 3;
}
__host__ __device__ constexpr int word_col(int w)
{
    const int l = word_lane(w), b = word_block_row(w), off = word_pos(w) - quad_group_base(b);
    if (off == 0) return 3 * b + l;
    if (off == 1) return 3 * b + (l == 1 ? 1 : 2);
    return 3 * (b + 1 + (off - 2) / 3) + l;
}
// First word (a multiple of vw) from which on every stored element belongs to block-row >= b: the quads from
// there to the end are final once the levelled predict has finished block-row b.
__host__ __device__ constexpr int level_first_word(int b, int vw)
{
    int w0 = 120;
    for (int w = 119; w >= 0; --w) {
        if (word_block_row(w) < b) break;
        if (w % vw == 0) w0 = w;
    }
    return w0;
}
#define QLE_PS(i, j) P[::qle::sidx((i), (j))]

// Uniform (per-launch) parameters in the compute dtype; derived on the host by
// qle_params_derive == initialize_params (EKF.cpp:87-125).
template <typename T>
struct DevParams {
    T dT;            // dT_nom (EKF.cpp:90,356)
    T dTw;           // est_bias ? dT : 0   (F[th,wb] = -dT I only with est_bias, EKF.cpp:400)
    T bias_on;       // est_bias ? 1 : 0    (F[v,ab] EKF.cpp:399; bias injection EKF.cpp:494-498)
    T small_ang_tol; // EKF.cpp:80
    T g[3];          // EKF.cpp:81
    T q_vc[4];       // x,y,z,w, normalised (EKF.cpp:121)
    T C_vc[9];       // EKF.cpp:122
    T r_v_cv[3];     // EKF.cpp:55
    T Q[12];         // diag(Q_a,Q_w,Q_ab,Q_wb) (EKF.cpp:100-112); zero where !est_bias
    T R[6];          // diag(R_r,R_ang) (EKF.cpp:116-118)
    T ab_static[3];  // EKF.cpp:357
    T wb_static[3];  // EKF.cpp:358
    int32_t compact; // state records hold the 9 x 9 pose block of P only (est_bias = false, EKF.cpp:92; ekf_kernels.hpp)
};

// Noise / static-bias values a tick actually uses: shared or per filter (cfg 5).
template <typename T>
struct Noise {
    T Q[12];
    T ab_static[3];
    T wb_static[3];
    T R[6];
};

// sin(h) / (2 h) and cos(h): the vector scale and the scalar part of exp(phi), h = |phi| / 2 (QH.cpp:9-28).  Series in h^2 on
// |h| <= pi/4 (truncation < 3e-10 in fp32, < 3e-20 in fp64).  Larger half-angles (no physical rate: more than 90 degrees per tick) are
// halved until they fit and the result is doubled back -- sin 2a / 2a = (sin a / a) cos a, cos 2a = 1 - 2 a^2 (sin a / a)^2, no square
// root.  The number of halvings is WAVE-UNIFORM (the largest any lane needs; normally zero, and then nothing below differs from the
// plain series): a per-lane branch to the library's sincos here put a divergent region into the middle of kernels that sit at the
// register limit, and the backend placed register copies at that region's join in front of the EXEC restore -- copies that never
// happened for the lanes that had skipped the branch (profiles/r04_tuning.md section 1: the GPU memory access faults).
template <typename T>
__host__ __device__ __forceinline__ void half_angle_sinc_cos(T h2, T& k, T& ch)
{
    constexpr T kLim = T(0.6168502750680849);   // (pi/4)^2
    int halvings = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    while (halvings < 600 && __any(h2 > kLim)) { h2 *= T(0.25); ++halvings; }
#else
    while (halvings < 600 && h2 > kLim) { h2 *= T(0.25); ++halvings; }
#endif
    T sc, cc;   // sin(a) / a and cos(a), a^2 = h2
    if (sizeof(T) == 4) {
        // sin h / h = 1 - h2/6 + h2^2/120 - h2^3/5040 + h2^4/362880 - h2^5/39916800
        T s = T(-2.505210838544172e-08);
        s = s * h2 + T(2.755731922398589e-06);
        s = s * h2 + T(-1.984126984126984e-04);
        s = s * h2 + T(8.333333333333333e-03);
        s = s * h2 + T(-1.666666666666667e-01);
        s = s * h2 + T(1);
        sc = s;
        T c = T(2.08767569878681e-09);
        c = c * h2 + T(-2.755731922398589e-07);
        c = c * h2 + T(2.48015873015873e-05);
        c = c * h2 + T(-1.388888888888889e-03);
        c = c * h2 + T(4.166666666666666e-02);
        c = c * h2 + T(-0.5);
        cc = c * h2 + T(1);
    } else {
        // 1/(2n+1)! and 1/(2n)! down to n = 10
        T s = T(1.957294106339126e-20);
        s = s * h2 + T(-8.22063524662433e-18);
        s = s * h2 + T(2.811457254345521e-15);
        s = s * h2 + T(-7.647163731819816e-13);
        s = s * h2 + T(1.605904383682161e-10);
        s = s * h2 + T(-2.505210838544172e-08);
        s = s * h2 + T(2.755731922398589e-06);
        s = s * h2 + T(-1.984126984126984e-04);
        s = s * h2 + T(8.333333333333333e-03);
        s = s * h2 + T(-1.666666666666667e-01);
        s = s * h2 + T(1);
        sc = s;
        T c = T(4.110317623312165e-19);
        c = c * h2 + T(-1.561920696858623e-16);
        c = c * h2 + T(4.779477332387385e-14);
        c = c * h2 + T(-1.147074559772972e-11);
        c = c * h2 + T(2.08767569878681e-09);
        c = c * h2 + T(-2.755731922398589e-07);
        c = c * h2 + T(2.48015873015873e-05);
        c = c * h2 + T(-1.388888888888889e-03);
        c = c * h2 + T(4.166666666666666e-02);
        c = c * h2 + T(-0.5);
        cc = c * h2 + T(1);
    }
    for (int j = 0; j < halvings; ++j) {   // wave-uniform trip count
        const T s2 = sc * sc * h2;
        sc = sc * cc;
        cc = T(1) - T(2) * s2;
        h2 = h2 * T(4);
    }
    k = T(0.5) * sc;
    ch = cc;
}

// ------------------------------------------------------ quaternion helpers
// quaternion_norm, QH.cpp:61-73: normalise, then flip to the w >= -0.75 cover.
template <typename T>
__device__ __forceinline__ void quat_norm(T (&q)[4])
{
    T n = t_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    T inv = T(1) / n;
    T s = (q[3] * inv < T(-0.75)) ? -inv : inv;
    q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s;
}

// quaternion_exp, QH.cpp:9-33 (including the final quaternion_norm at :30).
// sin(|v|/2)/|v| and cos(|v|/2) come from half_angle_sinc_cos: the series the reference's small-angle branch truncates (QH.cpp:19-24),
// valid for every |v| <= pi/2 without a square root, a division or a branch; larger arguments take the library path inside it.
template <typename T>
__device__ __forceinline__ void quat_exp(const T (&v)[3], T (&q)[4])
{
    const T n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    T k, ch;
    half_angle_sinc_cos(T(0.25) * n2, k, ch);
    q[0] = v[0] * k; q[1] = v[1] * k; q[2] = v[2] * k; q[3] = ch;
    quat_norm(q);
}

// quaternion_log, QH.cpp:36-58.
template <typename T>
__device__ __forceinline__ void quat_log(const T (&q)[4], T (&v)[3])
{
    T m = t_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    bool small = m < T(1E-10);
    T mw = m / q[3];
    T k_small = T(2) / q[3] * (T(1) - mw * mw * (T(1) / T(3)));
    T k_full = T(2) * t_atan2(m, q[3]) / (small ? T(1) : m);
    T k = small ? k_small : k_full;
    v[0] = k * q[0]; v[1] = k * q[1]; v[2] = k * q[2];
}

// Hamilton product, storage x,y,z,w (Eigen operator*, EKF.cpp:367,431,448,488).
template <typename T>
__device__ __forceinline__ void quat_mul(const T (&a)[4], const T (&b)[4], T (&o)[4])
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

// Rotation matrix of a unit quaternion (Eigen toRotationMatrix, EKF.cpp:359,429).
template <typename T>
__device__ __forceinline__ void quat_to_rot(const T (&q)[4], T (&C)[9])
{
    T x = q[0], y = q[1], z = q[2], w = q[3];
    T tx = x + x, ty = y + y, tz = z + z;
    T twx = tx * w, twy = ty * w, twz = tz * w;
    T txx = tx * x, txy = ty * x, txz = tz * x;
    T tyy = ty * y, tyz = tz * y, tzz = tz * z;
    C[0] = T(1) - (tyy + tzz); C[1] = txy - twz;           C[2] = txz + twy;
    C[3] = txy + twz;           C[4] = T(1) - (txx + tzz); C[5] = tyz - twx;
    C[6] = txz - twy;           C[7] = tyz + twx;           C[8] = T(1) - (txx + tyy);
}

// ------------------------------------------------------------------ predict
// The parts of prediction_step (EKF.cpp:346-415) that do not touch the covariance: see PredictCtx below.
template <typename T>
struct PredictCtx {
    T a[3], dw[3];   // bias-corrected specific force, rotation increment dT (w - wb)
    T C[9];          // rotation matrix of the OLD attitude (EKF.cpp:359)
    T X[3][6];       // [A | Bm], A = -dT C [a]x (EKF.cpp:381), Bm = -dT C with est_bias (EKF.cpp:399)
    T Rt[3][3];      // F[th,th], EKF.cpp:383-395
    T dT, dTw;
};

// Nominal state and the blocks of F in ONE pass without a transcendental call (used by every predict of the engine).
// exp(phi) is needed for the nominal state anyway (EKF.cpp:367); F[th,th] = AngleAxis(-|phi|, phi/|phi|) (EKF.cpp:383-395) is the
// rotation matrix of its conjugate, so the second sine / cosine (of the whole angle) and the normalised axis are not needed, and the
// half-angle pair comes from half_angle_sinc_cos: no libm slow-path branch splits the block, the scheduler can interleave the chain with
// the covariance work next to it.  The small-angle branches of the reference (QH.cpp:19-28, EKF.cpp:385-389) are the same series
// truncated (agreement 1e-20).
template <typename T>
__device__ __forceinline__ void predict_nominal_lean(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], const T (&u)[6], T (&accel)[3],
                                                     PredictCtx<T>& c)
{
    const T dT = p.dT;
    c.dT = dT; c.dTw = p.dTw;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c.a[i] = u[i] - x[10 + i] - nz.ab_static[i];                   // EKF.cpp:357
        c.dw[i] = dT * (u[3 + i] - x[13 + i] - nz.wb_static[i]);       // EKF.cpp:358, :367
    }
    const T q[4] = {x[6], x[7], x[8], x[9]};
    quat_to_rot(q, c.C);                                               // EKF.cpp:359
#pragma unroll
    for (int i = 0; i < 3; ++i) accel[i] = (c.C[3 * i] * c.a[0] + c.C[3 * i + 1] * c.a[1] + c.C[3 * i + 2] * c.a[2]) + p.g[i];   // EKF.cpp:362
    const T n2 = c.dw[0] * c.dw[0] + c.dw[1] * c.dw[1] + c.dw[2] * c.dw[2];
    T k, ch;
    half_angle_sinc_cos(T(0.25) * n2, k, ch);                          // QH.cpp:9-28
    T qe[4] = {c.dw[0] * k, c.dw[1] * k, c.dw[2] * k, ch};
    quat_norm(qe);                                                     // QH.cpp:30
    T qn[4];
    quat_mul(q, qe, qn);                                               // EKF.cpp:367
    quat_norm(qn);                                                     // EKF.cpp:371
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x[i] += dT * x[3 + i];                                         // EKF.cpp:365
        x[3 + i] += dT * accel[i];                                     // EKF.cpp:366
    }
    x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
    {   // Rt = R(exp(phi))^T
        T Re[9];
        quat_to_rot(qe, Re);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int m = 0; m < 3; ++m) c.Rt[i][m] = Re[3 * m + i];
        }
    }
    const T mdT = -dT, mdTb = -dT * p.bias_on;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T c0 = c.C[3 * i], c1 = c.C[3 * i + 1], c2 = c.C[3 * i + 2];
        c.X[i][0] = mdT * (c1 * c.a[2] - c2 * c.a[1]);                 // -dT C [a]x, EKF.cpp:381
        c.X[i][1] = mdT * (c2 * c.a[0] - c0 * c.a[2]);
        c.X[i][2] = mdT * (c0 * c.a[1] - c1 * c.a[0]);
        c.X[i][3] = mdTb * c0; c.X[i][4] = mdTb * c1; c.X[i][5] = mdTb * c2;   // EKF.cpp:399
    }
}

// Nominal state, EKF.cpp:356-371: x is advanced in place, accel = pose_accel (EKF.cpp:362).
template <typename T>
__device__ __forceinline__ void predict_nominal(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], const T (&u)[6], T (&accel)[3],
                                                PredictCtx<T>& c)
{
    const T dT = p.dT;
    c.dT = dT; c.dTw = p.dTw;
    T w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        c.a[i] = u[i] - x[10 + i] - nz.ab_static[i];
        w[i] = u[3 + i] - x[13 + i] - nz.wb_static[i];
    }
    T q[4] = {x[6], x[7], x[8], x[9]};
    quat_to_rot(q, c.C);
#pragma unroll
    for (int i = 0; i < 3; ++i) accel[i] = (c.C[3 * i] * c.a[0] + c.C[3 * i + 1] * c.a[1] + c.C[3 * i + 2] * c.a[2]) + p.g[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.dw[i] = dT * w[i];
    T qe[4], qn[4];
    quat_exp(c.dw, qe);
    quat_mul(q, qe, qn);
    quat_norm(qn);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x[i] += dT * x[3 + i];
        x[3 + i] += dT * accel[i];
    }
    x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
}

// The blocks of F that are not identity: X = [A | Bm] and Rt.
template <typename T>
__device__ __forceinline__ void predict_jacobians(const DevParams<T>& p, PredictCtx<T>& c)
{
    const T mdT = -c.dT, mdTb = -c.dT * p.bias_on;
    const T (&a)[3] = c.a;
    const T (&dw)[3] = c.dw;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T c0 = c.C[3 * i], c1 = c.C[3 * i + 1], c2 = c.C[3 * i + 2];
        c.X[i][0] = mdT * (c1 * a[2] - c2 * a[1]);
        c.X[i][1] = mdT * (c2 * a[0] - c0 * a[2]);
        c.X[i][2] = mdT * (c0 * a[1] - c1 * a[0]);
        c.X[i][3] = mdTb * c0; c.X[i][4] = mdTb * c1; c.X[i][5] = mdTb * c2;
    }
    T (&Rt)[3][3] = c.Rt;
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

// The covariance part of prediction_step, in place: P <- L3 (L2 (L1 P L1^T) L2^T) L3^T + W Q W^T as three symmetric congruences on the
// packed upper triangle.
template <typename T>
__device__ __forceinline__ void predict_cov_inplace_noq(const PredictCtx<T>& ctx, T (&P)[120])
{
    const T dT = ctx.dT, dTw = ctx.dTw;
    const T (&X)[3][6] = ctx.X;
    const T (&Rt)[3][3] = ctx.Rt;
    // ---- congruence 1: r <- r + dT v -------------------------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int c = 6; c < 15; ++c) QLE_PS(i, c) += dT * QLE_PS(3 + i, c);
    }
    {
        T rv_old[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                rv_old[i][j] = QLE_PS(i, 3 + j);
                QLE_PS(i, 3 + j) = rv_old[i][j] + dT * QLE_PS(3 + i, 3 + j);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = i; j < 3; ++j) QLE_PS(i, j) += dT * (rv_old[j][i] + QLE_PS(i, 3 + j));
        }
    }

    // ---- congruence 2: v <- v + X [th; ab] -------------------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // P_rv += P_r,[th ab] X^T
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T s = QLE_PS(i, 3 + j);
#pragma unroll
            for (int m = 0; m < 6; ++m) s += X[j][m] * QLE_PS(i, 6 + m);
            QLE_PS(i, 3 + j) = s;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {  // P_v,wb += X P_[th ab],wb
#pragma unroll
        for (int c = 12; c < 15; ++c) {
            T s = QLE_PS(3 + i, c);
#pragma unroll
            for (int m = 0; m < 6; ++m) s += X[i][m] * QLE_PS(6 + m, c);
            QLE_PS(3 + i, c) = s;
        }
    }
    {
        T vj_old[3][6], N[3][6];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 6; ++c) vj_old[i][c] = QLE_PS(3 + i, 6 + c);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                T s = vj_old[i][c];
#pragma unroll
                for (int m = 0; m < 6; ++m) s += X[i][m] * QLE_PS(6 + m, 6 + c);
                N[i][c] = s;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = i; k < 3; ++k) {
                T s = QLE_PS(3 + i, 3 + k);
#pragma unroll
                for (int m = 0; m < 6; ++m) s += X[i][m] * vj_old[k][m];
#pragma unroll
                for (int m = 0; m < 6; ++m) s += N[i][m] * X[k][m];
                QLE_PS(3 + i, 3 + k) = s;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 6; ++c) QLE_PS(3 + i, 6 + c) = N[i][c];
        }
    }

    // ---- congruence 3: th <- Rt th - dTw wb ------------------------------
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // rows r,v: P_k,th <- P_k,th Rt^T - dTw P_k,wb
        T o0 = QLE_PS(i, 6), o1 = QLE_PS(i, 7), o2 = QLE_PS(i, 8);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            QLE_PS(i, 6 + j) = (Rt[j][0] * o0 + Rt[j][1] * o1 + Rt[j][2] * o2) - dTw * QLE_PS(i, 12 + j);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // P_th,ab <- Rt P_th,ab - dTw P_wb,ab
        T o0 = QLE_PS(6, 9 + c), o1 = QLE_PS(7, 9 + c), o2 = QLE_PS(8, 9 + c);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            QLE_PS(6 + i, 9 + c) = (Rt[i][0] * o0 + Rt[i][1] * o1 + Rt[i][2] * o2) - dTw * QLE_PS(9 + c, 12 + i);
    }
    {
        T N[3][3], tw[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {  // N = Rt P_th,th - dTw P_wb,th   (old P_th,wb)
#pragma unroll
            for (int m = 0; m < 3; ++m)
                N[i][m] = (Rt[i][0] * QLE_PS(6, 6 + m) + Rt[i][1] * QLE_PS(7, 6 + m) + Rt[i][2] * QLE_PS(8, 6 + m)) -
                          dTw * QLE_PS(6 + m, 12 + i);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {  // P_th,wb <- Rt P_th,wb - dTw P_wb,wb
            T o0 = QLE_PS(6, 12 + c), o1 = QLE_PS(7, 12 + c), o2 = QLE_PS(8, 12 + c);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                tw[i][c] = (Rt[i][0] * o0 + Rt[i][1] * o1 + Rt[i][2] * o2) - dTw * QLE_PS(12 + i, 12 + c);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = i; k < 3; ++k)
                QLE_PS(6 + i, 6 + k) = (N[i][0] * Rt[k][0] + N[i][1] * Rt[k][1] + N[i][2] * Rt[k][2]) - dTw * tw[i][k];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) QLE_PS(6 + i, 12 + c) = tw[i][c];
        }
    }

}
template <typename T>
__device__ __forceinline__ void predict_cov_inplace(const PredictCtx<T>& ctx, const Noise<T>& nz, T (&P)[120])
{
    predict_cov_inplace_noq<T>(ctx, P);
    const T (&C)[9] = ctx.C;
    // ---- W Q W^T, EKF.cpp:402-414: blockdiag(0, C Qa C^T, Qw, Qab, Qwb) ----
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T cq0 = C[3 * i] * nz.Q[0], cq1 = C[3 * i + 1] * nz.Q[1], cq2 = C[3 * i + 2] * nz.Q[2];
#pragma unroll
        for (int k = i; k < 3; ++k) QLE_PS(3 + i, 3 + k) += cq0 * C[3 * k] + cq1 * C[3 * k + 1] + cq2 * C[3 * k + 2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        QLE_PS(6 + i, 6 + i) += nz.Q[3 + i];
        QLE_PS(9 + i, 9 + i) += nz.Q[6 + i];
        QLE_PS(12 + i, 12 + i) += nz.Q[9 + i];
    }
}

// prediction_step, EKF.cpp:346-415.  x and P are updated in place; accel is pose_accel (EKF.cpp:362).
template <typename T>
__device__ __forceinline__ void ekf_predict(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], T (&P)[120],
                                            const T (&u)[6], T (&accel)[3])
{
    PredictCtx<T> c;
    predict_nominal_lean<T>(p, nz, x, u, accel, c);
    predict_cov_inplace<T>(c, nz, P);
}

// ------------------------------------------------- predict, levelled variant
// Same arithmetic as ekf_predict, organised for a memory-bound single wave per SIMD: the new
// covariance Pn is computed block-row by block-row from the OLD P
//   level 0: rows ab, wb    (need old rows ab, wb)
//   level 1: rows th        (need old rows th, ab, wb)
//   level 2: rows v         (need old rows v, th, ab, wb)
//   level 3: rows r         (need old rows r, v)
// F = L3 L2 L1 as in ekf_predict; the formulas below are the composition written out per block.  The levels read only the old
// covariance, so a caller may run them in any order: ekf_predict_levels goes bottom-up (stores of a level overlap the loads of the
// rows above), the fused tick (ekf_fused.hpp) takes the rows the innovation covariance needs first.
#define QLE_PN(i, j) Pn[::qle::sidx((i), (j))]
// level 0: rows ab (9..11), wb (12..14)
template <typename T>
__device__ __forceinline__ void predict_level0(const Noise<T>& nz, const T (&P)[120], T (&Pn)[120])
{
#pragma unroll
    for (int i = 9; i < 15; ++i) {
#pragma unroll
        for (int k = i; k < 15; ++k) QLE_PN(i, k) = QLE_PS(i, k);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        QLE_PN(9 + i, 9 + i) += nz.Q[6 + i];
        QLE_PN(12 + i, 12 + i) += nz.Q[9 + i];
    }
}
// level 1: rows th (6..8)
template <typename T>
__device__ __forceinline__ void predict_level1(const PredictCtx<T>& c, const Noise<T>& nz, const T (&P)[120], T (&Pn)[120])
{
    const T (&Rt)[3][3] = c.Rt;
    const T dTw = c.dTw;
    T Mtt[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            // th,wb and th,ab: Rt O_t* - dTw O_w*
            QLE_PN(6 + i, 12 + cc) = (Rt[i][0] * QLE_PS(6, 12 + cc) + Rt[i][1] * QLE_PS(7, 12 + cc) + Rt[i][2] * QLE_PS(8, 12 + cc)) - dTw * QLE_PS(12 + i, 12 + cc);
            QLE_PN(6 + i, 9 + cc) = (Rt[i][0] * QLE_PS(6, 9 + cc) + Rt[i][1] * QLE_PS(7, 9 + cc) + Rt[i][2] * QLE_PS(8, 9 + cc)) - dTw * QLE_PS(12 + i, 9 + cc);
            Mtt[i][cc] = (Rt[i][0] * QLE_PS(6, 6 + cc) + Rt[i][1] * QLE_PS(7, 6 + cc) + Rt[i][2] * QLE_PS(8, 6 + cc)) - dTw * QLE_PS(12 + i, 6 + cc);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = i; k < 3; ++k)
            QLE_PN(6 + i, 6 + k) = (Mtt[i][0] * Rt[k][0] + Mtt[i][1] * Rt[k][1] + Mtt[i][2] * Rt[k][2]) - dTw * QLE_PN(6 + i, 12 + k) +
                                   ((i == k) ? nz.Q[3 + i] : T(0));
    }
}
// level 2: rows v (3..5)
template <typename T>
__device__ __forceinline__ void predict_level2(const PredictCtx<T>& c, const Noise<T>& nz, const T (&P)[120], T (&Pn)[120])
{
    const T (&Rt)[3][3] = c.Rt;
    const T (&X)[3][6] = c.X;
    const T (&C)[9] = c.C;
    const T dTw = c.dTw;
    T Mv[3][9];  // M_v,[th ab wb] = O_v* + X O_[th ab],*
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int cc = 0; cc < 9; ++cc) {
            T acc = QLE_PS(3 + i, 6 + cc);
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += X[i][m] * QLE_PS(6 + m, 6 + cc);
            Mv[i][cc] = acc;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            QLE_PN(3 + i, 12 + cc) = Mv[i][6 + cc];
            QLE_PN(3 + i, 9 + cc) = Mv[i][3 + cc];
            QLE_PN(3 + i, 6 + cc) = (Mv[i][0] * Rt[cc][0] + Mv[i][1] * Rt[cc][1] + Mv[i][2] * Rt[cc][2]) - dTw * Mv[i][6 + cc];
        }
#pragma unroll
        for (int k = i; k < 3; ++k) {
            T acc = QLE_PS(3 + i, 3 + k);
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += X[i][m] * QLE_PS(3 + k, 6 + m);
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += Mv[i][m] * X[k][m];
            acc += C[3 * i] * nz.Q[0] * C[3 * k] + C[3 * i + 1] * nz.Q[1] * C[3 * k + 1] + C[3 * i + 2] * nz.Q[2] * C[3 * k + 2];
            QLE_PN(3 + i, 3 + k) = acc;
        }
    }
}
// level 3: rows r (0..2)
template <typename T>
__device__ __forceinline__ void predict_level3(const PredictCtx<T>& c, const T (&P)[120], T (&Pn)[120])
{
    const T (&Rt)[3][3] = c.Rt;
    const T (&X)[3][6] = c.X;
    const T dT = c.dT, dTw = c.dTw;
    T M1[3][12];  // M1_r,[v th ab wb] = O_r* + dT O_v*
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int cc = 0; cc < 12; ++cc) M1[i][cc] = QLE_PS(i, 3 + cc) + dT * QLE_PS(3 + i, 3 + cc);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            QLE_PN(i, 12 + cc) = M1[i][9 + cc];
            QLE_PN(i, 9 + cc) = M1[i][6 + cc];
            QLE_PN(i, 6 + cc) = (M1[i][3] * Rt[cc][0] + M1[i][4] * Rt[cc][1] + M1[i][5] * Rt[cc][2]) - dTw * M1[i][9 + cc];
            T acc = M1[i][cc];
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += M1[i][3 + m] * X[cc][m];
            QLE_PN(i, 3 + cc) = acc;
        }
#pragma unroll
        for (int k = i; k < 3; ++k) QLE_PN(i, k) = QLE_PS(i, k) + dT * (QLE_PS(k, 3 + i) + M1[i][k]);
    }
}
#undef QLE_PN

// `done(level)` is called after each level (-1: x is final) so the caller can issue that level's stores while the loads of the
// rows above are still in flight (the caller issues the loads bottom-up too).
template <typename T, typename Done>
__device__ __forceinline__ void ekf_predict_levels(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], const T (&P)[120],
                                                   const T (&u)[6], T (&accel)[3], T (&Pn)[120], Done done)
{
    PredictCtx<T> c;
    predict_nominal_lean<T>(p, nz, x, u, accel, c);
    done(-1);  // x is final
    predict_level0<T>(nz, P, Pn);
    done(0);
    predict_level1<T>(c, nz, P, Pn);
    done(1);
    predict_level2<T>(c, nz, P, Pn);
    done(2);
    predict_level3<T>(c, P, Pn);
    done(3);
}

// ------------------------------------------------------------------- update
// correction_step, EKF.cpp:417-502, in two parts.  z = [r_c_tc(3), q_ct(x,y,z,w)(4)].
//
// ekf_update_prepare: everything that does not need the covariance -- the innovation (EKF.cpp:429-450), G (EKF.cpp:453-459),
//   R_k = N R N^T (EKF.cpp:462-472) and its L D L^T factor, the decorrelated measurement y' = L^-1 dy, G' = L^-1 G.  It is a long
//   DEPENDENT chain of scalar instructions (a lone wave pays ~9 cycles for each, profiles/r02_tuning.md), so the tick kernels run it
//   right after the predict's own scalar part, while the covariance loads are still in flight, instead of after the predict.
// ekf_update_apply: the six scalar fusions on the live covariance (EKF.cpp:475-481 in sequential form) and the injection
//   (EKF.cpp:486-501).
// obs receives r_t_vt_obs(3), q_tv_obs(4) (members written at EKF.cpp:431-443) through emit_obs(obs), called as soon as the reported
// observation exists so that a caller that only stores it does not keep seven more values alive.
template <typename T>
struct UpdatePrep {
    T dy[6];      // decorrelated innovation y'
    T d[6];       // diagonal of R' = D
    T Gm[6][6];   // G' = L^-1 G over the state columns J = {r0,r1,r2,th0,th1,th2}
};

template <typename T, bool DIRECT, typename EmitObs>
__device__ __forceinline__ void ekf_update_prepare(const DevParams<T>& p, const Noise<T>& nz, const T (&x)[16], const T (&z)[7],
                                                   UpdatePrep<T>& u, EmitObs&& emit_obs)
{
    T obs[7];
    T q[4] = {x[6], x[7], x[8], x[9]};
    T r[3] = {x[0], x[1], x[2]};
    T Cc[9];
    quat_to_rot(q, Cc);                                      // EKF.cpp:429
    T qo[4];
    {
        T qct[4] = {z[3], z[4], z[5], z[6]}, t[4];
        quat_mul(p.q_vc, qct, t);                            // EKF.cpp:431
        qo[0] = -t[0]; qo[1] = -t[1]; qo[2] = -t[2]; qo[3] = t[3];
        quat_norm(qo);                                       // EKF.cpp:432
    }
    T (&dy)[6] = u.dy;
    {
        // EKF.cpp:434-444: -(q * T_vc * r_c_tc) with q = q_tv_obs (direct) or q_check
        T Cq[9];
        if (DIRECT) quat_to_rot(qo, Cq);
        else {
#pragma unroll
            for (int i = 0; i < 9; ++i) Cq[i] = Cc[i];
        }
        T pv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            pv[i] = (p.C_vc[3 * i] * z[0] + p.C_vc[3 * i + 1] * z[1] + p.C_vc[3 * i + 2] * z[2]) + p.r_v_cv[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T ro = -(Cq[3 * i] * pv[0] + Cq[3 * i + 1] * pv[1] + Cq[3 * i + 2] * pv[2]);
            obs[i] = ro;
            dy[i] = ro - r[i];                               // EKF.cpp:447
        }
        obs[3] = qo[0]; obs[4] = qo[1]; obs[5] = qo[2]; obs[6] = qo[3];
        emit_obs(obs);
        T qc[4] = {-q[0], -q[1], -q[2], q[3]}, dq[4], dth[3];
        quat_mul(qc, qo, dq);                                // EKF.cpp:448
        quat_norm(dq);                                       // EKF.cpp:449
        quat_log(dq, dth);                                   // EKF.cpp:450
        dy[3] = dth[0]; dy[4] = dth[1]; dy[5] = dth[2];
    }

    // G restricted to the six state columns it touches, J = {r0,r1,r2,th0,th1,th2} = {0,1,2,6,7,8}:
    //   G = [I Gx; 0 I] over J, Gx = Cc [Cc^T r]x unless direct (EKF.cpp:453-459).
    T (&Gm)[6][6] = u.Gm;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = 0; b < 6; ++b) Gm[a][b] = (a == b) ? T(1) : T(0);
    }
    if (!DIRECT) {
        T b[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = Cc[i] * r[0] + Cc[3 + i] * r[1] + Cc[6 + i] * r[2];  // Cc^T r
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            T c0 = Cc[3 * i], c1 = Cc[3 * i + 1], c2 = Cc[3 * i + 2];
            Gm[i][3] = c1 * b[2] - c2 * b[1];
            Gm[i][4] = c2 * b[0] - c0 * b[2];
            Gm[i][5] = c0 * b[1] - c1 * b[0];
        }
    }
    // R_k = N R N^T (upper triangle), N = [-Cc C_vc, [r]x (direct) ; 0, C_vc]  (EKF.cpp:462-472)
    T Rk[6][6];
    {
        T N00[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                N00[i][j] = -(Cc[3 * i] * p.C_vc[j] + Cc[3 * i + 1] * p.C_vc[3 + j] + Cc[3 * i + 2] * p.C_vc[6 + j]);
        }
        // [r]x rows (EKF.cpp:465-468): (0,-r2,r1), (r2,0,-r0), (-r1,r0,0)
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
                Rk[i][3 + j] = DIRECT ? Sr[i][0] * nz.R[3] * p.C_vc[3 * j] + Sr[i][1] * nz.R[4] * p.C_vc[3 * j + 1] +
                                            Sr[i][2] * nz.R[5] * p.C_vc[3 * j + 2]
                                      : T(0);
        }
    }
    // Decorrelate the measurement: R_k = L D L^T (unit lower L); y' = L^-1 dy, G' = L^-1 G,
    // R' = D is diagonal, so the six components can be fused one scalar at a time and
    // h = P g'^T is read from the live covariance (no 15x6 copy of P G^T is kept).
    // In the direct method G' stays unit lower triangular: only entries m <= c are touched.
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        u.d[c] = Rk[c][c];
        const T invd = T(1) / u.d[c];
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
}

template <typename T, bool DIRECT>
__device__ __forceinline__ void ekf_update_apply(const DevParams<T>& p, T (&x)[16], T (&P)[120], const UpdatePrep<T>& u)
{
    constexpr int J[6] = {0, 1, 2, 6, 7, 8};
    const T (&Gm)[6][6] = u.Gm;
    // Six scalar updates (EKF.cpp:475-481 in sequential form): for component c
    //   h = P g'_c^T, s = g'_c h + d_c, k = h/s, dx += k (y'_c - g'_c dx), P -= k h^T.
    T dx[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) dx[k] = T(0);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        T h[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            T acc = T(0);
#pragma unroll
            for (int m = 0; m < 6; ++m)
                if (!DIRECT || m <= c) acc += Gm[c][m] * QLE_PS(k, J[m]);
            h[k] = acc;
        }
        T s = u.d[c], nu = u.dy[c];
#pragma unroll
        for (int m = 0; m < 6; ++m)
            if (!DIRECT || m <= c) { s += Gm[c][m] * h[J[m]]; nu -= Gm[c][m] * dx[J[m]]; }
        // gain k = h / s is never stored: dx += h (nu/s), P(i,k) -= (h_i/s) h_k.  In fp32 the sign goes into the scaled h_i once, so
        // that the 120 updates are plain accumulations (v_fmac_f32, 4 bytes) and not v_fma_f32 with a negated operand (8 bytes): a
        // lone wave on long straight-line code is bound by instruction supply (profiles/r02_tuning.md; k_step 15.4 -> 14.7 us with the
        // fast fp32 division).  fp64 keeps the subtraction: there the other form costs registers (38 -> 41.6 us at 65 536 filters).
        const T inv = T(1) / s;
        const T c_nu = inv * nu;
#pragma unroll
        for (int k = 0; k < 15; ++k) dx[k] += h[k] * c_nu;
        if (sizeof(T) == 4) {
            const T ninv = -inv;
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                const T nhi = h[i] * ninv;
#pragma unroll
                for (int k = i; k < 15; ++k) QLE_PS(i, k) += nhi * h[k];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                const T hi = h[i] * inv;
#pragma unroll
                for (int k = i; k < 15; ++k) QLE_PS(i, k) -= hi * h[k];
            }
        }
    }

    // inject, EKF.cpp:486-501
    T q[4] = {x[6], x[7], x[8], x[9]};
    T dth[3] = {dx[6], dx[7], dx[8]}, qe[4], qn[4];
    quat_exp(dth, qe);
    quat_mul(q, qe, qn);
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

template <typename T, bool DIRECT, typename EmitObs>
__device__ __forceinline__ void ekf_update_emit(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], T (&P)[120],
                                                const T (&z)[7], EmitObs&& emit_obs)
{
    UpdatePrep<T> u;
    ekf_update_prepare<T, DIRECT>(p, nz, x, z, u, emit_obs);
    ekf_update_apply<T, DIRECT>(p, x, P, u);
}
template <typename T, bool DIRECT>
__device__ __forceinline__ void ekf_update(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], T (&P)[120],
                                           const T (&z)[7], T (&obs)[7])
{
    ekf_update_emit<T, DIRECT>(p, nz, x, P, z, [&](const T (&o)[7]) {
#pragma unroll
        for (int k = 0; k < 7; ++k) obs[k] = o[k];
    });
}

}  // namespace qle
