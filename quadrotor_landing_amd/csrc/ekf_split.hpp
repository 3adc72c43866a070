// ekf_split.hpp -- prediction_step (EKF.cpp:346-415) and correction_step (EKF.cpp:417-502) on a covariance that is SPLIT between
// registers and the LDS: the form the fp64 multirate replay (k_step_mr<double>, EKF.cpp:196-236) keeps a filter in.  gfx950 (CDNA4).
//
// Why.  One lane owns one filter and the replay keeps its covariance on chip for a dozen ticks.  In fp64 the packed triangle alone is
// 240 of the 512 registers a lone wave has, and only the 256 architectural ones can be VALU operands; with the correction's gain
// vectors next to it the compiler spilled 1.2-1.5 KB per lane (profiles/r03_tuning.md: 240 us per launch, 0.078 of the roofline).
// Here the two top block-rows of P -- rows r and v, 75 of the 120 values -- live in the LDS (element-major: element k of the wave's
// 64 filters is one 512-byte row, so every ds_read_b64 / ds_write_b64 is conflict-free), the three bottom block-rows (th, ab, wb:
// 45 values) stay in registers, and both the predict and the correction stream the top rows through registers one 3x3 block at a
// time.  4 waves x 75 x 512 B = 150 KiB of the CU's 160 KiB; nothing of P is ever live twice and no array reaches scratch.
//
//   F = L3 L2 L1 (ekf_device.hpp); with M = F P the new block-row b is (M_b,:) F^T and needs only OLD block-rows at or below b:
//     row r :  M1_c = P_rc + dT P_vc;   P'_rr = P_rr + dT (P_vr + M1_v),  P'_rv = M1_v + M1_th A^T + M1_ab B^T,
//              P'_rth = M1_th Rt^T - dTw M1_wb,  P'_rab = M1_ab,  P'_rwb = M1_wb
//     row v :  M2_c = P_vc + A P_thc + B P_abc;   P'_vv = M2_v + M2_th A^T + M2_ab B^T + C Qa C^T,  P'_vth = M2_th Rt^T - dTw M2_wb, ...
//     row th:  M3_c = Rt P_thc - dTw P_wbc;       P'_thth = M3_th Rt^T - dTw M3_wb + Qw, ...
//   so the step runs top-down in place, and while row r is formed only one block of row v is in registers at a time.
//   Correction: the batch form of ekf_fused.hpp -- S = G P G^T + R_k = L D L^T, V = (P G^T) L^-T, P <- P - V D^-1 V^T,
//   dx = V D^-1 L^-1 dy -- with the scalar parts of ekf_quad.hpp; the downdate passes over the LDS rows once.
// Same expressions as predict_cov_inplace_noq / ekf_step_fused, evaluated from another home of the operands.
//
// `Top` is where the two top block-rows live: ld(k) / st(k, v) with compile-time k in [0, kTopWords).  The device hands in an LDS
// window (LdsTop), the host build of the test suite a plain array (ArrayTop).
#pragma once

#include "ekf_device.hpp"
#include "ekf_quad.hpp"
#include "ekf_packed.hpp"

// Between two phases of the split algebra: nothing is scheduled across (the backend would otherwise hoist the next phase's LDS reads over
// the current one and hold both block sets in registers).
#ifndef QLE_FENCE_MASK
#define QLE_FENCE_MASK 0   // sched_barrier mask: which instruction classes MAY still cross (0: none; 2 | 4: VALU and SALU)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define QLE_PHASE_FENCE() __builtin_amdgcn_sched_barrier(QLE_FENCE_MASK)
#else
#define QLE_PHASE_FENCE() do { } while (0)
#endif

// Diagnostic build only (make dbg): s_memtime at the phase boundaries of the LAST split predict a wave ran (lane 0 writes), read back
// through qle_debug_split_clocks (tu_misc.hip).  Slots: 0 entry, 1 row r formed, 2 row v formed, 3 rows th / ab / wb done.
#if defined(QLE_MR_STAMPS) && defined(__HIPCC__)
namespace qle { static __device__ unsigned long long qle_dbg_split_clock[4096 * 8]; }
#endif
#if defined(QLE_MR_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define QLE_SPLIT_STAMP(k, dep)                                                                                             \
    do {                                                                                                                    \
        unsigned long long t_;                                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : "v"(dep) : "memory");                              \
        const unsigned w_ = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                                   \
        if ((threadIdx.x & 63) == 0 && w_ < 4096) ::qle::qle_dbg_split_clock[w_ * 8 + (k)] = t_;                            \
    } while (0)
#else
#define QLE_SPLIT_STAMP(k, dep) do { } while (0)
#endif

// The fences inside the predict (the per-tick part of the replay loop) have their own switch.  They are OFF: with MachineLICM off for the
// unit (Makefile) the predict needs none to stay out of scratch, and without them the backend overlaps the nominal-state chain with
// the LDS round trips of block-row r (k_step_mr<double> 94.0 -> 89.6 us, profiles/r04_tuning.md section 2).  The correction keeps its fences.
#ifndef QLE_PREDICT_FENCES
#define QLE_PREDICT_FENCES 0
#endif
// fences of the correction: 2 = between every 3 x 3 block of the LDS rows, 1 = between block-rows and halves only, 0 = none
#ifndef QLE_UPDATE_FENCES
#define QLE_UPDATE_FENCES 2
#endif
#if QLE_UPDATE_FENCES >= 2
#define QLE_BLOCK_FENCE() QLE_PHASE_FENCE()
#else
#define QLE_BLOCK_FENCE() do { } while (0)
#endif
#if QLE_UPDATE_FENCES >= 1
#define QLE_UPDATE_FENCE() QLE_PHASE_FENCE()
#else
#define QLE_UPDATE_FENCE() do { } while (0)
#endif
#if QLE_PREDICT_FENCES
#define QLE_PREDICT_FENCE() QLE_PHASE_FENCE()
#else
#define QLE_PREDICT_FENCE() do { } while (0)
#endif

namespace qle {

constexpr int kTopWords = 75;   // rows r (42) and v (33)
constexpr int kLoWords = 45;    // rows th (24), ab (15), wb (6)

// Upper triangle of a symmetric 3x3: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2).
__host__ __device__ constexpr int sym3(int i, int j) { return i <= j ? i * 3 - i * (i - 1) / 2 + (j - i) : j * 3 - j * (j - 1) / 2 + (i - j); }
// First word of block (b, c), b <= c, in its home: rows 0, 1 in `top`, rows 2..4 in `lo`.  Diagonal blocks hold 6 words, the others 9.
__host__ __device__ constexpr int split_base(int b, int c)
{
    return b == 0 ? (c == 0 ? 0 : 6 + 9 * (c - 1))
         : b == 1 ? (c == 1 ? 42 : 48 + 9 * (c - 2))
         : b == 2 ? (c == 2 ? 0 : 6 + 9 * (c - 3))
         : b == 3 ? (c == 3 ? 24 : 30)
                  : 39;
}
// Word of element (i, k), i <= k, in its home.
__host__ __device__ constexpr int split_word(int i, int k)
{
    return split_base(i / 3, k / 3) + (i / 3 == k / 3 ? sym3(i % 3, k % 3) : 3 * (i % 3) + k % 3);
}
enum : int {
    T_RR = 0, T_RV = 6, T_RT = 15, T_RA = 24, T_RW = 33, T_VV = 42, T_VT = 48, T_VA = 57, T_VW = 66,
    L_TT = 0, L_TA = 6, L_TW = 15, L_AA = 24, L_AW = 30, L_WW = 39
};

template <typename T>
struct ArrayTop {   // host checker
    T a[kTopWords];
    __host__ __device__ __forceinline__ T ld(int k) const { return a[k]; }
    __host__ __device__ __forceinline__ void st(int k, T v) { a[k] = v; }
};
#if defined(__HIPCC__)
template <typename T>
struct LdsTop {     // p = the wave's LDS window + lane; element k of the wave's 64 filters is the row at k * 64
    T* p;
    __device__ __forceinline__ T ld(int k) const { return p[k * 64]; }
    __device__ __forceinline__ void st(int k, T v) { p[k * 64] = v; }
};
#endif

template <typename T, class Top>
__host__ __device__ __forceinline__ void split_from_flat(const T (&P)[120], Top& top, T (&lo)[kLoWords])
{
#pragma unroll
    for (int i = 0; i < 15; ++i) {
#pragma unroll
        for (int k = i; k < 15; ++k) {
            if (i < 6) top.st(split_word(i, k), P[sidx(i, k)]);
            else lo[split_word(i, k)] = P[sidx(i, k)];
        }
    }
}
template <typename T, class Top>
__host__ __device__ __forceinline__ void split_to_flat(const Top& top, const T (&lo)[kLoWords], T (&P)[120])
{
#pragma unroll
    for (int i = 0; i < 15; ++i) {
#pragma unroll
        for (int k = i; k < 15; ++k) P[sidx(i, k)] = i < 6 ? top.ld(split_word(i, k)) : lo[split_word(i, k)];
    }
}

// The blocks of F one predicted tick needs, row-major (from the per-tick scalar part, packed_nominal in ekf_packed.hpp).
template <typename T>
struct SplitCtx {
    T A[9], B[9], R[9];   // F[v,th], F[v,ab], F[th,th]
    T CQC[6];             // C diag(Q_a) C^T, symmetric
    T dT, dTw;
};

// Nominal state (EKF.cpp:356-371) in place + the blocks of F.  Values: packed_nominal's.
template <typename T>
__host__ __device__ __forceinline__ void split_nominal(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], const T (&u)[6], T (&accel)[3], SplitCtx<T>& s)
{
    PackedCtx<T> c;
    packed_nominal<T>(p, nz, x, u, accel, c);
    s.dT = c.dT; s.dTw = c.dTw;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            s.A[3 * i + m] = c3_el(c.CA, i, m);
            s.B[3 * i + m] = c3_el(c.CB, i, m);
            s.R[3 * i + m] = c3_el(c.CR, i, m);
        }
#pragma unroll
        for (int k = i; k < 3; ++k) s.CQC[sym3(i, k)] = m3_elA(c.CQC, i, k);
    }
}

// P <- F P F^T + W Q W^T (EKF.cpp:412-414), top-down in place.
template <typename T, class Top>
__host__ __device__ __forceinline__ void split_predict_cov(const SplitCtx<T>& s, const Noise<T>& nz, Top& top, T (&lo)[kLoWords])
{
    const T dT = s.dT, dTw = s.dTw;
    const T (&A)[9] = s.A;
    const T (&B)[9] = s.B;
    const T (&R)[9] = s.R;
    QLE_SPLIT_STAMP(0, lo[0] + A[0]);
    // ---- block-row r ------------------------------------------------------------------------------------------------------------
    {
        T Mw[9], Ma[9], Mt[9], Mv[9], rvo[9], rr[6];
#pragma unroll
        for (int k = 0; k < 9; ++k) Mw[k] = top.ld(T_RW + k) + dT * top.ld(T_VW + k);
#pragma unroll
        for (int k = 0; k < 9; ++k) Ma[k] = top.ld(T_RA + k) + dT * top.ld(T_VA + k);
#pragma unroll
        for (int k = 0; k < 9; ++k) Mt[k] = top.ld(T_RT + k) + dT * top.ld(T_VT + k);
        {
            T vv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) vv[k] = top.ld(T_VV + k);
#pragma unroll
            for (int k = 0; k < 9; ++k) rvo[k] = top.ld(T_RV + k);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) Mv[3 * i + j] = rvo[3 * i + j] + dT * vv[sym3(i, j)];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = i; j < 3; ++j) rr[sym3(i, j)] = top.ld(T_RR + sym3(i, j)) + dT * (rvo[3 * j + i] + Mv[3 * i + j]);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) top.st(T_RR + k, rr[k]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                T sv = Mv[3 * i + j];
#pragma unroll
                for (int m = 0; m < 3; ++m) sv += A[3 * j + m] * Mt[3 * i + m];
#pragma unroll
                for (int m = 0; m < 3; ++m) sv += B[3 * j + m] * Ma[3 * i + m];
                top.st(T_RV + 3 * i + j, sv);
                const T st = (R[3 * j] * Mt[3 * i] + R[3 * j + 1] * Mt[3 * i + 1] + R[3 * j + 2] * Mt[3 * i + 2]) - dTw * Mw[3 * i + j];
                top.st(T_RT + 3 * i + j, st);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) top.st(T_RA + k, Ma[k]);
#pragma unroll
        for (int k = 0; k < 9; ++k) top.st(T_RW + k, Mw[k]);
    }
    QLE_SPLIT_STAMP(1, lo[0]);
    QLE_PREDICT_FENCE();
    // ---- block-row v ------------------------------------------------------------------------------------------------------------
    {
        T Mw[9], Ma[9], Mt[9], vto[9], vao[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                T sw = top.ld(T_VW + 3 * i + c);
#pragma unroll
                for (int m = 0; m < 3; ++m) sw += A[3 * i + m] * lo[L_TW + 3 * m + c];
#pragma unroll
                for (int m = 0; m < 3; ++m) sw += B[3 * i + m] * lo[L_AW + 3 * m + c];
                Mw[3 * i + c] = sw;
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) { vto[k] = top.ld(T_VT + k); vao[k] = top.ld(T_VA + k); }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                T st = vto[3 * i + c], sa = vao[3 * i + c];
#pragma unroll
                for (int m = 0; m < 3; ++m) { st += A[3 * i + m] * lo[L_TT + sym3(m, c)]; sa += A[3 * i + m] * lo[L_TA + 3 * m + c]; }
#pragma unroll
                for (int m = 0; m < 3; ++m) { st += B[3 * i + m] * lo[L_TA + 3 * c + m]; sa += B[3 * i + m] * lo[L_AA + sym3(m, c)]; }
                Mt[3 * i + c] = st;
                Ma[3 * i + c] = sa;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = i; k < 3; ++k) {
                T sv = top.ld(T_VV + sym3(i, k));
#pragma unroll
                for (int m = 0; m < 3; ++m) sv += A[3 * i + m] * vto[3 * k + m];
#pragma unroll
                for (int m = 0; m < 3; ++m) sv += B[3 * i + m] * vao[3 * k + m];
#pragma unroll
                for (int m = 0; m < 3; ++m) sv += Mt[3 * i + m] * A[3 * k + m];
#pragma unroll
                for (int m = 0; m < 3; ++m) sv += Ma[3 * i + m] * B[3 * k + m];
                top.st(T_VV + sym3(i, k), sv + s.CQC[sym3(i, k)]);
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const T st = (R[3 * j] * Mt[3 * i] + R[3 * j + 1] * Mt[3 * i + 1] + R[3 * j + 2] * Mt[3 * i + 2]) - dTw * Mw[3 * i + j];
                top.st(T_VT + 3 * i + j, st);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) top.st(T_VA + k, Ma[k]);
#pragma unroll
        for (int k = 0; k < 9; ++k) top.st(T_VW + k, Mw[k]);
    }
    QLE_SPLIT_STAMP(2, lo[0]);
    QLE_PREDICT_FENCE();
    // ---- block-row th (registers) -----------------------------------------------------------------------------------------------
    {
        T N[9], tw[9], ta[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ta[3 * i + c] = (R[3 * i] * lo[L_TA + c] + R[3 * i + 1] * lo[L_TA + 3 + c] + R[3 * i + 2] * lo[L_TA + 6 + c]) - dTw * lo[L_AW + 3 * c + i];
                N[3 * i + c] = (R[3 * i] * lo[L_TT + sym3(0, c)] + R[3 * i + 1] * lo[L_TT + sym3(1, c)] + R[3 * i + 2] * lo[L_TT + sym3(2, c)]) -
                               dTw * lo[L_TW + 3 * c + i];
                tw[3 * i + c] = (R[3 * i] * lo[L_TW + c] + R[3 * i + 1] * lo[L_TW + 3 + c] + R[3 * i + 2] * lo[L_TW + 6 + c]) - dTw * lo[L_WW + sym3(i, c)];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = i; k < 3; ++k)
                lo[L_TT + sym3(i, k)] = (N[3 * i] * R[3 * k] + N[3 * i + 1] * R[3 * k + 1] + N[3 * i + 2] * R[3 * k + 2]) - dTw * tw[3 * i + k];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) { lo[L_TA + k] = ta[k]; lo[L_TW + k] = tw[k]; }
    }
    // ---- W Q W^T on the diagonal blocks of th, ab, wb (C Qa C^T went into P_vv above) ------------------------------------------------
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        lo[L_TT + sym3(i, i)] += nz.Q[3 + i];
        lo[L_AA + sym3(i, i)] += nz.Q[6 + i];
        lo[L_WW + sym3(i, i)] += nz.Q[9 + i];
    }
    QLE_SPLIT_STAMP(3, lo[L_TT] + lo[L_WW]);
}

// prediction_step, EKF.cpp:346-415, on the split covariance.
template <typename T, class Top>
__host__ __device__ __forceinline__ void ekf_predict_split(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], Top& top, T (&lo)[kLoWords],
                                                           const T (&u)[6], T (&accel)[3])
{
    SplitCtx<T> s;
    split_nominal<T>(p, nz, x, u, accel, s);
    split_predict_cov<T>(s, nz, top, lo);
}

// correction_step, EKF.cpp:417-502, on the split covariance (batch form).  x: the predicted nominal state in, the corrected one out.
template <typename T, bool DIRECT, class Top, typename EmitObs>
__host__ __device__ __forceinline__ void ekf_update_split(const DevParams<T>& p, const Noise<T>& nzl, T (&x)[16], Top& top, T (&lo)[kLoWords],
                                                          const T (&z)[7], EmitObs&& emit_obs)
{
    using SQ = quad::ScalarQ<T>;
    quad::UpdU<T> f;
    T rk12[9];   // R_k[0:3, 3:6]
    {
        quad::NoiseV<T> nz;
#pragma unroll
        for (int k = 0; k < 6; ++k) nz.R[k] = nzl.R[k];
        quad::FactorIn<T> in;
        quad::update_innovation<SQ, T, DIRECT>(p, x, z, in.dy_, emit_obs);
        quad::update_noise<SQ, T, DIRECT>(p, nz, x, in.gx, in.rk);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                in.frr[3 * i + k] = top.ld(T_RR + sym3(i, k));
                in.frt[3 * i + k] = top.ld(T_RT + 3 * i + k);
                in.ftt[3 * i + k] = lo[L_TT + sym3(i, k)];
            }
        }
        quad::update_factor<SQ, DIRECT>(in, f);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) rk12[3 * i + k] = in.rk[quad::rk_idx(i, 3 + k)];
        }
    }
    // P <- P - V D^-1 V^T with V = (P G^T) L^-T as TWO rank-3 downdates, so that only half of V (45 values) is ever live:
    //   V1 = W1 L11^-T from the columns W1 = P G1^T (G1 = [I 0 Gx 0 0]) of the covariance as it is;
    //   P1 = P - V1 D1^-1 V1^T;
    //   V2 = (W2 - V1 L21^T) L22^-T, and W2 - V1 L21^T = P1 G2^T - V1 Cx with Cx = D1^-1 L11^-1 R_k[0:3, 3:6] (G2 = [0 0 I 0 0]: the th
    //   columns of the ALREADY downdated P1), because L21 D1 L11^T = S21 = G2 W1 + R21;
    //   P' = P1 - V2 D2^-1 V2^T;  dx = V1 yd[0:3] + V2 yd[3:6].
    T Cx[3][3];   // Cx[m][k], m < 3: row m of D1^-1 L11^-1 R12
    {
        T r12[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int k = 0; k < 3; ++k) r12[i][k] = rk12[3 * i + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // forward substitution with the unit lower L11, column by column
            r12[1][k] -= f.Lm[quad::lm_idx(1, 0)] * r12[0][k];
            r12[2][k] -= f.Lm[quad::lm_idx(2, 0)] * r12[0][k] + f.Lm[quad::lm_idx(2, 1)] * r12[1][k];
#pragma unroll
            for (int m = 0; m < 3; ++m) Cx[m][k] = r12[m][k] * f.invd[m];
        }
    }
    T V[15][3], dx[15];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            V[a][k] = top.ld(T_RR + sym3(a, k));
            V[3 + a][k] = top.ld(T_RV + 3 * k + a);
            V[6 + a][k] = top.ld(T_RT + 3 * k + a);
            V[9 + a][k] = top.ld(T_RA + 3 * k + a);
            V[12 + a][k] = top.ld(T_RW + 3 * k + a);
        }
    }
    if (!DIRECT) {   // + P(a, th) Gx^T
#pragma unroll
        for (int a = 0; a < 15; ++a) {
            T t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                t[k] = a < 3 ? top.ld(T_RT + 3 * a + k) : a < 6 ? top.ld(T_VT + 3 * (a - 3) + k) : a < 9 ? lo[L_TT + sym3(a - 6, k)]
                     : a < 12 ? lo[L_TA + 3 * k + (a - 9)] : lo[L_TW + 3 * k + (a - 12)];
#pragma unroll
            for (int k = 0; k < 3; ++k)   // pinned evaluation order: compact and full records must agree bit for bit
                V[a][k] += fused_fma(t[2], f.Gx[3 * k + 2], fused_fma(t[1], f.Gx[3 * k + 1], t[0] * f.Gx[3 * k]));
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        QLE_UPDATE_FENCE();
        if (half == 1) {   // the th columns of P1, less V1 Cx, take V1's place row by row
#pragma unroll
            for (int a = 0; a < 15; ++a) {
                T w[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T pk = a < 3 ? top.ld(T_RT + 3 * a + k) : a < 6 ? top.ld(T_VT + 3 * (a - 3) + k) : a < 9 ? lo[L_TT + sym3(a - 6, k)]
                               : a < 12 ? lo[L_TA + 3 * k + (a - 9)] : lo[L_TW + 3 * k + (a - 12)];
                    w[k] = pk - fused_fma(V[a][2], Cx[2][k], fused_fma(V[a][1], Cx[1][k], V[a][0] * Cx[0][k]));
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) V[a][k] = w[k];
            }
        }
        // eliminate with the unit lower diagonal block of L (L11 / L22), then this half's share of dx
#pragma unroll
        for (int a = 0; a < 15; ++a) {
            T (&v)[3] = V[a];
            v[1] -= f.Lm[quad::lm_idx(3 * half + 1, 3 * half)] * v[0];
            v[2] -= f.Lm[quad::lm_idx(3 * half + 2, 3 * half)] * v[0] + f.Lm[quad::lm_idx(3 * half + 2, 3 * half + 1)] * v[1];
            const T acc = v[0] * f.yd[3 * half] + v[1] * f.yd[3 * half + 1] + v[2] * f.yd[3 * half + 2];
            dx[a] = half == 0 ? acc : dx[a] + acc;
        }
        // P(i, k) += sum_m (-V(i, m) / d_m) V(k, m), block-row by block-row; the LDS rows pass through registers one block at a time
#pragma unroll
        for (int b = 0; b < 5; ++b) {
            QLE_UPDATE_FENCE();
            T NV[3][3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int m = 0; m < 3; ++m) NV[i][m] = V[3 * b + i][m] * (-f.invd[3 * half + m]);
            }
#pragma unroll
            for (int c = b; c < 5; ++c) {
                if (b < 2 && c > b) QLE_BLOCK_FENCE();
#pragma unroll
                for (int i = 0; i < 3; ++i) {
#pragma unroll
                    for (int k = (c == b ? i : 0); k < 3; ++k) {
                        const int w = split_word(3 * b + i, 3 * c + k);
                        T acc = b < 2 ? top.ld(w) : lo[w];
#pragma unroll
                        for (int m = 0; m < 3; ++m) acc += NV[i][m] * V[3 * c + k][m];
                        if (b < 2) top.st(w, acc);
                        else lo[w] = acc;
                    }
                }
            }
        }
    }
    QLE_UPDATE_FENCE();
    quad::update_inject<SQ, T>(p, x, dx);   // EKF.cpp:486-501
}

}  // namespace qle
