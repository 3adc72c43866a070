// ekf_packed.hpp -- prediction_step (EKF.cpp:346-415) for a covariance that STAYS IN REGISTERS over many ticks: the replay loop of the
// multirate filter (k_step_mr, EKF.cpp:222-226) and the on-chip-resident kernel.  gfx950 (CDNA4) only.
//
// Why a second formulation.  At 65 536 filters a SIMD holds exactly one wave, and a lone wave issues one VALU instruction per 4-5
// cycles whatever the instruction is (profiles/r02_tuning.md section 10): a kernel that keeps P on chip is bound by the NUMBER of
// instructions per filter.  The one-lane predict of ekf_device.hpp costs ~1 370 of them per replayed tick (711 FMA-class, ~100 for the
// two sincosf with their slow paths, selects, copies).  Here the same algebra is laid out so that almost every FMA is one half of a
// v_pk_fma_f32 (two FMAs per issue slot) and the per-tick scalar part needs no transcendental call:
//
//   * P is held as its fifteen 3x3 blocks (r, v, th, ab, wb; upper block triangle, diagonal blocks in full), every block in ONE
//     register layout M3: three column pairs a[i] = (m(i,0), m(i,1)), the row pair c = (m(0,2), m(1,2)) and the scalar d = m(2,2).
//     Read the same registers as a[j] = (m(0,j), m(1,j)), c = (m(2,0), m(2,1)) and they hold the TRANSPOSE ("layout R" of m ==
//     layout A of m^T); toggle() converts one into the other with two pair moves.
//   * With that, each of the three products the block-structured congruence is made of runs as 5 chains (4 packed + 1 scalar) instead
//     of 9 scalar ones, every broadcast operand is an op_sel of an existing register (hipcc folds it, no move):
//        lmulAA  o(A) += C S(A)        lmulRA  o(R) += C S(A)        rmulAR  o(A) += S(R) C^T
//     C a 3x3 coefficient matrix held as column pairs over its row index (C3: cc[m] = (C(0,m), C(1,m)), c2[m] = C(2,m)).
//   * F = L3 L2 L1 as in ekf_device.hpp; the new block-rows are formed FROM THE OLD P top-down (r rows, then v, th, biases): a level
//     reads only block-rows at or below its own and writes its own, so the whole step is in place (no second copy of P).
//   * F[th,th] = AngleAxis(-|phi|, phi/|phi|) (EKF.cpp:383-395) is the rotation matrix of the conjugate of exp(phi), which the nominal
//     state needs anyway (EKF.cpp:367): one half-angle sine / cosine per tick, as branch-free polynomials for |phi|/2 <= pi/4 (any
//     physical rate: 0.79 rad per tick) with the libm path behind a wave-level branch beyond.  The small-angle branches of the
//     reference (QH.cpp:19-28, EKF.cpp:385-389) are the same series truncated; they agree to 1e-20.
// ~900 instructions per tick in fp32 (of which ~330 packed) against ~1 370.  Device code runs the block form in fp32 only (k_step_mr<float>);
// the fp64 replay keeps its covariance split between the LDS and registers (ekf_split.hpp) and the on-chip-resident kernel runs the in-place
// congruences of ekf_device.hpp.  The host build (test suite) instantiates the block form in both dtypes.
#pragma once

#include "ekf_device.hpp"

namespace qle {

template <typename T> struct PairOf;
template <> struct PairOf<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct PairOf<double> { typedef double type __attribute__((ext_vector_type(2))); };
template <typename T> using pair_t = typename PairOf<T>::type;

template <typename T> __device__ __forceinline__ pair_t<T> pk_bc(T s) { pair_t<T> r = {s, s}; return r; }
template <typename T> __device__ __forceinline__ pair_t<T> pk_mk(T a, T b) { pair_t<T> r = {a, b}; return r; }

// One 3x3 block.  Layout A: a[i] = (m(i,0), m(i,1)), c = (m(0,2), m(1,2)), d = m(2,2).
//                Layout R: a[j] = (m(0,j), m(1,j)), c = (m(2,0), m(2,1)), d = m(2,2)   (== layout A of the transpose).
template <typename T>
struct M3 {
    pair_t<T> a[3];
    pair_t<T> c;
    T d;
};
// 3x3 coefficient matrix: cc[m] = (C(0,m), C(1,m)), c2[m] = C(2,m).
template <typename T>
struct C3 {
    pair_t<T> cc[3];
    T c2[3];
};

template <typename T> __device__ __forceinline__ T m3_elA(const M3<T>& m, int i, int j)
{
    return j < 2 ? (j == 0 ? m.a[i].x : m.a[i].y) : (i < 2 ? (i == 0 ? m.c.x : m.c.y) : m.d);
}
template <typename T> __device__ __forceinline__ T m3_elR(const M3<T>& m, int i, int j) { return m3_elA(m, j, i); }
template <typename T> __device__ __forceinline__ void m3_setA(M3<T>& m, int i, int j, T v)
{
    if (j < 2) { if (j == 0) m.a[i].x = v; else m.a[i].y = v; }
    else if (i < 2) { if (i == 0) m.c.x = v; else m.c.y = v; }
    else m.d = v;
}
template <typename T> __device__ __forceinline__ T c3_el(const C3<T>& c, int i, int m) { return i == 0 ? c.cc[m].x : (i == 1 ? c.cc[m].y : c.c2[m]); }
template <typename T> __device__ __forceinline__ void c3_set(C3<T>& c, int i, int m, T v)
{
    if (i == 0) c.cc[m].x = v; else if (i == 1) c.cc[m].y = v; else c.c2[m] = v;
}
template <typename T> __device__ __forceinline__ void m3_zero(M3<T>& m)
{
    m.a[0] = m.a[1] = m.a[2] = m.c = pk_bc(T(0));
    m.d = T(0);
}
// layout A <-> layout R of the same matrix (an involution): two pair moves, the rest is renaming
template <typename T> __device__ __forceinline__ M3<T> m3_toggle(const M3<T>& s)
{
    M3<T> o;
    o.a[0] = pk_mk(s.a[0].x, s.a[1].x);
    o.a[1] = pk_mk(s.a[0].y, s.a[1].y);
    o.a[2] = s.c;
    o.c = s.a[2];
    o.d = s.d;
    return o;
}
// o += s * S, both in the same layout.  SYM: only the registers that hold the upper triangle of a symmetric block in layout A.
template <typename T, bool SYM = false> __device__ __forceinline__ void m3_axpy(M3<T>& o, T s, const M3<T>& S)
{
    const pair_t<T> sp = pk_bc(s);
    o.a[0] += sp * S.a[0];
    o.a[1] += sp * S.a[1];
    if (!SYM) o.a[2] += sp * S.a[2];
    o.c += sp * S.c;
    o.d += s * S.d;
}
// acc (+)= a * b: SET starts the accumulator with the product (no zero to initialise, no add of a zero)
template <bool SET, typename V> __device__ __forceinline__ void pk_mac(V& acc, const V& a, const V& b)
{
    if (SET) acc = a * b;
    else acc += a * b;
}
// o(A) += C S(A)      (SET: o(A) = C S(A))
template <typename T, bool SYM = false, bool SET = false> __device__ __forceinline__ void m3_lmulAA(M3<T>& o, const C3<T>& C, const M3<T>& S)
{
    {
        const T s2 = m3_elA(S, 0, 2);
        pk_mac<SET>(o.a[0], pk_bc(c3_el(C, 0, 0)), S.a[0]);
        pk_mac<SET>(o.a[1], pk_bc(c3_el(C, 1, 0)), S.a[0]);
        if (!SYM) pk_mac<SET>(o.a[2], pk_bc(c3_el(C, 2, 0)), S.a[0]);
        pk_mac<SET>(o.c, C.cc[0], pk_bc(s2));
        pk_mac<SET>(o.d, C.c2[0], s2);
    }
#pragma unroll
    for (int m = 1; m < 3; ++m) {
        const T s2 = m3_elA(S, m, 2);
        o.a[0] += pk_bc(c3_el(C, 0, m)) * S.a[m];
        o.a[1] += pk_bc(c3_el(C, 1, m)) * S.a[m];
        if (!SYM) o.a[2] += pk_bc(c3_el(C, 2, m)) * S.a[m];
        o.c += C.cc[m] * pk_bc(s2);
        o.d += C.c2[m] * s2;
    }
}
// o(R) += C S(A)      (SET: o(R) = C S(A))
template <typename T, bool SET = false> __device__ __forceinline__ void m3_lmulRA(M3<T>& o, const C3<T>& C, const M3<T>& S)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) pk_mac<SET>(o.a[j], C.cc[0], pk_bc(m3_elA(S, 0, j)));
    pk_mac<SET>(o.c, pk_bc(C.c2[0]), S.a[0]);
    pk_mac<SET>(o.d, C.c2[0], m3_elA(S, 0, 2));
#pragma unroll
    for (int m = 1; m < 3; ++m) {
#pragma unroll
        for (int j = 0; j < 3; ++j) o.a[j] += C.cc[m] * pk_bc(m3_elA(S, m, j));
        o.c += pk_bc(C.c2[m]) * S.a[m];
        o.d += C.c2[m] * m3_elA(S, m, 2);
    }
}
// o(A) += S C^T with S given in layout R:  o(i,j) = sum_m S(i,m) C(j,m)      (SET: o(A) = S C^T)
template <typename T, bool SYM = false, bool SET = false> __device__ __forceinline__ void m3_rmulAR(M3<T>& o, const M3<T>& S, const C3<T>& C)
{
    pk_mac<SET>(o.a[0], pk_bc(m3_elR(S, 0, 0)), C.cc[0]);
    pk_mac<SET>(o.a[1], pk_bc(m3_elR(S, 1, 0)), C.cc[0]);
    if (!SYM) pk_mac<SET>(o.a[2], pk_bc(m3_elR(S, 2, 0)), C.cc[0]);
    pk_mac<SET>(o.c, S.a[0], pk_bc(C.c2[0]));
    pk_mac<SET>(o.d, m3_elR(S, 2, 0), C.c2[0]);
#pragma unroll
    for (int m = 1; m < 3; ++m) {
        o.a[0] += pk_bc(m3_elR(S, 0, m)) * C.cc[m];
        o.a[1] += pk_bc(m3_elR(S, 1, m)) * C.cc[m];
        if (!SYM) o.a[2] += pk_bc(m3_elR(S, 2, m)) * C.cc[m];
        o.c += S.a[m] * pk_bc(C.c2[m]);
        o.d += m3_elR(S, 2, m) * C.c2[m];
    }
}
// a symmetric block in layout A after its upper registers were formed: (1,0) <- (0,1), row 2 <- column 2
template <typename T> __device__ __forceinline__ void m3_symmetrise(M3<T>& m)
{
    m.a[1].x = m.a[0].y;
    m.a[2] = m.c;
}

// The covariance as blocks: blk(b, c), b <= c, over (r, v, th, ab, wb), all in layout A; diagonal blocks hold the full symmetric 3x3.
template <typename T>
struct PackedCov {
    M3<T> B[15];
    static __host__ __device__ constexpr int idx(int b, int c) { return b * 5 - b * (b - 1) / 2 + (c - b); }
    __device__ __forceinline__ M3<T>& blk(int b, int c) { return B[idx(b, c)]; }
    __device__ __forceinline__ const M3<T>& blk(int b, int c) const { return B[idx(b, c)]; }
};

// packed memory order (sidx, ekf_device.hpp) <-> blocks: register renaming plus the moves that make pairs adjacent
template <typename T>
__device__ __forceinline__ void cov_pack(const T (&P)[120], PackedCov<T>& S)
{
#pragma unroll
    for (int b = 0; b < 5; ++b) {
#pragma unroll
        for (int c = b; c < 5; ++c) {
            M3<T>& m = S.blk(b, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) m3_setA(m, i, j, P[sidx(3 * b + i, 3 * c + j)]);   // sidx is symmetric in its arguments
            }
        }
    }
}
template <typename T>
__device__ __forceinline__ void cov_unpack(const PackedCov<T>& S, T (&P)[120])
{
#pragma unroll
    for (int b = 0; b < 5; ++b) {
#pragma unroll
        for (int c = b; c < 5; ++c) {
            const M3<T>& m = S.blk(b, c);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = (b == c ? i : 0); j < 3; ++j) P[sidx(3 * b + i, 3 * c + j)] = m3_elA(m, i, j);
            }
        }
    }
}

// What one predicted tick needs besides P: the blocks of F that are not identity, as coefficient matrices, and C Qa C^T.
template <typename T>
struct PackedCtx {
    C3<T> CA;     // A = -dT C [a]x                   (F[v,th], EKF.cpp:381)
    C3<T> CB;     // Bm = -dT C with est_bias          (F[v,ab], EKF.cpp:399)
    C3<T> CR;     // Rt = F[th,th]                     (EKF.cpp:383-395)
    M3<T> CQC;    // C diag(Q_a) C^T, layout A, upper registers (W Q W^T, EKF.cpp:402-414)
    T dT, dTw;
};

// Nominal state (EKF.cpp:356-371) advanced in place, accel = pose_accel (EKF.cpp:362), and the coefficient matrices of this tick.
template <typename T>
__device__ __forceinline__ void packed_nominal(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], const T (&u)[6], T (&accel)[3], PackedCtx<T>& c)
{
    const T dT = p.dT;
    c.dT = dT; c.dTw = p.dTw;
    T a[3], dw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = u[i] - x[10 + i] - nz.ab_static[i];                     // EKF.cpp:357
        dw[i] = dT * (u[3 + i] - x[13 + i] - nz.wb_static[i]);         // EKF.cpp:358, :367
    }
    const T q[4] = {x[6], x[7], x[8], x[9]};
    T C[9];
    quat_to_rot(q, C);                                                 // EKF.cpp:359
#pragma unroll
    for (int i = 0; i < 3; ++i) accel[i] = (C[3 * i] * a[0] + C[3 * i + 1] * a[1] + C[3 * i + 2] * a[2]) + p.g[i];   // EKF.cpp:362
    // exp(phi), QH.cpp:9-33, including its final quaternion_norm
    const T n2 = dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2];
    T k, ch;
    half_angle_sinc_cos(T(0.25) * n2, k, ch);
    T qe[4] = {dw[0] * k, dw[1] * k, dw[2] * k, ch};
    quat_norm(qe);
    T qn[4];
    quat_mul(q, qe, qn);                                               // EKF.cpp:367
    quat_norm(qn);                                                     // EKF.cpp:371
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        x[i] += dT * x[3 + i];                                         // EKF.cpp:365
        x[3 + i] += dT * accel[i];                                     // EKF.cpp:366
    }
    x[6] = qn[0]; x[7] = qn[1]; x[8] = qn[2]; x[9] = qn[3];
    // F[th,th] = rotation by -|phi| about phi (EKF.cpp:383-395) = R(conj(exp(phi))) = R(exp(phi))^T
    {
        T Re[9];
        quat_to_rot(qe, Re);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int m = 0; m < 3; ++m) c3_set(c.CR, i, m, Re[3 * m + i]);
        }
    }
    const T mdT = -dT, mdTb = -dT * p.bias_on;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T c0 = C[3 * i], c1 = C[3 * i + 1], c2 = C[3 * i + 2];
        c3_set(c.CA, i, 0, mdT * (c1 * a[2] - c2 * a[1]));             // -dT C [a]x, EKF.cpp:381
        c3_set(c.CA, i, 1, mdT * (c2 * a[0] - c0 * a[2]));
        c3_set(c.CA, i, 2, mdT * (c0 * a[1] - c1 * a[0]));
        c3_set(c.CB, i, 0, mdTb * c0); c3_set(c.CB, i, 1, mdTb * c1); c3_set(c.CB, i, 2, mdTb * c2);   // EKF.cpp:399
    }
    // C diag(Qa) C^T: (C diag(Qa)) in layout R, times C^T
    {
        C3<T> CC;
        M3<T> S;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int m = 0; m < 3; ++m) c3_set(CC, i, m, C[3 * i + m]);
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) S.a[m] = CC.cc[m] * pk_bc(nz.Q[m]);
        S.c = pk_mk(C[6] * nz.Q[0], C[7] * nz.Q[1]);
        S.d = C[8] * nz.Q[2];
        m3_rmulAR<T, true, true>(c.CQC, S, CC);
        c.CQC.a[2] = c.CQC.c;
    }
}

// P <- F P F^T + W Q W^T (EKF.cpp:412-414) on the blocks, in place, top-down.
template <typename T>
__device__ __forceinline__ void packed_cov_predict(const PackedCtx<T>& c, const Noise<T>& nz, PackedCov<T>& S)
{
    constexpr int R = 0, V = 1, TH = 2, AB = 3, WB = 4;
    const T dT = c.dT, mdTw = -c.dTw;
    // ---- level 3: rows r.  r' = r + dT v, then the column maps of v and th ----------------------------------------------------
    {
        M3<T> M1v = S.blk(R, V), M1t = S.blk(R, TH), M1a = S.blk(R, AB), M1w = S.blk(R, WB);
        m3_axpy(M1v, dT, S.blk(V, V));
        m3_axpy(M1t, dT, S.blk(V, TH));
        m3_axpy(M1a, dT, S.blk(V, AB));
        m3_axpy(M1w, dT, S.blk(V, WB));
        const M3<T> M1tR = m3_toggle(M1t);
        M3<T> Nrt;
        m3_rmulAR<T, false, true>(Nrt, M1tR, c.CR);
        m3_axpy(Nrt, mdTw, M1w);
        M3<T> Nrv = M1v;
        m3_rmulAR(Nrv, M1tR, c.CA);
        m3_rmulAR(Nrv, m3_toggle(M1a), c.CB);
        // P_rr + dT (P_vr + M1v), P_vr = P_rv^T
        M3<T> Nrr = S.blk(R, R);
        m3_axpy<T, true>(Nrr, dT, m3_toggle(S.blk(R, V)));
        m3_axpy<T, true>(Nrr, dT, M1v);
        m3_symmetrise(Nrr);
        S.blk(R, R) = Nrr; S.blk(R, V) = Nrv; S.blk(R, TH) = Nrt; S.blk(R, AB) = M1a; S.blk(R, WB) = M1w;
    }
    // ---- level 2: rows v.  v' = v + A th + Bm ab ------------------------------------------------------------------------------
    {
        M3<T> Mvw = S.blk(V, WB), Mva = S.blk(V, AB);
        m3_lmulAA(Mvw, c.CA, S.blk(TH, WB));
        m3_lmulAA(Mvw, c.CB, S.blk(AB, WB));
        m3_lmulAA(Mva, c.CA, S.blk(TH, AB));
        m3_lmulAA(Mva, c.CB, S.blk(AB, AB));
        const M3<T> Ptv = m3_toggle(S.blk(V, TH));   // layout R of P_v,th == layout A of P_th,v
        const M3<T> Pav = m3_toggle(S.blk(V, AB));
        M3<T> MvtR = Ptv;                            // M_v,th in layout R
        m3_lmulRA(MvtR, c.CA, S.blk(TH, TH));
        m3_lmulRA(MvtR, c.CB, m3_toggle(S.blk(TH, AB)));
        M3<T> Nvt;
        m3_rmulAR<T, false, true>(Nvt, MvtR, c.CR);
        m3_axpy(Nvt, mdTw, Mvw);
        M3<T> Nvv = S.blk(V, V);
        m3_lmulAA<T, true>(Nvv, c.CA, Ptv);
        m3_lmulAA<T, true>(Nvv, c.CB, Pav);
        m3_rmulAR<T, true>(Nvv, MvtR, c.CA);
        m3_rmulAR<T, true>(Nvv, m3_toggle(Mva), c.CB);
        m3_axpy<T, true>(Nvv, T(1), c.CQC);
        m3_symmetrise(Nvv);
        S.blk(V, V) = Nvv; S.blk(V, TH) = Nvt; S.blk(V, AB) = Mva; S.blk(V, WB) = Mvw;
    }
    // ---- level 1: rows th.  th' = Rt th - dTw wb ------------------------------------------------------------------------------
    {
        M3<T> Ntw, Nta, MttR, Ntt;
        m3_lmulAA<T, false, true>(Ntw, c.CR, S.blk(TH, WB));
        m3_axpy(Ntw, mdTw, S.blk(WB, WB));
        m3_lmulAA<T, false, true>(Nta, c.CR, S.blk(TH, AB));
        m3_axpy(Nta, mdTw, m3_toggle(S.blk(AB, WB)));
        m3_lmulRA<T, true>(MttR, c.CR, S.blk(TH, TH));
        m3_axpy(MttR, mdTw, S.blk(TH, WB));          // layout R of P_wb,th == layout A of P_th,wb: the registers as they are
        m3_rmulAR<T, true, true>(Ntt, MttR, c.CR);
        m3_axpy<T, true>(Ntt, mdTw, Ntw);
        Ntt.a[0].x += nz.Q[3]; Ntt.a[1].y += nz.Q[4]; Ntt.d += nz.Q[5];
        m3_symmetrise(Ntt);
        S.blk(TH, TH) = Ntt; S.blk(TH, AB) = Nta; S.blk(TH, WB) = Ntw;
    }
    // ---- level 0: the bias blocks only gain their process noise --------------------------------------------------------------------
    {
        M3<T>& Paa = S.blk(AB, AB);
        M3<T>& Pww = S.blk(WB, WB);
        Paa.a[0].x += nz.Q[6]; Paa.a[1].y += nz.Q[7]; Paa.d += nz.Q[8];
        Pww.a[0].x += nz.Q[9]; Pww.a[1].y += nz.Q[10]; Pww.d += nz.Q[11];
    }
}

// prediction_step, EKF.cpp:346-415, on the blocks.  Device code instantiates it for fp32 only: fp64 has no packed FMA to gain and the full
// diagonal blocks cost 30 registers more than the packed triangle (the one fp64 instantiation that was built, round 3, spilled 1.5 KB per
// lane; its GPU memory fault was the backend's, see profiles/r04_tuning.md section 1) -- the fp64 replay runs ekf_split.hpp.  The host
// build (test suite) runs both dtypes.
template <typename T>
__device__ __forceinline__ void ekf_predict_packed(const DevParams<T>& p, const Noise<T>& nz, T (&x)[16], PackedCov<T>& S, const T (&u)[6], T (&accel)[3])
{
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(T) == 4, "the register-block predict is an fp32 device path");
#endif
    PackedCtx<T> c;
    packed_nominal<T>(p, nz, x, u, accel, c);
    packed_cov_predict<T>(c, nz, S);
}

}  // namespace qle
