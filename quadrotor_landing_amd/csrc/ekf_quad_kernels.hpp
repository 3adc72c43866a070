// ekf_quad_kernels.hpp -- the workgroup-cooperative tick kernel (arithmetic: ekf_quad.hpp), gfx950.
//
// Same HBM layout as the one-lane-per-filter kernels (wave tiles of 64 filters, ekf_kernels.hpp) and the same
// per-tick traffic.  A 256-thread workgroup is ONE tile and every thread has up to two roles:
//
//   quad role   (all four waves)  thread t is lane t%4 of the quad of filter t/4: lanes 0..2 load / store the ten
//                                 16-byte quads 3m + j of the packed covariance (column j of every 3x3 block, sidx in
//                                 ekf_device.hpp) and run the covariance algebra on them (quad::predict_P / update_P);
//   scalar role (waves 0, 1, 2)   thread t of wave w owns filter t % 64: wave 0 loads x, u, z (and the per-filter parameters), runs
//                                 quad::predict_scalar, later update_factor and update_inject, and stores x; on ticks with tag
//                                 poses waves 1 and 2 meanwhile compute the parts of the correction that need only the predicted
//                                 nominal state -- the innovation (wave 1) and Gx, R_k (wave 2) -- so that the three dependent
//                                 scalar chains run side by side instead of one after the other.
//
// The two roles talk through a per-filter LDS record (kLdsStride words): predict_scalar -> {A, Bm, Rt, C Qa C^T, Q diag}
// -> predict_P -> {P_rr, P_rt, P_tt} -> update_scalar -> {L, D^-1, D^-1 L^-1 dy, Gx} -> update_P -> {dx} -> update_inject,
// one workgroup barrier per arrow.  65 536 filters are 1 024 workgroups = four per CU = four waves per SIMD instead
// of the single 250-VGPR wave of the one-lane kernels: while one workgroup's wave 0 is in a scalar phase (its other
// three waves parked at the barrier) the other workgroups' waves keep the SIMDs and the memory system busy, and
// fp64 needs no scratch (45 covariance values per lane instead of 120).
#pragma once

#include "ekf_kernels.hpp"
#include "ekf_quad.hpp"

namespace qle {

// Per-filter LDS record, in words of the compute dtype.  The stride is odd: the 16 filters of a wave (quad role) and the
// 64 lanes of wave 0 (scalar role) then fall on distinct banks.
constexpr int kLdsU1 = 0;     // PredU (45 words); later UpdU (36 words)
constexpr int kLdsFlag = 45;  // bit 0: filter initialised and in range, bit 1: it corrects on this tick
constexpr int kLdsU2 = 46;    // P_rr, P_rt, P_tt (27 words); later dx (15 words)
constexpr int kLdsPark = 73;  // wave 0's predicted x (16) while the quads work
constexpr int kLdsPre = 89;   // from waves 1 and 2: dy (6), Gx (9), R_k upper triangle (21), reported observation (7)
constexpr int kLdsZ = 132;    // quarter-tile workgroups: the tag pose (7) of a filter that corrects, from the scalar role to the helper waves
constexpr int kLdsStride = 141;

// The quads' view of the scalar results: every value is fetched from the filter's LDS record where it is used.
template <typename T>
struct LdsPredQ {
    const T* rec;
    int j;
    __device__ __forceinline__ T A(int k) const { return rec[kLdsU1 + k]; }
    __device__ __forceinline__ T Bm(int k) const { return rec[kLdsU1 + 9 + k]; }
    __device__ __forceinline__ T Rt(int k) const { return rec[kLdsU1 + 18 + k]; }
    // row j of A, Bm, Rt; column j of C Qa C^T; the j-th diagonal noise terms (lane 3 reads neighbouring words it never uses)
    __device__ __forceinline__ T AR(int m) const { return rec[kLdsU1 + 3 * j + m]; }
    __device__ __forceinline__ T BR(int m) const { return rec[kLdsU1 + 9 + 3 * j + m]; }
    __device__ __forceinline__ T RtR(int m) const { return rec[kLdsU1 + 18 + 3 * j + m]; }
    __device__ __forceinline__ T CQc(int m) const { return rec[kLdsU1 + 27 + 3 * m + j]; }
    __device__ __forceinline__ T qw() const { return rec[kLdsU1 + 36 + j]; }
    __device__ __forceinline__ T qab() const { return rec[kLdsU1 + 39 + j]; }
    __device__ __forceinline__ T qwb() const { return rec[kLdsU1 + 42 + j]; }
};
template <typename T>
struct LdsFactorIn {
    const T* rec;
    __device__ __forceinline__ T Frr(int k) const { return rec[kLdsU2 + k]; }
    __device__ __forceinline__ T Frt(int k) const { return rec[kLdsU2 + 9 + k]; }
    __device__ __forceinline__ T Ftt(int k) const { return rec[kLdsU2 + 18 + k]; }
    __device__ __forceinline__ T dy(int k) const { return rec[kLdsPre + k]; }
    __device__ __forceinline__ T Gx(int k) const { return rec[kLdsPre + 6 + k]; }
    __device__ __forceinline__ T Rk(int k) const { return rec[kLdsPre + 15 + k]; }
};
template <typename T, bool DIRECT>
struct LdsUpdQ {
    const T* rec;
    __device__ __forceinline__ T Lm(int k) const { return rec[kLdsU1 + k]; }
    __device__ __forceinline__ T invd(int k) const { return rec[kLdsU1 + 15 + k]; }
    __device__ __forceinline__ T yd(int k) const { return rec[kLdsU1 + 21 + k]; }
    __device__ __forceinline__ T Gx(int k) const { return DIRECT ? T(0) : rec[kLdsU1 + 27 + k]; }
};

// Lane j < 3: the m-th quad of its covariance list is record words kXW + 4 (3m + j) .. +3.
template <typename T, int M0, int M1, int NTL>
__device__ __forceinline__ void quad_load_P(const T* __restrict__ tb, int f, int j, T (&L)[quad::kList])
{
    using QT = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
#pragma unroll
    for (int m = M1 - 1; m >= M0; --m) {
#pragma unroll
        for (int h = 0; h < 4 / VW; ++h) {
            const int row = (kXW + 4 * (3 * m + j)) / VW + h;
            const QT v = ld_quad<NTL>(reinterpret_cast<const QT*>(tb + (row * kTile + f) * VW));
            unpack_quad(v, &L[4 * m + h * VW]);
        }
    }
}
template <typename T, int M0, int M1, int NTS>
__device__ __forceinline__ void quad_store_P(T* __restrict__ tb, int f, int j, const T (&L)[quad::kList])
{
    using QT = typename Quad<T>::type;
    constexpr int VW = Quad<T>::VW;
#pragma unroll
    for (int m = M1 - 1; m >= M0; --m) {
#pragma unroll
        for (int h = 0; h < 4 / VW; ++h) {
            const int row = (kXW + 4 * (3 * m + j)) / VW + h;
            st_quad<NTS>(reinterpret_cast<QT*>(tb + (row * kTile + f) * VW), pack_quad(&L[4 * m + h * VW]));
        }
    }
}
// Register budget: at least WgWaves waves per SIMD (512 / WgWaves registers per lane).
// Measured (profiles/r02_tuning.md): fp32 fits 128 registers (four waves per SIMD) only for the predict-only instantiation;
// fp64 needs the whole file (256 VGPRs + AGPRs, one wave per SIMD) to stay out of scratch memory.
#ifndef QLE_WG_WAVES_F32
#define QLE_WG_WAVES_F32 2
#endif
#ifndef QLE_WG_WAVES_F64
#define QLE_WG_WAVES_F64 1
#endif
template <typename T> struct WgWaves { static constexpr int value = sizeof(T) == 4 ? QLE_WG_WAVES_F32 : QLE_WG_WAVES_F64; };

// One tick of tile `tile` by its workgroup: predict, and with STEP correct where the tag record's mask word is set (GATE:
// where filter_update's decision logic says so, EKF.cpp:147-186).  In place on `st`.  A filter whose stored quaternion is
// all zero has not been initialised (EKF.cpp:129-130) and is left untouched.
// FPW = filters per workgroup.  64: the workgroup is one whole tile, wave 0 carries the main scalar role next to its quads, waves 1 and 2
// the helper roles.  16: the workgroup is a QUARTER of a tile (filters f0 .. f0+15 of it) -- wave 0 holds the 16 quads, lanes 0..15 of
// waves 1, 2, 3 the main scalar role and the two helper roles -- so that a batch of a few thousand filters spreads over all 256 CUs
// instead of 64: the tick of such a batch is bound by what ONE CU can load and store (a tile's 60 KiB of fp64 covariance each way at
// ~11 bytes per cycle: 8 400 + 5 400 of the 25 000 cycles of a tick at 4 096 fp64 filters, profiles/r03_tuning.md), not by the chip.
template <typename T, bool DIRECT, bool PFP, bool GATE, bool STEP, int NT, int FPW>
__device__ __forceinline__ void wg_tick(const DevParams<T>& p, const GateParams& gp, T* st, const T* __restrict__ us, const T* __restrict__ zs,
                                        const T* __restrict__ pfp, T* __restrict__ aux_accel, T* __restrict__ aux_obs,
                                        int32_t* __restrict__ last_corr, uint8_t* __restrict__ flags, int64_t B, int64_t tile, int f0, T* lds)
{
    using DQ = quad::DevQ<T>;
    using SQ = quad::ScalarQ<T>;
    static_assert(FPW == 64 || FPW == 16, "a workgroup is a tile or a quarter of one");
    constexpr int NTL = NtLd<NT, kSW>::value, NTS = NtSt<NT>::value;
    constexpr int SM = FPW == 64 ? 0 : 1;            // the wave of the main scalar role
    const int t = (int)threadIdx.x, w = t >> 6, l = t & 63;
    const bool scalar_wave = w == SM && l < FPW;
    const int lfq = t >> 2, j = t & 3;               // quad role: local filter, lane of its quad
    const int fq = f0 + lfq;                         // that filter's position in the tile
    const bool quad_thread = t < 4 * FPW;
    const bool in_q = quad_thread && tile * kTile + fq < B;
    T* tb = st + tile * (int64_t)(kSW * kTile);

    // diagnostic build (make dbg): the first thread of the quad role / of the main scalar role stamps the phases
    // (profiles/r03_scripts/kw_timeline.py)
    const int dbg_id = (int)(tile * (kTile / FPW)) + f0 / FPW;
    (void)dbg_id;
#define QLE_KW_STAMP(k, dep) QLE_STAMPW(dbg_id, (((k) == 0 || (k) == 3 || (k) == 4 || (k) == 7 || (k) == 8) ? t == 0 : t == SM * 64), k, dep)
    QLE_KW_STAMP(0, (T)t);
    // ---- loads.  The records of the scalar roles (x, u, z: a few rows) are requested BEFORE the covariance: a wave's loads return in
    // order, so with the covariance first the scalar chain waited for all 60 KiB of the tile (5 200 of the tick's 25 600 cycles at
    // 4 096 fp64 filters, profiles/r03_tuning.md) instead of running while the covariance is in flight.
    const int64_t js = tile * kTile + f0 + l;                          // the filter of this thread's scalar role (lane l of its wave)
    const int64_t is = js;
    T* mine = lds + l * kLdsStride;   // scalar roles: the record of local filter l
    const bool helper_wave = STEP && (w == SM + 1 || w == SM + 2) && l < FPW;   // ticks with tag poses: the two waves after the main one
    const bool helper_first = w == SM + 1;
    T sx[kXW], su[kUW], szr[kZW], sfp[kFW];
    // (the helper waves of a quarter-tile workgroup load nothing: they take the predicted x and the tag pose from the LDS behind the
    // first barrier -- 30 fewer vector memory instructions per workgroup, each of which costs the CU's address unit 50-90 cycles
    // whatever the number of active lanes (profiles/micro/wave_load_split.hip), and no second and third predict of the nominal state)
    constexpr bool kHelpersLate = FPW == 16;
    if ((scalar_wave || (helper_wave && !kHelpersLate)) && js < B) {
        load_rec<T, kUW, 0, kUW, NT>(us, js, su);
        load_rec<T, kSW, 0, kXW, NT>(st, js, sx);
        if (STEP) load_rec<T, kZW, 0, kZW, NT>(zs, js, szr);
        if (PFP) load_rec<T, kFW, 0, kFW, NT>(pfp, js, sfp);
    } else if (PFP && helper_wave && js < B) {
        load_rec<T, kFW, 0, kFW, NT>(pfp, js, sfp);
    }
    T L[quad::kList];
    if (in_q && j < 3) quad_load_P<T, 0, 10, NTL>(tb, fq, j, L);
    else {
#pragma unroll
        for (int k = 0; k < quad::kList; ++k) L[k] = T(0);
    }

    // ---- helper roles (ticks with tag poses): what the correction needs of the predicted NOMINAL state only -- the innovation and the
    // reported observation (first helper wave), Gx and R_k (second) -- for the factor phase behind the SECOND barrier.
    // A whole-tile workgroup runs them side by side with the scalar role, in front of the first barrier, from their own copies of x, u, z
    // (its helper waves carry quads as well and would otherwise delay predict_P).  A quarter-tile workgroup runs them BEHIND the first
    // barrier, beside the quads' predict_P (its helper waves hold no quads), from what the scalar role left in the LDS.
    // cfg 2 (4 096 fp64 filters): 10.0-10.25 -> 9.75 us per tick (profiles/r04_tuning.md section 9).
    auto helper_work = [&]() {
        // the two waves behind the main scalar role's, lane = local filter
        // INVARIANT (no barrier orders these waves' load of x against wave 0's in-place store of the predicted x): wave 0 stores x
        // early only for a filter that does NOT correct on this tick (gate refused, mask clear), and everything these waves write for
        // such a filter -- dy, Gx, R_k, the reported observation in kLdsPre -- is read only where flag bit 1 (corrects) is set.  A torn
        // or already-predicted x can therefore only produce values nobody reads; any new consumer of kLdsPre must keep to bit 1.
        T* rec_s = mine;
        if (js < B) {
            T (&x)[kXW] = sx;
            T (&u)[kUW] = su;
            T (&zr)[kZW] = szr;
            quad::NoiseV<T> nz;
            if (PFP) {
                T (&fp)[kFW] = sfp;
#pragma unroll
                for (int k = 0; k < 3; ++k) { nz.ab_static[k] = fp[12 + k]; nz.wb_static[k] = fp[15 + k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) nz.R[k] = fp[18 + k];
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) { nz.ab_static[k] = p.ab_static[k]; nz.wb_static[k] = p.wb_static[k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) nz.R[k] = p.R[k];
            }
            bool act;
            if (kHelpersLate) {   // the scalar role's decision, its predicted x and the tag pose
                act = ((int)rec_s[kLdsFlag] & 2) != 0;
                if (act) {
#pragma unroll
                    for (int k = 0; k < kXW; ++k) x[k] = rec_s[kLdsPark + k];
#pragma unroll
                    for (int k = 0; k < 7; ++k) zr[k] = rec_s[kLdsZ + k];
                }
            } else {
                act = !filter_uninitialised(x) && zr[7] != T(0);
                if (act) quad::predict_nominal<SQ, T>(p, nz, x, u);
            }
            if (act) {
                if (helper_first) {
                    const T z[7] = {zr[0], zr[1], zr[2], zr[3], zr[4], zr[5], zr[6]};
                    T dy[6];
                    quad::update_innovation<SQ, T, DIRECT>(p, x, z, dy, [&](const T (&obs)[7]) {
#pragma unroll
                        for (int k = 0; k < 7; ++k) rec_s[kLdsPre + 36 + k] = obs[k];
                    });
#pragma unroll
                    for (int k = 0; k < 6; ++k) rec_s[kLdsPre + k] = dy[k];
                } else {
                    T Gx[9], Rk[quad::kRkWords];
                    quad::update_noise<SQ, T, DIRECT>(p, nz, x, Gx, Rk);
#pragma unroll
                    for (int k = 0; k < 9; ++k) rec_s[kLdsPre + 6 + k] = Gx[k];
#pragma unroll
                    for (int k = 0; k < quad::kRkWords; ++k) rec_s[kLdsPre + 15 + k] = Rk[k];
                }
            }
        }
    };

    // ---- scalar role: nominal state, blocks of F, the decision whether this filter corrects
    if (scalar_wave) {
        T (&x)[kXW] = sx;
        T (&zr)[kZW] = szr;
        quad::NoiseV<T> nz;
        bool live = false, corr = false;
        if (is < B) {
            T (&u)[kUW] = su;
            if (PFP) {
                T (&fp)[kFW] = sfp;
#pragma unroll
                for (int k = 0; k < 12; ++k) nz.Q[k] = fp[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) { nz.ab_static[k] = fp[12 + k]; nz.wb_static[k] = fp[15 + k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) nz.R[k] = fp[18 + k];
            } else {
#pragma unroll
                for (int k = 0; k < 12; ++k) nz.Q[k] = p.Q[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) { nz.ab_static[k] = p.ab_static[k]; nz.wb_static[k] = p.wb_static[k]; }
#pragma unroll
                for (int k = 0; k < 6; ++k) nz.R[k] = p.R[k];
            }
            live = !filter_uninitialised(x);
            QLE_KW_STAMP(1, x[9] + u[5]);
            if (live) {
                if (STEP) {
                    corr = zr[7] != T(0);
                    if (GATE) {
                        const bool consume = corr && (!gp.limit || (gp.tick - last_corr[is]) >= gp.upd_per_meas);
                        bool ok = consume;
                        if (consume && gp.corner_enbl) {
                            const double zd[7] = {(double)zr[0], (double)zr[1], (double)zr[2], (double)zr[3], (double)zr[4], (double)zr[5], (double)zr[6]};
                            ok = corner_gate(gp, zd);
                        }
                        corr = ok;
                        if (ok) last_corr[is] = gp.tick;
                        flags[is] = (uint8_t)((ok ? 1 : 0) | (consume ? 2 : 0));
                    }
                }
                quad::PredU<T> pu;
                T accel[3];
                quad::predict_scalar<SQ, T>(p, nz, x, u, accel, pu);
                QLE_KW_STAMP(2, pu.Rt[8] + pu.A[0] + x[9]);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    mine[kLdsU1 + k] = pu.A[k]; mine[kLdsU1 + 9 + k] = pu.Bm[k]; mine[kLdsU1 + 18 + k] = pu.Rt[k];
                    mine[kLdsU1 + 27 + k] = pu.CQ[k]; mine[kLdsU1 + 36 + k] = pu.Qd[k];
                }
                if (aux_accel) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) aux_accel[is * 3 + k] = accel[k];
                }
                // (see the INVARIANT at the helper waves below: this early store is only for filters that do not correct)
                if (!(STEP && corr)) store_rec<T, kSW, 0, kXW, NT>(st, is, x);
                else {   // the predicted nominal state waits in LDS while the quads work (keeps the quad phases' register count down)
#pragma unroll
                    for (int k = 0; k < kXW; ++k) mine[kLdsPark + k] = x[k];
                    if (kHelpersLate) {
#pragma unroll
                        for (int k = 0; k < 7; ++k) mine[kLdsZ + k] = zr[k];
                    }
                }
            }
        }
        mine[kLdsFlag] = T((live ? 1 : 0) | (corr ? 2 : 0));
        // "does any filter of this workgroup correct": decided here, once, by the wave that knows (read by everyone after the first
        // barrier; a vote at the second barrier instead -- __syncthreads_or -- is two barriers)
        const bool any_corr = __ballot(corr) != 0;   // evaluated by every lane of the scalar role, written by one
        if (STEP && l == 0) lds[FPW * kLdsStride] = T(any_corr ? 1 : 0);
    } else if (helper_wave && !kHelpersLate) {
        helper_work();
    }
    __syncthreads();
    if (kHelpersLate && helper_wave && lds[FPW * kLdsStride] != T(0)) helper_work();   // (the word: "somebody in this workgroup corrects")

    // ---- quad role: P <- F P F^T + Q; the words of a block-row are stored as soon as they are final unless a correction follows
    T* rec = lds + (quad_thread ? lfq : 0) * kLdsStride;
    const int fl = (int)rec[kLdsFlag];
    QLE_KW_STAMP(3, (T)fl + L[0] + L[39]);
    const bool live_q = in_q && (fl & 1) != 0, corr_q = STEP && (fl & 2) != 0;
    T Ln[quad::kList], Prr[3], Ptt[3];
    if (live_q) {
        const LdsPredQ<T> g{rec, j};
        quad::predict_P<DQ, T>(p, g, L, Ln, Prr, Ptt, [&](int level) {
            if (corr_q || j == 3) return;
            if (level == 0) quad_store_P<T, 9, 10, NTS>(tb, fq, j, Ln);
            else if (level == 1) quad_store_P<T, 7, 9, NTS>(tb, fq, j, Ln);
            else if (level == 2) quad_store_P<T, 4, 7, NTS>(tb, fq, j, Ln);
            else quad_store_P<T, 0, 4, NTS>(tb, fq, j, Ln);
        });
    }
    QLE_KW_STAMP(4, Ln[0] + Ln[39]);
    if (!STEP) return;
    if (live_q && corr_q && j < 3) {   // the blocks of the predicted P over {r, th}: column j of each
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            rec[kLdsU2 + 3 * i + j] = Prr[i];
            rec[kLdsU2 + 9 + 3 * i + j] = Ln[QLE_QO(0, 2, i)];
            rec[kLdsU2 + 18 + 3 * i + j] = Ptt[i];
        }
    }
    if (lds[FPW * kLdsStride] == T(0)) return;   // nobody in this workgroup corrects (wave-uniform, written before the first barrier)
    __syncthreads();

    // ---- scalar role: innovation, S = G P G^T + R_k = L D L^T
    const bool corr_s = scalar_wave && ((int)mine[kLdsFlag] & 2) != 0;
    QLE_KW_STAMP(5, mine[kLdsFlag]);
    if (corr_s) {
        quad::UpdU<T> uu;
        quad::update_factor<SQ, DIRECT>(LdsFactorIn<T>{mine}, uu);
        QLE_KW_STAMP(6, uu.invd[5] + uu.yd[5]);
        if (aux_accel) {
#pragma unroll
            for (int k = 0; k < 7; ++k) aux_obs[is * 7 + k] = mine[kLdsPre + 36 + k];
        }
#pragma unroll
        for (int k = 0; k < 15; ++k) mine[kLdsU1 + k] = uu.Lm[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { mine[kLdsU1 + 15 + k] = uu.invd[k]; mine[kLdsU1 + 21 + k] = uu.yd[k]; }
        if (!DIRECT) {
#pragma unroll
            for (int k = 0; k < 9; ++k) mine[kLdsU1 + 27 + k] = uu.Gx[k];
        }
    }
    __syncthreads();

    // ---- quad role: P <- P - V D^-1 V^T, rows of dx
    QLE_KW_STAMP(7, rec[kLdsU1]);
    if (live_q && corr_q) {
        const LdsUpdQ<T, DIRECT> g{rec};
        T dxo[5];
        quad::update_P<DQ, DIRECT>(g, Ln, Prr, Ptt, dxo);
        QLE_KW_STAMP(8, dxo[0] + dxo[4] + Ln[0] + Ln[39]);
        if (j < 3) {
#pragma unroll
            for (int b = 0; b < 5; ++b) rec[kLdsU2 + 3 * b + j] = dxo[b];
            quad_store_P<T, 0, 10, NTS>(tb, fq, j, Ln);
        }
    }
    // The injection needs dx from LDS and nothing from memory: a barrier that waits for the LDS writes only.  __syncthreads() also
    // waits for every outstanding global access (s_waitcnt vmcnt(0)), i.e. for the drain of the 60 KiB of covariance stores just
    // issued: 5 500 of the tick's 25 600 cycles at 4 096 fp64 filters before the last scalar phase could start.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- scalar role: inject the error state (EKF.cpp:486-501)
    QLE_KW_STAMP(9, mine[kLdsU2]);
    if (corr_s) {
        T x[kXW], dx[15];
#pragma unroll
        for (int k = 0; k < kXW; ++k) x[k] = mine[kLdsPark + k];
#pragma unroll
        for (int k = 0; k < 15; ++k) dx[k] = mine[kLdsU2 + k];
        quad::update_inject<SQ, T>(p, x, dx);
        QLE_KW_STAMP(10, x[9] + x[0]);
        store_rec<T, kSW, 0, kXW, NT>(st, is, x);
    }
    QLE_KW_STAMP(11, (T)t);
#undef QLE_KW_STAMP
}

// Grid: one 256-thread workgroup per FPW filters (a tile, or a quarter of one).  NT as in k_predict / k_step (3 = cached / streamed split
// per workgroup).
template <typename T, bool DIRECT, bool PFP, bool GATE, bool STEP, int NT, int FPW>
__global__ __launch_bounds__(kBlock, WgWaves<T>::value) void kw_tick(T* st, const T* __restrict__ us, const T* __restrict__ zs, int64_t B, int32_t grid_x, int32_t split,   // (argument order: see k_predict)
                                                                      const T* __restrict__ pfp, T* __restrict__ aux_accel, T* __restrict__ aux_obs,
                                                                      int32_t* __restrict__ last_corr, uint8_t* __restrict__ flags, DevParams<T> p, GateParams gp)
{
    __shared__ T lds[FPW * kLdsStride + 2];   // the per-filter records + the workgroup's "somebody corrects" word
    QLE_ARGS_EARLY(st, us, zs, B, grid_x);
    constexpr int PER = kTile / FPW;
    const int64_t wg = batch_block((unsigned)grid_x);
    const int64_t tile = wg / PER;
    const int f0 = (int)(wg % PER) * FPW;
    if (NT == 3) {
        if (cached_workgroup(split)) wg_tick<T, DIRECT, PFP, GATE, STEP, 0, FPW>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, B, tile, f0, lds);
        else wg_tick<T, DIRECT, PFP, GATE, STEP, 2, FPW>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, B, tile, f0, lds);
    } else {
        wg_tick<T, DIRECT, PFP, GATE, STEP, NT, FPW>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, B, tile, f0, lds);
    }
}

}  // namespace qle
