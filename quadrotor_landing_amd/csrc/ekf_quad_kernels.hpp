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
constexpr int kLdsStride = 133;

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
template <typename T, bool DIRECT, bool PFP, bool GATE, bool STEP, int NT>
__device__ __forceinline__ void wg_tick(const DevParams<T>& p, const GateParams& gp, T* st, const T* __restrict__ us, const T* __restrict__ zs,
                                        const T* __restrict__ pfp, T* __restrict__ aux_accel, T* __restrict__ aux_obs,
                                        int32_t* __restrict__ last_corr, uint8_t* __restrict__ flags, int64_t B, int64_t tile, T* lds)
{
    using DQ = quad::DevQ<T>;
    using SQ = quad::ScalarQ<T>;
    constexpr int NTL = NtLd<NT, kSW>::value, NTS = NtSt<NT>::value;
    const int t = (int)threadIdx.x;
    const bool scalar_wave = t < kTile;
    const int fq = t >> 2, j = t & 3;
    const bool in_q = tile * kTile + fq < B;
    T* tb = st + tile * (int64_t)(kSW * kTile);

    // ---- covariance loads (quad role), issued first: the scalar phase below runs while they are in flight
    T L[quad::kList];
    if (in_q && j < 3) quad_load_P<T, 0, 10, NTL>(tb, fq, j, L);
    else {
#pragma unroll
        for (int k = 0; k < quad::kList; ++k) L[k] = T(0);
    }

    // ---- scalar role: nominal state, blocks of F, the decision whether this filter corrects
    const int64_t is = tile * kTile + t;
    T* mine = lds + t * kLdsStride;   // wave 0: the record of filter t
    if (scalar_wave) {
        T x[kXW], zr[kZW];
        quad::NoiseV<T> nz;
        bool live = false, corr = false;
        if (is < B) {
            T u[kUW];
            load_rec<T, kUW, 0, kUW, NT>(us, is, u);
            load_rec<T, kSW, 0, kXW, NT>(st, is, x);
            if (STEP) load_rec<T, kZW, 0, kZW, NT>(zs, is, zr);
            if (PFP) {
                T fp[kFW];
                load_rec<T, kFW, 0, kFW, NT>(pfp, is, fp);
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
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    mine[kLdsU1 + k] = pu.A[k]; mine[kLdsU1 + 9 + k] = pu.Bm[k]; mine[kLdsU1 + 18 + k] = pu.Rt[k];
                    mine[kLdsU1 + 27 + k] = pu.CQ[k]; mine[kLdsU1 + 36 + k] = pu.Qd[k];
                }
                if (aux_accel) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) aux_accel[is * 3 + k] = accel[k];
                }
                if (!(STEP && corr)) store_rec<T, kSW, 0, kXW, NT>(st, is, x);
                else {   // the predicted nominal state waits in LDS while the quads work (keeps the quad phases' register count down)
#pragma unroll
                    for (int k = 0; k < kXW; ++k) mine[kLdsPark + k] = x[k];
                }
            }
        }
        mine[kLdsFlag] = T((live ? 1 : 0) | (corr ? 2 : 0));
    } else if (STEP && t < 3 * kTile) {
        // waves 1 and 2, filter t % 64: what the correction needs of the predicted NOMINAL state only, side by side with wave 0
        const int fs = t & (kTile - 1);
        const int64_t js = tile * kTile + fs;
        T* rec_s = lds + fs * kLdsStride;
        if (js < B) {
            T x[kXW], u[kUW], zr[kZW];
            load_rec<T, kUW, 0, kUW, NT>(us, js, u);
            load_rec<T, kSW, 0, kXW, NT>(st, js, x);
            load_rec<T, kZW, 0, kZW, NT>(zs, js, zr);
            quad::NoiseV<T> nz;
            if (PFP) {
                T fp[kFW];
                load_rec<T, kFW, 0, kFW, NT>(pfp, js, fp);
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
            if (!filter_uninitialised(x) && zr[7] != T(0)) {
                quad::predict_nominal<SQ, T>(p, nz, x, u);
                if (t < 2 * kTile) {
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
    }
    __syncthreads();

    // ---- quad role: P <- F P F^T + Q; the words of a block-row are stored as soon as they are final unless a correction follows
    T* rec = lds + fq * kLdsStride;
    const int fl = (int)rec[kLdsFlag];
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
    if (!STEP) return;
    if (live_q && corr_q && j < 3) {   // the blocks of the predicted P over {r, th}: column j of each
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            rec[kLdsU2 + 3 * i + j] = Prr[i];
            rec[kLdsU2 + 9 + 3 * i + j] = Ln[QLE_QO(0, 2, i)];
            rec[kLdsU2 + 18 + 3 * i + j] = Ptt[i];
        }
    }
    if (!__syncthreads_or(corr_q ? 1 : 0)) return;   // nobody in this tile corrects

    // ---- scalar role: innovation, S = G P G^T + R_k = L D L^T
    const bool corr_s = scalar_wave && ((int)mine[kLdsFlag] & 2) != 0;
    if (corr_s) {
        quad::UpdU<T> uu;
        quad::update_factor<SQ, DIRECT>(LdsFactorIn<T>{mine}, uu);
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
    if (live_q && corr_q) {
        const LdsUpdQ<T, DIRECT> g{rec};
        T dxo[5];
        quad::update_P<DQ, DIRECT>(g, Ln, Prr, Ptt, dxo);
        if (j < 3) {
#pragma unroll
            for (int b = 0; b < 5; ++b) rec[kLdsU2 + 3 * b + j] = dxo[b];
            quad_store_P<T, 0, 10, NTS>(tb, fq, j, Ln);
        }
    }
    __syncthreads();

    // ---- scalar role: inject the error state (EKF.cpp:486-501)
    if (corr_s) {
        T x[kXW], dx[15];
#pragma unroll
        for (int k = 0; k < kXW; ++k) x[k] = mine[kLdsPark + k];
#pragma unroll
        for (int k = 0; k < 15; ++k) dx[k] = mine[kLdsU2 + k];
        quad::update_inject<SQ, T>(p, x, dx);
        store_rec<T, kSW, 0, kXW, NT>(st, is, x);
    }
}

// Grid: one 256-thread workgroup per tile.  NT as in k_predict / k_step (3 = cached / streamed split per workgroup).
template <typename T, bool DIRECT, bool PFP, bool GATE, bool STEP, int NT>
__global__ __launch_bounds__(kBlock, WgWaves<T>::value) void kw_tick(DevParams<T> p, GateParams gp, T* st, const T* __restrict__ us, const T* __restrict__ zs,
                                                                      const T* __restrict__ pfp, T* __restrict__ aux_accel, T* __restrict__ aux_obs,
                                                                      int32_t* __restrict__ last_corr, uint8_t* __restrict__ flags, int64_t B, int32_t split)
{
    __shared__ T lds[kTile * kLdsStride];
    const int64_t tile = batch_block();
    if (NT == 3) {
        if (cached_workgroup(split)) wg_tick<T, DIRECT, PFP, GATE, STEP, 0>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, B, tile, lds);
        else wg_tick<T, DIRECT, PFP, GATE, STEP, 2>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, B, tile, lds);
    } else {
        wg_tick<T, DIRECT, PFP, GATE, STEP, NT>(p, gp, st, us, zs, pfp, aux_accel, aux_obs, last_corr, flags, B, tile, lds);
    }
}

}  // namespace qle
