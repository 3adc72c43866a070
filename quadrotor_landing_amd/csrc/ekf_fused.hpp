// ekf_fused.hpp -- one single-rate filter_update tick (predict + correction, EKF.cpp:238-249, 265-290) for one lane as ONE
// straight-line schedule.  gfx950 (CDNA4).  Written so that a host compiler accepts it too: the test suite compiles the arithmetic
// headers into a CPU checker of its own; the product never does.
//
// Why: a lone wave per SIMD (65 536 filters) pays ~9 cycles for every DEPENDENT instruction and 2.5-4 for an independent one
// (profiles/r02_tuning.md section 3).  The tick has four long dependent scalar chains -- the nominal-state propagation (quaternion
// exponential), the innovation (quaternion logarithm), R_k with the L D L^T factor of S = G P G^T + R_k (six dependent divisions)
// and the injection -- and ~1 900 independent covariance FMAs.  Everything up to the factor is written unconditionally, in an order
// that puts independent covariance work next to each chain, and the downdate streams its stores.  Measured (section 7 there): the
// streamed stores and the lower register pressure are what pays (fp32 15.4 -> 13.3 us at 65 536 filters, fp64 38.3 -> 28.0 us and no
// scratch); the placement of the chains does not -- the compiler sinks the factor back under the branch and a lone wave is bound by
// the number of instructions it issues, not by their dependencies.
//
//   nominal state            | (loads of P in flight)
//   R_k, Gx                  | need the predicted nominal state only
//   level 3 (rows r), level 1 (rows th) of the levelled predict: the blocks S needs -- P(r,r), P(r,th), P(th,th)
//   S = L D L^T              | next to level 2 (rows v, the largest level) and level 0
//   lanes that correct:  P <- P - V D^-1 V^T bottom-up in memory order, V = (P G^T) L^-T formed row-block by row-block when the
//                        downdate first needs it, every 16-byte quad stored as soon as it is final; then -- under the drain of those
//                        stores -- the innovation dy, its elimination yd = D^-1 L^-1 dy, dx = V yd and the injection
//   lanes that do not:   store the predicted state
//
// Forming V lazily works because the entries W = P G^T is read from -- P(r, *), P(th, *), P(v, th) -- all sit in block-rows at
// or above the one whose rows of V are being formed, and the downdate reaches those words later (descending memory order is
// block-row 4 first).  At most 120 + 36 covariance-sized values are live at the start of the downdate and P shrinks as V grows,
// instead of P + all of V (210) in the ascending form.
//
// Values: the same expressions as ekf_predict_levels + quad::update_* (the cooperative kernel's scalar parts), evaluated in another
// order.
//
// Registers (csrc/resources.py): the DIRECT orientation method (EKF.cpp:441; both shipped configurations and every BASELINE config)
// needs no scratch in either dtype.  The CONVENTIONAL method (direct_orien_method = false, EKF.cpp:443-444, the constructor's default)
// carries Gx through the sweep and does spill: fp32 0-40 B per lane, fp64 140-236 B per lane (256 VGPR + 256 AGPR).  Measured at
// 65 536 filters with every filter correcting (profiles/r03_kernel_times.jsonl, HIP-event period of back-to-back launches,
// QLE_TIME_DIRECT=0): fp32 12.75 us, the same as the direct method; fp64 32.8 us against 26.8 us.  Compact records (est_bias = false)
// have none in either method.
#pragma once

#include <cmath>
#include <type_traits>

#include "ekf_device.hpp"
#include "ekf_quad.hpp"

namespace qle {

// f(std::integral_constant<int, I>) for I = B .. E-1, unrolled by the template machinery rather than by the loop unroller.
template <int B, int E, typename F>
__host__ __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Last word (in memory order) that holds an element of block-row b: the first one a descending sweep meets.
__host__ __device__ constexpr int block_row_last_word(int b)
{
    int r = -1;
    for (int w = 0; w < 120; ++w)
        if (word_block_row(w) == b) r = w;
    return r;
}

// Gx = Cc [Cc^T r]x of the conventional orientation method (EKF.cpp:455-458).  Every product-sum is an EXPLICIT fma chain: this function
// is evaluated several times per tick and in several instantiations (full and compact records) whose results must agree bit for bit, and
// `a*b - c*d` left to the backend contracts as fma(a, b, -(c*d)) or as fma(-c, d, a*b) depending on what surrounds it.
template <typename T>
__host__ __device__ __forceinline__ void conventional_gx(const T (&x)[16], T (&Gx)[9])
{
    const T qx = x[6], qy = x[7], qz = x[8], qw = x[9];
    const T tx = qx + qx, ty = qy + qy, tz = qz + qz;
    const T twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, tyy = ty * qy, tzz = tz * qz;
    // Eigen toRotationMatrix (EKF.cpp:429), off-diagonals as one fma each
    T Cc[9];
    Cc[0] = T(1) - (tyy + tzz);      Cc[1] = fused_fma(ty, qx, -twz); Cc[2] = fused_fma(tz, qx, twy);
    Cc[3] = fused_fma(ty, qx, twz);  Cc[4] = T(1) - (txx + tzz);      Cc[5] = fused_fma(tz, qy, -twx);
    Cc[6] = fused_fma(tz, qx, -twy); Cc[7] = fused_fma(tz, qy, twx);  Cc[8] = T(1) - (txx + tyy);
    T b[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = fused_fma(Cc[6 + i], x[2], fused_fma(Cc[3 + i], x[1], Cc[i] * x[0]));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const T c0 = Cc[3 * i], c1 = Cc[3 * i + 1], c2 = Cc[3 * i + 2];
        Gx[3 * i] = fused_fma(c1, b[2], -(c2 * b[1]));
        Gx[3 * i + 1] = fused_fma(c2, b[0], -(c0 * b[2]));
        Gx[3 * i + 2] = fused_fma(c0, b[1], -(c1 * b[0]));
    }
}

// x: state at tick n-1 in, state at tick n out.  Po: covariance at tick n-1.  `corr`: this lane fuses the tag pose z; `live`: the
// filter is initialised (a lane that is not stores nothing: corr implies live).
// emit_accel(accel) / emit_obs(obs): side outputs as soon as they exist.  store_quad(q4 as a std::integral_constant, ptr to 4 final words): 4-word group q4 of
// the new covariance is final (called for q4 = 29 .. 0 on every lane); store_x(): x is final.
// tag_pose(zz): the tag pose z of this lane, asked for AFTER the covariance sweep, where the innovation needs it -- a caller may read it
// from memory again there (fp64: 14 registers that do not have to stay live through the sweep) or hand over what it loaded at the start.
// park(x) / unpark(x): called right before and right after the covariance sweep of a correcting lane; parked = whether park moved
// the nominal state out of the registers (fp64: into the LDS) for the duration of the sweep.  The sweep itself needs r and q only, and
// only in the conventional method (Gx per block-row): unpark(xl) before each of those evaluations then reads them back into a copy.
struct FusedNoPark {
    static constexpr bool parked = false;
    template <typename X> __host__ __device__ __forceinline__ void operator()(X&) const {}
};
template <typename T, bool DIRECT, typename TagPose, typename EmitAccel, typename EmitObs, typename StoreX, typename StoreQuad,
          typename Park = FusedNoPark, typename Unpark = FusedNoPark>
__device__ __forceinline__ void ekf_step_fused_z(const DevParams<T>& p, const Noise<T>& nzl, T (&x)[16], const T (&Po)[120], const T (&u)[6],
                                                 TagPose&& tag_pose, bool corr, bool live, EmitAccel&& emit_accel, EmitObs&& emit_obs,
                                                 StoreX&& store_x, StoreQuad&& store_quad, Park&& park = Park(), Unpark&& unpark = Unpark())
{
    using SQ = quad::ScalarQ<T>;
    constexpr int kPW_ = 120;
    T Pn[kPW_];
    PredictCtx<T> c;
    {
        T accel[3];
        predict_nominal_lean<T>(p, nzl, x, u, accel, c);
        emit_accel(accel);
    }
    // correction, the parts that need only the predicted nominal state
    quad::NoiseV<T> nz;
#pragma unroll
    for (int k = 0; k < 6; ++k) nz.R[k] = nzl.R[k];
    quad::FactorIn<T> in;
#pragma unroll
    for (int k = 0; k < 6; ++k) in.dy_[k] = T(0);   // the factor below is wanted for L and D only; dy is eliminated after the sweep
    quad::update_noise<SQ, T, DIRECT>(p, nz, x, in.gx, in.rk);
    // the rows S is read from
    predict_level3<T>(c, Po, Pn);
    predict_level1<T>(c, nzl, Po, Pn);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            in.frr[3 * i + k] = Pn[sidx(i, k)]; in.frt[3 * i + k] = Pn[sidx(i, 6 + k)]; in.ftt[3 * i + k] = Pn[sidx(6 + i, 6 + k)];
        }
    }
    quad::UpdU<T> f;
    quad::update_factor<SQ, DIRECT>(in, f);
    predict_level2<T>(c, nzl, Po, Pn);
    predict_level0<T>(nzl, Po, Pn);

    if (corr) {
        T V[15][6], NV[15][6];
        park(x);
        static_for<0, kPW_>([&](auto wc) {   // a compile-time loop: every index below must be a constant (no array may reach scratch)
            constexpr int w = kPW_ - 1 - decltype(wc)::value;
            constexpr int i = word_row(w), k = word_col(w), b = i / 3;
            if constexpr (w == block_row_last_word(b)) {   // rows 3b .. 3b+2 of V = (P G^T) L^-T and of -V D^-1, from words not yet touched
                // conventional method: Gx = Cc [Cc^T r]x (EKF.cpp:455-458) is formed again from the predicted nominal state for every
                // block-row (five times ~30 operations) instead of nine values staying live through the whole sweep -- in fp64 they were
                // most of what the kernel spilled (188 B per lane)
                T gx[9];
                if constexpr (!DIRECT && sizeof(T) == 4) {   // fp32 has the registers: Gx as the factor step left it
#pragma unroll
                    for (int kk = 0; kk < 9; ++kk) gx[kk] = f.Gx[kk];
                }
                if constexpr (!DIRECT && sizeof(T) == 8) {
                    T xl[16];
                    if constexpr (std::remove_reference_t<Unpark>::parked) unpark(xl);
                    else {
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) xl[kk] = x[kk];
                    }
#if defined(__HIP_DEVICE_COMPILE__)
                    // opaque copies of the seven words Gx depends on, tied to a covariance word of THIS block-row: the backend would otherwise
                    // fold the five evaluations back into one and keep its result alive
                    asm volatile("" : "+v"(xl[0]), "+v"(xl[1]), "+v"(xl[2]), "+v"(xl[6]), "+v"(xl[7]), "+v"(xl[8]), "+v"(xl[9]) : "v"(Pn[w]));
#endif
                    conventional_gx<T>(xl, gx);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int a = 3 * b + r;
                    T (&v)[6] = V[a];
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) { v[kk] = Pn[sidx(a, kk)]; v[3 + kk] = Pn[sidx(a, 6 + kk)]; }
                    if (!DIRECT) {
#pragma unroll
                        for (int kk = 0; kk < 3; ++kk)   // pinned evaluation order (see conventional_gx)
                            v[kk] += fused_fma(v[5], gx[3 * kk + 2], fused_fma(v[4], gx[3 * kk + 1], v[3] * gx[3 * kk]));
                    }
#pragma unroll
                    for (int m = 1; m < 6; ++m) {
#pragma unroll
                        for (int m2 = 0; m2 < m; ++m2) v[m] -= f.Lm[quad::lm_idx(m, m2)] * v[m2];
                    }
#pragma unroll
                    for (int m = 0; m < 6; ++m) NV[a][m] = v[m] * (-f.invd[m]);
                }
            }
            // P(i,k) += sum_m (-V(i,m)/d_m) V(k,m)
            T acc = Pn[w];
#pragma unroll
            for (int m = 0; m < 6; ++m) acc += NV[i][m] * V[k][m];
            Pn[w] = acc;
            if constexpr (w % 4 == 0) store_quad(std::integral_constant<int, w / 4>{}, &Pn[w]);
        });
        unpark(x);
        // Nothing the downdate needs depends on the innovation: dy, its elimination yd = D^-1 L^-1 dy, dx and the injection come AFTER the
        // sweep, where they run under the drain of the covariance stores.  The tag pose is made to depend on the last covariance word so
        // that the scheduler cannot pull the chain (quaternion logarithm: ~150 dependent instructions) in front of the first store.
        {
            T zz[7];
            tag_pose(zz);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(zz[6]) : "v"(Pn[0]));
#endif
            T xp[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) xp[k] = x[k];
            T dy[6];
            quad::update_innovation<SQ, T, DIRECT>(p, xp, zz, dy, emit_obs);
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) {
#pragma unroll
                for (int j = cc + 1; j < 6; ++j) dy[j] = dy[j] - f.Lm[quad::lm_idx(j, cc)] * dy[cc];
            }
#pragma unroll
            for (int m = 0; m < 6; ++m) f.yd[m] = dy[m] * f.invd[m];
        }
        // inject, EKF.cpp:486-501
        T dx[15];
#pragma unroll
        for (int a = 0; a < 15; ++a) {
            T acc = V[a][0] * f.yd[0];
#pragma unroll
            for (int m = 1; m < 6; ++m) acc += V[a][m] * f.yd[m];
            dx[a] = acc;
        }
        quad::update_inject<SQ, T>(p, x, dx);
        store_x();
    } else if (live) {
        store_x();
        static_for<0, kPW_ / 4>([&](auto qc) {
            constexpr int q4 = kPW_ / 4 - 1 - decltype(qc)::value;
            store_quad(std::integral_constant<int, q4>{}, &Pn[4 * q4]);
        });
    }
}

template <typename T, bool DIRECT, typename EmitAccel, typename EmitObs, typename StoreX, typename StoreQuad>
__device__ __forceinline__ void ekf_step_fused(const DevParams<T>& p, const Noise<T>& nzl, T (&x)[16], const T (&Po)[120], const T (&u)[6],
                                               const T (&z)[7], bool corr, bool live, EmitAccel&& emit_accel, EmitObs&& emit_obs,
                                               StoreX&& store_x, StoreQuad&& store_quad)
{
    ekf_step_fused_z<T, DIRECT>(p, nzl, x, Po, u, [&](T (&zz)[7]) {
#pragma unroll
        for (int k = 0; k < 7; ++k) zz[k] = z[k];
    }, corr, live, emit_accel, emit_obs, store_x, store_quad);
}

}  // namespace qle
