// ekf_structured_cpu.cpp -- CPU build of the engine's per-filter arithmetic (TEST / BASELINE ONLY).
//
// quadrotor_landing_amd/csrc/ekf_device.hpp (block-structured predict, decorrelated sequential update)
// is compiled here, unmodified, by g++ -- the HIP headers define __device__/__forceinline__ away for a
// host compiler.  Purpose:
//   1. the second, stronger CPU baseline of bench.py ("structure-exploiting CPU variant", BASELINE.md
//      section 3) next to the dense reference-shaped oracle;
//   2. a no-GPU check of the engine's algebra against the dense oracle (tests/test_oracle.py).
// It is part of libekf_oracle_structured.so only; the product library never contains or calls it
// (the product has no CPU path: qle_create fails without a HIP device).
#include <cstdint>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <cmath>

#include "../quadrotor_landing_amd/csrc/ekf_device.hpp"
#include "../quadrotor_landing_amd/csrc/ekf_quad.hpp"
#include "../quadrotor_landing_amd/csrc/ekf_fused.hpp"
#include "../quadrotor_landing_amd/csrc/ekf_packed.hpp"
#include "../quadrotor_landing_amd/csrc/ekf_split.hpp"
#include "ekf_oracle.h"

using namespace qle;

template <typename T>
static DevParams<T> to_dev(const orc_params* p)
{
    DevParams<T> o;
    o.dT = (T)p->dT_nom;
    o.dTw = p->est_bias ? (T)p->dT_nom : T(0);
    o.bias_on = p->est_bias ? T(1) : T(0);
    o.small_ang_tol = (T)p->small_ang_tol;
    for (int i = 0; i < 3; ++i) { o.g[i] = (T)p->g[i]; o.r_v_cv[i] = (T)p->r_v_cv[i]; o.ab_static[i] = (T)p->ab_static[i]; o.wb_static[i] = (T)p->wb_static[i]; }
    for (int i = 0; i < 4; ++i) o.q_vc[i] = (T)p->q_vc[i];
    for (int i = 0; i < 9; ++i) o.C_vc[i] = (T)p->C_vc[i];
    for (int i = 0; i < 12; ++i) o.Q[i] = (T)p->Q[i];
    for (int i = 0; i < 6; ++i) o.R[i] = (T)p->R[i];
    return o;
}

template <typename T>
static int64_t run_batch_t(const orc_params* p, int64_t B, int64_t Tn, double* x, double* P, const double* u, const double* z,
                           const uint8_t* mask, int levels, int n_threads)
{
    const int n = p->num_states;
    const DevParams<T> dp = to_dev<T>(p);
    Noise<T> nz;
    for (int k = 0; k < 12; ++k) nz.Q[k] = dp.Q[k];
    for (int k = 0; k < 3; ++k) { nz.ab_static[k] = dp.ab_static[k]; nz.wb_static[k] = dp.wb_static[k]; }
    for (int k = 0; k < 6; ++k) nz.R[k] = dp.R[k];
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        T xs[16], Pp[120], Pn[120], acc[3], obs[7];
        for (int k = 0; k < 16; ++k) xs[k] = (T)x[16 * i + k];
        const double* Pi = P + (int64_t)n * n * i;
        for (int a = 0; a < 15; ++a)
            for (int b = a; b < 15; ++b) Pp[sidx(a, b)] = (a < n && b < n) ? (T)(0.5 * (Pi[a * n + b] + Pi[b * n + a])) : T(0);
        PackedCov<T> Sb;                     // levels == 3: the covariance as register blocks (ekf_packed.hpp), kept across ticks
        if (levels == 3) cov_pack<T>(Pp, Sb);
        ArrayTop<T> top;                     // levels == 4: the covariance split between "LDS" rows (an array here) and registers (ekf_split.hpp)
        T lo[kLoWords];
        if (levels == 4) split_from_flat<T>(Pp, top, lo);
        for (int64_t t = 0; t < Tn; ++t) {
            const double* ut = u + (t * B + i) * 6;
            const T uu[6] = {(T)ut[0], (T)ut[1], (T)ut[2], (T)ut[3], (T)ut[4], (T)ut[5]};
            if (levels == 4) {   // what the fp64 multirate replay loop runs: predict and batch-form correction on the split covariance
                ekf_predict_split<T>(dp, nz, xs, top, lo, uu, acc);
                if (mask && mask[t * B + i]) {
                    const double* zt = z + (t * B + i) * 7;
                    const T zz[7] = {(T)zt[0], (T)zt[1], (T)zt[2], (T)zt[3], (T)zt[4], (T)zt[5], (T)zt[6]};
                    auto none7 = [](const T (&)[7]) {};
                    if (p->direct_orien_method) ekf_update_split<T, true>(dp, nz, xs, top, lo, zz, none7);
                    else ekf_update_split<T, false>(dp, nz, xs, top, lo, zz, none7);
                }
                if (t + 1 == Tn) split_to_flat<T>(top, lo, Pp);
                continue;
            }
            if (levels == 3) {   // what the multirate replay loop runs: packed predict, blocks unpacked only for a correction
                ekf_predict_packed<T>(dp, nz, xs, Sb, uu, acc);
                if (mask && mask[t * B + i]) {
                    const double* zt = z + (t * B + i) * 7;
                    const T zz[7] = {(T)zt[0], (T)zt[1], (T)zt[2], (T)zt[3], (T)zt[4], (T)zt[5], (T)zt[6]};
                    cov_unpack<T>(Sb, Pp);
                    if (p->direct_orien_method) ekf_update<T, true>(dp, nz, xs, Pp, zz, obs);
                    else ekf_update<T, false>(dp, nz, xs, Pp, zz, obs);
                    cov_pack<T>(Pp, Sb);
                }
                if (t + 1 == Tn) cov_unpack<T>(Sb, Pp);
                continue;
            }
            if (levels == 2) {   // the whole tick as one schedule (ekf_fused.hpp), what k_step runs
                const bool corr = mask && mask[t * B + i];
                T zz[7] = {T(0), T(0), T(0), T(0), T(0), T(0), T(1)};
                if (corr) {
                    const double* zt = z + (t * B + i) * 7;
                    for (int k = 0; k < 7; ++k) zz[k] = (T)zt[k];
                }
                int groups = 0;
                auto none3 = [](const T (&)[3]) {};
                auto none7 = [](const T (&)[7]) {};
                auto storex = []() {};
                auto storeq = [&](int q4, const T* w4) { std::memcpy(&Pn[4 * q4], w4, 4 * sizeof(T)); ++groups; };
                if (p->direct_orien_method) ekf_step_fused<T, true>(dp, nz, xs, Pp, uu, zz, corr, true, none3, none7, storex, storeq);
                else ekf_step_fused<T, false>(dp, nz, xs, Pp, uu, zz, corr, true, none3, none7, storex, storeq);
                if (groups != 30) xs[0] = (T)NAN;   // every 4-word group must be handed out exactly once
                std::memcpy(Pp, Pn, sizeof(Pp));
                continue;
            }
            if (levels) {
                ekf_predict_levels<T>(dp, nz, xs, Pp, uu, acc, Pn, [](int) {});
                std::memcpy(Pp, Pn, sizeof(Pp));
            } else {
                ekf_predict<T>(dp, nz, xs, Pp, uu, acc);
            }
            if (mask && mask[t * B + i]) {
                const double* zt = z + (t * B + i) * 7;
                const T zz[7] = {(T)zt[0], (T)zt[1], (T)zt[2], (T)zt[3], (T)zt[4], (T)zt[5], (T)zt[6]};
                if (p->direct_orien_method) ekf_update<T, true>(dp, nz, xs, Pp, zz, obs);
                else ekf_update<T, false>(dp, nz, xs, Pp, zz, obs);
            }
        }
        for (int k = 0; k < 16; ++k) x[16 * i + k] = (double)xs[k];
        double* Po = P + (int64_t)n * n * i;
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) Po[a * n + b] = (double)Pp[sidx(a, b)];
    }
    return B * Tn;
}

// ---- four-lanes-per-filter arithmetic (ekf_quad.hpp) on an emulated quad --------------------------------------
// L4<T> holds the values the four lanes of a quad hold in one register; HostQ<T> implements the cross-lane reads
// the device does with quad_perm DPP.  The algebra header is compiled unmodified.
template <typename T>
struct L4 {
    T v[4];
    L4() = default;
    template <typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    L4(S s) { for (int i = 0; i < 4; ++i) v[i] = (T)s; }
    L4 operator-() const { L4 o; for (int i = 0; i < 4; ++i) o.v[i] = -v[i]; return o; }
};
#define QLE_L4_OP(OP)                                                                                          \
    template <typename T> static inline L4<T> operator OP(const L4<T>& a, const L4<T>& b)                      \
    { L4<T> o; for (int i = 0; i < 4; ++i) o.v[i] = a.v[i] OP b.v[i]; return o; }
QLE_L4_OP(+) QLE_L4_OP(-) QLE_L4_OP(*) QLE_L4_OP(/)
#undef QLE_L4_OP
struct M4 { bool v[4]; };
template <typename T>
struct HostQ {
    using V = L4<T>;
    using M = M4;
    template <int K> static V bc(const V& a) { V o; for (int i = 0; i < 4; ++i) o.v[i] = a.v[K]; return o; }
    static V rot1(const V& a) { V o; o.v[0] = a.v[1]; o.v[1] = a.v[2]; o.v[2] = a.v[0]; o.v[3] = a.v[3]; return o; }
    static V rot2(const V& a) { V o; o.v[0] = a.v[2]; o.v[1] = a.v[0]; o.v[2] = a.v[1]; o.v[3] = a.v[3]; return o; }
    static V pick3(const V& a, const V& b, const V& c) { V o; o.v[0] = a.v[0]; o.v[1] = b.v[1]; o.v[2] = c.v[2]; o.v[3] = a.v[3]; return o; }
};

template <typename T>
static int64_t quad_run_batch_t(const orc_params* p, int64_t B, int64_t Tn, double* x, double* P, const double* u, const double* z,
                                const uint8_t* mask)
{
    using Q = HostQ<T>;
    using S = quad::ScalarQ<T>;
    using V = L4<T>;
    const int n = p->num_states;
    const DevParams<T> dp = to_dev<T>(p);
    quad::NoiseV<T> nz;
    for (int k = 0; k < 12; ++k) nz.Q[k] = dp.Q[k];
    for (int k = 0; k < 3; ++k) { nz.ab_static[k] = dp.ab_static[k]; nz.wb_static[k] = dp.wb_static[k]; }
    for (int k = 0; k < 6; ++k) nz.R[k] = dp.R[k];
    const T poison = (T)NAN;   // lane 3 of a quad carries no covariance: nothing may depend on what it holds
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        T Pp[120], xs[16], acc[3];
        V L[quad::kList], Ln[quad::kList], Prr[3], Ptt[3];
        for (int k = 0; k < 16; ++k) xs[k] = (T)x[16 * i + k];
        const double* Pi = P + (int64_t)n * n * i;
        for (int a = 0; a < 15; ++a)
            for (int b = a; b < 15; ++b) Pp[sidx(a, b)] = (a < n && b < n) ? (T)(0.5 * (Pi[a * n + b] + Pi[b * n + a])) : T(0);
        for (int k = 0; k < quad::kList; ++k)
            for (int l = 0; l < 4; ++l) L[k].v[l] = l < 3 ? Pp[quad_word(l, k)] : poison;
        for (int64_t t = 0; t < Tn; ++t) {
            const double* ut = u + (t * B + i) * 6;
            const T uu[6] = {(T)ut[0], (T)ut[1], (T)ut[2], (T)ut[3], (T)ut[4], (T)ut[5]};
            // scalar part, once per filter ...
            quad::PredU<T> pu;
            quad::predict_scalar<S, T>(dp, nz, xs, uu, acc, pu);
            // ... handed to the quad: full matrices in every lane, row / column j in lane j
            quad::PredQ<V> pq;
            for (int k = 0; k < 9; ++k) { pq.a_[k] = V(pu.A[k]); pq.bm_[k] = V(pu.Bm[k]); pq.rt_[k] = V(pu.Rt[k]); }
            for (int m = 0; m < 3; ++m)
                for (int l = 0; l < 4; ++l) {
                    pq.ar_[m].v[l] = l < 3 ? pu.A[3 * l + m] : poison;
                    pq.br_[m].v[l] = l < 3 ? pu.Bm[3 * l + m] : poison;
                    pq.rtr_[m].v[l] = l < 3 ? pu.Rt[3 * l + m] : poison;
                    pq.cqc_[m].v[l] = l < 3 ? pu.CQ[3 * m + l] : poison;
                }
            for (int l = 0; l < 4; ++l) { pq.qw_.v[l] = l < 3 ? pu.Qd[l] : poison; pq.qab_.v[l] = l < 3 ? pu.Qd[3 + l] : poison; pq.qwb_.v[l] = l < 3 ? pu.Qd[6 + l] : poison; }
            quad::predict_P<Q, T>(dp, pq, L, Ln, Prr, Ptt, [](int) {});
            if (mask && mask[t * B + i]) {
                const double* zt = z + (t * B + i) * 7;
                const T zz[7] = {(T)zt[0], (T)zt[1], (T)zt[2], (T)zt[3], (T)zt[4], (T)zt[5], (T)zt[6]};
                T Frr[9], Frt[9], Ftt[9];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) { Frr[3 * r + c] = Prr[r].v[c]; Ftt[3 * r + c] = Ptt[r].v[c]; Frt[3 * r + c] = Ln[QLE_QO(0, 2, r)].v[c]; }
                quad::UpdU<T> us;
                if (p->direct_orien_method) quad::update_scalar<S, T, true>(dp, nz, xs, zz, Frr, Frt, Ftt, us, [](const T (&)[7]) {});
                else quad::update_scalar<S, T, false>(dp, nz, xs, zz, Frr, Frt, Ftt, us, [](const T (&)[7]) {});
                quad::UpdQ<V> uq;
                for (int k = 0; k < 15; ++k) uq.u.Lm[k] = V(us.Lm[k]);
                for (int k = 0; k < 6; ++k) { uq.u.invd[k] = V(us.invd[k]); uq.u.yd[k] = V(us.yd[k]); }
                for (int k = 0; k < 9; ++k) uq.u.Gx[k] = V(us.Gx[k]);
                V dxo[5];
                if (p->direct_orien_method) quad::update_P<Q, true>(uq, Ln, Prr, Ptt, dxo);
                else quad::update_P<Q, false>(uq, Ln, Prr, Ptt, dxo);
                T dx[15];
                for (int b = 0; b < 5; ++b)
                    for (int l = 0; l < 3; ++l) dx[3 * b + l] = dxo[b].v[l];
                quad::update_inject<S, T>(dp, xs, dx);
            }
            for (int k = 0; k < quad::kList; ++k) L[k] = Ln[k];
        }
        for (int k = 0; k < quad::kList; ++k)
            for (int l = 0; l < 3; ++l) Pp[quad_word(l, k)] = L[k].v[l];
        for (int k = 0; k < 16; ++k) x[16 * i + k] = (double)xs[k];
        double* Po = P + (int64_t)n * n * i;
        for (int a = 0; a < n; ++a)
            for (int b = 0; b < n; ++b) Po[a * n + b] = (double)Pp[sidx(a, b)];
    }
    return B * Tn;
}

extern "C" {
// The engine's quaternion_exp (ekf_device.hpp: half-angle series with halving / doubling beyond pi/4) for n rotation vectors.
void orc_structured_quat_exp(const double* v, double* q, int64_t n, int dtype)
{
    for (int64_t i = 0; i < n; ++i) {
        if (dtype == 0) {
            const float vv[3] = {(float)v[3 * i], (float)v[3 * i + 1], (float)v[3 * i + 2]};
            float qq[4];
            quat_exp<float>(vv, qq);
            for (int k = 0; k < 4; ++k) q[4 * i + k] = qq[k];
        } else {
            const double vv[3] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
            double qq[4];
            quat_exp<double>(vv, qq);
            for (int k = 0; k < 4; ++k) q[4 * i + k] = qq[k];
        }
    }
}
// The quad arithmetic (ekf_quad.hpp) on an emulated quad; same contract as orc_run_batch.
int64_t orc_quad_run_batch(const orc_params* p, int64_t B, int64_t T, double* x, double* P, const double* u, const double* z,
                           const uint8_t* mask, int dtype)
{
    return dtype == 0 ? quad_run_batch_t<float>(p, B, T, x, P, u, z, mask) : quad_run_batch_t<double>(p, B, T, x, P, u, z, mask);
}
// Same contract as orc_run_batch (ekf_oracle.h).  dtype 0 = fp32 arithmetic, 1 = fp64.
// levels: 1 the levelled predict (the one k_predict runs) + sequential update, 0 the in-place congruences + sequential update,
// 2 the fused tick of k_step (ekf_fused.hpp).
int64_t orc_structured_run_batch(const orc_params* p, int64_t B, int64_t T, double* x, double* P, const double* u, const double* z,
                                 const uint8_t* mask, int dtype, int levels, int n_threads)
{
    return dtype == 0 ? run_batch_t<float>(p, B, T, x, P, u, z, mask, levels, n_threads)
                      : run_batch_t<double>(p, B, T, x, P, u, z, mask, levels, n_threads);
}
}
