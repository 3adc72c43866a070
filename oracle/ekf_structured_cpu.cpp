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

#include "../quadrotor_landing_amd/csrc/ekf_device.hpp"
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
        for (int64_t t = 0; t < Tn; ++t) {
            const double* ut = u + (t * B + i) * 6;
            const T uu[6] = {(T)ut[0], (T)ut[1], (T)ut[2], (T)ut[3], (T)ut[4], (T)ut[5]};
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

extern "C" {
// Same contract as orc_run_batch (ekf_oracle.h).  dtype 0 = fp32 arithmetic, 1 = fp64.
// levels != 0 uses the levelled predict (the one k_predict/k_step run), 0 the in-place congruences.
int64_t orc_structured_run_batch(const orc_params* p, int64_t B, int64_t T, double* x, double* P, const double* u, const double* z,
                                 const uint8_t* mask, int dtype, int levels, int n_threads)
{
    return dtype == 0 ? run_batch_t<float>(p, B, T, x, P, u, z, mask, levels, n_threads)
                      : run_batch_t<double>(p, B, T, x, P, u, z, mask, levels, n_threads);
}
}
