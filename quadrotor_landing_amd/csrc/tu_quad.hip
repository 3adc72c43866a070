// tu_quad.hip -- launcher of kw_tick (workgroup-cooperative tick: scalar wave + covariance quads, ekf_quad_kernels.hpp)
// Compiled once per compute dtype (-DQLE_TU_T=float|double); see ekf_host.hpp.
#include "ekf_host.hpp"
#include "ekf_quad_kernels.hpp"
#ifndef QLE_TU_T
#error "compile with -DQLE_TU_T=float or -DQLE_TU_T=double"
#endif

template <typename T, bool STEP, bool DIRECT, bool GATE>
static int launch_quad_dg(qle_batch* h, const void* u, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const GateParams gp = make_gate(h);
    const dim3 g((unsigned)(h->Bp / kTile)), b(kBlock);
    T *st = (T*)state_cur(h), *acc = h->aux ? (T*)h->aux_accel : (T*)nullptr, *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
#define QLE_QT_LAUNCH(F, N) hipLaunchKernelGGL((kw_tick<T, DIRECT, F, GATE, STEP, N>), g, b, 0, h->stream, p, gp, st, (const T*)u, (const T*)z, pfp, acc, obs, h->last_corr, h->flags, h->B, h->split)
#define QLE_QT_N(N) do { if (h->pfp_on) QLE_QT_LAUNCH(true, N); else QLE_QT_LAUNCH(false, N); } while (0)
    const int nt = effective_nt(h);
    if (nt == 3) QLE_QT_N(3); else if (nt == 2) QLE_QT_N(2); else if (nt == 1) QLE_QT_N(1); else QLE_QT_N(0);
#undef QLE_QT_N
#undef QLE_QT_LAUNCH
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
template <typename T>
int launch_quad(qle_batch* h, const void* u, const void* z)
{
    if (!z) return launch_quad_dg<T, false, false, false>(h, u, nullptr);
    if (h->pub.direct_orien_method) return h->gating ? launch_quad_dg<T, true, true, true>(h, u, z) : launch_quad_dg<T, true, true, false>(h, u, z);
    return h->gating ? launch_quad_dg<T, true, false, true>(h, u, z) : launch_quad_dg<T, true, false, false>(h, u, z);
}


template int launch_quad<QLE_TU_T>(qle_batch*, const void*, const void*);
