// tu_step.hip -- launcher of k_step (one lane per filter, fused predict + masked update)
// Compiled once per compute dtype (-DQLE_TU_T=float|double); see ekf_host.hpp.
#include "ekf_host.hpp"

#ifndef QLE_TU_T
#error "compile with -DQLE_TU_T=float or -DQLE_TU_T=double"
#endif

template <typename T, bool DIRECT, bool GATE>
static int launch_step_dg(qle_batch* h, const void* u, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const GateParams gp = make_gate(h);
    const dim3 b(h->block);
    T *st = (T*)state_cur(h), *acc = h->aux ? (T*)h->aux_accel : (T*)nullptr, *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
#define QLE_STEP_LAUNCH(F, N) for_chunks(h, h->block, [&](dim3 gc, int64_t i0, int64_t end) { hipLaunchKernelGGL((k_step<T, DIRECT, F, GATE, N>), gc, b, 0, h->stream, st, (const T*)u, (const T*)z, end, i0, (int32_t)gc.x, (int32_t)h->block, h->split, pfp, acc, obs, h->last_corr, h->flags, p, gp); })
#define QLE_STEP_N(N) do { if (h->pfp_on) QLE_STEP_LAUNCH(true, N); else QLE_STEP_LAUNCH(false, N); } while (0)
    const int nt = effective_nt(h);
    if (nt == 3) QLE_STEP_N(3); else if (nt == 2) QLE_STEP_N(2); else if (nt == 1) QLE_STEP_N(1); else QLE_STEP_N(0);
#undef QLE_STEP_N
#undef QLE_STEP_LAUNCH
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}

template <typename T>
int launch_step_lane(qle_batch* h, const void* u, const void* z)
{
    if (h->compact) return launch_step_compact<T>(h, u, z);
    if (h->pub.direct_orien_method) return h->gating ? launch_step_dg<T, true, true>(h, u, z) : launch_step_dg<T, true, false>(h, u, z);
    return h->gating ? launch_step_dg<T, false, true>(h, u, z) : launch_step_dg<T, false, false>(h, u, z);
}
template int launch_step_lane<QLE_TU_T>(qle_batch*, const void*, const void*);
