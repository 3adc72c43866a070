// tu_compact.hip -- launchers of the lane-per-filter kernels on COMPACT records (est_bias = false, relative_pose_EKF.cpp:92: the record
// keeps the 9 x 9 pose block of P only; ekf_kernels.hpp).  Separate instantiations, so that the full-record kernels are compiled
// exactly as before.  Compiled once per compute dtype (-DQLE_TU_T=float|double); see ekf_host.hpp.
#include "ekf_host.hpp"

#ifndef QLE_TU_T
#error "compile with -DQLE_TU_T=float or -DQLE_TU_T=double"
#endif

template <typename T>
int launch_predict_compact(qle_batch* h, const void* u, const void* src, void* dst)
{
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T* acc = h->aux ? (T*)h->aux_accel : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
#define QLE_PRED(F, N) hipLaunchKernelGGL((k_predict<T, F, N, false, true>), g, b, 0, h->stream, (const T*)src, (T*)dst, (const T*)u, h->B, (int64_t)0, (int32_t)g.x, (int32_t)b.x, h->split, 0, pfp, acc, (T*)nullptr, (T*)nullptr, p)
#define QLE_PRED_N(N) do { if (h->pfp_on) QLE_PRED(true, N); else QLE_PRED(false, N); } while (0)
    const int nt = effective_nt(h);
    if (nt == 3) QLE_PRED_N(3); else if (nt == 2) QLE_PRED_N(2); else if (nt == 1) QLE_PRED_N(1); else QLE_PRED_N(0);
#undef QLE_PRED_N
#undef QLE_PRED
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}

template <typename T, bool DIRECT, bool GATE>
static int launch_step_compact_dg(qle_batch* h, const void* u, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const GateParams gp = make_gate(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T *st = (T*)state_cur(h), *acc = h->aux ? (T*)h->aux_accel : (T*)nullptr, *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
#define QLE_STEP_LAUNCH(F, N) hipLaunchKernelGGL((k_step<T, DIRECT, F, GATE, N, true>), g, b, 0, h->stream, st, (const T*)u, (const T*)z, h->B, (int64_t)0, (int32_t)g.x, (int32_t)b.x, h->split, pfp, acc, obs, h->last_corr, h->flags, p, gp)
#define QLE_STEP_N(N) do { if (h->pfp_on) QLE_STEP_LAUNCH(true, N); else QLE_STEP_LAUNCH(false, N); } while (0)
    const int nt = effective_nt(h);
    if (nt == 3) QLE_STEP_N(3); else if (nt == 2) QLE_STEP_N(2); else if (nt == 1) QLE_STEP_N(1); else QLE_STEP_N(0);
#undef QLE_STEP_N
#undef QLE_STEP_LAUNCH
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
template <typename T>
int launch_step_compact(qle_batch* h, const void* u, const void* z)
{
    if (h->pub.direct_orien_method) return h->gating ? launch_step_compact_dg<T, true, true>(h, u, z) : launch_step_compact_dg<T, true, false>(h, u, z);
    return h->gating ? launch_step_compact_dg<T, false, true>(h, u, z) : launch_step_compact_dg<T, false, false>(h, u, z);
}

template <typename T>
int launch_update_compact(qle_batch* h, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T *st = (T*)state_cur(h), *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
    const size_t lds = split_lds<T>(h);
#define QLE_UPD(D, F) QLE_ASK_LDS((k_update<T, D, F, true>), lds); hipLaunchKernelGGL((k_update<T, D, F, true>), g, b, lds, h->stream, st, (const T*)z, h->B, (int32_t)g.x, (int32_t)b.x, pfp, obs, p)
    if (h->pub.direct_orien_method) { if (h->pfp_on) { QLE_UPD(true, true); } else { QLE_UPD(true, false); } }
    else { if (h->pfp_on) { QLE_UPD(false, true); } else { QLE_UPD(false, false); } }
#undef QLE_UPD
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}

template <typename T>
int run_resident_compact(qle_batch* h, const qle_inputs* in, int64_t t0, int64_t n)
{
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    const T* pfp = (const T*)h->pfp;
    const int64_t pu = (int64_t)(in->pitch_u / h->wsz), pz = (int64_t)(in->pitch_z / h->wsz);
    const size_t lds = split_lds<T>(h);   // fp64 keeps the covariance split between the LDS and registers here too (k_run_resident)
#define QLE_RES(D, F) QLE_ASK_LDS((k_run_resident<T, D, F, true>), lds); hipLaunchKernelGGL((k_run_resident<T, D, F, true>), g, b, lds, h->stream, p, (T*)state_cur(h), (const T*)in->u, (const T*)in->z, (const int32_t*)in->d_slot, pu, pz, in->T, t0, n, pfp, h->B)
    if (h->pub.direct_orien_method) { if (h->pfp_on) { QLE_RES(true, true); } else { QLE_RES(true, false); } }
    else { if (h->pfp_on) { QLE_RES(false, true); } else { QLE_RES(false, false); } }
#undef QLE_RES
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}

template int launch_predict_compact<QLE_TU_T>(qle_batch*, const void*, const void*, void*);
template int launch_step_compact<QLE_TU_T>(qle_batch*, const void*, const void*);
template int launch_update_compact<QLE_TU_T>(qle_batch*, const void*);
template int run_resident_compact<QLE_TU_T>(qle_batch*, const qle_inputs*, int64_t, int64_t);
