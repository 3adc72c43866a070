// tu_misc.hip -- launchers of k_step_mr, k_update, k_run_resident
// Compiled once per compute dtype (-DQLE_TU_T=float|double); see ekf_host.hpp.
#include "ekf_host.hpp"
#ifndef QLE_TU_T
#error "compile with -DQLE_TU_T=float or -DQLE_TU_T=double"
#endif

#if defined(QLE_MR_STAMPS)
// diagnostic build only (make dbg): the per-wave s_memtime stamps of the last k_step_mr launch of this dtype's translation unit
#define QLE_CAT2(a, b) a##b
#define QLE_CAT(a, b) QLE_CAT2(a, b)
extern "C" int QLE_CAT(qle_debug_clocks_, QLE_TU_T)(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qle::qle_dbg_clock), sizeof(unsigned long long) * (size_t)n);
}
extern "C" int QLE_CAT(qle_debug_split_clocks_, QLE_TU_T)(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qle::qle_dbg_split_clock), sizeof(unsigned long long) * (size_t)n);
}
#endif

// One multirate tick that carries tag poses (predict-only multirate ticks go through launch_predict).
template <typename T>
int launch_step_mr(qle_batch* h, const void* u, const void* z)
{
    QLE_TRY(mr_prepare(h));
    const DevParams<T>& p = dev<T>(h);
    const GateParams gp = make_gate(h);
    const MrParams m = make_mr(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T *acc = h->aux ? (T*)h->aux_accel : (T*)nullptr, *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
    const double* stamp = (h->have_stamps && h->pub.dynamic_meas_delay) ? h->stamp : nullptr;
    const size_t lds = split_lds<T>(h);
#define QLE_MR_LAUNCH(D, F) do { QLE_ASK_LDS((k_step_mr<T, D, F>), lds); QLE_MR_LAUNCH1(D, F); } while (0)
#define QLE_MR_LAUNCH1(D, F) hipLaunchKernelGGL((k_step_mr<T, D, F>), g, b, lds, h->stream, (T*)state_cur(h), (const T*)u, (const T*)z, h->B, (int32_t)g.x, (int32_t)b.x, h->hist_first, (T*)h->mr_u, (T*)h->mr_ckpt, (T*)h->mr_anchor, pfp, stamp, acc, obs, h->last_corr, h->flags, h->delay_cur, p, gp, m)
#ifdef QLE_DEBUG_PTRS   // diagnostic builds only: where every buffer of the launch lies (to place a fault address)
    {
        const size_t sb = slot_bytes(h);
        std::fprintf(stderr, "[qle] k_step_mr tick %lld B %lld block %d lds %zu e_tick %lld\n", (long long)h->tick, (long long)h->B, h->block, lds, (long long)h->e_tick);
        auto rng = [](const char* n, const void* p0, size_t bytes) { std::fprintf(stderr, "[qle]   %-10s %p .. %p (%zu B)\n", n, p0, (const char*)p0 + bytes, bytes); };
        rng("cur", state_cur(h), sb);
        rng("mr_u", h->mr_u, (size_t)h->mr_Cu * kHW * (size_t)h->Bp * h->wsz);
        rng("mr_ckpt", h->mr_ckpt, (size_t)(h->mr_Nc + 1) * sb);
        rng("mr_anchor", h->mr_anchor, sb);
        rng("u", u, (size_t)kUW * h->Bp * h->wsz);
        rng("z", z, (size_t)kZW * h->Bp * h->wsz);
        rng("hist_first", h->hist_first, 4 * (size_t)h->Bp);
        rng("last_corr", h->last_corr, h->last_corr ? 4 * (size_t)h->Bp : 0);
        rng("flags", h->flags, h->flags ? (size_t)h->Bp : 0);
        rng("delay_cur", h->delay_cur, 8 * (size_t)h->Bp);
        rng("stamp", h->stamp, 8 * (size_t)h->Bp);
        std::fflush(stderr);
    }
#endif
    if (h->pub.direct_orien_method) { if (h->pfp_on) QLE_MR_LAUNCH(true, true); else QLE_MR_LAUNCH(true, false); }
    else { if (h->pfp_on) QLE_MR_LAUNCH(false, true); else QLE_MR_LAUNCH(false, false); }
#ifdef QLE_DEBUG_PTRS
    { hipError_t e_ = hipStreamSynchronize(h->stream); std::fprintf(stderr, "[qle]   launch done: %s\n", hipGetErrorString(e_)); }
#endif
#undef QLE_MR_LAUNCH
#undef QLE_MR_LAUNCH1
    HIP_TRY(hipGetLastError());
    mr_schedule_extra(h);
    return QLE_OK;
}


template <typename T, bool DIRECT>
static int launch_update_d(qle_batch* h, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T *st = (T*)state_cur(h), *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
    const size_t lds = split_lds<T>(h);
    if (h->pfp_on) { QLE_ASK_LDS((k_update<T, DIRECT, true>), lds); hipLaunchKernelGGL((k_update<T, DIRECT, true>), g, b, lds, h->stream, st, (const T*)z, h->B, (int32_t)g.x, (int32_t)b.x, pfp, obs, p); }
    else { QLE_ASK_LDS((k_update<T, DIRECT, false>), lds); hipLaunchKernelGGL((k_update<T, DIRECT, false>), g, b, lds, h->stream, st, (const T*)z, h->B, (int32_t)g.x, (int32_t)b.x, pfp, obs, p); }
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
template <typename T>
int launch_update(qle_batch* h, const void* z)
{
    if (h->compact) return launch_update_compact<T>(h, z);
    return h->pub.direct_orien_method ? launch_update_d<T, true>(h, z) : launch_update_d<T, false>(h, z);
}

// On-chip-resident variant: ONE launch advances every filter by n ticks with x and P held in
// registers; HBM traffic is the state once plus the inputs.  Not the unit of work of the headline
// metric (one launch per tick); reported separately.
template <typename T>
int run_resident_t(qle_batch* h, const qle_inputs* in, int64_t t0, int64_t n)
{
    if (h->compact) return run_resident_compact<T>(h, in, t0, n);
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    const T* pfp = (const T*)h->pfp;
    const int64_t pu = (int64_t)(in->pitch_u / h->wsz), pz = (int64_t)(in->pitch_z / h->wsz);
    const size_t lds = split_lds<T>(h);
#define QLE_RES(D, F) QLE_ASK_LDS((k_run_resident<T, D, F>), lds); hipLaunchKernelGGL((k_run_resident<T, D, F>), g, b, lds, h->stream, p, (T*)state_cur(h), (const T*)in->u, (const T*)in->z, (const int32_t*)in->d_slot, pu, pz, in->T, t0, n, pfp, h->B)
    if (h->pub.direct_orien_method) { if (h->pfp_on) { QLE_RES(true, true); } else { QLE_RES(true, false); } }
    else { if (h->pfp_on) { QLE_RES(false, true); } else { QLE_RES(false, false); } }
#undef QLE_RES
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
template int launch_step_mr<QLE_TU_T>(qle_batch*, const void*, const void*);
template int launch_update<QLE_TU_T>(qle_batch*, const void*);
template int run_resident_t<QLE_TU_T>(qle_batch*, const qle_inputs*, int64_t, int64_t);
