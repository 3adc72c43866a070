// tu_predict.hip -- launcher of k_predict (one lane per filter, predict-only tick)
// Compiled once per compute dtype (-DQLE_TU_T=float|double); see ekf_host.hpp.
#include "ekf_host.hpp"

#ifndef QLE_TU_T
#error "compile with -DQLE_TU_T=float or -DQLE_TU_T=double"
#endif


// prediction_step from `src` into `dst`; keep_u: the record also stores the IMU sample (multirate history).
template <typename T>
int launch_predict_sd(qle_batch* h, const void* u, const void* src, void* dst, bool history)
{
    if (h->compact) return launch_predict_compact<T>(h, u, src, dst);   // never with the multirate history
    const DevParams<T>& p = dev<T>(h);
    const dim3 b(h->block);
    T* acc = h->aux ? (T*)h->aux_accel : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
    // multirate history of this tick: the IMU sample's ring slot and, on checkpoint ticks, the checkpoint slot
    T* hu = history ? (T*)mr_u_slot_host(h, h->tick) : (T*)nullptr;
    bool extra_ck = false;
    T* hc = history ? (T*)mr_ck_for_predict(h, h->tick, &extra_ck) : (T*)nullptr;
    // the extra checkpoint stays in the Infinity Cache when it fits there next to the state (cached stores), else it is streamed
    const int32_t ck_cached = extra_ck && 2 * slot_bytes(h) <= ((size_t)200 << 20) ? 1 : 0;
    // QLE_LDS_PAD=bytes (experiments): dynamic LDS the kernel never touches, to cap the workgroups a CU holds (occupancy experiments
    // on the 131 072 ... 524 288-filter plateau, profiles/r03_tuning.md)
    static const size_t lds_pad = [] { const char* s = std::getenv("QLE_LDS_PAD"); return s ? (size_t)std::atoll(s) : (size_t)0; }();
    // "loads first" (predict_tick): the fp32 tick of a batch that gives every SIMD at most one wave
    const bool lf = sizeof(T) == 4 && h->loads_first;
#define QLE_PRED_L(F, N, M, L) for_chunks(h, h->block, [&](dim3 gc, int64_t i0, int64_t end) { hipLaunchKernelGGL((k_predict<T, F, N, M, false, L>), gc, b, lds_pad, h->stream, (const T*)src, (T*)dst, (const T*)u, end, i0, (int32_t)gc.x, (int32_t)h->block, h->split, ck_cached, pfp, acc, hu, hc, p); })
#define QLE_PRED(F, N, M) do { if constexpr (sizeof(T) == 4) { if (lf) QLE_PRED_L(F, N, M, true); else QLE_PRED_L(F, N, M, false); } else QLE_PRED_L(F, N, M, false); } while (0)
#define QLE_PRED_N(N, M) do { if (h->pfp_on) QLE_PRED(true, N, M); else QLE_PRED(false, N, M); } while (0)
    const int nt = effective_nt(h);
#define QLE_PRED_M(M) do { if (nt == 2) QLE_PRED_N(2, M); else if (nt == 1) QLE_PRED_N(1, M); else QLE_PRED_N(0, M); } while (0)
    if (history) { if (nt == 3) QLE_PRED_N(3, true); else QLE_PRED_M(true); }
    else if (nt == 3) QLE_PRED_N(3, false);
    else QLE_PRED_M(false);
#undef QLE_PRED_M
#undef QLE_PRED_N
#undef QLE_PRED
#undef QLE_PRED_L
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}

template int launch_predict_sd<QLE_TU_T>(qle_batch*, const void*, const void*, void*, bool);
