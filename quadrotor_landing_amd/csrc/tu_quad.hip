// tu_quad.hip -- launcher of kw_tick (workgroup-cooperative tick: scalar wave + covariance quads, ekf_quad_kernels.hpp)
// Compiled once per compute dtype (-DQLE_TU_T=float|double); see ekf_host.hpp.
#include "ekf_host.hpp"
#include "ekf_quad_kernels.hpp"
#ifndef QLE_TU_T
#error "compile with -DQLE_TU_T=float or -DQLE_TU_T=double"
#endif

#if defined(QLE_MR_STAMPS)
// diagnostic build only (make dbg): the per-workgroup s_memtime stamps of the last kw_tick launch of this dtype
#define QLE_CAT2(a, b) a##b
#define QLE_CAT(a, b) QLE_CAT2(a, b)
extern "C" int QLE_CAT(qle_debug_clocks_kw_, QLE_TU_T)(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(qle::qle_dbg_clock), sizeof(unsigned long long) * (size_t)n);
}
#endif

template <typename T, bool STEP, bool DIRECT, bool GATE>
static int launch_quad_dg(qle_batch* h, const void* u, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const GateParams gp = make_gate(h);
    // a quarter tile per workgroup while that gives at most one workgroup per CU (up to 4 096 filters), else a tile: measured
    // (profiles/r03_tuning.md) 4 096 fp64 filters 12.0 -> 9.8 us per correcting tick, 1 024: 11.6 -> 9.5; 8 192 (two quarter-tile
    // workgroups per CU) 17.9 against 12.5 us with whole tiles
    static const int fpw_env = [] { const char* e = std::getenv("QLE_WG_FILTERS"); return e ? std::atoi(e) : 0; }();
    const int64_t tiles = h->Bp / kTile;
    const int fpw = fpw_env == 16 || fpw_env == 64 ? fpw_env : (tiles * 4 <= 256 ? 16 : 64);
    const dim3 g((unsigned)(tiles * (kTile / fpw))), b(kBlock);
    T *st = (T*)state_cur(h), *acc = h->aux ? (T*)h->aux_accel : (T*)nullptr, *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
#define QLE_QT_LAUNCH_W(F, N, W) hipLaunchKernelGGL((kw_tick<T, DIRECT, F, GATE, STEP, N, W>), g, b, 0, h->stream, st, (const T*)u, (const T*)z, h->B, (int32_t)g.x, h->split, pfp, acc, obs, h->last_corr, h->flags, p, gp)
#define QLE_QT_LAUNCH(F, N) do { if (fpw == 16) QLE_QT_LAUNCH_W(F, N, 16); else QLE_QT_LAUNCH_W(F, N, 64); } while (0)
#define QLE_QT_N(N) do { if (h->pfp_on) QLE_QT_LAUNCH(true, N); else QLE_QT_LAUNCH(false, N); } while (0)
    // the "split" policy (3) belongs to states larger than the Infinity Cache, where this kernel is never selected (<= 4 096 filters);
    // under a QLE_NT=3 override it runs with cached accesses
    const int nt = effective_nt(h);
    if (nt == 2) QLE_QT_N(2); else if (nt == 1) QLE_QT_N(1); else QLE_QT_N(0);
#undef QLE_QT_N
#undef QLE_QT_LAUNCH
#undef QLE_QT_LAUNCH_W
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
