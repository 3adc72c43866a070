// ekf_host.hpp -- host side shared by the translation units of libqle_ekf.so: the handle, its helpers and the
// declarations of the kernel launchers.  The kernels are instantiated in separate translation units
// (tu_predict / tu_step / tu_quad / tu_misc .hip, each compiled once per compute dtype) so that the library builds
// in parallel; ekf_capi.hip holds the C-ABI.  There is deliberately no CPU compute path.
#pragma once

#include "../../include/qle_ekf.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ekf_kernels.hpp"

using namespace qle;

// ------------------------------------------------------------------ errors

int qle_fail(int code, const char* fmt, ...);   // sets the thread-local message (ekf_capi.hip), returns code
#define fail qle_fail
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(QLE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define QLE_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != QLE_OK) return rc_; \
    } while (0)

// Nothing may throw across the C ABI: entry points that allocate host memory run under this guard.
#define QLE_GUARD_BEGIN try {
#define QLE_GUARD_END                                                                   \
    }                                                                                   \
    catch (const std::bad_alloc&) { return fail(QLE_ERR_NOMEM, "host allocation failed"); } \
    catch (const std::exception& e_) { return fail(QLE_ERR_INVALID, "unexpected exception: %s", e_.what()); } \
    catch (...) { return fail(QLE_ERR_INVALID, "unexpected exception"); }

// ------------------------------------------------------------------ handle
struct qle_batch {
    int64_t B = 0;
    int32_t dtype = QLE_F32;
    int32_t device = 0;
    int32_t block = 256;
    int32_t split = 0;        // nt == 3: which workgroups keep their tiles cached (cached_workgroup() in ekf_kernels.hpp)
    int32_t nt_refresh = 0;   // > 0: nt == 1 and the state is <= 40 MiB: non-temporal stores, cached-store tick every nt_refresh ticks
    int32_t nt = 0;        // cache policy of the hot kernels' state accesses: 0 cached, 1 L2-sized scheme (effective_nt), 2 non-temporal, 3 split
    int64_t chunk = 0;     // > 0: the lane-per-filter single-rate ticks are launched in chunks of this many filters (choose_cache_policy)
    bool quad_auto = true; // quad follows the rules of qle_create / qle_set_params (false: QLE_QUAD given)
    int32_t quad = 0;      // workgroup-cooperative tick kernel (ekf_quad_kernels.hpp): bit 0 ticks with tag poses, bit 1 predict-only ticks
    size_t wsz = 4;
    qle_params pub;
    qle_derived der;
    DevParams<float> pf;
    DevParams<double> pd;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t Bp = 0;        // batch padded to whole 64-filter tiles
    // The state: one array of 144-word records (x 16, packed P 120, 8 pad), updated in place.
    void* ring = nullptr;
    void* pfp = nullptr;   // [24 words] per-filter params, wave tiles
    bool pfp_on = false;
    bool loads_first = false;   // k_predict<float>: every load requested before the arithmetic starts (batches of at most one wave per SIMD)
    bool aux = false;
    void* aux_accel = nullptr;  // AoS [B][3], compute dtype
    void* aux_obs = nullptr;    // AoS [B][7]
    void* tick_u = nullptr;     // one tick of inputs in device layout
    void* tick_z = nullptr;
    double* stage = nullptr;    // AoS fp64 staging, kStageFilters filters
    uint8_t* stage_mask = nullptr;
    unsigned long long* counter = nullptr;
    bool state_set = false;
    bool compact = false;       // records hold the 9 x 9 pose block of P only (est_bias = false; qle_set_params, ekf_kernels.hpp)
    // device-side measurement gating (EKF.cpp:147-186)
    bool gating = false;
    int32_t* last_corr = nullptr;  // [B] index of each filter's last correcting tick, -1 = never
    uint8_t* flags = nullptr;      // [B] bit0 performed_correction, bit1 measurement consumed (last measurement tick)
    int64_t flags_tick = -1;       // tick whose launch last wrote the flag bytes (predict-only ticks do not)
    int64_t tick = 0;              // filter_update ticks executed so far (since the last origin shift)
    int64_t tick_origin = 0;       // ticks removed by origin shifts (reporting only)
    int64_t rebase_at = (int64_t)1 << 30;  // shift the tick origin when the counter reaches this (QLE_TICK_REBASE for tests)
    // multirate EKF (EKF.cpp:196-236, 251-264)
    bool mr = false;               // pub.multirate_ekf
    bool hist_dirty = true;        // state was overwritten: restart the history at the next tick
    int32_t* hist_first = nullptr; // [B] tick of each filter's oldest valid history entry
    void* mr_u = nullptr;          // IMU ring: mr_Cu slots of kHW words per filter
    void* mr_ckpt = nullptr;       // state checkpoints: mr_Nc slots, one per mr_k ticks
    void* mr_anchor = nullptr;     // one state slot: every filter's corrected entry at hist_first
    int32_t mr_k = 32, mr_Nc = 0, mr_Cu = 0;
    // the extra checkpoint (slot mr_Nc), placed at the expected entry of the next measurement (k_step_mr, ekf_kernels.hpp)
    int64_t e_tick = -1;           // tick whose state the slot holds, -1 = none
    int64_t e_want = -1;           // the predict launch of this tick fills it, -1 = none scheduled
    int64_t last_mr_launch = -1;   // tick of the last launch of k_step_mr (the cadence of the tag poses as the host sees it)
    double* stamp = nullptr;       // [B] apriltag_time per filter (dynamic delay)
    double* delay_cur = nullptr;   // [B] measurement_delay_curr (EKF.hpp:86)
    double t_curr = 0.0, uniform_age = 0.0;
    bool have_stamps = false;
};

struct qle_inputs {
    qle_batch* h = nullptr;  // owner; only dereferenced by calls that also take the handle or run before its destroy
    int32_t device = 0;      // copied so that destroy never touches the (possibly already destroyed) handle
    int64_t T = 0;
    int64_t n_slots = 0;
    std::vector<int32_t> slot;  // per tick: measurement slot or -1
    size_t pitch_u = 0, pitch_z = 0;
    void* u = nullptr;
    void* z = nullptr;
    void* truth = nullptr;      // AoS [B][7] fp64: r(3), q(4) at the end of the sequence
    void* truth_bias = nullptr; // AoS [B][6] fp64
    int32_t* d_slot = nullptr;  // the slot table on the device (qle_run_resident, generator)
    bool has_truth = false;
};

static constexpr int64_t kStageFilters = 32768;
static constexpr int64_t kStageDoubles = kStageFilters * 225;

template <typename T> static DevParams<T> make_dev(const qle_params& p, const qle_derived& d)
{
    DevParams<T> o;
    o.dT = (T)d.dT_nom;
    o.dTw = p.est_bias ? (T)d.dT_nom : T(0);
    o.bias_on = p.est_bias ? T(1) : T(0);
    o.small_ang_tol = (T)p.small_ang_tol;
    for (int i = 0; i < 3; ++i) { o.g[i] = (T)p.g[i]; o.r_v_cv[i] = (T)p.r_v_cv[i]; o.ab_static[i] = (T)p.ab_static[i]; o.wb_static[i] = (T)p.wb_static[i]; }
    for (int i = 0; i < 4; ++i) o.q_vc[i] = (T)d.q_vc[i];
    for (int i = 0; i < 9; ++i) o.C_vc[i] = (T)d.C_vc[i];
    for (int i = 0; i < 12; ++i) o.Q[i] = (T)d.Q[i];
    for (int i = 0; i < 6; ++i) o.R[i] = (T)d.R[i];
    o.compact = 0;   // set by qle_set_params
    return o;
}
template <typename T> static const DevParams<T>& dev(const qle_batch* h);
template <> const DevParams<float>& dev<float>(const qle_batch* h) { return h->pf; }
template <> const DevParams<double>& dev<double>(const qle_batch* h) { return h->pd; }

static inline dim3 grid_for(const qle_batch* h, int block) { return dim3((unsigned)((h->B + block - 1) / block)); }
// One tick as a sequence of launches over [i0, end): the whole batch at once, or h->chunk filters at a time.
template <typename F> static inline void for_chunks(const qle_batch* h, int block, F&& launch)
{
    const int64_t step = h->chunk > 0 ? h->chunk : h->B;
    for (int64_t i0 = 0; i0 < h->B; i0 += step) {
        const int64_t end = std::min(h->B, i0 + step);
        launch(dim3((unsigned)((end - i0 + block - 1) / block)), i0, end);
    }
}

static inline size_t slot_bytes(const qle_batch* h) { return (size_t)kSW * (size_t)h->Bp * h->wsz; }
// the state: one record array, updated in place by every tick
static inline void* state_cur(const qle_batch* h) { return h->ring; }

#define BY_DTYPE(h, FN, ...) ((h)->dtype == QLE_F32 ? FN<float>(__VA_ARGS__) : FN<double>(__VA_ARGS__))

// Kernel cache policy of this tick.  For an L2-sized state (h->nt == 1) the fastest sustained scheme measured
// (profiles/r01_tuning.md section 5) is: non-temporal loads AND stores -- the stores update the lines the state
// already has in the Infinity Cache and leave no dirty L2 to flush at the kernel boundary -- plus one tick with
// cached stores every nt_refresh (128) ticks, which re-allocates the state in the Infinity Cache.  Without the refresh the
// state drifts out of the cache within ~3 000 ticks and every tick streams from HBM (9.3 -> 10.9 us per predict
// at 65 536 filters); cached stores on every tick cost 9.9 us.  QLE_REFRESH=R overrides (0: cached stores always).
static inline int effective_nt(const qle_batch* h)
{
    if (h->nt == 1 && h->nt_refresh > 0) return (h->tick % h->nt_refresh) == 0 ? 1 : 2;
    return h->nt;
}

// Workgroup-cooperative tick kernel (ekf_quad_kernels.hpp): the single-rate tick in place, one 256-thread workgroup per tile.
static inline bool use_quad(const qle_batch* h, int bit) { return (h->quad & bit) != 0 && !h->mr && !h->compact; }

static inline GateParams make_gate(const qle_batch* h)
{
    GateParams g;
    std::memset(&g, 0, sizeof(g));
    g.limit = h->pub.limit_measurement_freq;
    g.upd_per_meas = h->der.upd_per_meas;
    g.corner_enbl = h->pub.corner_margin_enbl;
    g.n_tags = h->pub.n_tags;
    g.tick = (int32_t)h->tick;
    for (int i = 0; i < 9; ++i) g.K[i] = h->pub.camera_K[i];
    const double m = h->pub.tag_in_view_margin;
    g.x_lo = h->pub.camera_width * m;  g.x_hi = h->pub.camera_width * (1 - m);    // EKF.cpp:175-178
    g.y_lo = h->pub.camera_height * m; g.y_hi = h->pub.camera_height * (1 - m);
    for (int i = 0; i < QLE_MAX_TAGS; ++i) {
        g.hw[i] = h->pub.tag_widths[i] / 2;
        g.px[i] = h->pub.tag_positions[3 * i];
        g.py[i] = h->pub.tag_positions[3 * i + 1];
    }
    return g;
}

static inline MrParams make_mr(const qle_batch* h)
{
    MrParams m;
    std::memset(&m, 0, sizeof(m));
    m.k = h->mr_k;
    m.Nc = h->mr_Nc;
    m.Cu = h->mr_Cu;
    m.tick = (int32_t)h->tick;
    m.fixed_step = h->der.measurement_step_delay;
    m.dynamic = h->pub.dynamic_meas_delay;
    m.gate = h->gating ? 1 : 0;
    m.e_tick = h->e_tick >= 0 ? (int32_t)h->e_tick : -(1 << 30);
    m.slot_words = (int64_t)kSW * h->Bp;
    m.u_words = (int64_t)kHW * h->Bp;
    m.dT = h->der.dT_nom;
    m.offset = h->pub.dyn_measurement_delay_offset;
    m.delay_max = h->pub.measurement_delay_max;
    m.t_curr = h->t_curr;
    m.uniform_age = h->uniform_age;
    return m;
}
// where tick t's IMU sample goes / where the state after checkpoint tick t goes (nullptr: not a checkpoint tick)
static inline void* mr_u_slot_host(const qle_batch* h, int64_t t)
{
    int64_t s = t % h->mr_Cu;
    if (s < 0) s += h->mr_Cu;
    return (char*)h->mr_u + (size_t)s * (size_t)kHW * (size_t)h->Bp * h->wsz;
}
static inline void* mr_ck_slot_host(const qle_batch* h, int64_t t)
{
    if (t < 0 || t % h->mr_k != 0) return nullptr;
    return (char*)h->mr_ckpt + slot_bytes(h) * (size_t)((t / h->mr_k) % h->mr_Nc);
}
// the checkpoint copy a predict launch of tick t makes, if any: the grid slot, or the extra slot when t is the tick it was scheduled for
static inline void* mr_ck_for_predict(qle_batch* h, int64_t t, bool* extra)
{
    *extra = false;
    if (void* grid = mr_ck_slot_host(h, t)) {
        if (t == h->e_want) h->e_want = -1;          // the grid checkpoint of this tick serves
        return grid;
    }
    if (t >= 0 && t == h->e_want) {
        h->e_want = -1;
        h->e_tick = t;
        *extra = true;
        return (char*)h->mr_ckpt + slot_bytes(h) * (size_t)h->mr_Nc;
    }
    return nullptr;
}
// after a launch of k_step_mr at tick n: where will the entry of the NEXT tag poses be?
static inline void mr_schedule_extra(qle_batch* h)
{
    const int64_t n = h->tick;
    const int64_t period = h->last_mr_launch >= 0 ? n - h->last_mr_launch : (int64_t)h->der.upd_per_meas;
    h->last_mr_launch = n;
    int64_t step = h->der.measurement_step_delay;                                      // EKF.cpp:93
    if (h->pub.dynamic_meas_delay && !h->have_stamps)                                  // EKF.cpp:199-200 with the uniform age
        step = std::max<int64_t>((int64_t)(std::min(h->uniform_age + h->pub.dyn_measurement_delay_offset, h->pub.measurement_delay_max) / h->der.dT_nom + 0.5), 1);
    const int64_t e = n + period - step;
    // only an entry that lies AFTER this tick is worth a copy (a longer delay than the cadence puts it inside the replayed range,
    // one tick after the anchor), and only while the cadence is short enough for the IMU ring to still hold the samples
    // (per-filter stamps with dynamic_meas_delay: every filter has its own step delay, the fixed-step guess serves none of them reliably)
    const bool per_filter_delays = h->pub.dynamic_meas_delay && h->have_stamps;
    h->e_want = (e > n && period > 1 && period + step < h->mr_Cu && !per_filter_delays) ? e : -1;
}

// fp64 kernels that keep the covariance split between the LDS and registers (ekf_split.hpp) are launched with 37.5 KiB of dynamic LDS per
// wave: 150 KiB for a 256-thread workgroup, more than the 64 KiB a launch gets without asking.
template <typename T> static inline size_t split_lds(const qle_batch* h) { return sizeof(T) == 8 ? (size_t)(h->block / kTile) * kMrLdsPerWave : 0; }
#define QLE_ASK_LDS(KERNEL, BYTES)                                                                                                    \
    do {                                                                                                                               \
        if ((BYTES) > 65536) {                                                                                                         \
            static bool asked = false;   /* per call site = per instantiation; the attribute is a property of the kernel */             \
            if (!asked) {                                                                                                              \
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES))); \
                asked = true;                                                                                                          \
            }                                                                                                                          \
        }                                                                                                                              \
    } while (0)

// ---- kernel launchers, defined and explicitly instantiated for float and double in the tu_*.hip files ----
int mr_prepare(qle_batch* h);                                                                  // tu_misc
template <typename T> int launch_step_mr(qle_batch* h, const void* u, const void* z);          // tu_misc: k_step_mr
template <typename T> int launch_update(qle_batch* h, const void* z);                          // tu_misc: k_update
template <typename T> int run_resident_t(qle_batch* h, const qle_inputs* in, int64_t t0, int64_t n);   // tu_misc: k_run_resident
template <typename T> int launch_predict_sd(qle_batch* h, const void* u, const void* src, void* dst, bool history);   // tu_predict: k_predict
template <typename T> int launch_step_lane(qle_batch* h, const void* u, const void* z);        // tu_step: k_step
template <typename T> int launch_quad(qle_batch* h, const void* u, const void* z);             // tu_quad: kw_tick
// tu_compact: the same lane-per-filter kernels instantiated for compact records (h->compact)
template <typename T> int launch_predict_compact(qle_batch* h, const void* u, const void* src, void* dst);
template <typename T> int launch_step_compact(qle_batch* h, const void* u, const void* z);
template <typename T> int launch_update_compact(qle_batch* h, const void* z);
template <typename T> int run_resident_compact(qle_batch* h, const qle_inputs* in, int64_t t0, int64_t n);
