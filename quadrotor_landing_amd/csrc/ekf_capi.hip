// ekf_capi.hip -- C-ABI (include/qle_ekf.h) over the HIP kernels.
// Host side of the batched relative-pose EKF engine: handle and device-memory
// management, parameter derivation (initialize_params, EKF.cpp:87-125 of the
// reference), AoS<->quad-row staging, launches on the handle's own stream.
// There is deliberately no CPU compute path in this file.
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
#include "ekf_rows.hpp"
#include "synth_kernels.hpp"

using namespace qle;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
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

extern "C" const char* qle_last_error(void) { return g_err.c_str(); }
extern "C" const char* qle_version(void) { return "quadrotor_landing_amd 0.1 (gfx950)"; }

extern "C" int qle_device_count(int32_t* count)
{
    if (!count) return fail(QLE_ERR_INVALID, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(QLE_ERR_NO_DEVICE, "no HIP device available (%s); this engine has no CPU fallback", hipGetErrorString(e)); }
    *count = n;
    return QLE_OK;
}

// -------------------------------------------------------------- parameters
extern "C" int qle_params_default(qle_params* p)
{   // RelativePoseEKF::RelativePoseEKF(), EKF.cpp:28-81; cov_init from NODE.cpp:89-93
    if (!p) return fail(QLE_ERR_INVALID, "params is null");
    std::memset(p, 0, sizeof(*p));
    p->update_freq = 100.0;
    p->measurement_freq = 10.0;
    p->measurement_delay = 0.010;
    p->measurement_delay_max = 0.200;
    p->dyn_measurement_delay_offset = 0.0;
    p->est_bias = 1;
    p->limit_measurement_freq = 0;
    p->corner_margin_enbl = 1;
    p->direct_orien_method = 0;
    p->multirate_ekf = 0;
    p->dynamic_meas_delay = 0;
    p->r_cov_init = 0.1; p->v_cov_init = 0.1; p->ang_cov_init = 0.15; p->ab_cov_init = 0.5; p->wb_cov_init = 0.1;
    for (int i = 0; i < 3; ++i) { p->Q_a[i] = 0.005; p->Q_w[i] = 0.0005; p->Q_ab[i] = 5E-5; p->Q_wb[i] = 5E-6; }
    p->R_r[0] = 0.005; p->R_r[1] = 0.005; p->R_r[2] = 0.015;
    p->R_ang[0] = 0.0025; p->R_ang[1] = 0.0025; p->R_ang[2] = 0.025;
    p->r_v_cv[2] = -0.073;
    p->q_vc[0] = 0.70711; p->q_vc[1] = -0.70711;  // Quaterniond(w=0, x=0.70711, y=-0.70711, z=0), EKF.cpp:56
    p->camera_K[0] = 241.4268; p->camera_K[2] = 376.5; p->camera_K[4] = 241.4268; p->camera_K[5] = 240.5; p->camera_K[8] = 1.0;
    p->camera_width = 752; p->camera_height = 480;
    p->n_tags = 1;
    p->tag_in_view_margin = 0.02;
    p->tag_widths[0] = 0.8;
    p->small_ang_tol = 1E-10;
    p->g[2] = -9.8;
    return QLE_OK;
}

extern "C" int qle_params_derive(const qle_params* p, qle_derived* d)
{   // RelativePoseEKF::initialize_params(), EKF.cpp:87-125
    if (!p || !d) return fail(QLE_ERR_INVALID, "null argument");
    if (!(p->update_freq > 0.0) || !(p->measurement_freq > 0.0)) return fail(QLE_ERR_INVALID, "update_freq and measurement_freq must be > 0");
    if (p->n_tags < 0 || p->n_tags > QLE_MAX_TAGS) return fail(QLE_ERR_INVALID, "n_tags out of range [0,%d]", QLE_MAX_TAGS);
    std::memset(d, 0, sizeof(*d));
    d->dT_nom = 1.0 / p->update_freq;                                               // :90
    d->upd_per_meas = (int32_t)std::ceil(p->update_freq / p->measurement_freq);     // :91
    d->num_states = p->est_bias ? 15 : 9;                                           // :92
    d->measurement_step_delay = std::max((int32_t)(p->measurement_delay / d->dT_nom + 0.5), 1);  // :93
    for (int i = 0; i < 3; ++i) {
        d->Q[i] = p->Q_a[i];
        d->Q[3 + i] = p->Q_w[i];
        d->cov_init[i] = p->r_cov_init;
        d->cov_init[3 + i] = p->v_cov_init;
        d->cov_init[6 + i] = p->ang_cov_init;
        if (p->est_bias) {
            d->Q[6 + i] = p->Q_ab[i];
            d->Q[9 + i] = p->Q_wb[i];
            d->cov_init[9 + i] = p->ab_cov_init;
            d->cov_init[12 + i] = p->wb_cov_init;
        }
        d->R[i] = p->R_r[i];
        d->R[3 + i] = p->R_ang[i];
    }
    // quaternion_norm(q_vc) (:121, QH.cpp:61-73) and C_vc = q_vc.toRotationMatrix() (:122)
    double n = std::sqrt(p->q_vc[0] * p->q_vc[0] + p->q_vc[1] * p->q_vc[1] + p->q_vc[2] * p->q_vc[2] + p->q_vc[3] * p->q_vc[3]);
    if (!(n > 0.0)) return fail(QLE_ERR_INVALID, "q_vc has zero norm");
    for (int i = 0; i < 4; ++i) d->q_vc[i] = p->q_vc[i] / n;
    if (d->q_vc[3] < -0.75)
        for (int i = 0; i < 4; ++i) d->q_vc[i] = -d->q_vc[i];
    const double x = d->q_vc[0], y = d->q_vc[1], z = d->q_vc[2], w = d->q_vc[3];
    double* C = d->C_vc;
    C[0] = 1 - 2 * (y * y + z * z); C[1] = 2 * (x * y - w * z);     C[2] = 2 * (x * z + w * y);
    C[3] = 2 * (x * y + w * z);     C[4] = 1 - 2 * (x * x + z * z); C[5] = 2 * (y * z - w * x);
    C[6] = 2 * (x * z - w * y);     C[7] = 2 * (y * z + w * x);     C[8] = 1 - 2 * (x * x + y * y);
    return QLE_OK;
}

// ------------------------------------------------------------------ handle
struct qle_batch {
    int64_t B = 0;
    int32_t dtype = QLE_F32;
    int32_t device = 0;
    int32_t block = 256;
    int32_t split = 0;        // nt == 3: which workgroups keep their tiles cached (cached_workgroup() in ekf_kernels.hpp)
    int32_t nt_refresh = 0;   // > 0: nt == 1 and the state is <= 40 MiB: non-temporal stores, cached-store tick every nt_refresh ticks
    int32_t nt = 0;        // cache policy of the hot kernels' state accesses: 0 cached, 1 L2-sized scheme (effective_nt), 2 non-temporal, 3 split
    int64_t rows_max = 0;  // batches up to this size may use the rows-across-lanes kernel (ekf_rows.hpp)
    bool rows_forced = false;  // QLE_ROWS_MAX set: use it for every eligible tick (tests, experiments)
    size_t wsz = 4;
    qle_params pub;
    qle_derived der;
    DevParams<float> pf;
    DevParams<double> pd;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t Bp = 0;        // batch padded to whole 64-filter tiles
    // State storage: ring of C arrays of 144-word state records, slot = tick % C.  C = 1 for the
    // single-rate filter (in place); C = max step delay + 1 for the multirate filter, where the
    // ring is the history of EKF.hpp:62-64.  The state "now" is slot (tick-1) % C.
    void* ring = nullptr;
    int32_t C = 0;
    void* pfp = nullptr;   // [24 words] per-filter params, wave tiles
    bool pfp_on = false;
    bool aux = false;
    void* aux_accel = nullptr;  // AoS [B][3], compute dtype
    void* aux_obs = nullptr;    // AoS [B][7]
    void* tick_u = nullptr;     // one tick of inputs in device layout
    void* tick_z = nullptr;
    double* stage = nullptr;    // AoS fp64 staging, kStageFilters filters
    uint8_t* stage_mask = nullptr;
    unsigned long long* counter = nullptr;
    bool state_set = false;
    // device-side measurement gating (EKF.cpp:147-186)
    bool gating = false;
    int32_t* last_corr = nullptr;  // [B] index of each filter's last correcting tick, -1 = never
    uint8_t* flags = nullptr;      // [B] bit0 performed_correction, bit1 measurement consumed (last measurement tick)
    int64_t tick = 0;              // filter_update ticks executed so far (since the last origin shift)
    int64_t tick_origin = 0;       // ticks removed by origin shifts (reporting only)
    int64_t rebase_at = (int64_t)1 << 30;  // shift the tick origin when the counter reaches this (QLE_TICK_REBASE for tests)
    // multirate EKF (EKF.cpp:196-236, 251-264)
    bool mr = false;               // pub.multirate_ekf
    bool hist_dirty = true;        // state was overwritten: restart the history at the next tick
    int32_t* hist_first = nullptr; // [B] tick of each filter's oldest valid history entry
    int32_t* fresh_from = nullptr; // [B] tick of the entry written by the filter's last correction tick (entries between are stale)
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
    return o;
}
template <typename T> static const DevParams<T>& dev(const qle_batch* h);
template <> const DevParams<float>& dev<float>(const qle_batch* h) { return h->pf; }
template <> const DevParams<double>& dev<double>(const qle_batch* h) { return h->pd; }

static inline dim3 grid_for(const qle_batch* h, int block) { return dim3((unsigned)((h->B + block - 1) / block)); }

static inline size_t slot_bytes(const qle_batch* h) { return (size_t)kSW * (size_t)h->Bp * h->wsz; }
static inline int32_t slot_of(const qle_batch* h, int64_t tick)
{
    int64_t s = tick % h->C;
    return (int32_t)(s < 0 ? s + h->C : s);
}
// state after the last executed tick / state the next tick writes
static inline void* state_cur(const qle_batch* h) { return (char*)h->ring + slot_bytes(h) * (size_t)slot_of(h, h->tick - 1); }
static inline void* state_next(const qle_batch* h) { return (char*)h->ring + slot_bytes(h) * (size_t)slot_of(h, h->tick); }

static int check_handle(const qle_batch* h)
{
    if (!h) return fail(QLE_ERR_INVALID, "handle is null");
    (void)hipGetLastError();  // drop any stale sticky error of this thread: launches below check their own
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(QLE_ERR_HIP, "hipSetDevice(%d): %s", h->device, hipGetErrorString(e));
    return QLE_OK;
}

extern "C" int qle_set_params(qle_batch* h, const qle_params* p)
{
    QLE_TRY(check_handle(h));
    if (!p) return fail(QLE_ERR_INVALID, "params is null");
    qle_derived d;
    QLE_TRY(qle_params_derive(p, &d));
    // State ring: C = 1 (single-rate) or 2 x the largest reachable step delay + 1 (multirate, EKF.cpp:199-201).
    // Allocate first; the handle's parameters change only once everything needed exists.
    const bool mr = p->multirate_ekf != 0;
    int32_t C = 1;
    if (mr) {
        int32_t step_max = d.measurement_step_delay;
        if (p->dynamic_meas_delay) step_max = std::max((int32_t)(p->measurement_delay_max / d.dT_nom + 0.5), 1);
        C = 2 * step_max + 1;   // lazy history (k_step_mr): a measurement in a stale zone restarts from the entry before it
    }
    if (mr && !h->hist_first) {
        hipError_t e = hipMalloc((void**)&h->hist_first, sizeof(int32_t) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMalloc((void**)&h->fresh_from, sizeof(int32_t) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMalloc((void**)&h->stamp, sizeof(double) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMalloc((void**)&h->delay_cur, sizeof(double) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMemsetAsync(h->delay_cur, 0, sizeof(double) * (size_t)h->Bp, h->stream);
        if (e != hipSuccess) return fail(QLE_ERR_NOMEM, "multirate bookkeeping arrays: %s", hipGetErrorString(e));
    }
    if (C != h->C) {
        void* nr = nullptr;
        hipError_t e = hipMalloc(&nr, slot_bytes(h) * (size_t)C);
        if (e != hipSuccess) return fail(QLE_ERR_NOMEM, "hipMalloc of the state ring (%d slots x %lld filters): %s", C, (long long)h->Bp, hipGetErrorString(e));
        if (h->ring) {  // keep the current state: it moves to the slot the new ring assigns to tick-1
            int64_t sn = (h->tick - 1) % C;
            if (sn < 0) sn += C;
            e = hipMemcpyAsync((char*)nr + slot_bytes(h) * (size_t)sn, state_cur(h), slot_bytes(h), hipMemcpyDeviceToDevice, h->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e == hipSuccess) e = hipFree(h->ring);
        } else {
            e = hipMemsetAsync(nr, 0, slot_bytes(h) * (size_t)C, h->stream);
        }
        if (e != hipSuccess) { (void)hipFree(nr); return fail(QLE_ERR_HIP, "state ring setup: %s", hipGetErrorString(e)); }
        h->ring = nr;
        h->C = C;
    }
    h->pub = *p;
    h->der = d;
    h->pf = make_dev<float>(*p, d);
    h->pd = make_dev<double>(*p, d);
    h->mr = mr;
    if (mr) h->uniform_age = p->measurement_delay;
    h->hist_dirty = true;  // the multirate history restarts from the current state
    return QLE_OK;
}

extern "C" int qle_destroy(qle_batch* h)
{
    if (!h) return QLE_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* bufs[] = {h->ring, h->pfp, h->aux_accel, h->aux_obs, h->tick_u, h->tick_z, h->stage, h->stage_mask, h->counter, h->last_corr, h->flags, h->hist_first, h->fresh_from, h->stamp,
                    h->delay_cur};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return QLE_OK;
}

extern "C" int qle_create(qle_batch** out, int64_t batch, int32_t dtype, int32_t device, const qle_params* p)
{
    if (!out) return fail(QLE_ERR_INVALID, "out is null");
    *out = nullptr;
    if (batch <= 0) return fail(QLE_ERR_INVALID, "batch must be > 0 (got %lld)", (long long)batch);
    if (dtype != QLE_F32 && dtype != QLE_F64) return fail(QLE_ERR_INVALID, "dtype must be QLE_F32 or QLE_F64");
    if (!p) return fail(QLE_ERR_INVALID, "params is null");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(QLE_ERR_NO_DEVICE, "no HIP device available (%s); this engine has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(QLE_ERR_NO_DEVICE, "device %d out of range (have %d)", device, ndev);
    qle_batch* h = new (std::nothrow) qle_batch();
    if (!h) return fail(QLE_ERR_NOMEM, "host allocation failed");
    h->B = batch;
    h->Bp = padded_filters(batch);
    h->dtype = dtype;
    h->device = device;
    h->wsz = dtype == QLE_F32 ? 4 : 8;
    {   // Cache policy of the state accesses, from SUSTAINED rates on MI355X (profiles/r01_tuning.md section 5;
        // the input records are always read non-temporally):
        //   state <= 40 MiB (about the aggregate L2): non-temporal loads and stores with a cached-store tick every
        //     128 ticks that keeps the state allocated in the Infinity Cache (effective_nt below);
        //   up to 48 MiB: non-temporal loads, cached stores;
        //   up to 300 MiB: cached loads and stores (Infinity-Cache resident from tick to tick);
        //   beyond: "split" -- a fixed ~216 MiB of the state stays cached, the rest streams non-temporally, so the
        //     Infinity Cache and HBM serve the tick side by side (+25 % at 432 and 576 MiB, +17 % at 1.1 GB, +8 % at
        //     2.3 GB over streaming everything).
        // QLE_NT=0|1|2|3 overrides (0 cached, 1 the L2-sized scheme, 2 non-temporal loads+stores, 3 split with
        // QLE_SPLIT=-k: k of every 64 workgroup groups cached).
        const double state_mib = (double)kSW * (double)h->Bp * (double)h->wsz / (1024.0 * 1024.0);
        h->nt = state_mib <= 48.0 ? 1 : (state_mib <= 300.0 ? 0 : 3);
        h->nt_refresh = state_mib <= 40.0 ? 128 : 0;   // at 45 MiB the refresh scheme loses (18.2 vs 16.3 us), plain policy 1 wins
        if (const char* s = std::getenv("QLE_REFRESH")) h->nt_refresh = std::max(0, std::atoi(s));
        // larger than the cache: keep about 216 MiB of the state cached (k of every 64 workgroup groups, interleaved
        // over the batch and spread evenly over the XCDs) and stream the rest
        const int k64 = (int)std::lround(64.0 * 216.0 / std::max(state_mib, 1.0));
        h->split = -std::min(63, std::max(1, k64));
        if (state_mib > 64.0 * 240.0) h->nt = 2;       // even 1/64 of it would not fit: stream everything
        if (const char* s = std::getenv("QLE_NT")) h->nt = std::min(3, std::max(0, std::atoi(s)));
        if (const char* s = std::getenv("QLE_SPLIT")) h->split = std::atoi(s);
    }
    // Rows-across-lanes kernel (16 lanes per filter).  Measured (profiles/r01_tuning.md section 3): its per-wave
    // instruction stream is as long as the one-lane-per-filter kernels', so it only pays where those spill:
    // fp64 ticks that carry corrections, up to ~6k filters (BASELINE cfg 2: 15 vs 23 us per tick).
    h->rows_max = dtype == QLE_F64 ? 6144 : 0;
    if (const char* s = std::getenv("QLE_ROWS_MAX")) { h->rows_max = std::atoll(s); h->rows_forced = true; }
    if (const char* s = std::getenv("QLE_TICK_REBASE")) {
        const long long v = std::atoll(s);
        if (v >= 16) h->rebase_at = v;
    }
    // Workgroup size of the hot kernels: 256 threads (4 tiles) up to 131 072 filters; one wave per workgroup beyond
    // (finer dispatch: +2-3 % at 262 144 and 524 288 filters, +1.5 % at 1-2 M, level below; profiles/r01_tuning.md section 4).
    h->block = batch >= 262144 ? 64 : kBlock;
    if (const char* s = std::getenv("QLE_BLOCK")) {
        int b = std::atoi(s);
        if (b == 64 || b == 128 || b == 256) h->block = b;
    }
    int rc = QLE_OK;
    auto bail = [&](int code) { qle_destroy(h); return code; };
    if (hipSetDevice(device) != hipSuccess) return bail(fail(QLE_ERR_HIP, "hipSetDevice(%d) failed", device));
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(fail(QLE_ERR_HIP, "hipStreamCreate failed"));
    if ((rc = qle_set_params(h, p)) != QLE_OK) return bail(rc);
#define ALLOC(ptr, bytes)                                                                               \
    do {                                                                                                \
        hipError_t ea_ = hipMalloc((void**)&(ptr), (bytes));                                            \
        if (ea_ != hipSuccess) return bail(fail(QLE_ERR_NOMEM, "hipMalloc(%zu B) for %s: %s", (size_t)(bytes), #ptr, hipGetErrorString(ea_))); \
    } while (0)
    if (hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) return bail(fail(QLE_ERR_HIP, "hipEventCreate failed"));
    const size_t B = (size_t)h->Bp, w = h->wsz;
    ALLOC(h->tick_u, kUW * B * w);
    ALLOC(h->tick_z, kZW * B * w);
    ALLOC(h->stage, (size_t)kStageDoubles * sizeof(double));
    ALLOC(h->stage_mask, (size_t)kStageFilters);
    ALLOC(h->counter, sizeof(unsigned long long));
#undef ALLOC
    *out = h;
    return QLE_OK;
}

extern "C" int64_t qle_batch_size(const qle_batch* h) { return h ? h->B : 0; }
extern "C" int32_t qle_dtype(const qle_batch* h) { return h ? h->dtype : -1; }
extern "C" int32_t qle_num_states(const qle_batch* h) { return h ? h->der.num_states : 0; }

extern "C" int64_t qle_algorithmic_bytes(const qle_batch* h, int32_t kind)
{   // packed P, SoA, one streamed tick
    if (!h) return 0;
    // a multirate predict tick also writes the IMU sample into the new history entry (+6 words, +2 pad)
    const int64_t wr = (16 + 120) + ((h->mr && kind == 0) ? 8 : 0);
    int64_t words = kind == 0 ? (16 + 120 + 6) + wr : kind == 1 ? (16 + 120 + 6 + 7) + wr : (16 + 120 + 7) + wr;
    if (h->pfp_on) words += kFW;
    return words * (int64_t)h->wsz * h->B;
}

extern "C" int qle_synchronize(qle_batch* h)
{
    QLE_TRY(check_handle(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return QLE_OK;
}
extern "C" int qle_timer_begin(qle_batch* h)
{
    QLE_TRY(check_handle(h));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return QLE_OK;
}
extern "C" int qle_timer_end(qle_batch* h, float* ms)
{
    QLE_TRY(check_handle(h));
    if (!ms) return fail(QLE_ERR_INVALID, "elapsed_ms is null");
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return QLE_OK;
}

// --------------------------------------------------- staging (not hot path)
// Chunked AoS fp64 host -> device tiles.  `W` words per filter taken from a
// host row of `stride` doubles go to words [w0, w0+W) of the WT-word record.
template <typename T>
static int pack_rows(qle_batch* h, const double* host, int stride, int W, void* dst, int WT, int w0)
{
    const int64_t chunk = std::min<int64_t>(kStageFilters, kStageDoubles / std::max(stride, 1));
    for (int64_t i0 = 0; i0 < h->B; i0 += chunk) {
        const int64_t n = std::min(chunk, h->B - i0);
        HIP_TRY(hipMemcpyAsync(h->stage, host + i0 * stride, (size_t)n * stride * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL((k_pack_off<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->stage, stride, W,
                           (T*)dst, WT, w0, i0, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));  // staging buffer is reused by the next chunk
    }
    return QLE_OK;
}
template <typename T>
static int unpack_rows(qle_batch* h, const void* src, int stride, int W, double* host, int WT, int w0)
{
    const int64_t chunk = std::min<int64_t>(kStageFilters, kStageDoubles / std::max(stride, 1));
    for (int64_t i0 = 0; i0 < h->B; i0 += chunk) {
        const int64_t n = std::min(chunk, h->B - i0);
        hipLaunchKernelGGL((k_unpack_off<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const T*)src, stride, W, h->stage,
                           WT, w0, i0, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host + i0 * stride, h->stage, (size_t)n * stride * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return QLE_OK;
}
template <typename T>
static int pack_z(qle_batch* h, const double* z, const uint8_t* mask, void* dst)
{
    for (int64_t i0 = 0; i0 < h->B; i0 += kStageFilters) {
        const int64_t n = std::min(kStageFilters, h->B - i0);
        if (z) HIP_TRY(hipMemcpyAsync(h->stage, z + i0 * 7, (size_t)n * 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (mask) HIP_TRY(hipMemcpyAsync(h->stage_mask, mask + i0, (size_t)n, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL((k_pack_z_off<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, z ? (const double*)h->stage : nullptr,
                           mask ? (const uint8_t*)h->stage_mask : nullptr, (T*)dst, i0, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return QLE_OK;
}
template <typename T>
static int pack_P(qle_batch* h, const double* P, void* dst)
{
    const int n = h->der.num_states;
    for (int64_t i0 = 0; i0 < h->B; i0 += kStageFilters) {
        const int64_t m = std::min(kStageFilters, h->B - i0);
        HIP_TRY(hipMemcpyAsync(h->stage, P + i0 * n * n, (size_t)m * n * n * sizeof(double), hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL((k_pack_P_off<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->stage, n, (T*)dst,
                           i0, m);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return QLE_OK;
}
template <typename T>
static int unpack_P(qle_batch* h, const void* src, double* P)
{
    const int n = h->der.num_states;
    for (int64_t i0 = 0; i0 < h->B; i0 += kStageFilters) {
        const int64_t m = std::min(kStageFilters, h->B - i0);
        hipLaunchKernelGGL((k_unpack_P_off<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, (const T*)src, n, h->stage, i0, m);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(P + i0 * n * n, h->stage, (size_t)m * n * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return QLE_OK;
}

#define BY_DTYPE(h, FN, ...) ((h)->dtype == QLE_F32 ? FN<float>(__VA_ARGS__) : FN<double>(__VA_ARGS__))

extern "C" int qle_set_state(qle_batch* h, const double* x, const double* P)
{
    QLE_TRY(check_handle(h));
    if (!x || !P) return fail(QLE_ERR_INVALID, "x and P must be non-null");
    QLE_TRY(BY_DTYPE(h, pack_rows, h, x, kXW, kXW, state_cur(h), kSW, 0));
    QLE_TRY(BY_DTYPE(h, pack_P, h, P, state_cur(h)));
    h->state_set = true;
    h->hist_dirty = true;
    return QLE_OK;
}
extern "C" int qle_get_state(qle_batch* h, double* x, double* P)
{
    QLE_TRY(check_handle(h));
    if (x) QLE_TRY(BY_DTYPE(h, unpack_rows, h, state_cur(h), kXW, kXW, x, kSW, 0));
    if (P) QLE_TRY(BY_DTYPE(h, unpack_P, h, state_cur(h), P));
    return QLE_OK;
}

extern "C" int qle_set_filter_params(qle_batch* h, const double* pfp)
{
    QLE_TRY(check_handle(h));
    if (!pfp) { h->pfp_on = false; return QLE_OK; }
    if (!h->pfp) HIP_TRY(hipMalloc(&h->pfp, kFW * (size_t)h->Bp * h->wsz));
    QLE_TRY(BY_DTYPE(h, pack_rows, h, pfp, kFW, kFW, h->pfp, kFW, 0));
    h->pfp_on = true;
    return QLE_OK;
}

extern "C" int qle_enable_aux(qle_batch* h, int32_t on)
{
    QLE_TRY(check_handle(h));
    if (on && !h->aux_accel) {
        HIP_TRY(hipMalloc(&h->aux_accel, 3 * (size_t)h->B * h->wsz));
        HIP_TRY(hipMalloc(&h->aux_obs, 7 * (size_t)h->B * h->wsz));
        HIP_TRY(hipMemsetAsync(h->aux_accel, 0, 3 * (size_t)h->B * h->wsz, h->stream));
        HIP_TRY(hipMemsetAsync(h->aux_obs, 0, 7 * (size_t)h->B * h->wsz, h->stream));
    }
    h->aux = on != 0;
    return QLE_OK;
}
template <typename T>
static int get_aux_t(qle_batch* h, double* accel, double* obs)
{
    std::vector<T> tmp((size_t)h->B * 7);
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (accel) {
        HIP_TRY(hipMemcpy(tmp.data(), h->aux_accel, (size_t)h->B * 3 * sizeof(T), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < h->B * 3; ++k) accel[k] = (double)tmp[k];
    }
    if (obs) {
        HIP_TRY(hipMemcpy(tmp.data(), h->aux_obs, (size_t)h->B * 7 * sizeof(T), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < h->B * 7; ++k) obs[k] = (double)tmp[k];
    }
    return QLE_OK;
}
extern "C" int qle_get_aux(qle_batch* h, double* accel_rel, double* obs)
{
    QLE_TRY(check_handle(h));
    if (!h->aux_accel) return fail(QLE_ERR_STATE, "aux outputs are not enabled (qle_enable_aux)");
    QLE_GUARD_BEGIN
    return BY_DTYPE(h, get_aux_t, h, accel_rel, obs);
    QLE_GUARD_END
}

// -------------------------------------------------------------- hot launches
// Restart the multirate history with the single entry "state now" (EKF.cpp:337-339).
static int mr_prepare(qle_batch* h)
{
    if (h->mr && h->hist_dirty) {
        hipLaunchKernelGGL(k_fill_i32, grid_for(h, 256), dim3(256), 0, h->stream, h->hist_first, (int32_t)(h->tick - 1), h->B);
        hipLaunchKernelGGL(k_fill_i32, grid_for(h, 256), dim3(256), 0, h->stream, h->fresh_from, (int32_t)(h->tick - 1), h->B);
        HIP_TRY(hipGetLastError());
    }
    h->hist_dirty = false;
    return QLE_OK;
}

// The rows-across-lanes kernel covers the plain single-rate tick (no device gating, no multirate history,
// no side outputs); everything else runs on the one-lane-per-filter kernels.
static inline bool use_rows(const qle_batch* h, bool has_meas)
{
    return h->B <= h->rows_max && (has_meas || h->rows_forced) && !h->mr && !h->gating && !h->aux;
}

template <typename T>
static int launch_rows(qle_batch* h, const void* u, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const int64_t lanes = h->B * kRowLanes;
    const dim3 g((unsigned)((lanes + kBlock - 1) / kBlock)), b(kBlock);
    const T* pfp = (const T*)h->pfp;
#define QLE_ROWS(D, F) hipLaunchKernelGGL((k_rows<T, D, F>), g, b, 0, h->stream, p, (T*)state_cur(h), (const T*)u, (const T*)z, pfp, h->B)
    if (h->pub.direct_orien_method) { if (h->pfp_on) QLE_ROWS(true, true); else QLE_ROWS(true, false); }
    else { if (h->pfp_on) QLE_ROWS(false, true); else QLE_ROWS(false, false); }
#undef QLE_ROWS
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}

// Kernel cache policy of this tick.  For an L2-sized state (h->nt == 1) the fastest sustained scheme measured
// (profiles/r01_tuning.md section 5) is: non-temporal loads AND stores -- the stores update the lines the state
// already has in the Infinity Cache and leave no dirty L2 to flush at the kernel boundary -- plus one tick with
// cached stores every nt_refresh (128) ticks, which re-allocates the state in the Infinity Cache.  Without the refresh the
// state drifts out of the cache within ~3 000 ticks and every tick streams from HBM (9.3 -> 10.9 us per predict
// at 65 536 filters); cached stores on every tick cost 9.9 us.  QLE_REFRESH=R overrides (0: cached stores always).
static inline int effective_nt(const qle_batch* h)
{
    if (h->nt == 1 && h->nt_refresh > 0) return (h->tick % h->nt_refresh) == 0 ? 1 : 2;
    if (h->nt == 3 && h->mr) return 2;   // the split policy is for the in-place single-rate state
    return h->nt;
}

// prediction_step from `src` into `dst`; keep_u: the record also stores the IMU sample (multirate history).
template <typename T>
static int launch_predict_sd(qle_batch* h, const void* u, const void* src, void* dst, bool keep_u)
{
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T* acc = h->aux ? (T*)h->aux_accel : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
#define QLE_PRED(F, N, M) hipLaunchKernelGGL((k_predict<T, F, N, M>), g, b, 0, h->stream, p, (const T*)src, (T*)dst, (const T*)u, pfp, acc, h->B, h->split)
#define QLE_PRED_N(N, M) do { if (h->pfp_on) QLE_PRED(true, N, M); else QLE_PRED(false, N, M); } while (0)
    const int nt = effective_nt(h);
#define QLE_PRED_M(M) do { if (nt == 2) QLE_PRED_N(2, M); else if (nt == 1) QLE_PRED_N(1, M); else QLE_PRED_N(0, M); } while (0)
    if (keep_u) QLE_PRED_M(true);
    else if (nt == 3) QLE_PRED_N(3, false);
    else QLE_PRED_M(false);
#undef QLE_PRED_M
#undef QLE_PRED_N
#undef QLE_PRED
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
// the predict of one filter tick: slot n-1 -> slot n (the same slot when C == 1)
template <typename T>
static int launch_predict(qle_batch* h, const void* u)
{
    if (use_rows(h, false)) return launch_rows<T>(h, u, nullptr);
    QLE_TRY(mr_prepare(h));
    return launch_predict_sd<T>(h, u, state_cur(h), state_next(h), h->mr);
}

static GateParams make_gate(const qle_batch* h)
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

template <typename T, bool DIRECT, bool GATE>
static int launch_step_dg(qle_batch* h, const void* u, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const GateParams gp = make_gate(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T *st = (T*)state_cur(h), *acc = h->aux ? (T*)h->aux_accel : (T*)nullptr, *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
#define QLE_STEP_LAUNCH(F, N) hipLaunchKernelGGL((k_step<T, DIRECT, F, GATE, N>), g, b, 0, h->stream, p, gp, st, (const T*)u, (const T*)z, pfp, acc, obs, h->last_corr, h->flags, h->B, h->split)
#define QLE_STEP_N(N) do { if (h->pfp_on) QLE_STEP_LAUNCH(true, N); else QLE_STEP_LAUNCH(false, N); } while (0)
    const int nt = effective_nt(h);
    if (nt == 3) QLE_STEP_N(3); else if (nt == 2) QLE_STEP_N(2); else if (nt == 1) QLE_STEP_N(1); else QLE_STEP_N(0);
#undef QLE_STEP_N
#undef QLE_STEP_LAUNCH
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
static MrParams make_mr(const qle_batch* h)
{
    MrParams m;
    std::memset(&m, 0, sizeof(m));
    m.C = h->C;
    m.tick = (int32_t)h->tick;
    m.fixed_step = h->der.measurement_step_delay;
    m.dynamic = h->pub.dynamic_meas_delay;
    m.gate = h->gating ? 1 : 0;
    m.slot_words = (int64_t)kSW * h->Bp;
    m.dT = h->der.dT_nom;
    m.offset = h->pub.dyn_measurement_delay_offset;
    m.delay_max = h->pub.measurement_delay_max;
    m.t_curr = h->t_curr;
    m.uniform_age = h->uniform_age;
    return m;
}

// One multirate tick that carries tag poses (predict-only multirate ticks go through launch_predict).
template <typename T>
static int launch_step_mr(qle_batch* h, const void* u, const void* z)
{
    QLE_TRY(mr_prepare(h));
    const DevParams<T>& p = dev<T>(h);
    const GateParams gp = make_gate(h);
    const MrParams m = make_mr(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T *acc = h->aux ? (T*)h->aux_accel : (T*)nullptr, *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
    const double* stamp = (h->have_stamps && h->pub.dynamic_meas_delay) ? h->stamp : nullptr;
#define QLE_MR_LAUNCH(D, F) hipLaunchKernelGGL((k_step_mr<T, D, F>), g, b, 0, h->stream, p, gp, m, (T*)h->ring, (const T*)u, (const T*)z, pfp, stamp, acc, obs, h->hist_first, h->fresh_from, h->last_corr, h->flags, h->delay_cur, h->B)
    if (h->pub.direct_orien_method) { if (h->pfp_on) QLE_MR_LAUNCH(true, true); else QLE_MR_LAUNCH(true, false); }
    else { if (h->pfp_on) QLE_MR_LAUNCH(false, true); else QLE_MR_LAUNCH(false, false); }
#undef QLE_MR_LAUNCH
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}

template <typename T>
static int launch_step(qle_batch* h, const void* u, const void* z)
{
    if (use_rows(h, true)) return launch_rows<T>(h, u, z);
    if (h->mr) return launch_step_mr<T>(h, u, z);
    if (h->pub.direct_orien_method) return h->gating ? launch_step_dg<T, true, true>(h, u, z) : launch_step_dg<T, true, false>(h, u, z);
    return h->gating ? launch_step_dg<T, false, true>(h, u, z) : launch_step_dg<T, false, false>(h, u, z);
}

template <typename T, bool DIRECT>
static int launch_update_d(qle_batch* h, const void* z)
{
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    T *st = (T*)state_cur(h), *obs = h->aux ? (T*)h->aux_obs : (T*)nullptr;
    const T* pfp = (const T*)h->pfp;
    if (h->pfp_on) hipLaunchKernelGGL((k_update<T, DIRECT, true>), g, b, 0, h->stream, p, st, (const T*)z, pfp, obs, h->B);
    else hipLaunchKernelGGL((k_update<T, DIRECT, false>), g, b, 0, h->stream, p, st, (const T*)z, pfp, obs, h->B);
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
template <typename T>
static int launch_update(qle_batch* h, const void* z)
{
    return h->pub.direct_orien_method ? launch_update_d<T, true>(h, z) : launch_update_d<T, false>(h, z);
}

// One filter_update tick has been launched.  Tick indices are 32-bit on the device
// (last_corr, ring slot = tick % C); long before they could wrap, shift the origin by a
// multiple of the ring capacity so that slots and differences are unchanged.
static int advance_tick(qle_batch* h)
{
    h->tick++;
    if (h->tick >= h->rebase_at) {
        const int64_t C = h->C > 0 ? h->C : 1;
        const int64_t shift = ((h->rebase_at / 2) / C) * C;  // a multiple of C: ring slots (tick % C) are unchanged
        if (shift <= 0) return QLE_OK;
        int32_t* arrs[3] = {h->last_corr, h->hist_first, h->fresh_from};
        for (int32_t* a : arrs)
            if (a) {
                hipLaunchKernelGGL(k_rebase_ticks, grid_for(h, 256), dim3(256), 0, h->stream, a, (int32_t)shift, h->B);
                HIP_TRY(hipGetLastError());
            }
        h->tick -= shift;
        h->tick_origin += shift;
    }
    return QLE_OK;
}

static int need_state(const qle_batch* h)
{
    if (!h->state_set) return fail(QLE_ERR_STATE, "state not initialised: call qle_set_state or qle_initialize_state first (EKF.cpp:129-130)");
    return QLE_OK;
}

extern "C" int qle_predict(qle_batch* h, const double* u)
{
    QLE_TRY(check_handle(h));
    QLE_TRY(need_state(h));
    if (!u) return fail(QLE_ERR_INVALID, "u is null");
    QLE_TRY(BY_DTYPE(h, pack_rows, h, u, kUW, kUW, h->tick_u, kUW, 0));
    h->hist_dirty = true;  // a bare prediction_step is not a filter tick: the multirate history restarts
    return BY_DTYPE(h, launch_predict_sd, h, h->tick_u, state_cur(h), state_cur(h), false);
}
extern "C" int qle_update(qle_batch* h, const double* z, const uint8_t* mask)
{
    QLE_TRY(check_handle(h));
    QLE_TRY(need_state(h));
    if (!z) return fail(QLE_ERR_INVALID, "z is null");
    QLE_TRY(BY_DTYPE(h, pack_z, h, z, mask, h->tick_z));
    h->hist_dirty = true;
    return BY_DTYPE(h, launch_update, h, h->tick_z);
}
extern "C" int qle_step(qle_batch* h, const double* u, const double* z, const uint8_t* mask)
{
    QLE_TRY(check_handle(h));
    QLE_TRY(need_state(h));
    if (!u) return fail(QLE_ERR_INVALID, "u is null");
    QLE_TRY(BY_DTYPE(h, pack_rows, h, u, kUW, kUW, h->tick_u, kUW, 0));
    if (!z) {
        QLE_TRY(BY_DTYPE(h, launch_predict, h, h->tick_u));
    } else {
        QLE_TRY(BY_DTYPE(h, pack_z, h, z, mask, h->tick_z));
        QLE_TRY(BY_DTYPE(h, launch_step, h, h->tick_u, h->tick_z));
    }
    return advance_tick(h);
}

// ---- device-side gating: the full single-rate filter_update decision logic ----
extern "C" int qle_enable_gating(qle_batch* h, int32_t on)
{
    QLE_TRY(check_handle(h));
    if (on && !h->last_corr) {
        HIP_TRY(hipMalloc((void**)&h->last_corr, sizeof(int32_t) * (size_t)h->Bp));
        HIP_TRY(hipMalloc((void**)&h->flags, (size_t)h->Bp));
        HIP_TRY(hipMemsetAsync(h->last_corr, 0xFF, sizeof(int32_t) * (size_t)h->Bp, h->stream));  // -1
        HIP_TRY(hipMemsetAsync(h->flags, 0, (size_t)h->Bp, h->stream));
    }
    h->gating = on != 0;
    return QLE_OK;
}

extern "C" int qle_filter_update(qle_batch* h, const double* u, const double* z, const uint8_t* measurement_ready)
{
    QLE_TRY(check_handle(h));
    if (!h->gating) return fail(QLE_ERR_STATE, "gating is not enabled (qle_enable_gating)");
    if (z == nullptr && h->flags) HIP_TRY(hipMemsetAsync(h->flags, 0, (size_t)h->Bp, h->stream));  // performed_correction = false
    h->have_stamps = false;
    return qle_step(h, u, z, measurement_ready);
}

extern "C" int qle_filter_update_stamped(qle_batch* h, const double* u, const double* z, const uint8_t* measurement_ready, double t_curr,
                                         const double* apriltag_time)
{
    QLE_TRY(check_handle(h));
    if (!h->gating) return fail(QLE_ERR_STATE, "gating is not enabled (qle_enable_gating)");
    if (z == nullptr && h->flags) HIP_TRY(hipMemsetAsync(h->flags, 0, (size_t)h->Bp, h->stream));
    h->t_curr = t_curr;
    h->have_stamps = false;
    if (h->mr && apriltag_time && z) {
        HIP_TRY(hipMemcpyAsync(h->stamp, apriltag_time, sizeof(double) * (size_t)h->B, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        h->have_stamps = true;
    }
    return qle_step(h, u, z, measurement_ready);
}

extern "C" int qle_get_measurement_delay(qle_batch* h, double* measurement_delay_curr)
{
    QLE_TRY(check_handle(h));
    if (!measurement_delay_curr) return fail(QLE_ERR_INVALID, "output is null");
    if (!h->mr || !h->pub.dynamic_meas_delay) {  // fixed delay (EKF.cpp:199)
        for (int64_t i = 0; i < h->B; ++i) measurement_delay_curr[i] = h->pub.measurement_delay;
        return QLE_OK;
    }
    HIP_TRY(hipMemcpyAsync(measurement_delay_curr, h->delay_cur, sizeof(double) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return QLE_OK;
}

extern "C" int qle_set_uniform_measurement_age(qle_batch* h, double seconds)
{
    QLE_TRY(check_handle(h));
    h->uniform_age = seconds;
    return QLE_OK;
}

extern "C" int qle_get_tick_flags(qle_batch* h, uint8_t* performed_correction, uint8_t* consumed, int32_t* upds_since_correction)
{
    QLE_TRY(check_handle(h));
    if (!h->last_corr) return fail(QLE_ERR_STATE, "gating is not enabled (qle_enable_gating)");
    QLE_GUARD_BEGIN
    std::vector<uint8_t> f((size_t)h->B);
    std::vector<int32_t> lc((size_t)h->B);
    HIP_TRY(hipMemcpyAsync(f.data(), h->flags, (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(lc.data(), h->last_corr, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int64_t i = 0; i < h->B; ++i) {
        if (performed_correction) performed_correction[i] = f[(size_t)i] & 1;
        if (consumed) consumed[i] = (f[(size_t)i] >> 1) & 1;
        if (upds_since_correction) upds_since_correction[i] = (int32_t)(h->tick - 1 - lc[(size_t)i]);  // EKF.cpp:292-299
    }
    return QLE_OK;
    QLE_GUARD_END
}

template <typename T>
static int seed_t(qle_batch* h, int reinit)
{
    const qle_derived& d = h->der;
    hipLaunchKernelGGL((k_seed<T>), grid_for(h, 256), dim3(256), 0, h->stream, dev<T>(h), (const T*)h->tick_z, (T*)state_cur(h),
                       (T)d.cov_init[0], (T)d.cov_init[3], (T)d.cov_init[6], (T)d.cov_init[9], (T)d.cov_init[12], reinit, h->B);
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
extern "C" int qle_initialize_state(qle_batch* h, const double* z, int32_t reinit_bias)
{
    QLE_TRY(check_handle(h));
    if (!z) return fail(QLE_ERR_INVALID, "z is null");
    QLE_TRY(BY_DTYPE(h, pack_z, h, z, (const uint8_t*)nullptr, h->tick_z));
    QLE_TRY(BY_DTYPE(h, seed_t, h, reinit_bias));
    h->state_set = true;
    h->hist_dirty = true;
    return QLE_OK;
}

// ---------------------------------------------------------------- reporting
template <typename T>
static int report_t(qle_batch* h, double* pose, double* cov, double* vel, double* bias)
{
    // staged per chunk: 7 + 36 + 3 + 6 = 52 doubles per filter
    for (int64_t i0 = 0; i0 < h->B; i0 += kStageFilters) {
        const int64_t n = std::min(kStageFilters, h->B - i0);
        double* s_pose = h->stage;
        double* s_cov = s_pose + n * 7;
        double* s_vel = s_cov + n * 36;
        double* s_bias = s_vel + n * 3;
        hipLaunchKernelGGL((k_report_off<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, dev<T>(h), (const T*)state_cur(h),
                           h->pfp_on ? (const T*)h->pfp : (const T*)nullptr, s_pose, s_cov, s_vel, s_bias, i0, n);
        HIP_TRY(hipGetLastError());
        if (pose) HIP_TRY(hipMemcpyAsync(pose + i0 * 7, s_pose, (size_t)n * 7 * 8, hipMemcpyDeviceToHost, h->stream));
        if (cov) HIP_TRY(hipMemcpyAsync(cov + i0 * 36, s_cov, (size_t)n * 36 * 8, hipMemcpyDeviceToHost, h->stream));
        if (vel) HIP_TRY(hipMemcpyAsync(vel + i0 * 3, s_vel, (size_t)n * 3 * 8, hipMemcpyDeviceToHost, h->stream));
        if (bias) HIP_TRY(hipMemcpyAsync(bias + i0 * 6, s_bias, (size_t)n * 6 * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return QLE_OK;
}
extern "C" int qle_get_report(qle_batch* h, double* pose, double* pose_cov, double* vel, double* bias)
{
    QLE_TRY(check_handle(h));
    return BY_DTYPE(h, report_t, h, pose, pose_cov, vel, bias);
}

template <typename T>
static int nonfinite_t(qle_batch* h)
{
    hipLaunchKernelGGL((k_count_nonfinite<T>), grid_for(h, 256), dim3(256), 0, h->stream, (const T*)state_cur(h), h->counter, h->B);
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
extern "C" int qle_count_nonfinite(qle_batch* h, int64_t* count)
{
    QLE_TRY(check_handle(h));
    if (!count) return fail(QLE_ERR_INVALID, "count is null");
    HIP_TRY(hipMemsetAsync(h->counter, 0, sizeof(unsigned long long), h->stream));
    QLE_TRY(BY_DTYPE(h, nonfinite_t, h));
    unsigned long long c = 0;
    HIP_TRY(hipMemcpyAsync(&c, h->counter, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *count = (int64_t)c;
    return QLE_OK;
}

// ------------------------------------------------- device-resident sequences
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" int qle_inputs_destroy(qle_inputs* in)
{
    if (!in) return QLE_OK;
    (void)hipSetDevice(in->device);
    void* bufs[] = {in->u, in->z, in->truth, in->truth_bias, in->d_slot};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    delete in;
    return QLE_OK;
}

extern "C" int qle_inputs_create(qle_batch* h, int64_t n_ticks, const uint8_t* tick_has_meas, qle_inputs** out)
{
    QLE_TRY(check_handle(h));
    if (!out) return fail(QLE_ERR_INVALID, "out is null");
    *out = nullptr;
    if (n_ticks <= 0) return fail(QLE_ERR_INVALID, "n_ticks must be > 0");
    qle_inputs* in = new (std::nothrow) qle_inputs();
    if (!in) return fail(QLE_ERR_NOMEM, "host allocation failed");
    in->h = h;
    in->device = h->device;
    in->T = n_ticks;
    try {
        in->slot.assign((size_t)n_ticks, -1);
    } catch (...) {
        delete in;
        return fail(QLE_ERR_NOMEM, "host allocation of the slot table for %lld ticks failed", (long long)n_ticks);
    }
    for (int64_t t = 0; t < n_ticks; ++t)
        if (tick_has_meas && tick_has_meas[t]) in->slot[(size_t)t] = (int32_t)in->n_slots++;
    in->pitch_u = align_up(kUW * (size_t)h->Bp * h->wsz, 256);
    in->pitch_z = align_up(kZW * (size_t)h->Bp * h->wsz, 256);
    hipError_t e = hipMalloc(&in->u, in->pitch_u * (size_t)n_ticks);
    if (e == hipSuccess && in->n_slots) e = hipMalloc(&in->z, in->pitch_z * (size_t)in->n_slots);
    if (e == hipSuccess) e = hipMalloc(&in->truth, (size_t)h->B * 7 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&in->truth_bias, (size_t)h->B * 6 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&in->d_slot, sizeof(int32_t) * (size_t)n_ticks);
    if (e == hipSuccess) e = hipMemcpy(in->d_slot, in->slot.data(), sizeof(int32_t) * (size_t)n_ticks, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        qle_inputs_destroy(in);
        return fail(QLE_ERR_NOMEM, "hipMalloc for %lld ticks of inputs: %s", (long long)n_ticks, hipGetErrorString(e));
    }
    *out = in;
    return QLE_OK;
}

static int check_tick(const qle_inputs* in, int64_t t)
{
    if (!in) return fail(QLE_ERR_INVALID, "inputs is null");
    if (t < 0 || t >= in->T) return fail(QLE_ERR_INVALID, "tick %lld out of range [0,%lld)", (long long)t, (long long)in->T);
    return QLE_OK;
}
static inline void* u_at(const qle_inputs* in, int64_t t) { return (char*)in->u + in->pitch_u * (size_t)t; }
static inline void* z_at(const qle_inputs* in, int32_t s) { return (char*)in->z + in->pitch_z * (size_t)s; }

extern "C" int qle_inputs_upload_tick(qle_inputs* in, int64_t t, const double* u, const double* z, const uint8_t* mask)
{
    QLE_TRY(check_tick(in, t));
    qle_batch* h = in->h;
    QLE_TRY(check_handle(h));
    if (!u) return fail(QLE_ERR_INVALID, "u is null");
    QLE_TRY(BY_DTYPE(h, pack_rows, h, u, kUW, kUW, u_at(in, t), kUW, 0));
    const int32_t s = in->slot[(size_t)t];
    if (s >= 0) {
        if (!z) return fail(QLE_ERR_INVALID, "tick %lld has a measurement slot but z is null", (long long)t);
        QLE_TRY(BY_DTYPE(h, pack_z, h, z, mask, z_at(in, s)));
    } else if (z) {
        return fail(QLE_ERR_INVALID, "tick %lld has no measurement slot", (long long)t);
    }
    return QLE_OK;
}

template <typename T>
static int unpack_z(qle_batch* h, const void* src, double* z, uint8_t* mask)
{
    for (int64_t i0 = 0; i0 < h->B; i0 += kStageFilters) {
        const int64_t n = std::min(kStageFilters, h->B - i0);
        hipLaunchKernelGGL((k_unpack_z_off<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const T*)src, h->stage, h->stage_mask,
                           i0, n);
        HIP_TRY(hipGetLastError());
        if (z) HIP_TRY(hipMemcpyAsync(z + i0 * 7, h->stage, (size_t)n * 7 * 8, hipMemcpyDeviceToHost, h->stream));
        if (mask) HIP_TRY(hipMemcpyAsync(mask + i0, h->stage_mask, (size_t)n, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return QLE_OK;
}
extern "C" int qle_inputs_download_tick(qle_inputs* in, int64_t t, double* u, double* z, uint8_t* mask)
{
    QLE_TRY(check_tick(in, t));
    qle_batch* h = in->h;
    QLE_TRY(check_handle(h));
    if (u) QLE_TRY(BY_DTYPE(h, unpack_rows, h, u_at(in, t), kUW, kUW, u, kUW, 0));
    const int32_t s = in->slot[(size_t)t];
    if (s >= 0 && (z || mask)) QLE_TRY(BY_DTYPE(h, unpack_z, h, z_at(in, s), z, mask));
    if (s < 0 && mask) std::memset(mask, 0, (size_t)h->B);
    return QLE_OK;
}

extern "C" int qle_run(qle_batch* h, const qle_inputs* in, int64_t t0, int64_t n)
{
    QLE_TRY(check_handle(h));
    QLE_TRY(need_state(h));
    if (!in || in->h != h) return fail(QLE_ERR_INVALID, "inputs do not belong to this handle");
    if (t0 < 0 || n < 0) return fail(QLE_ERR_INVALID, "t0 and n must be >= 0");
    for (int64_t k = 0; k < n; ++k) {
        const int64_t t = (t0 + k) % in->T;
        const int32_t s = in->slot[(size_t)t];
        if (s < 0) {
            QLE_TRY(BY_DTYPE(h, launch_predict, h, u_at(in, t)));
        } else {
            QLE_TRY(BY_DTYPE(h, launch_step, h, u_at(in, t), z_at(in, s)));
        }
        QLE_TRY(advance_tick(h));
    }
    return QLE_OK;
}

// On-chip-resident variant: ONE launch advances every filter by n ticks with x and P held in
// registers; HBM traffic is the state once plus the inputs.  Not the unit of work of the headline
// metric (one launch per tick); reported separately.
template <typename T>
static int run_resident_t(qle_batch* h, const qle_inputs* in, int64_t t0, int64_t n)
{
    const DevParams<T>& p = dev<T>(h);
    const dim3 g = grid_for(h, h->block), b(h->block);
    const T* pfp = (const T*)h->pfp;
    const int64_t pu = (int64_t)(in->pitch_u / h->wsz), pz = (int64_t)(in->pitch_z / h->wsz);
#define QLE_RES(D, F) hipLaunchKernelGGL((k_run_resident<T, D, F>), g, b, 0, h->stream, p, (T*)state_cur(h), (const T*)in->u, (const T*)in->z, (const int32_t*)in->d_slot, pu, pz, in->T, t0, n, pfp, h->B)
    if (h->pub.direct_orien_method) { if (h->pfp_on) QLE_RES(true, true); else QLE_RES(true, false); }
    else { if (h->pfp_on) QLE_RES(false, true); else QLE_RES(false, false); }
#undef QLE_RES
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
extern "C" int qle_run_resident(qle_batch* h, const qle_inputs* in, int64_t t0, int64_t n)
{
    QLE_TRY(check_handle(h));
    QLE_TRY(need_state(h));
    if (!in || in->h != h) return fail(QLE_ERR_INVALID, "inputs do not belong to this handle");
    if (t0 < 0 || n < 0) return fail(QLE_ERR_INVALID, "t0 and n must be >= 0");
    if (h->mr || h->gating) return fail(QLE_ERR_STATE, "qle_run_resident covers the single-rate filter with explicit masks (no multirate_ekf, no device gating)");
    if (n == 0) return QLE_OK;
    QLE_TRY(BY_DTYPE(h, run_resident_t, h, in, t0, n));
    for (int64_t k = 0; k < n; ++k) QLE_TRY(advance_tick(h));
    return QLE_OK;
}

// ------------------------------------------------------ synthetic generator
extern "C" int qle_synth_cfg_default(qle_synth_cfg* c)
{
    if (!c) return fail(QLE_ERR_INVALID, "cfg is null");
    std::memset(c, 0, sizeof(*c));
    c->seed = 0xE4F00003ULL;
    c->ab_true_sigma = 0.1;
    c->wb_true_sigma = 0.01;
    c->meas_noise_scale = 1.0;
    c->imu_noise_scale = 1.0;
    return QLE_OK;
}

template <typename T>
static int synth_t(qle_batch* h, qle_inputs* in, const qle_synth_cfg* c)
{
    SynthArgs a;
    a.seed = c->seed;
    a.filter_offset = c->filter_offset;
    a.ab_sigma = c->ab_true_sigma;
    a.wb_sigma = c->wb_true_sigma;
    a.meas_scale = c->meas_noise_scale;
    a.imu_scale = c->imu_noise_scale;
    a.perturb = c->perturb_filter_params;
    a.meas_delay_ticks = c->meas_delay_ticks < 0 ? 0 : (c->meas_delay_ticks > kSynthMaxDelay ? kSynthMaxDelay : c->meas_delay_ticks);
    a.dT = h->der.dT_nom;
    for (int i = 0; i < 12; ++i) a.Q[i] = h->der.Q[i];
    for (int i = 0; i < 6; ++i) a.R[i] = h->der.R[i];
    for (int i = 0; i < 3; ++i) { a.g[i] = h->pub.g[i]; a.r_v_cv[i] = h->pub.r_v_cv[i]; a.ab_static[i] = h->pub.ab_static[i]; a.wb_static[i] = h->pub.wb_static[i]; }
    for (int i = 0; i < 4; ++i) a.q_vc[i] = h->der.q_vc[i];
    for (int i = 0; i < 9; ++i) a.C_vc[i] = h->der.C_vc[i];
    a.est_bias = h->pub.est_bias;
    a.T = in->T;
    a.B = h->B;
    a.pitch_u_words = (int64_t)(in->pitch_u / h->wsz);
    a.pitch_z_words = (int64_t)(in->pitch_z / h->wsz);
    const int32_t* d_slot = in->d_slot;
    hipError_t e = hipSuccess;
    {
        if (c->perturb_filter_params && !h->pfp) e = hipMalloc(&h->pfp, kFW * (size_t)h->Bp * h->wsz);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL((k_synth<T>), grid_for(h, 64), dim3(64), 0, h->stream, a, (const int32_t*)d_slot, (T*)in->u, (T*)in->z, (T*)h->tick_z,
                           c->perturb_filter_params ? (T*)h->pfp : (T*)nullptr, (double*)in->truth, (double*)in->truth_bias);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(QLE_ERR_HIP, "synthetic generator: %s", hipGetErrorString(e));
    if (c->perturb_filter_params) h->pfp_on = true;
    in->has_truth = true;
    return QLE_OK;
}

extern "C" int qle_synth_generate(qle_batch* h, qle_inputs* in, const qle_synth_cfg* c)
{
    QLE_TRY(check_handle(h));
    if (!in || in->h != h) return fail(QLE_ERR_INVALID, "inputs do not belong to this handle");
    if (!c) return fail(QLE_ERR_INVALID, "cfg is null");
    QLE_TRY(BY_DTYPE(h, synth_t, h, in, c));
    // seed every filter from the generator's first (pre-sequence) tag pose, left in tick_z
    QLE_TRY(BY_DTYPE(h, seed_t, h, 1));
    h->state_set = true;
    h->hist_dirty = true;
    return QLE_OK;
}

template <typename T>
static int rmse_t(qle_batch* h, const qle_inputs* in, double* d_out)
{
    hipLaunchKernelGGL((k_rmse<T>), grid_for(h, 256), dim3(256), 0, h->stream, (const T*)state_cur(h), (const double*)in->truth, d_out, h->B);
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
extern "C" int qle_synth_rmse(qle_batch* h, const qle_inputs* in, double out[3])
{
    QLE_TRY(check_handle(h));
    if (!in || in->h != h || !out) return fail(QLE_ERR_INVALID, "bad arguments");
    if (!in->has_truth) return fail(QLE_ERR_STATE, "inputs hold no generated truth (qle_synth_generate)");
    double* d_out = h->stage;  // 3 doubles of the staging buffer
    HIP_TRY(hipMemsetAsync(d_out, 0, 3 * sizeof(double), h->stream));
    QLE_TRY(BY_DTYPE(h, rmse_t, h, in, d_out));
    HIP_TRY(hipMemcpyAsync(out, d_out, 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return QLE_OK;
}
