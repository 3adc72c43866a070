// ekf_capi.hip -- C-ABI (include/qle_ekf.h) over the HIP kernels.
// Host side of the batched relative-pose EKF engine: handle and device-memory
// management, parameter derivation (initialize_params, EKF.cpp:87-125 of the
// reference), AoS<->quad-row staging, launches on the handle's own stream.
// There is deliberately no CPU compute path in this file.
#include "ekf_host.hpp"
#include "synth_kernels.hpp"

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

int qle_fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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
    // small_ang_tol (EKF.hpp:131): below it the reference switches exp / log / F[th,th] to first-order forms (QH.cpp:19-24, :44-49,
    // EKF.cpp:385-389).  The engine evaluates the full series at every angle, which agrees with those forms to 1e-20 at the
    // reference's 1e-10 and stops agreeing as the tolerance grows (4e-7 in the covariance at 1e-3): a larger value is refused rather than
    // silently ignored.
    if (!(p->small_ang_tol >= 0.0) || p->small_ang_tol > 1e-8)
        return fail(QLE_ERR_INVALID, "small_ang_tol = %g is not supported: the engine evaluates the exact series at every angle, which matches the "
                                     "reference's small-angle forms only for tolerances <= 1e-8 (reference default 1e-10)", p->small_ang_tol);
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


static int check_handle(const qle_batch* h)
{
    if (!h) return fail(QLE_ERR_INVALID, "handle is null");
    (void)hipGetLastError();  // drop any stale sticky error of this thread: launches below check their own
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(QLE_ERR_HIP, "hipSetDevice(%d): %s", h->device, hipGetErrorString(e));
    return QLE_OK;
}

static void choose_cache_policy(qle_batch* h);
extern "C" int qle_set_params(qle_batch* h, const qle_params* p)
{
    QLE_TRY(check_handle(h));
    if (!p) return fail(QLE_ERR_INVALID, "params is null");
    qle_derived d;
    QLE_TRY(qle_params_derive(p, &d));
    // The state is one record array, updated in place.  The multirate EKF (EKF.cpp:196-236) keeps its history next to it: an IMU
    // ring, a state checkpoint every mr_k ticks and one anchor slot (k_step_mr, ekf_kernels.hpp), sized for the largest step delay
    // the parameters allow (EKF.cpp:199-201).  Everything is allocated into locals; the handle changes only when all of it exists.
    const bool mr = p->multirate_ekf != 0;
    int32_t Nc = 0, Cu = 0;
    if (mr) {
        int32_t step_max = d.measurement_step_delay;
        if (p->dynamic_meas_delay) step_max = std::max((int32_t)(p->measurement_delay_max / d.dT_nom + 0.5), 1);
        Nc = (step_max + h->mr_k + 1 + h->mr_k - 1) / h->mr_k + 1;
        Cu = Nc * h->mr_k;
    }
    if (!h->ring) {
        void* nr = nullptr;
        hipError_t e = hipMalloc(&nr, slot_bytes(h));
        if (e != hipSuccess) return fail(QLE_ERR_NOMEM, "hipMalloc of the state (%lld filters): %s", (long long)h->Bp, hipGetErrorString(e));
        e = hipMemsetAsync(nr, 0, slot_bytes(h), h->stream);   // all-zero records = filters not initialised
        if (e != hipSuccess) { (void)hipFree(nr); return fail(QLE_ERR_HIP, "state setup: %s", hipGetErrorString(e)); }
        h->ring = nr;
    }
    if (mr && (!h->hist_first || Nc != h->mr_Nc)) {
        int32_t* hf = nullptr;
        double *stp = nullptr, *dc = nullptr;
        void *mu = nullptr, *mc = nullptr, *ma = nullptr;
        const size_t ub = (size_t)Cu * kHW * (size_t)h->Bp * h->wsz, cb = (size_t)(Nc + 1) * slot_bytes(h);   // Nc grid checkpoints + the extra one
        hipError_t e = hipMalloc((void**)&hf, sizeof(int32_t) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMalloc((void**)&stp, sizeof(double) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMalloc((void**)&dc, sizeof(double) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMalloc(&mu, ub);
        if (e == hipSuccess) e = hipMalloc(&mc, cb);
        if (e == hipSuccess) e = hipMalloc(&ma, slot_bytes(h));
        if (e == hipSuccess) e = hipMemsetAsync(dc, 0, sizeof(double) * (size_t)h->Bp, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(stp, 0, sizeof(double) * (size_t)h->Bp, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(mu, 0, ub, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(mc, 0, cb, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(ma, 0, slot_bytes(h), h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            void* tmp[] = {hf, stp, dc, mu, mc, ma};
            for (void* b : tmp)
                if (b) (void)hipFree(b);
            return fail(QLE_ERR_NOMEM, "multirate history (%d checkpoint slots, %d IMU slots x %lld filters): %s", Nc, Cu, (long long)h->Bp, hipGetErrorString(e));
        }
        void* old[] = {h->hist_first, h->stamp, h->delay_cur, h->mr_u, h->mr_ckpt, h->mr_anchor};
        for (void* b : old)
            if (b) (void)hipFree(b);
        h->hist_first = hf; h->stamp = stp; h->delay_cur = dc; h->mr_u = mu; h->mr_ckpt = mc; h->mr_anchor = ma;
        h->mr_Nc = Nc; h->mr_Cu = Cu;
    }
    // Record layout (ekf_kernels.hpp): est_bias = false without the multirate history keeps only the 9 x 9 pose block of P (compact
    // records, 64 words moved per direction instead of 136) on the batch sizes the lane-per-filter kernels serve; the workgroup-cooperative
    // kernels of the small batches (latency-bound, not byte-bound) and the multirate history work on full records.  QLE_COMPACT=0|1 forces it.
    // fp32 small batches: the cooperative kernel only where ticks with tag poses are frequent (see qle_create)
    if (h->quad_auto && h->dtype == QLE_F32 && h->B <= 4096) h->quad = d.upd_per_meas <= 3 ? 1 : 0;
    bool compact = !p->est_bias && !mr && h->quad == 0;
    if (const char* s = std::getenv("QLE_COMPACT")) compact = std::atoi(s) != 0 && !p->est_bias && !mr;
    if (compact != h->compact && h->state_set) {   // a live state changes layout with the parameters
        if (h->dtype == QLE_F32) hipLaunchKernelGGL((k_relayout_P<float>), grid_for(h, 256), dim3(256), 0, h->stream, (float*)h->ring, (int)h->compact, (int)compact, h->B);
        else hipLaunchKernelGGL((k_relayout_P<double>), grid_for(h, 256), dim3(256), 0, h->stream, (double*)h->ring, (int)h->compact, (int)compact, h->B);
        HIP_TRY(hipGetLastError());
    }
    if (compact != h->compact) { h->compact = compact; choose_cache_policy(h); }
    h->pub = *p;
    h->der = d;
    h->pf = make_dev<float>(*p, d);
    h->pd = make_dev<double>(*p, d);
    h->pf.compact = h->pd.compact = compact ? 1 : 0;
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
    void* bufs[] = {h->ring, h->pfp, h->aux_accel, h->aux_obs, h->tick_u, h->tick_z, h->stage, h->stage_mask, h->counter, h->last_corr, h->flags, h->hist_first, h->stamp,
                    h->delay_cur, h->mr_u, h->mr_ckpt, h->mr_anchor};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return QLE_OK;
}

static void choose_cache_policy(qle_batch* h)
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
        // (compact records: only 64 of a record's 144 words are ever touched, qle_set_params)
        const double state_mib = (double)(h->compact ? kXW + kPWc : kSW) * (double)h->Bp * (double)h->wsz / (1024.0 * 1024.0);
        h->nt = state_mib <= 48.0 ? 1 : (state_mib <= 300.0 ? 0 : 3);
        h->nt_refresh = state_mib <= 40.0 ? 128 : 0;   // at 45 MiB the refresh scheme loses (18.2 vs 16.3 us), plain policy 1 wins
        if (const char* s = std::getenv("QLE_REFRESH")) h->nt_refresh = std::max(0, std::atoi(s));
        // larger than the cache: keep about 216 MiB of the state cached (k of every 64 workgroup groups, interleaved
        // over the batch and spread evenly over the XCDs) and stream the rest
        const int k64 = (int)std::lround(64.0 * 216.0 / std::max(state_mib, 1.0));
        h->split = -std::min(63, std::max(1, k64));
        if (state_mib > 64.0 * 240.0) h->nt = 2;       // even 1/64 of it would not fit: stream everything
        // QLE_CHUNK=n (experiments; profiles/r04_tuning.md section 4): the lane-per-filter single-rate ticks are launched n filters at a time and
        // the cache policy is the one of an n-filter state
        h->chunk = 0;
        if (const char* s = std::getenv("QLE_CHUNK")) {
            const int64_t c = std::atoll(s);
            if (c >= 256 && c % 256 == 0 && c < h->B) {
                h->chunk = c;
                const double chunk_mib = (double)(h->compact ? kXW + kPWc : kSW) * (double)c * (double)h->wsz / (1024.0 * 1024.0);
                h->nt = chunk_mib <= 48.0 ? 1 : (chunk_mib <= 300.0 ? 0 : 3);
                h->nt_refresh = chunk_mib <= 40.0 ? 128 : 0;
            }
        }
        if (const char* s = std::getenv("QLE_NT")) h->nt = std::min(3, std::max(0, std::atoi(s)));
        if (const char* s = std::getenv("QLE_SPLIT")) h->split = std::atoi(s);
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
    choose_cache_policy(h);
    // Workgroup-cooperative tick kernel (ekf_quad_kernels.hpp: scalar waves + covariance quads, 45 covariance values per lane).
    // Measured (profiles/r02_tuning.md section 2): it wins on ticks that carry corrections while the chip is not full -- up to 16 384
    // filters in both dtypes (fp64 4 096 filters, BASELINE cfg 2: 12.9 vs 18.9 us per tick; fp32: 8.2 vs 9.5 us) -- and loses beyond
    // (32 768 filters: fp32 13.0 vs 12.0 us, fp64 28.2 vs 25.5; 65 536: 25 vs 14.7 and 60 vs 38), so it is selected only there and
    // only for ticks with tag poses (predict-only ticks take the same time on both).  QLE_QUAD=bits overrides (1: ticks with tag
    // poses, 2: predict-only ticks, 0: never).
    // up to 4 096 filters (quarter-tile workgroups, one per CU) it also takes the predict-only ticks: 6.3 against 7.75 us at 4 096 fp64
    // Round 3, with one-wave workgroups for the lane-per-filter kernels below 65 536 filters (h->block below): the cooperative kernel
    // keeps every tick up to 4 096 filters (with tag poses: fp64 9.7 against 11.3 us, fp32 6.4 against 7.25; predict-only: fp64 5.56 against
    // 5.74, fp32 4.45 against 4.1 in kernel time -- but at 4-5 us per tick the host's launch rate is the limit and a schedule that stays on
    // one kernel family runs faster end to end: cfg 3 at 4 096 fp32 filters 4.75 against 5.05 us per tick); from 8 192 filters on the lane
    // kernels win every tick kind in both dtypes (8 192 fp64: 11.9 against 12.5 us with tag poses, 6.3 against 8.05 predict-only; 16 384:
    // 13.25 / 14.3, 7.5 / 9.3) -- profiles/r03_small_family.log.
    // Round 4, after the entry changes of the lane kernels (arguments preloaded, profiles/r04_tuning.md section 10): in fp32 the lane kernel
    // now wins the predict-only tick of the small batches as well, end to end (cfg 3 schedule at 4 096 fp32 filters: 4.00 us per tick on the
    // lane kernels alone, 4.11-4.25 with the cooperative kernel on the ticks with tag poses, 4.25-4.29 with it on every tick; 1 024 filters:
    // 3.93-4.03 / 3.91-3.95 / 4.06-4.13), while a tick WITH tag poses is still faster on the cooperative kernel (6.2 against 6.7 us).  So fp32
    // keeps it for the ticks with tag poses only, and only where the parameters say they are frequent (qle_set_params: at least every third
    // tick); fp64 keeps it for every tick (predict-only 5.2 against 5.6 us, with tag poses 9.6 against 11.6).  profiles/r04_small_quad_rule.log
    h->quad = batch <= 4096 ? (dtype == QLE_F64 ? 3 : 1) : 0;
    h->quad_auto = true;
    if (const char* s = std::getenv("QLE_QUAD")) { h->quad = std::atoi(s) & 7; h->quad_auto = false; }
    // Multirate history: a state checkpoint every mr_k ticks: a predict tick streams 136/k extra words, a correction replays
    // (k-1)/2 extra predictions on average.  Measured on cfg 3 with a 12-tick camera latency (profiles/r02_tuning.md): k = 4 / 8 / 16
    // -> predict tick 11.7 / 10.9 / 10.4 us, whole schedule 16.1 / 15.1 / 15.1 us per tick; 16 ships (history 0.6 GB).
    if (const char* s = std::getenv("QLE_MR_K")) h->mr_k = std::min(64, std::max(1, std::atoi(s)));
    if (const char* s = std::getenv("QLE_TICK_REBASE")) {
        const long long v = std::atoll(s);
        if (v >= 16) h->rebase_at = v;
    }
    // Workgroup size of the lane-per-filter kernels: 256 threads (4 tiles) at 65 536 and 131 072 filters; one wave per workgroup from
    // 262 144 filters on (finer dispatch: +2-3 % at 262 144 and 524 288 filters, +1.5 % at 1-2 M; profiles/r01_tuning.md section 4) and
    // BELOW 65 536 filters, where 256-thread workgroups leave CUs without work (32 768 filters are 128 of them on 256 CUs): predict tick
    // 7.2 -> 6.5 us at 32 768 fp32 filters, 6.0 -> 4.95 at 16 384, fp64 9.5 -> 7.55 at 16 384 (profiles/r03_tuning.md section 7).
    h->block = (batch >= 262144 || batch < 65536) ? 64 : kBlock;
    // k_predict<float> with every load in front of the arithmetic: where a SIMD holds one wave (predict_tick, ekf_kernels.hpp); QLE_LOADS_FIRST=0|1 forces
    h->loads_first = batch <= 65536;
    if (const char* s = std::getenv("QLE_LOADS_FIRST")) h->loads_first = std::atoi(s) != 0;
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
    const int64_t rec = h->compact ? 16 + kPWc : 136;   // state words a tick reads and writes (compact records: x + the 9 x 9 pose block of P)
    int64_t words = kind == 0 ? (rec + 6) + rec : kind == 1 ? (rec + 6 + 7) + rec : (rec + 7) + rec;
    if (h->pfp_on) words += kFW;
    int64_t bytes = words * (int64_t)h->wsz * h->B;
    // a multirate predict tick also appends to the history: the IMU sample (6 words + 2 pad) and, every mr_k-th tick, a checkpoint
    // (the extra checkpoint at the expected entry of the next tag pose, one more record per measurement cycle, belongs to the correcting
    // tick's account: bench.py)
    if (h->mr && kind == 0) bytes += (int64_t)((kHW + 136.0 / h->mr_k) * (double)h->wsz * (double)h->B);
    return bytes;
}

extern "C" int qle_get_policy(const qle_batch* h, qle_policy* out)
{
    if (!h || !out) return fail(QLE_ERR_INVALID, "null argument");
    out->state_policy = h->nt;
    out->refresh_period = (h->nt == 1) ? h->nt_refresh : 0;
    out->split_k64 = h->nt == 3 ? -h->split : 0;
    out->block = h->block;
    out->coop_ticks = h->mr ? 0 : h->quad;
    out->ring_slots = h->mr ? h->mr_Nc + 1 : 1;
    out->record_words = h->compact ? kXW + kPWc : kXW + kPW;
    out->reserved = 0;
    out->state_bytes = (int64_t)slot_bytes(h);
    // what a tick touches again later: the state itself, plus (multirate) the history it streams to
    out->ring_bytes = (int64_t)slot_bytes(h) * (h->mr ? (3 + h->mr_Nc) : 1) + (h->mr ? (int64_t)h->mr_Cu * kHW * h->Bp * (int64_t)h->wsz : 0);
    return QLE_OK;
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
                           i0, m, (int)h->compact);
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
        hipLaunchKernelGGL((k_unpack_P_off<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->stream, (const T*)src, n, h->stage, i0, m, (int)h->compact);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(P + i0 * n * n, h->stage, (size_t)m * n * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return QLE_OK;
}


extern "C" int qle_set_state(qle_batch* h, const double* x, const double* P)
{
    QLE_TRY(check_handle(h));
    if (!x || !P) return fail(QLE_ERR_INVALID, "x and P must be non-null");
    QLE_TRY(BY_DTYPE(h, pack_rows, h, x, kXW, kXW, state_cur(h), kSW, 0));
    QLE_TRY(BY_DTYPE(h, pack_P, h, P, state_cur(h)));
    if (!h->state_set && h->last_corr) {   // first state of the handle: upds_since_correction = 0 now (EKF.cpp:77)
        hipLaunchKernelGGL(k_fill_i32<int32_t>, grid_for(h, 256), dim3(256), 0, h->stream, h->last_corr, (int32_t)(h->tick - 1), h->B);
        HIP_TRY(hipGetLastError());
    }
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

extern "C" int qle_get_filter_params(qle_batch* h, double* pfp)
{
    QLE_TRY(check_handle(h));
    if (!pfp) return fail(QLE_ERR_INVALID, "pfp is null");
    if (!h->pfp || !h->pfp_on) return fail(QLE_ERR_STATE, "no per-filter parameters are set (qle_set_filter_params, or qle_synth_generate with perturb_filter_params)");
    return BY_DTYPE(h, unpack_rows, h, h->pfp, kFW, kFW, pfp, kFW, 0);
}

extern "C" int qle_enable_aux(qle_batch* h, int32_t on)
{
    QLE_TRY(check_handle(h));
    if (on && !h->aux_accel) {
        void *a = nullptr, *o = nullptr;
        hipError_t e = hipMalloc(&a, 3 * (size_t)h->B * h->wsz);
        if (e == hipSuccess) e = hipMalloc(&o, 7 * (size_t)h->B * h->wsz);
        if (e == hipSuccess) e = hipMemsetAsync(a, 0, 3 * (size_t)h->B * h->wsz, h->stream);
        if (e == hipSuccess) e = hipMemsetAsync(o, 0, 7 * (size_t)h->B * h->wsz, h->stream);
        if (e != hipSuccess) {
            if (a) (void)hipFree(a);
            if (o) (void)hipFree(o);
            return fail(QLE_ERR_NOMEM, "side-output buffers: %s", hipGetErrorString(e));
        }
        h->aux_accel = a; h->aux_obs = o;
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
int mr_prepare(qle_batch* h)
{
    if (h->mr && h->hist_dirty) {   // every filter's history = the single entry "state now": anchor <- state, hist_first = tick-1
        hipLaunchKernelGGL(k_fill_i32<int32_t>, grid_for(h, 256), dim3(256), 0, h->stream, h->hist_first, (int32_t)(h->tick - 1), h->B);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->mr_anchor, state_cur(h), slot_bytes(h), hipMemcpyDeviceToDevice, h->stream));
        h->e_tick = h->e_want = h->last_mr_launch = -1;
    }
    h->hist_dirty = false;
    return QLE_OK;
}

// the predict of one filter tick: slot n-1 -> slot n (the same slot when C == 1)
template <typename T>
static int launch_predict(qle_batch* h, const void* u)
{
    if (use_quad(h, 2)) return launch_quad<T>(h, u, nullptr);
    QLE_TRY(mr_prepare(h));
    return launch_predict_sd<T>(h, u, state_cur(h), state_cur(h), h->mr);
}


template <typename T>
static int launch_step(qle_batch* h, const void* u, const void* z)
{
    if (h->gating) h->flags_tick = h->tick;   // this tick writes the per-filter flag bytes (gated kernels only)
    if (use_quad(h, 1)) return launch_quad<T>(h, u, z);
    if (h->mr) return launch_step_mr<T>(h, u, z);
    return launch_step_lane<T>(h, u, z);
}


// One filter_update tick has been launched.  Tick indices are 32-bit on the device
// (last_corr, hist_first, history slots = tick modulo the ring sizes); long before they could wrap, shift the origin by a
// multiple of the ring capacity so that slots and differences are unchanged.
static int advance_tick(qle_batch* h)
{
    h->tick++;
    if (h->tick >= h->rebase_at) {
        const int64_t C = h->mr ? h->mr_Cu : 1;
        const int64_t shift = ((h->rebase_at / 2) / C) * C;  // a multiple of the IMU ring (= k x checkpoint slots): history slots are unchanged
        if (shift <= 0) return QLE_OK;
        int32_t* arrs[2] = {h->last_corr, h->hist_first};
        for (int32_t* a : arrs)
            if (a) {
                hipLaunchKernelGGL(k_rebase_ticks<int32_t>, grid_for(h, 256), dim3(256), 0, h->stream, a, (int32_t)shift, h->B);
                HIP_TRY(hipGetLastError());
            }
        h->tick -= shift;
        h->flags_tick -= shift;
        for (int64_t* tk : {&h->e_tick, &h->e_want, &h->last_mr_launch})
            if (*tk >= 0) *tk = std::max<int64_t>(*tk - shift, -1);
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
    if ((h->quad & 2) && !h->compact) return BY_DTYPE(h, launch_quad, h, h->tick_u, nullptr);
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
        int32_t* lc = nullptr;
        uint8_t* fl = nullptr;
        hipError_t e = hipMalloc((void**)&lc, sizeof(int32_t) * (size_t)h->Bp);
        if (e == hipSuccess) e = hipMalloc((void**)&fl, (size_t)h->Bp);
        if (e == hipSuccess) e = hipMemsetAsync(fl, 0, (size_t)h->Bp, h->stream);
        if (e != hipSuccess) {
            if (lc) (void)hipFree(lc);
            if (fl) (void)hipFree(fl);
            return fail(QLE_ERR_NOMEM, "gating arrays: %s", hipGetErrorString(e));
        }
        h->last_corr = lc; h->flags = fl;
        // upds_since_correction = 0 before the next tick (EKF.cpp:77): as if tick-1 had corrected
        hipLaunchKernelGGL(k_fill_i32<int32_t>, grid_for(h, 256), dim3(256), 0, h->stream, h->last_corr, (int32_t)(h->tick - 1), h->B);
        HIP_TRY(hipGetLastError());
    }
    h->gating = on != 0;
    return QLE_OK;
}

extern "C" int qle_filter_update(qle_batch* h, const double* u, const double* z, const uint8_t* measurement_ready)
{
    QLE_TRY(check_handle(h));
    if (!h->gating) return fail(QLE_ERR_STATE, "gating is not enabled (qle_enable_gating)");
    h->have_stamps = false;
    return qle_step(h, u, z, measurement_ready);
}

extern "C" int qle_filter_update_stamped(qle_batch* h, const double* u, const double* z, const uint8_t* measurement_ready, double t_curr,
                                         const double* apriltag_time)
{
    QLE_TRY(check_handle(h));
    if (!h->gating) return fail(QLE_ERR_STATE, "gating is not enabled (qle_enable_gating)");
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

template <typename T>
static int upds_since_t(qle_batch* h, int32_t* d_out)
{
    hipLaunchKernelGGL((k_upds_since<T>), grid_for(h, 256), dim3(256), 0, h->stream, (const T*)state_cur(h), (const int32_t*)h->last_corr, (int32_t)h->tick, d_out, h->B);
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
// NODE.cpp:192-281 in one call: the gathers above composed into one struct per filter (host side; the device work is theirs).
extern "C" int qle_get_node_report(qle_batch* h, qle_node_report* out)
{
    QLE_TRY(check_handle(h));
    if (!out) return fail(QLE_ERR_INVALID, "out is null");
    QLE_GUARD_BEGIN
    const size_t B = (size_t)h->B;
    std::vector<double> pose(B * 7), cov(B * 36), vel(B * 3), bias(B * 6), acc(B * 3, 0.0), obs(B * 7, 0.0), dly(B, 0.0);
    std::vector<uint8_t> pc(B, 0), co(B, 0), init(B, 0);
    std::vector<int32_t> ups(B, -1);
    QLE_TRY(qle_get_report(h, pose.data(), cov.data(), vel.data(), bias.data()));
    QLE_TRY(qle_get_state_initialized(h, init.data()));
    if (h->aux) QLE_TRY(qle_get_aux(h, acc.data(), obs.data()));
    if (h->last_corr) QLE_TRY(qle_get_tick_flags(h, pc.data(), co.data(), ups.data()));
    if (h->mr) QLE_TRY(qle_get_measurement_delay(h, dly.data()));
    for (size_t i = 0; i < B; ++i) {
        qle_node_report& r = out[i];
        std::memcpy(r.pose, &pose[i * 7], sizeof(r.pose));
        std::memcpy(r.pose_cov, &cov[i * 36], sizeof(r.pose_cov));
        std::memcpy(r.vel, &vel[i * 3], sizeof(r.vel));
        std::memcpy(r.accel, &acc[i * 3], sizeof(r.accel));
        std::memcpy(r.bias, &bias[i * 6], sizeof(r.bias));
        std::memcpy(r.obs, &obs[i * 7], sizeof(r.obs));
        r.measurement_delay_curr = dly[i];
        r.upds_since_correction = ups[i];
        r.performed_correction = pc[i];
        r.measurement_consumed = co[i];
        r.state_initialized = init[i];
        r.reserved = 0;
    }
    return QLE_OK;
    QLE_GUARD_END
}

extern "C" int qle_get_tick_flags(qle_batch* h, uint8_t* performed_correction, uint8_t* consumed, int32_t* upds_since_correction)
{
    QLE_TRY(check_handle(h));
    if (!h->last_corr) return fail(QLE_ERR_STATE, "gating is not enabled (qle_enable_gating)");
    QLE_GUARD_BEGIN
    std::vector<uint8_t> f((size_t)h->B);
    HIP_TRY(hipMemcpyAsync(f.data(), h->flags, (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
    if (upds_since_correction) {   // EKF.cpp:292-299; staged in chunks of the AoS staging buffer
        int32_t* d_out = reinterpret_cast<int32_t*>(h->stage);
        if ((size_t)h->B * sizeof(int32_t) <= (size_t)kStageDoubles * sizeof(double)) {
            QLE_TRY(BY_DTYPE(h, upds_since_t, h, d_out));
            HIP_TRY(hipMemcpyAsync(upds_since_correction, d_out, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream));
        } else {
            int32_t* tmp = nullptr;
            HIP_TRY(hipMalloc((void**)&tmp, sizeof(int32_t) * (size_t)h->B));
            int rc = BY_DTYPE(h, upds_since_t, h, tmp);
            hipError_t e = rc == QLE_OK ? hipMemcpyAsync(upds_since_correction, tmp, sizeof(int32_t) * (size_t)h->B, hipMemcpyDeviceToHost, h->stream) : hipSuccess;
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            (void)hipFree(tmp);
            if (rc != QLE_OK) return rc;
            if (e != hipSuccess) return fail(QLE_ERR_HIP, "tick flags: %s", hipGetErrorString(e));
        }
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    // the flag bytes are written by ticks that carry tag poses; after a predict-only tick nothing was performed or consumed
    const bool fresh = h->flags_tick == h->tick - 1;
    for (int64_t i = 0; i < h->B; ++i) {
        if (performed_correction) performed_correction[i] = fresh ? (f[(size_t)i] & 1) : 0;
        if (consumed) consumed[i] = fresh ? ((f[(size_t)i] >> 1) & 1) : 0;
    }
    return QLE_OK;
    QLE_GUARD_END
}

template <typename T>
static int seed_t(qle_batch* h, int reinit)
{
    const qle_derived& d = h->der;
    QLE_TRY(mr_prepare(h));   // a pending whole-batch history restart first; the seeded filters then restart theirs
    hipLaunchKernelGGL((k_seed<T>), grid_for(h, 256), dim3(256), 0, h->stream, dev<T>(h), (const T*)h->tick_z, (T*)state_cur(h),
                       (T)d.cov_init[0], (T)d.cov_init[3], (T)d.cov_init[6], (T)d.cov_init[9], (T)d.cov_init[12], reinit, (int32_t)h->tick,
                       h->last_corr, h->mr ? h->hist_first : (int32_t*)nullptr, h->mr ? (T*)h->mr_anchor : (T*)nullptr, h->B);
    HIP_TRY(hipGetLastError());
    return QLE_OK;
}
extern "C" int qle_initialize_state_masked(qle_batch* h, const double* z, const uint8_t* mask, int32_t reinit_bias)
{
    QLE_TRY(check_handle(h));
    if (!z) return fail(QLE_ERR_INVALID, "z is null");
    QLE_TRY(BY_DTYPE(h, pack_z, h, z, mask, h->tick_z));
    QLE_TRY(BY_DTYPE(h, seed_t, h, reinit_bias));
    h->state_set = true;
    return QLE_OK;
}
extern "C" int qle_initialize_state(qle_batch* h, const double* z, int32_t reinit_bias)
{
    return qle_initialize_state_masked(h, z, nullptr, reinit_bias);
}
extern "C" int qle_get_state_initialized(qle_batch* h, uint8_t* state_initialized)
{
    QLE_TRY(check_handle(h));
    if (!state_initialized) return fail(QLE_ERR_INVALID, "output is null");
    QLE_GUARD_BEGIN
    std::vector<double> x((size_t)h->B * kXW);
    QLE_TRY(BY_DTYPE(h, unpack_rows, h, state_cur(h), kXW, kXW, x.data(), kSW, 0));
    for (int64_t i = 0; i < h->B; ++i) {
        const double* q = &x[(size_t)i * kXW + 6];
        state_initialized[i] = (q[0] != 0.0 || q[1] != 0.0 || q[2] != 0.0 || q[3] != 0.0) ? 1 : 0;
    }
    return QLE_OK;
    QLE_GUARD_END
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
    hipLaunchKernelGGL((k_count_nonfinite<T>), grid_for(h, 256), dim3(256), 0, h->stream, (const T*)state_cur(h), h->counter, h->B, h->compact ? kXW + kPWc : kXW + kPW);
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
    c->view_scale = 1.0;
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
    a.view_scale = (c->view_scale > 0.0 && c->view_scale <= 1.0) ? c->view_scale : 1.0;
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
    return QLE_OK;
}

extern "C" int qle_synth_get_truth(qle_batch* h, const qle_inputs* in, double* pose, double* imu_bias)
{
    QLE_TRY(check_handle(h));
    if (!in || in->h != h) return fail(QLE_ERR_INVALID, "inputs do not belong to this handle");
    if (!in->has_truth) return fail(QLE_ERR_STATE, "inputs hold no generated truth (qle_synth_generate)");
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (pose) HIP_TRY(hipMemcpy(pose, in->truth, (size_t)h->B * 7 * sizeof(double), hipMemcpyDeviceToHost));
    if (imu_bias) HIP_TRY(hipMemcpy(imu_bias, in->truth_bias, (size_t)h->B * 6 * sizeof(double), hipMemcpyDeviceToHost));
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
