/*
 * qle_ekf.h -- C-ABI of the MI355X-native batched relative-pose EKF engine.
 *
 * Drop-in boundary for the hot path of mbrymer/quadrotor_landing's
 * quad_state_estimation filter core.  The reference has no FFI; its boundary
 * is the public surface of `class RelativePoseEKF`
 * (quad_state_estimation/include/relative_pose_EKF.hpp:20-141) as used by its
 * one caller (src/relative_pose_EKF_node.cpp:144-182).  Each entry point below
 * cites the reference interface it replaces.  Paths are relative to
 * quad_state_estimation/ in the reference:
 *   EKF.hpp  = include/relative_pose_EKF.hpp
 *   EKF.cpp  = src/relative_pose_EKF.cpp
 *   NODE.cpp = src/relative_pose_EKF_node.cpp
 *
 * Conventions fixed by this ABI
 *   - plain pointers and sizes only; no C++/Eigen/torch types cross it.
 *   - host buffers are caller-owned, row-major AoS, always fp64 (the
 *     reference's type); the handle owns device memory in its compute dtype.
 *   - quaternions are x,y,z,w everywhere (EKF.cpp:329, QH.cpp:88-93).
 *   - state x = [r(3) v(3) q(4) ab(3) wb(3)] = 16 doubles (EKF.cpp:244-245).
 *   - covariance P is n x n row-major, n = num_states = 15 (est_bias) or 9
 *     (EKF.cpp:92).  The engine stores the symmetric part, packed (120 words).
 *   - measurement z = [r_c_tc(3) q_ct(x,y,z,w)(4)] = 7 doubles.
 *   - every call returns 0 (QLE_OK) or a negative error class; the message is
 *     in qle_last_error() (thread-local).  Nothing throws across the ABI.
 *     The reference returns void everywhere and fails by Eigen assertion or
 *     silent NaN (EKF.cpp:475); qle_count_nonfinite() makes the latter visible.
 *   - a handle is single-writer; calls on one handle are ordered on that
 *     handle's HIP stream and are asynchronous unless they return host data.
 *   - there is NO CPU fallback: without a usable HIP device qle_create fails
 *     with QLE_ERR_NO_DEVICE.
 */
#ifndef QLE_EKF_H
#define QLE_EKF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QLE_OK 0
#define QLE_ERR_INVALID (-1)   /* bad argument / shape / null pointer        */
#define QLE_ERR_HIP (-2)       /* a HIP runtime call failed                  */
#define QLE_ERR_NOMEM (-3)     /* host or device allocation failed           */
#define QLE_ERR_STATE (-4)     /* call order: state not initialised, etc.    */
#define QLE_ERR_NO_DEVICE (-5) /* no HIP device / device index out of range  */

#define QLE_F32 0
#define QLE_F64 1

#define QLE_MAX_TAGS 16
#define QLE_X_DIM 16
#define QLE_U_DIM 6
#define QLE_Z_DIM 7
#define QLE_PFP_DIM 24 /* per-filter params: Q diag 12, ab_static 3, wb_static 3, R diag 6 */

/* Public parameters of the filter == the reference's public data members
 * (EKF.hpp:66-133) as the node fills them (NODE.cpp:52-136).  Defaults:
 * qle_params_default(). */
typedef struct qle_params {
    double update_freq;                  /* EKF.hpp:66  */
    double measurement_freq;             /* EKF.hpp:68  */
    double measurement_delay;            /* EKF.hpp:69  */
    double measurement_delay_max;        /* EKF.hpp:70  */
    double dyn_measurement_delay_offset; /* EKF.hpp:71  */
    int32_t est_bias;                    /* EKF.hpp:74  */
    int32_t limit_measurement_freq;      /* EKF.hpp:75  */
    int32_t corner_margin_enbl;          /* EKF.hpp:76  */
    int32_t direct_orien_method;         /* EKF.hpp:77  */
    int32_t multirate_ekf;               /* EKF.hpp:78  */
    int32_t dynamic_meas_delay;          /* EKF.hpp:79  */
    double r_cov_init, v_cov_init, ang_cov_init, ab_cov_init, wb_cov_init; /* EKF.hpp:88-92 */
    double Q_a[3], Q_w[3], Q_ab[3], Q_wb[3];   /* EKF.hpp:96-99  */
    double R_r[3], R_ang[3];                   /* EKF.hpp:102-103 */
    double ab_static[3], wb_static[3];         /* EKF.hpp:55-56  */
    double r_v_cv[3];                          /* EKF.hpp:107 */
    double q_vc[4];                            /* EKF.hpp:108, x,y,z,w (NODE.cpp:109) */
    double camera_K[9];                        /* EKF.hpp:112, row-major (NODE.cpp:117) */
    int32_t camera_width, camera_height;       /* EKF.hpp:113-114 */
    int32_t n_tags;                            /* EKF.hpp:117 */
    int32_t _pad0;
    double tag_in_view_margin;                 /* EKF.hpp:118 */
    double tag_widths[QLE_MAX_TAGS];           /* EKF.hpp:120 */
    double tag_positions[3 * QLE_MAX_TAGS];    /* EKF.hpp:121, x,y,z per tag (NODE.cpp:130-136) */
    double small_ang_tol;                      /* EKF.hpp:131; must be <= 1e-8: the engine evaluates the exact series at every angle
                                                  (equal to the reference's small-angle forms there), larger values are refused */
    double g[3];                               /* EKF.hpp:132 */
} qle_params;

/* Values initialize_params() derives (EKF.cpp:87-125). */
typedef struct qle_derived {
    double dT_nom;                  /* EKF.cpp:90 */
    int32_t upd_per_meas;           /* EKF.cpp:91 */
    int32_t num_states;             /* EKF.cpp:92 */
    int32_t measurement_step_delay; /* EKF.cpp:93 */
    int32_t _pad0;
    double Q[12];                   /* diag of Q, EKF.cpp:100-112 (entries 6..11 zero if !est_bias) */
    double R[6];                    /* diag of R, EKF.cpp:116-118 */
    double cov_init[15];            /* diag of cov_init, EKF.cpp:102-113 */
    double q_vc[4];                 /* normalised, EKF.cpp:121 */
    double C_vc[9];                 /* row-major, EKF.cpp:122 */
} qle_derived;

typedef struct qle_batch qle_batch;   /* B filters on one device            */
typedef struct qle_inputs qle_inputs; /* device-resident IMU/tag sequences  */

/* ---- library / errors ---------------------------------------------------- */
const char *qle_last_error(void);
const char *qle_version(void);
int qle_device_count(int32_t *count);

/* ---- parameters (host only, no device needed) ----------------------------- */
/* RelativePoseEKF::RelativePoseEKF() defaults, EKF.cpp:8-85, plus the node's
 * cov_init defaults NODE.cpp:89-93 (the constructor leaves those uninitialised). */
int qle_params_default(qle_params *p);
/* RelativePoseEKF::initialize_params(), EKF.cpp:87-125. */
int qle_params_derive(const qle_params *p, qle_derived *d);

/* ---- handle --------------------------------------------------------------- */
/* Replaces constructing a RelativePoseEKF (EKF.cpp:8-85) + the node's parameter
 * overwrite and initialize_params() call (NODE.cpp:52-138), for `batch`
 * independent filters on HIP device `device`.  dtype = QLE_F32 | QLE_F64. */
int qle_create(qle_batch **out, int64_t batch, int32_t dtype, int32_t device, const qle_params *p);
int qle_destroy(qle_batch *h);
/* Re-run initialize_params() after changing public members (NODE.cpp:138). */
int qle_set_params(qle_batch *h, const qle_params *p);
/* Per-filter overrides of Q, static biases and R (BASELINE cfg 5):
 * pfp = [batch][QLE_PFP_DIM] or NULL to return to the shared parameters. */
int qle_set_filter_params(qle_batch *h, const double *pfp);
/* Read the per-filter overrides back ([batch][QLE_PFP_DIM], in the compute dtype's rounding): what
 * qle_set_filter_params stored or what qle_synth_generate drew with perturb_filter_params (BASELINE cfg 5), so that
 * a checker can run the same population. */
int qle_get_filter_params(qle_batch *h, double *pfp);
int64_t qle_batch_size(const qle_batch *h);
int32_t qle_dtype(const qle_batch *h);
int32_t qle_num_states(const qle_batch *h);

/* ---- state I/O ------------------------------------------------------------ */
/* Write r_nom,v_nom,q_nom,ab_nom,wb_nom and cov_pert (EKF.hpp:47-53) for all
 * filters.  x = [batch][16]; P = [batch][n][n] (symmetrised as (P+P^T)/2). */
int qle_set_state(qle_batch *h, const double *x, const double *P);
/* Read them back (NODE.cpp:195-220 reads these members after a tick). */
int qle_get_state(qle_batch *h, double *x, double *P);
/* RelativePoseEKF::initialize_state(reinit_bias), EKF.cpp:305-344, batched:
 * seeds every filter from its own first tag pose z = [batch][7]. */
int qle_initialize_state(qle_batch *h, const double *z, int32_t reinit_bias);
/* The same for the filters with mask[i] != 0 only (mask NULL = all).  The reference keeps state_initialized per
 * filter object: filter_update returns at once while it is false (EKF.cpp:129-130) and the node seeds a filter from
 * ITS first detection (NODE.cpp:169-174).  Here a filter is "not initialised" until initialize_state / set_state has
 * written its state (its stored quaternion is all zero until then); every tick entry point leaves such filters
 * untouched -- no predict, no counter, no history entry -- and a filter seeded later starts with
 * upds_since_correction = 0 (EKF.cpp:77) and a one-entry history (EKF.cpp:337-339). */
int qle_initialize_state_masked(qle_batch *h, const double *z, const uint8_t *mask, int32_t reinit_bias);
/* state_initialized (EKF.hpp:125) of every filter: [batch]. */
int qle_get_state_initialized(qle_batch *h, uint8_t *state_initialized);
/* Side outputs of the last tick: accel_rel (EKF.hpp:49; [batch][3]) and the
 * reported observation r_t_vt_obs,q_tv_obs (EKF.hpp:58-59; [batch][7]).
 * Only maintained while aux output is enabled (costs extra HBM writes). */
int qle_enable_aux(qle_batch *h, int32_t on);
int qle_get_aux(qle_batch *h, double *accel_rel, double *obs);

/* ---- hot path, host-buffer inputs (the reference hands inputs over by value) */
/* RelativePoseEKF::prediction_step, EKF.cpp:346-415 (decl EKF.hpp:137-138):
 * x,P <- predict(x,P,u) for every filter.  u = [batch][6] = accel, gyro. */
int qle_predict(qle_batch *h, const double *u);
/* RelativePoseEKF::correction_step, EKF.cpp:417-502 (decl EKF.hpp:140-141):
 * x,P <- correct(x,P,z) where mask[i] != 0 (mask NULL = all).  z = [batch][7]. */
int qle_update(qle_batch *h, const double *z, const uint8_t *mask);
/* One single-rate tick of filter_update, EKF.cpp:238-249 + 265-290: predict,
 * then correct where mask[i] != 0, fused in one pass over P.
 * z/mask may be NULL for a predict-only tick. */
int qle_step(qle_batch *h, const double *u, const double *z, const uint8_t *mask);

/* ---- the decision logic of filter_update on the device ----------------------
 * With gating enabled, the mask of qle_step / qle_filter_update / the mask word
 * of a sequence slot means "measurement_ready" (EKF.hpp:125), and the engine
 * decides per filter as filter_update does (EKF.cpp:147-186):
 *   consume = ready && (!limit_measurement_freq || upds_since_correction+1 >= upd_per_meas)
 *   perform = consume && (!corner_margin_enbl || a tag of the bundle projects inside the image margins)
 * and maintains upds_since_correction / performed_correction (EKF.cpp:292-301). */
int qle_enable_gating(qle_batch *h, int32_t on);
/* RelativePoseEKF::filter_update(t), single-rate branch (EKF.cpp:127-193,238-303):
 * u = latest IMU sample per filter [batch][6] (NODE.cpp:144-151), z = latest tag
 * pose [batch][7] or NULL, measurement_ready = [batch] or NULL (= all ready). */
int qle_filter_update(qle_batch *h, const double *u, const double *z, const uint8_t *measurement_ready);
/* The same with the time stamps the multirate EKF needs for dynamic_meas_delay
 * (EKF.cpp:199): t_curr = the tick's time, apriltag_time = [batch] header stamp of each
 * filter's latest tag pose (EKF.hpp:43, NODE.cpp:167).
 *
 * multirate_ekf = true (EKF.cpp:196-236, 251-264): every filter keeps a history of
 * (x, u, P) per tick in HBM; a correction is applied to the entry the measurement
 * belongs to, step = max(int(delay/dT_nom + 0.5), 1) ticks back, and the predictions
 * since are replayed from the stored IMU samples.  All tick entry points (qle_step,
 * qle_filter_update*, qle_run) follow that branch while the parameter is set;
 * qle_set_state / qle_initialize_state / bare qle_predict / qle_update restart the
 * history with a single entry (EKF.cpp:337-339). */
int qle_filter_update_stamped(qle_batch *h, const double *u, const double *z, const uint8_t *measurement_ready,
                              double t_curr, const double *apriltag_time);
/* measurement_delay_curr of each filter's last correction (EKF.hpp:86, EKF.cpp:199). */
int qle_get_measurement_delay(qle_batch *h, double *measurement_delay_curr);
/* Measurement age used for dynamic_meas_delay when no per-filter stamps are given
 * (qle_step, qle_run): default = measurement_delay. */
int qle_set_uniform_measurement_age(qle_batch *h, double seconds);
/* After a tick: performed_correction (EKF.hpp:126), whether the pending measurement
 * was consumed (the reference clears measurement_ready, EKF.cpp:152) and
 * upds_since_correction (EKF.hpp:128).  Any pointer may be NULL. */
int qle_get_tick_flags(qle_batch *h, uint8_t *performed_correction, uint8_t *consumed, int32_t *upds_since_correction);

/* ---- device-resident input sequences -------------------------------------- */
/* n_ticks of IMU input for the handle's batch; tick_has_meas[t] != 0 reserves
 * a tag-pose slot (z + per-filter mask) for tick t. */
int qle_inputs_create(qle_batch *h, int64_t n_ticks, const uint8_t *tick_has_meas, qle_inputs **out);
int qle_inputs_destroy(qle_inputs *in);
/* Upload one tick: u = [batch][6]; z = [batch][7] and mask = [batch] (or NULL =
 * all) only if that tick has a measurement slot. */
int qle_inputs_upload_tick(qle_inputs *in, int64_t t, const double *u, const double *z, const uint8_t *mask);
int qle_inputs_download_tick(qle_inputs *in, int64_t t, double *u, double *z, uint8_t *mask);
/* Run ticks [t0, t0+n) of the sequence (tick index wraps modulo n_ticks):
 * a predict launch on ticks without a measurement slot, a fused
 * predict+update launch on ticks with one.  Asynchronous. */
int qle_run(qle_batch *h, const qle_inputs *in, int64_t t0, int64_t n);

/* The same ticks as qle_run in ONE launch, with every filter's x and P held in registers for
 * all n ticks (single-rate filter, explicit masks).  HBM traffic is the state once plus the
 * inputs, so this is not the streamed one-launch-per-tick unit of work the headline metric and
 * the roofline figure are defined on; it is reported separately. */
int qle_run_resident(qle_batch *h, const qle_inputs *in, int64_t t0, int64_t n);

/* ---- synthetic truth + IMU + tag-pose generator (replaces the ROS inputs) --
 * SURVEY.md section 8(d) "synthetic inputs".  Values depend only on
 * (seed, global filter index, tick, channel): any sharding gives the same data. */
typedef struct qle_synth_cfg {
    uint64_t seed;
    int64_t filter_offset; /* global index of this handle's filter 0 */
    double ab_true_sigma, wb_true_sigma; /* true IMU bias spread (0.1, 0.01) */
    double meas_noise_scale;             /* 1 = N(0,R) on tag poses */
    double imu_noise_scale;              /* 1 = N(0,Q_a), N(0,Q_w) on IMU */
    int32_t perturb_filter_params;       /* cfg 5: also fill per-filter Q scale and static biases */
    int32_t meas_delay_ticks;            /* multirate runs: a tag pose delivered at tick t shows the pose after tick
                                            t - meas_delay_ticks (0 = no latency, single-rate) */
    double view_scale;                   /* 1 = the free flight of the BASELINE configs; < 1 shrinks the lateral offsets and
                                            amplitudes of the trajectory and its attitude excursions by this factor, so that the
                                            tag bundle stays inside the image -- a landing approach, what the node sees while it
                                            has detections at all (the corner gate of EKF.cpp:147-186 then passes) */
} qle_synth_cfg;
int qle_synth_cfg_default(qle_synth_cfg *c);
/* Fill `in` with a generated sequence and seed the filters from the first
 * (tick -1) noisy tag pose via initialize_state; keeps the truth pose at the
 * last tick on device for qle_synth_rmse. */
int qle_synth_generate(qle_batch *h, qle_inputs *in, const qle_synth_cfg *c);
/* Per-device error sums against the generator's truth at the end of the
 * sequence: out = { sum |r_err|^2, sum |theta_err|^2, count } (cfg 5). */
int qle_synth_rmse(qle_batch *h, const qle_inputs *in, double out[3]);
/* The generator's truth at the end of the sequence: pose = [batch][7] (r, q xyzw), imu_bias = [batch][6] (accel, gyro
 * bias).  Either pointer may be NULL.  Lets a host computation check qle_synth_rmse and the filters' tracking. */
int qle_synth_get_truth(qle_batch *h, const qle_inputs *in, double *pose, double *imu_bias);

/* ---- reporting (what the node publishes after a tick, NODE.cpp:192-220) ---- */
/* pose = [batch][7] (r, q xyzw); pose_cov = [batch][36]: rows/cols {0-2,6-8} of
 * cov_pert, row-major (NODE.cpp:203-210); vel = [batch][3];
 * bias = [batch][6] = ab_nom+ab_static, wb_nom+wb_static (NODE.cpp:215-220).
 * Any pointer may be NULL. */
int qle_get_report(qle_batch *h, double *pose, double *pose_cov, double *vel, double *bias);
/* Everything the node puts on the wire after a tick (NODE.cpp:192-281), one struct per filter, in ONE call:
 * rel_pose_state (pose + the 6x6 pose covariance, :192-211), rel_vel_state (:212-214), IMU_bias (:215-220), rel_accel (:222-226,
 * zero unless qle_enable_aux), upds_since_correction (:228-232) and, on a tick that corrected, the observation it fused and
 * measurement_delay_curr (:240-275).  The flag and counter fields need qle_enable_gating (else 0 / -1); the delay needs the
 * multirate EKF (else 0).  `out` holds batch structs. */
typedef struct qle_node_report {
    double pose[7];                 /* r_nom, q_nom (x,y,z,w) */
    double pose_cov[36];            /* rows/cols {0-2, 6-8} of cov_pert, row-major */
    double vel[3];                  /* v_nom */
    double accel[3];                /* accel_rel */
    double bias[6];                 /* ab_nom + ab_static, wb_nom + wb_static */
    double obs[7];                  /* r_t_vt_obs, q_tv_obs of the correction this tick performed */
    double measurement_delay_curr;  /* EKF.hpp:86 */
    int32_t upds_since_correction;  /* EKF.hpp:128; -1 without gating */
    uint8_t performed_correction;   /* EKF.hpp:126 */
    uint8_t measurement_consumed;   /* the tick cleared measurement_ready (EKF.cpp:186) */
    uint8_t state_initialized;      /* EKF.hpp:125 */
    uint8_t reserved;
} qle_node_report;
int qle_get_node_report(qle_batch *h, qle_node_report *out);
/* Number of filters whose x or P holds a NaN/Inf. */
int qle_count_nonfinite(qle_batch *h, int64_t *count);

/* ---- stream control / measurement ------------------------------------------ */
int qle_synchronize(qle_batch *h);
/* HIP events on the handle's own stream around a timed region. */
int qle_timer_begin(qle_batch *h);
int qle_timer_end(qle_batch *h, float *elapsed_ms); /* synchronises */
/* How this handle launches its ticks (chosen from the batch size and dtype at creation; measurement and reporting only). */
typedef struct qle_policy {
    int32_t state_policy;   /* cache policy of the state accesses: 0 cached loads+stores, 1 non-temporal loads, 2 non-temporal
                               loads+stores, 3 split (a fixed part cached, the rest streamed) */
    int32_t refresh_period; /* > 0: policy 2 with one cached-store tick every so many ticks (small states) */
    int32_t split_k64;      /* policy 3: this many of every 64 workgroup groups keep their tiles cached */
    int32_t block;          /* workgroup size of the one-lane-per-filter kernels */
    int32_t coop_ticks;     /* bit 0: ticks with tag poses, bit 1: predict-only ticks run on the workgroup-cooperative kernel */
    int32_t ring_slots;     /* state ring capacity (1 = single-rate, in place) */
    int64_t state_bytes;    /* bytes of one state slot (144 words x padded batch) */
    int64_t ring_bytes;     /* state_bytes x ring_slots */
    int32_t record_words;   /* state words a tick reads and writes per filter: 136 (x 16 + packed P 120), or 64 with compact records
                               (est_bias = false, relative_pose_EKF.cpp:92: x 16 + the 45 words of the 9 x 9 pose block + 3 pad) */
    int32_t reserved;
} qle_policy;
int qle_get_policy(const qle_batch *h, qle_policy *out);
/* Algorithmic HBM bytes one launch moves:
 * kind 0 = predict tick, 1 = fused predict+update tick, 2 = stand-alone update. */
int64_t qle_algorithmic_bytes(const qle_batch *h, int32_t kind);

#ifdef __cplusplus
}
#endif
#endif /* QLE_EKF_H */
