// relative_pose_ekf.hpp -- C++ host side of the engine, above the C-ABI (include/qle_ekf.h).
//
// Mirrors the public surface of the reference's `class RelativePoseEKF`
// (quad_state_estimation/include/relative_pose_EKF.hpp:20-141): same method names, same member
// names, same argument meaning and order.  Two classes:
//
//   qle_host::BatchedRelativePoseEKF  B filters on one MI355X (RAII over qle_batch)
//   qle_host::RelativePoseEKF         one filter, member-for-member drop-in for the reference
//                                     class: the node code of relative_pose_EKF_node.cpp:144-281
//                                     compiles against it after the type changes in INTEGRATION.md
//
// Data members use std::array / std::vector (quaternions x,y,z,w).  When <Eigen/Dense> is available
// (QLE_HAVE_EIGEN) the reference's exact Eigen-typed prediction_step / correction_step signatures
// (relative_pose_EKF.hpp:137-141) are provided as overloads.  Eigen is absent from the build image,
// so those overloads are compiled only on a system that has it.
//
// Nothing here computes filter arithmetic: every step is a call into libqle_ekf.so (HIP kernels).
#pragma once

#include <array>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/qle_ekf.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#include <Eigen/Geometry>
#define QLE_HAVE_EIGEN 1
#endif
#endif

namespace qle_host {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("qle error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc)
{
    if (rc != QLE_OK) throw Error(rc, qle_last_error());
}

// ------------------------------------------------------------------ batched
class BatchedRelativePoseEKF {
public:
    BatchedRelativePoseEKF(int64_t batch, int dtype, int device, const qle_params& p) : params(p)
    {
        check(qle_create(&h_, batch, dtype, device, &params));
        check(qle_params_derive(&params, &derived));
    }
    ~BatchedRelativePoseEKF() { qle_destroy(h_); }
    BatchedRelativePoseEKF(const BatchedRelativePoseEKF&) = delete;
    BatchedRelativePoseEKF& operator=(const BatchedRelativePoseEKF&) = delete;

    qle_batch* handle() { return h_; }
    int64_t batch() const { return qle_batch_size(h_); }
    int num_states() const { return derived.num_states; }

    void initialize_params()  // after editing `params` (relative_pose_EKF.cpp:87-125)
    {
        check(qle_set_params(h_, &params));
        check(qle_params_derive(&params, &derived));
    }
    void initialize_state(const double* z, bool reinit_bias) { check(qle_initialize_state(h_, z, reinit_bias ? 1 : 0)); }
    void set_state(const double* x, const double* P) { check(qle_set_state(h_, x, P)); }
    void get_state(double* x, double* P) { check(qle_get_state(h_, x, P)); }
    void predict(const double* u) { check(qle_predict(h_, u)); }                                    // prediction_step
    void update(const double* z, const uint8_t* mask = nullptr) { check(qle_update(h_, z, mask)); } // correction_step
    void step(const double* u, const double* z = nullptr, const uint8_t* mask = nullptr) { check(qle_step(h_, u, z, mask)); }
    void enable_gating(bool on) { check(qle_enable_gating(h_, on ? 1 : 0)); }
    void filter_update(const double* u, const double* z, const uint8_t* ready) { check(qle_filter_update(h_, u, z, ready)); }
    void synchronize() { check(qle_synchronize(h_)); }

    qle_params params;
    qle_derived derived;

private:
    qle_batch* h_ = nullptr;
};

// --------------------------------------------------------------- one filter
class RelativePoseEKF {
public:
    using Vec3 = std::array<double, 3>;
    using Quat = std::array<double, 4>;  // x, y, z, w

    explicit RelativePoseEKF(int device = 0, int dtype = QLE_F64) : device_(device), dtype_(dtype)
    {   // relative_pose_EKF.cpp:8-85
        qle_params p;
        check(qle_params_default(&p));
        from_params(p);
        IMU_accel = IMU_ang_vel = apriltag_pos = r_nom = v_nom = accel_rel = ab_nom = wb_nom = r_t_vt_obs = Vec3{0, 0, 0};
        apriltag_orien = q_nom = q_tv_obs = Quat{0, 0, 0, 1};
        initialize_params();
    }
    ~RelativePoseEKF() { qle_destroy(h_); }
    RelativePoseEKF(const RelativePoseEKF&) = delete;
    RelativePoseEKF& operator=(const RelativePoseEKF&) = delete;

    // Perform periodic EKF filter update (relative_pose_EKF.hpp:26, .cpp:127-303; single-rate and multirate)
    void filter_update(double t_curr)
    {
        if (!state_initialized) return;  // .cpp:129-130
        double u[6] = {IMU_accel[0], IMU_accel[1], IMU_accel[2], IMU_ang_vel[0], IMU_ang_vel[1], IMU_ang_vel[2]};
        double z[7] = {apriltag_pos[0], apriltag_pos[1], apriltag_pos[2], apriltag_orien[0], apriltag_orien[1], apriltag_orien[2], apriltag_orien[3]};
        const uint8_t ready = measurement_ready ? 1 : 0;
        check(qle_filter_update_stamped(h_, u, ready ? z : nullptr, ready ? &ready : nullptr, t_curr, &apriltag_time));
        uint8_t perf = 0, cons = 0;
        int32_t upds = 0;
        check(qle_get_tick_flags(h_, &perf, &cons, &upds));
        if (cons) measurement_ready = false;  // .cpp:152
        performed_correction = perf != 0;     // .cpp:301
        upds_since_correction = upds;         // .cpp:292-299
        if (perf && multirate_ekf) check(qle_get_measurement_delay(h_, &measurement_delay_curr));  // .cpp:199
        pull();
        filter_active = true;                 // .cpp:302
    }

    // Initialize state to last received AprilTag relative pose (relative_pose_EKF.hpp:28, .cpp:305-344)
    void initialize_state(bool reinit_bias)
    {
        double z[7] = {apriltag_pos[0], apriltag_pos[1], apriltag_pos[2], apriltag_orien[0], apriltag_orien[1], apriltag_orien[2], apriltag_orien[3]};
        push();  // keeps the biases when !reinit_bias
        check(qle_initialize_state(h_, z, reinit_bias ? 1 : 0));
        pull();
        state_initialized = true;
    }

    // Compute convenience values derived from parameters (relative_pose_EKF.hpp:30, .cpp:87-125)
    void initialize_params()
    {
        qle_params p = to_params();
        qle_derived d;
        check(qle_params_derive(&p, &d));
        if (!h_) {
            check(qle_create(&h_, 1, dtype_, device_, &p));
            check(qle_enable_aux(h_, 1));
            check(qle_enable_gating(h_, 1));
        } else {
            check(qle_set_params(h_, &p));
        }
        dT_nom = d.dT_nom; upd_per_meas = d.upd_per_meas; num_states = d.num_states; measurement_step_delay = d.measurement_step_delay;
        for (int i = 0; i < 4; ++i) q_vc[i] = d.q_vc[i];  // quaternion_norm(q_vc), .cpp:121
        for (int i = 0; i < 9; ++i) C_vc[i] = d.C_vc[i];
        cov_init.assign((size_t)(num_states * num_states), 0.0);
        for (int i = 0; i < num_states; ++i) cov_init[(size_t)(i * num_states + i)] = d.cov_init[i];
        cov_pert = cov_init;  // .cpp:114
        push();
    }

    // Prediction / correction steps (private in the reference, relative_pose_EKF.hpp:137-141).
    void prediction_step(const double* x_km1, const double* P_km1, const double* u, double* x_check, double* P_check, double* pose_accel)
    {
        check(qle_set_state(h_, x_km1, P_km1));
        check(qle_predict(h_, u));
        check(qle_get_state(h_, x_check, P_check));
        check(qle_get_aux(h_, pose_accel, nullptr));
    }
    void correction_step(const double* x_check, const double* P_check, const double* r_c_tc, const double* q_ct_xyzw, double* x_hat, double* P_hat)
    {
        double z[7] = {r_c_tc[0], r_c_tc[1], r_c_tc[2], q_ct_xyzw[0], q_ct_xyzw[1], q_ct_xyzw[2], q_ct_xyzw[3]};
        check(qle_set_state(h_, x_check, P_check));
        check(qle_update(h_, z, nullptr));
        check(qle_get_state(h_, x_hat, P_hat));
        double obs[7];
        check(qle_get_aux(h_, nullptr, obs));
        for (int i = 0; i < 3; ++i) r_t_vt_obs[i] = obs[i];
        for (int i = 0; i < 4; ++i) q_tv_obs[i] = obs[3 + i];
    }
#ifdef QLE_HAVE_EIGEN
    // The reference's own signatures (relative_pose_EKF.hpp:137-141); P is n x n, n = num_states.
    void prediction_step(Eigen::VectorXd x_km1, Eigen::MatrixXd P_km1, Eigen::VectorXd u, Eigen::VectorXd& x_check,
                         Eigen::MatrixXd& P_check, Eigen::VectorXd& pose_accel)
    {
        const int n = num_states;
        Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> Pin = P_km1, Pout(n, n);
        x_check.resize(16); pose_accel.resize(3);
        prediction_step(x_km1.data(), Pin.data(), u.data(), x_check.data(), Pout.data(), pose_accel.data());
        P_check = Pout;
    }
    void correction_step(Eigen::VectorXd x_check, Eigen::MatrixXd P_check, Eigen::VectorXd r_c_tc, Eigen::Quaterniond q_ct,
                         Eigen::VectorXd& x_hat, Eigen::MatrixXd& P_hat)
    {
        const int n = num_states;
        Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> Pin = P_check, Pout(n, n);
        const double q[4] = {q_ct.x(), q_ct.y(), q_ct.z(), q_ct.w()};
        x_hat.resize(16);
        correction_step(x_check.data(), Pin.data(), r_c_tc.data(), q, x_hat.data(), Pout.data());
        P_hat = Pout;
    }
#endif

    // ---- storage: same names as relative_pose_EKF.hpp:37-133 ----
    // Inputs
    Vec3 IMU_accel, IMU_ang_vel, apriltag_pos;
    Quat apriltag_orien;
    double apriltag_time = 0.0;
    // State  x = [r, v, theta, a_bias, w_bias]
    Vec3 r_nom, v_nom, accel_rel;
    Quat q_nom;
    Vec3 ab_nom, wb_nom;
    std::vector<double> cov_pert;  // num_states x num_states, row-major
    Vec3 ab_static, wb_static;
    Vec3 r_t_vt_obs;
    Quat q_tv_obs;
    // Filter parameters
    double update_freq, dT_nom = 0, measurement_freq, measurement_delay, measurement_delay_max, dyn_measurement_delay_offset = 0;
    double t_last_update = 0;
    bool est_bias, limit_measurement_freq, corner_margin_enbl, direct_orien_method, multirate_ekf, dynamic_meas_delay = false;
    int upd_per_meas = 0, num_states = 15, measurement_step_delay = 1;
    double measurement_delay_curr = 0.0;
    // Process and measurement noises
    double r_cov_init, v_cov_init, ang_cov_init, ab_cov_init, wb_cov_init;
    std::vector<double> cov_init;
    Vec3 Q_a, Q_w, Q_ab, Q_wb, R_r, R_ang;
    // Camera calibration
    Vec3 r_v_cv;
    Quat q_vc;
    std::array<double, 9> C_vc, camera_K;  // row-major
    int camera_width, camera_height;
    // Target configuration
    int n_tags;
    double tag_in_view_margin;
    std::vector<double> tag_widths;     // n_tags
    std::vector<double> tag_positions;  // 3 per tag: x, y, z (relative_pose_EKF_node.cpp:130-136)
    // Counters / flags
    bool state_initialized = false, measurement_ready = false, performed_correction = false, filter_active = false;
    int upds_since_correction = 0;
    // Tolerances and constants
    double small_ang_tol;
    Vec3 g;

private:
    void from_params(const qle_params& p)
    {
        update_freq = p.update_freq; measurement_freq = p.measurement_freq; measurement_delay = p.measurement_delay;
        measurement_delay_max = p.measurement_delay_max; dyn_measurement_delay_offset = p.dyn_measurement_delay_offset;
        est_bias = p.est_bias; limit_measurement_freq = p.limit_measurement_freq; corner_margin_enbl = p.corner_margin_enbl;
        direct_orien_method = p.direct_orien_method; multirate_ekf = p.multirate_ekf; dynamic_meas_delay = p.dynamic_meas_delay;
        r_cov_init = p.r_cov_init; v_cov_init = p.v_cov_init; ang_cov_init = p.ang_cov_init; ab_cov_init = p.ab_cov_init; wb_cov_init = p.wb_cov_init;
        for (int i = 0; i < 3; ++i) {
            Q_a[i] = p.Q_a[i]; Q_w[i] = p.Q_w[i]; Q_ab[i] = p.Q_ab[i]; Q_wb[i] = p.Q_wb[i]; R_r[i] = p.R_r[i]; R_ang[i] = p.R_ang[i];
            ab_static[i] = p.ab_static[i]; wb_static[i] = p.wb_static[i]; r_v_cv[i] = p.r_v_cv[i]; g[i] = p.g[i];
        }
        for (int i = 0; i < 4; ++i) q_vc[i] = p.q_vc[i];
        for (int i = 0; i < 9; ++i) camera_K[i] = p.camera_K[i];
        camera_width = p.camera_width; camera_height = p.camera_height; n_tags = p.n_tags; tag_in_view_margin = p.tag_in_view_margin;
        tag_widths.assign(p.tag_widths, p.tag_widths + p.n_tags);
        tag_positions.assign(p.tag_positions, p.tag_positions + 3 * p.n_tags);
        small_ang_tol = p.small_ang_tol;
    }
    qle_params to_params() const
    {
        qle_params p;
        std::memset(&p, 0, sizeof(p));
        p.update_freq = update_freq; p.measurement_freq = measurement_freq; p.measurement_delay = measurement_delay;
        p.measurement_delay_max = measurement_delay_max; p.dyn_measurement_delay_offset = dyn_measurement_delay_offset;
        p.est_bias = est_bias; p.limit_measurement_freq = limit_measurement_freq; p.corner_margin_enbl = corner_margin_enbl;
        p.direct_orien_method = direct_orien_method; p.multirate_ekf = multirate_ekf; p.dynamic_meas_delay = dynamic_meas_delay;
        p.r_cov_init = r_cov_init; p.v_cov_init = v_cov_init; p.ang_cov_init = ang_cov_init; p.ab_cov_init = ab_cov_init; p.wb_cov_init = wb_cov_init;
        for (int i = 0; i < 3; ++i) {
            p.Q_a[i] = Q_a[i]; p.Q_w[i] = Q_w[i]; p.Q_ab[i] = Q_ab[i]; p.Q_wb[i] = Q_wb[i]; p.R_r[i] = R_r[i]; p.R_ang[i] = R_ang[i];
            p.ab_static[i] = ab_static[i]; p.wb_static[i] = wb_static[i]; p.r_v_cv[i] = r_v_cv[i]; p.g[i] = g[i];
        }
        for (int i = 0; i < 4; ++i) p.q_vc[i] = q_vc[i];
        for (int i = 0; i < 9; ++i) p.camera_K[i] = camera_K[i];
        p.camera_width = camera_width; p.camera_height = camera_height; p.n_tags = n_tags; p.tag_in_view_margin = tag_in_view_margin;
        if (n_tags < 0 || n_tags > QLE_MAX_TAGS || (int)tag_widths.size() < n_tags || (int)tag_positions.size() < 3 * n_tags)
            throw Error(QLE_ERR_INVALID, "tag_widths / tag_positions do not match n_tags");
        for (int i = 0; i < n_tags; ++i) p.tag_widths[i] = tag_widths[(size_t)i];
        for (int i = 0; i < 3 * n_tags; ++i) p.tag_positions[i] = tag_positions[(size_t)i];
        p.small_ang_tol = small_ang_tol;
        return p;
    }
    void pack_x(double* x) const
    {   // relative_pose_EKF.cpp:244-245
        for (int i = 0; i < 3; ++i) { x[i] = r_nom[i]; x[3 + i] = v_nom[i]; x[10 + i] = ab_nom[i]; x[13 + i] = wb_nom[i]; }
        for (int i = 0; i < 4; ++i) x[6 + i] = q_nom[i];
    }
    void push()
    {
        double x[16];
        pack_x(x);
        cov_pert.resize((size_t)(num_states * num_states), 0.0);
        check(qle_set_state(h_, x, cov_pert.data()));
    }
    void pull()
    {   // relative_pose_EKF.cpp:273-290
        double x[16], acc[3], obs[7];
        cov_pert.resize((size_t)(num_states * num_states));
        check(qle_get_state(h_, x, cov_pert.data()));
        check(qle_get_aux(h_, acc, obs));
        for (int i = 0; i < 3; ++i) { r_nom[i] = x[i]; v_nom[i] = x[3 + i]; ab_nom[i] = x[10 + i]; wb_nom[i] = x[13 + i]; accel_rel[i] = acc[i]; }
        for (int i = 0; i < 4; ++i) q_nom[i] = x[6 + i];
        if (performed_correction) {
            for (int i = 0; i < 3; ++i) r_t_vt_obs[i] = obs[i];
            for (int i = 0; i < 4; ++i) q_tv_obs[i] = obs[3 + i];
        }
    }

    qle_batch* h_ = nullptr;
    int device_, dtype_;
};

}  // namespace qle_host
