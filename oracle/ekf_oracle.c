/*
 * ekf_oracle.c -- CPU restatement of the reference EKF hot path (plain C99).
 * TEST INFRASTRUCTURE ONLY; see ekf_oracle.h for scope, citations and the
 * parity-pinning statement.  Every function cites the reference lines it
 * follows (EKF.cpp = quad_state_estimation/src/relative_pose_EKF.cpp,
 * QH.cpp = quad_state_estimation/src/quaternion_helper.cpp).
 *
 * The arithmetic deliberately keeps the reference's shape: dense n x n
 * products with materialised transposes, a general (LU) 6x6 inverse and the
 * simple (I-KG)P covariance form -- no structure is exploited here, so this
 * file doubles as the "reference-shaped" CPU baseline of bench.py.
 */
#include "ekf_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ utils */
static void mat_mul(const double *A, const double *B, double *C, int m, int k, int n)
{ /* C[m x n] = A[m x k] * B[k x n], row-major, plain triple loop */
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int l = 0; l < k; ++l) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void mat_T(const double *A, double *AT, int m, int n)
{ /* AT[n x m] = A[m x n]^T */
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) AT[j * m + i] = A[i * n + j];
}
static void set_block(double *M, int ld, int r0, int c0, const double *B, int br, int bc, double scale)
{
    for (int i = 0; i < br; ++i)
        for (int j = 0; j < bc; ++j) M[(r0 + i) * ld + c0 + j] = scale * B[i * bc + j];
}
static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

/* General inverse by partial-pivot LU (Eigen's dynamic-size .inverse() is
 * PartialPivLU solved against the identity; EKF.cpp:475). */
static void mat_inverse_lu(const double *A, double *Ainv, int n)
{
    double LU[36];
    int perm[6];
    memcpy(LU, A, sizeof(double) * (size_t)(n * n));
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = fabs(LU[k * n + k]);
        for (int i = k + 1; i < n; ++i)
            if (fabs(LU[i * n + k]) > best) { best = fabs(LU[i * n + k]); piv = i; }
        if (piv != k) {
            for (int j = 0; j < n; ++j) { double t = LU[k * n + j]; LU[k * n + j] = LU[piv * n + j]; LU[piv * n + j] = t; }
            int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
        }
        for (int i = k + 1; i < n; ++i) {
            LU[i * n + k] /= LU[k * n + k];
            for (int j = k + 1; j < n; ++j) LU[i * n + j] -= LU[i * n + k] * LU[k * n + j];
        }
    }
    for (int c = 0; c < n; ++c) {
        double y[6];
        for (int i = 0; i < n; ++i) { /* forward: L y = P e_c */
            double s = (perm[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s -= LU[i * n + j] * y[j];
            y[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) { /* backward: U x = y */
            double s = y[i];
            for (int j = i + 1; j < n; ++j) s -= LU[i * n + j] * Ainv[j * n + c];
            Ainv[i * n + c] = s / LU[i * n + i];
        }
    }
}

/* ------------------------------------------------- quaternion helpers */
void orc_quaternion_norm(double q[4])
{ /* QH.cpp:61-73: normalise; flip to w >= -0.75 single cover */
    const double q_w_lim = -0.75;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
    if (q[3] < q_w_lim) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
}

void orc_quaternion_exp(const double v[3], double q[4])
{ /* QH.cpp:9-33 */
    const double norm_tol = 1E-10;
    double norm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    q[3] = cos(norm / 2);
    if (norm < norm_tol) {
        double s = (1 - pow(norm, 2) / 24);
        for (int i = 0; i < 3; ++i) q[i] = v[i] / 2 * s;
    } else {
        double s = sin(norm / 2);
        for (int i = 0; i < 3; ++i) q[i] = v[i] / norm * s;
    }
    orc_quaternion_norm(q); /* QH.cpp:30 */
}

void orc_quaternion_log(const double q[4], double v[3])
{ /* QH.cpp:36-58 */
    const double norm_tol = 1E-10;
    double vec_norm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (vec_norm < norm_tol) {
        double s = 2 / q[3] * (1 - pow(vec_norm / q[3], 2) / 3);
        for (int i = 0; i < 3; ++i) v[i] = s * q[i];
    } else {
        double phi = 2 * atan2(vec_norm, q[3]);
        double s = phi / vec_norm;
        for (int i = 0; i < 3; ++i) v[i] = s * q[i];
    }
}

void orc_skew_symm(const double v[3], double m[9])
{ /* QH.cpp:76-85 */
    m[0] = 0;     m[1] = -v[2]; m[2] = v[1];
    m[3] = v[2];  m[4] = 0;     m[5] = -v[0];
    m[6] = -v[1]; m[7] = v[0];  m[8] = 0;
}

void orc_quat_mul(const double a[4], const double b[4], double o[4])
{ /* Eigen quaternion product (Hamilton), storage x,y,z,w */
    double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

void orc_quat_to_rot(const double q[4], double C[9])
{ /* Eigen QuaternionBase::toRotationMatrix (no renormalisation) */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w;
    double txx = tx * x, txy = ty * x, txz = tz * x;
    double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    C[0] = 1 - (tyy + tzz); C[1] = txy - twz;       C[2] = txz + twy;
    C[3] = txy + twz;       C[4] = 1 - (txx + tzz); C[5] = tyz - twx;
    C[6] = txz - twy;       C[7] = tyz + twx;       C[8] = 1 - (txx + tyy);
}

void orc_angle_axis_to_rot(double angle, const double axis[3], double C[9])
{ /* Eigen AngleAxis::toRotationMatrix (EKF.cpp:394) */
    double s = sin(angle), c = cos(angle);
    double sa[3] = {s * axis[0], s * axis[1], s * axis[2]};
    double ca[3] = {(1 - c) * axis[0], (1 - c) * axis[1], (1 - c) * axis[2]};
    double tmp;
    tmp = ca[0] * axis[1]; C[1] = tmp - sa[2]; C[3] = tmp + sa[2];
    tmp = ca[0] * axis[2]; C[2] = tmp + sa[1]; C[6] = tmp - sa[1];
    tmp = ca[1] * axis[2]; C[5] = tmp - sa[0]; C[7] = tmp + sa[0];
    C[0] = ca[0] * axis[0] + c; C[4] = ca[1] * axis[1] + c; C[8] = ca[2] * axis[2] + c;
}

/* ------------------------------------------------------------ parameters */
void orc_params_default(orc_params *p)
{ /* EKF.cpp:28-81 constructor defaults; cov_init from NODE.cpp:89-93 */
    memset(p, 0, sizeof(*p));
    p->update_freq = 100; p->measurement_freq = 10;
    p->measurement_delay = 0.010; p->measurement_delay_max = 0.200;
    p->dyn_measurement_delay_offset = 0.0;
    p->est_bias = 1; p->limit_measurement_freq = 0; p->corner_margin_enbl = 1;
    p->direct_orien_method = 0; p->multirate_ekf = 0; p->dynamic_meas_delay = 0;
    p->r_cov_init = 0.1; p->v_cov_init = 0.1; p->ang_cov_init = 0.15;
    p->ab_cov_init = 0.5; p->wb_cov_init = 0.1;
    for (int i = 0; i < 3; ++i) { p->Q_a[i] = 0.005; p->Q_w[i] = 0.0005; p->Q_ab[i] = 5E-5; p->Q_wb[i] = 5E-6; }
    p->R_r[0] = 0.005; p->R_r[1] = 0.005; p->R_r[2] = 0.015;
    p->R_ang[0] = 0.0025; p->R_ang[1] = 0.0025; p->R_ang[2] = 0.025;
    p->r_v_cv[2] = -0.073;
    p->q_vc[0] = 0.70711; p->q_vc[1] = -0.70711; p->q_vc[2] = 0; p->q_vc[3] = 0; /* Quaterniond(0,0.70711,-0.70711,0) w-first */
    p->camera_K[0] = 241.4268; p->camera_K[2] = 376.5; p->camera_K[4] = 241.4268; p->camera_K[5] = 240.5; p->camera_K[8] = 1;
    p->camera_width = 752; p->camera_height = 480;
    p->n_tags = 1; p->tag_in_view_margin = 0.02; p->tag_widths[0] = 0.8;
    p->small_ang_tol = 1E-10;
    p->g[2] = -9.8;
    orc_initialize_params(p);
}

void orc_initialize_params(orc_params *p)
{ /* EKF.cpp:87-125 */
    p->dT_nom = 1 / p->update_freq;
    p->upd_per_meas = (int)ceil(p->update_freq / p->measurement_freq);
    p->num_states = p->est_bias ? 15 : 9;
    { int d = (int)(p->measurement_delay / p->dT_nom + 0.5); p->measurement_step_delay = d > 1 ? d : 1; }
    memset(p->Q, 0, sizeof(p->Q)); memset(p->cov_init, 0, sizeof(p->cov_init));
    for (int i = 0; i < 3; ++i) {
        p->Q[i] = p->Q_a[i]; p->Q[3 + i] = p->Q_w[i];
        p->cov_init[i] = p->r_cov_init; p->cov_init[3 + i] = p->v_cov_init; p->cov_init[6 + i] = p->ang_cov_init;
        if (p->est_bias) {
            p->Q[6 + i] = p->Q_ab[i]; p->Q[9 + i] = p->Q_wb[i];
            p->cov_init[9 + i] = p->ab_cov_init; p->cov_init[12 + i] = p->wb_cov_init;
        }
        p->R[i] = p->R_r[i]; p->R[3 + i] = p->R_ang[i];
    }
    orc_quaternion_norm(p->q_vc);       /* :121 */
    orc_quat_to_rot(p->q_vc, p->C_vc);  /* :122 */
    /* T_vc = Translation(r_v_cv) * q_vc (:123): linear C_vc, translation r_v_cv */
}

/* -------------------------------------------------------- prediction step */
void orc_prediction_step(const orc_params *p, const double x_km1[16], const double *P_km1,
                         const double u[6], double x_check[16], double *P_check, double pose_accel[3])
{ /* EKF.cpp:346-415 */
    const int n = p->num_states;
    const int nq = p->est_bias ? 12 : 6;
    const double *r_km1 = x_km1, *v_km1 = x_km1 + 3, *q_km1 = x_km1 + 6;
    const double *ab_km1 = x_km1 + 10, *wb_km1 = x_km1 + 13;
    const double dT = p->dT_nom; /* :356 */
    double a_nom[3], w_nom[3], C[9];
    for (int i = 0; i < 3; ++i) {
        a_nom[i] = u[i] - ab_km1[i] - p->ab_static[i];      /* :357 */
        w_nom[i] = u[3 + i] - wb_km1[i] - p->wb_static[i];  /* :358 */
    }
    orc_quat_to_rot(q_km1, C); /* :359 */
    for (int i = 0; i < 3; ++i)  /* :362 */
        pose_accel[i] = (C[3 * i] * a_nom[0] + C[3 * i + 1] * a_nom[1] + C[3 * i + 2] * a_nom[2]) + p->g[i];

    /* :365-371 nominal state */
    double dw[3] = {dT * w_nom[0], dT * w_nom[1], dT * w_nom[2]}, qe[4], q_check[4];
    orc_quaternion_exp(dw, qe);
    orc_quat_mul(q_km1, qe, q_check);
    orc_quaternion_norm(q_check);
    for (int i = 0; i < 3; ++i) {
        x_check[i] = r_km1[i] + dT * v_km1[i];
        x_check[3 + i] = v_km1[i] + dT * pose_accel[i];
        x_check[10 + i] = ab_km1[i];
        x_check[13 + i] = wb_km1[i];
    }
    for (int i = 0; i < 4; ++i) x_check[6 + i] = q_check[i];

    /* :378-409 Jacobians */
    double F[225], W[15 * 12];
    memset(F, 0, sizeof(F)); memset(W, 0, sizeof(W));
    for (int i = 0; i < n; ++i) F[i * n + i] = 1.0;
    set_block(F, n, 0, 3, I3, 3, 3, dT); /* :380 */
    {
        double S[9], CS[9];
        orc_skew_symm(a_nom, S);
        /* -dT*C*skew parses as ((-dT)*C)*S */
        double mC[9];
        for (int i = 0; i < 9; ++i) mC[i] = -dT * C[i];
        mat_mul(mC, S, CS, 3, 3, 3);
        set_block(F, n, 3, 6, CS, 3, 3, 1.0); /* :381 */
    }
    {
        double w_int_angle = sqrt(dw[0] * dw[0] + dw[1] * dw[1] + dw[2] * dw[2]);
        double Rm[9];
        if (w_int_angle < p->small_ang_tol) { /* :385-389 */
            double S[9];
            orc_skew_symm(dw, S);
            for (int i = 0; i < 9; ++i) Rm[i] = I3[i] - S[i];
        } else { /* :390-395 */
            double axis[3] = {dw[0] / w_int_angle, dw[1] / w_int_angle, dw[2] / w_int_angle};
            orc_angle_axis_to_rot(-w_int_angle, axis, Rm);
        }
        set_block(F, n, 6, 6, Rm, 3, 3, 1.0);
    }
    {
        double mC[9];
        for (int i = 0; i < 9; ++i) mC[i] = -C[i];
        if (p->est_bias) {
            double mdC[9];
            for (int i = 0; i < 9; ++i) mdC[i] = -dT * C[i];
            set_block(F, n, 3, 9, mdC, 3, 3, 1.0);   /* :399 */
            set_block(F, n, 6, 12, I3, 3, 3, -dT);   /* :400 */
        }
        set_block(W, nq, 3, 0, mC, 3, 3, 1.0);       /* :402 / :407 */
        for (int i = 0; i < n - 6; ++i) W[(6 + i) * nq + 3 + i] = 1.0; /* :403 / :408 */
    }

    /* :412-414  P_check = F*P*F^T + W*Q*W^T, dense, transposes materialised */
    double FT[225], WT[12 * 15], FP[225], FPF[225], Qd[144], WQ[15 * 12], WQW[225];
    mat_T(F, FT, n, n);
    mat_T(W, WT, n, nq);
    memset(Qd, 0, sizeof(Qd));
    for (int i = 0; i < nq; ++i) Qd[i * nq + i] = p->Q[i];
    mat_mul(F, P_km1, FP, n, n, n);
    mat_mul(FP, FT, FPF, n, n, n);
    mat_mul(W, Qd, WQ, n, nq, nq);
    mat_mul(WQ, WT, WQW, n, nq, n);
    for (int i = 0; i < n * n; ++i) P_check[i] = FPF[i] + WQW[i];
}

/* -------------------------------------------------------- correction step */
void orc_correction_step(const orc_params *p, const double x_check[16], const double *P_check,
                         const double r_c_tc[3], const double q_ct[4], double x_hat[16], double *P_hat,
                         double r_obs_out[3], double q_obs_out[4])
{ /* EKF.cpp:417-502 */
    const int n = p->num_states;
    const double *r_check = x_check, *v_check = x_check + 3, *q_check = x_check + 6;
    const double *ab_check = x_check + 10, *wb_check = x_check + 13;
    double C_check[9], C_check_T[9];
    orc_quat_to_rot(q_check, C_check);        /* :429 */
    mat_T(C_check, C_check_T, 3, 3);          /* :430 */
    double q_tv_obs[4];
    {
        double t[4];
        orc_quat_mul(p->q_vc, q_ct, t);       /* :431 */
        q_tv_obs[0] = -t[0]; q_tv_obs[1] = -t[1]; q_tv_obs[2] = -t[2]; q_tv_obs[3] = t[3];
        orc_quaternion_norm(q_tv_obs);        /* :432 */
    }
    /* :434-444  -(q * T_vc * r.homogeneous()):  q*T_vc is an affine map with
     * linear C(q)*C_vc and translation C(q)*r_v_cv (Eigen RotationBase * Transform). */
    double r_t_vt_obs[3];
    {
        double Cq[9], L[9], tr[3];
        orc_quat_to_rot(p->direct_orien_method ? q_tv_obs : q_check, Cq);
        mat_mul(Cq, p->C_vc, L, 3, 3, 3);
        mat_mul(Cq, p->r_v_cv, tr, 3, 3, 1);
        for (int i = 0; i < 3; ++i)
            r_t_vt_obs[i] = -((L[3 * i] * r_c_tc[0] + L[3 * i + 1] * r_c_tc[1] + L[3 * i + 2] * r_c_tc[2]) + tr[i]);
    }
    if (r_obs_out) memcpy(r_obs_out, r_t_vt_obs, sizeof(double) * 3);
    if (q_obs_out) memcpy(q_obs_out, q_tv_obs, sizeof(double) * 4);

    /* :447-450 observed perturbation */
    double delta_y[6];
    {
        double qc[4] = {-q_check[0], -q_check[1], -q_check[2], q_check[3]}, dq[4], dth[3];
        for (int i = 0; i < 3; ++i) delta_y[i] = r_t_vt_obs[i] - r_check[i];
        orc_quat_mul(qc, q_tv_obs, dq);
        orc_quaternion_norm(dq);              /* :449 */
        orc_quaternion_log(dq, dth);          /* :450 */
        for (int i = 0; i < 3; ++i) delta_y[3 + i] = dth[i];
    }

    /* :453-470 Jacobians */
    double G[6 * 15], GT[15 * 6], N[36], NT[36];
    memset(G, 0, sizeof(G)); memset(N, 0, sizeof(N));
    set_block(G, n, 0, 0, I3, 3, 3, 1.0);
    if (!p->direct_orien_method) { /* :455-458 */
        double Ctr[3], S[9], CS[9];
        mat_mul(C_check_T, r_check, Ctr, 3, 3, 1);
        orc_skew_symm(Ctr, S);
        mat_mul(C_check, S, CS, 3, 3, 3);
        set_block(G, n, 0, 6, CS, 3, 3, 1.0);
    }
    set_block(G, n, 3, 6, I3, 3, 3, 1.0);
    mat_T(G, GT, 6, n);
    {
        double mC[9], CCv[9];
        for (int i = 0; i < 9; ++i) mC[i] = -C_check[i];
        mat_mul(mC, p->C_vc, CCv, 3, 3, 3);
        set_block(N, 6, 0, 0, CCv, 3, 3, 1.0);
        set_block(N, 6, 3, 3, p->C_vc, 3, 3, 1.0);
        if (p->direct_orien_method) { /* :465-468 */
            double S[9];
            orc_skew_symm(r_check, S);
            set_block(N, 6, 0, 3, S, 3, 3, 1.0);
        }
    }
    mat_T(N, NT, 6, 6);

    /* :472  R_k = N*R*N^T */
    double Rd[36], NR[36], R_k[36];
    memset(Rd, 0, sizeof(Rd));
    for (int i = 0; i < 6; ++i) Rd[i * 6 + i] = p->R[i];
    mat_mul(N, Rd, NR, 6, 6, 6);
    mat_mul(NR, NT, R_k, 6, 6, 6);

    /* :475  K = P*G^T*((G*P*G^T + R_k).inverse()) */
    double GP[6 * 15], S[36], Sinv[36], PGT[15 * 6], K[15 * 6];
    mat_mul(G, P_check, GP, 6, n, n);
    mat_mul(GP, GT, S, 6, n, 6);
    for (int i = 0; i < 36; ++i) S[i] += R_k[i];
    mat_inverse_lu(S, Sinv, 6);
    mat_mul(P_check, GT, PGT, n, n, 6);
    mat_mul(PGT, Sinv, K, n, 6, 6);

    /* :480-481 */
    double KG[225], IKG[225], dx[15];
    mat_mul(K, G, KG, n, 6, n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) IKG[i * n + j] = ((i == j) ? 1.0 : 0.0) - KG[i * n + j];
    mat_mul(IKG, P_check, P_hat, n, n, n);
    memset(dx, 0, sizeof(dx));
    mat_mul(K, delta_y, dx, n, 6, 1);

    /* :486-501 inject */
    double qe[4], q_hat[4];
    orc_quaternion_exp(dx + 6, qe);
    orc_quat_mul(q_check, qe, q_hat);
    orc_quaternion_norm(q_hat);
    for (int i = 0; i < 3; ++i) {
        x_hat[i] = r_check[i] + dx[i];
        x_hat[3 + i] = v_check[i] + dx[3 + i];
        x_hat[10 + i] = p->est_bias ? ab_check[i] + dx[9 + i] : 0.0;
        x_hat[13 + i] = p->est_bias ? wb_check[i] + dx[12 + i] : 0.0;
    }
    for (int i = 0; i < 4; ++i) x_hat[6 + i] = q_hat[i];
}

/* ------------------------------------------------------ seeding and gate */
void orc_seed_pose(const orc_params *p, const double r_c_tc[3], const double q_ct[4],
                   double r_nom[3], double q_nom[4])
{ /* EKF.cpp:310-313 */
    double t[4], Cq[9], L[9], tr[3];
    orc_quat_mul(p->q_vc, q_ct, t);
    q_nom[0] = -t[0]; q_nom[1] = -t[1]; q_nom[2] = -t[2]; q_nom[3] = t[3];
    orc_quaternion_norm(q_nom);
    orc_quat_to_rot(q_nom, Cq);
    mat_mul(Cq, p->C_vc, L, 3, 3, 3);
    mat_mul(Cq, p->r_v_cv, tr, 3, 3, 1);
    for (int i = 0; i < 3; ++i)
        r_nom[i] = -((L[3 * i] * r_c_tc[0] + L[3 * i + 1] * r_c_tc[1] + L[3 * i + 2] * r_c_tc[2]) + tr[i]);
}

int orc_corner_gate(const orc_params *p, const double r_c_tc[3], const double q_ct[4])
{ /* EKF.cpp:154-186: T_ct = Translation(r_c_tc)*q_ct; corners at +-w/2 around tag_positions */
    double C[9];
    orc_quat_to_rot(q_ct, C);
    for (int i = 0; i < p->n_tags; ++i) {
        double hw = p->tag_widths[i] / 2;
        double px = p->tag_positions[3 * i], py = p->tag_positions[3 * i + 1];
        double cx[4] = {hw + px, -hw + px, -hw + px, hw + px};
        double cy[4] = {hw + py, hw + py, -hw + py, -hw + py};
        double minx = 0, miny = 0, maxx = 0, maxy = 0;
        for (int k = 0; k < 4; ++k) {
            /* :163-167 corner z = 0 (tag_positions z is not used), homogeneous 1 */
            double pc[3];
            for (int r = 0; r < 3; ++r) pc[r] = C[3 * r] * cx[k] + C[3 * r + 1] * cy[k] + C[3 * r + 2] * 0.0 + r_c_tc[r];
            double inv_z = 1.0 / pc[2];                                   /* :168 */
            double nx = pc[0] * inv_z, ny = pc[1] * inv_z, nz = pc[2] * inv_z; /* :169 */
            double u = p->camera_K[0] * nx + p->camera_K[1] * ny + p->camera_K[2] * nz; /* :170 */
            double v = p->camera_K[3] * nx + p->camera_K[4] * ny + p->camera_K[5] * nz;
            if (k == 0) { minx = maxx = u; miny = maxy = v; }
            else {
                if (u < minx) minx = u;
                if (u > maxx) maxx = u;
                if (v < miny) miny = v;
                if (v > maxy) maxy = v;
            }
        }
        int ok = (minx > p->camera_width * p->tag_in_view_margin &&
                  miny > p->camera_height * p->tag_in_view_margin &&
                  maxx < p->camera_width * (1 - p->tag_in_view_margin) &&
                  maxy < p->camera_height * (1 - p->tag_in_view_margin)); /* :175-178 */
        if (ok) return 1;
    }
    return 0;
}

/* ------------------------------------------------------------ full filter */
static void hist_reserve(orc_filter *f, int need)
{
    if (need <= f->hist_cap) return;
    int cap = f->hist_cap ? f->hist_cap : 64;
    while (cap < need) cap *= 2;
    f->x_hist = (double *)realloc(f->x_hist, sizeof(double) * 16 * (size_t)cap);
    f->u_hist = (double *)realloc(f->u_hist, sizeof(double) * 6 * (size_t)cap);
    f->P_hist = (double *)realloc(f->P_hist, sizeof(double) * 225 * (size_t)cap);
    f->hist_cap = cap;
}

void orc_filter_init(orc_filter *f, const orc_params *p)
{ /* EKF.cpp:8-85 with the node's parameter overwrite + initialize_params (NODE.cpp:138) */
    memset(f, 0, sizeof(*f));
    f->p = *p;
    orc_initialize_params(&f->p);
    f->apriltag_orien[3] = 1.0; f->q_nom[3] = 1.0; f->q_tv_obs[3] = 1.0;
    int n = f->p.num_states;
    for (int i = 0; i < n; ++i) f->cov_pert[i * n + i] = f->p.cov_init[i]; /* :114 */
}

void orc_filter_free(orc_filter *f)
{
    free(f->x_hist); free(f->u_hist); free(f->P_hist);
    f->x_hist = f->u_hist = f->P_hist = NULL; f->hist_cap = f->hist_len = 0;
}

static void pack_x(const orc_filter *f, double x[16])
{ /* :244-245 */
    memcpy(x, f->r_nom, 24); memcpy(x + 3, f->v_nom, 24); memcpy(x + 6, f->q_nom, 32);
    memcpy(x + 10, f->ab_nom, 24); memcpy(x + 13, f->wb_nom, 24);
}
static void unpack_x(orc_filter *f, const double x[16])
{ /* :273-277 etc. */
    memcpy(f->r_nom, x, 24); memcpy(f->v_nom, x + 3, 24); memcpy(f->q_nom, x + 6, 32);
    memcpy(f->ab_nom, x + 10, 24); memcpy(f->wb_nom, x + 13, 24);
}

void orc_filter_initialize_state(orc_filter *f, int reinit_bias)
{ /* EKF.cpp:305-344 */
    int n = f->p.num_states;
    orc_seed_pose(&f->p, f->apriltag_pos, f->apriltag_orien, f->r_nom, f->q_nom);
    memset(f->v_nom, 0, 24);
    if (reinit_bias) { memset(f->ab_nom, 0, 24); memset(f->wb_nom, 0, 24); }
    memset(f->cov_pert, 0, sizeof(f->cov_pert));
    for (int i = 0; i < n; ++i) f->cov_pert[i * n + i] = f->p.cov_init[i];
    hist_reserve(f, 1);
    memset(f->x_hist, 0, sizeof(double) * 16);
    memcpy(f->x_hist, f->r_nom, 24); /* v = 0 */
    memcpy(f->x_hist + 6, f->q_nom, 32);
    if (f->p.est_bias) { memcpy(f->x_hist + 10, f->ab_nom, 24); memcpy(f->x_hist + 13, f->wb_nom, 24); }
    memset(f->u_hist, 0, sizeof(double) * 6);
    memcpy(f->P_hist, f->cov_pert, sizeof(double) * (size_t)(n * n));
    f->hist_len = 1;
    f->state_initialized = 1;
}

void orc_filter_update(orc_filter *f, double t_curr)
{ /* EKF.cpp:127-303 */
    if (!f->state_initialized) return;
    orc_params *p = &f->p;
    const int n = p->num_states, nn = n * n;
    double u[6], r_c_tc[3] = {0, 0, 0}, q_ct[4] = {0, 0, 0, 1};
    memcpy(u, f->IMU_accel, 24); memcpy(u + 3, f->IMU_ang_vel, 24); /* :138-139 */
    int perform_correction = 0;
    if (f->measurement_ready && (!p->limit_measurement_freq || (f->upds_since_correction + 1) >= p->upd_per_meas)) { /* :147 */
        memcpy(r_c_tc, f->apriltag_pos, 24); memcpy(q_ct, f->apriltag_orien, 32);
        f->measurement_ready = 0;
        perform_correction = p->corner_margin_enbl ? orc_corner_gate(p, r_c_tc, q_ct) : 1; /* :156-186 */
    }
    if (p->multirate_ekf && perform_correction) { /* :196-236 */
        double d = p->dynamic_meas_delay
                       ? fmin(t_curr - f->apriltag_time + p->dyn_measurement_delay_offset, p->measurement_delay_max)
                       : p->measurement_delay;
        f->measurement_delay_curr = d;
        int step = (int)(d / p->dT_nom + 0.5); if (step < 1) step = 1;     /* :200 */
        int ind = f->hist_len - step; if (ind < 0) ind = 0;                /* :201 */
        double xh[16], Ph[225];
        orc_correction_step(p, f->x_hist + 16 * ind, f->P_hist + 225 * ind, r_c_tc, q_ct, xh, Ph, f->r_t_vt_obs, f->q_tv_obs);
        memcpy(f->x_hist + 16 * ind, xh, sizeof(xh));
        memcpy(f->P_hist + 225 * ind, Ph, sizeof(double) * (size_t)nn);
        if (ind > 0) { /* :214-219 */
            int keep = f->hist_len - ind;
            memmove(f->x_hist, f->x_hist + 16 * ind, sizeof(double) * 16 * (size_t)keep);
            memmove(f->u_hist, f->u_hist + 6 * ind, sizeof(double) * 6 * (size_t)keep);
            memmove(f->P_hist, f->P_hist + 225 * ind, sizeof(double) * 225 * (size_t)keep);
            f->hist_len = keep;
        }
        for (int i = 1; i < f->hist_len; ++i) { /* :222-226 */
            double foo[3], xo[16], Po[225];
            orc_prediction_step(p, f->x_hist + 16 * (i - 1), f->P_hist + 225 * (i - 1), f->u_hist + 6 * i, xo, Po, foo);
            memcpy(f->x_hist + 16 * i, xo, sizeof(xo));
            memcpy(f->P_hist + 225 * i, Po, sizeof(double) * (size_t)nn);
        }
        unpack_x(f, f->x_hist + 16 * (f->hist_len - 1));                   /* :229-233 */
        memcpy(f->cov_pert, f->P_hist + 225 * (f->hist_len - 1), sizeof(double) * (size_t)nn);
    }
    double x_km1[16], x_check[16], P_check[225];
    pack_x(f, x_km1);
    orc_prediction_step(p, x_km1, f->cov_pert, u, x_check, P_check, f->accel_rel); /* :249 */
    if (p->multirate_ekf) { /* :251-264 */
        hist_reserve(f, f->hist_len + 1);
        memcpy(f->x_hist + 16 * f->hist_len, x_check, sizeof(x_check));
        memcpy(f->u_hist + 6 * f->hist_len, u, sizeof(u));
        memcpy(f->P_hist + 225 * f->hist_len, P_check, sizeof(double) * (size_t)nn);
        f->hist_len++;
        unpack_x(f, x_check);
        memcpy(f->cov_pert, P_check, sizeof(double) * (size_t)nn);
    } else if (perform_correction) { /* :265-279 */
        double xh[16], Ph[225];
        orc_correction_step(p, x_check, P_check, r_c_tc, q_ct, xh, Ph, f->r_t_vt_obs, f->q_tv_obs);
        unpack_x(f, xh);
        memcpy(f->cov_pert, Ph, sizeof(double) * (size_t)nn);
    } else { /* :280-290 */
        unpack_x(f, x_check);
        memcpy(f->cov_pert, P_check, sizeof(double) * (size_t)nn);
    }
    if (perform_correction) f->upds_since_correction = 0; else f->upds_since_correction += 1; /* :292-299 */
    f->performed_correction = perform_correction;
    f->filter_active = 1;
}

/* ------------------------------------------------------------ batch runner */
int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int64_t orc_run_batch(const orc_params *p, int64_t B, int64_t T, double *x, double *P,
                      const double *u, const double *z, const uint8_t *mask,
                      const double *pfp, int n_threads)
{
    const int n = p->num_states, nn = n * n;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < B; ++i) {
        orc_params lp = *p;
        if (pfp) { /* per-filter Q diag 12, ab_static 3, wb_static 3, R diag 6 */
            const double *q = pfp + 24 * i;
            for (int k = 0; k < 12; ++k) lp.Q[k] = q[k];
            for (int k = 0; k < 3; ++k) { lp.ab_static[k] = q[12 + k]; lp.wb_static[k] = q[15 + k]; }
            for (int k = 0; k < 6; ++k) lp.R[k] = q[18 + k];
        }
        double xc[16], Pc[225], xh[16], Ph[225], acc[3];
        double *xi = x + 16 * i, *Pi = P + (int64_t)nn * i;
        for (int64_t t = 0; t < T; ++t) { /* single-rate branch, EKF.cpp:238-249,265-290 */
            orc_prediction_step(&lp, xi, Pi, u + (t * B + i) * 6, xc, Pc, acc);
            if (mask && mask[t * B + i]) {
                const double *zi = z + (t * B + i) * 7;
                orc_correction_step(&lp, xc, Pc, zi, zi + 3, xh, Ph, NULL, NULL);
                memcpy(xi, xh, sizeof(xh)); memcpy(Pi, Ph, sizeof(double) * (size_t)nn);
            } else {
                memcpy(xi, xc, sizeof(xc)); memcpy(Pi, Pc, sizeof(double) * (size_t)nn);
            }
        }
    }
    return B * T;
}
