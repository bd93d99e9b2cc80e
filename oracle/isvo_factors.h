/*
 * isvo_factors.h -- CPU ORACLE restatement of the IS-VINS cost functions (test infrastructure
 * only; parity unpinned -- see isvo_math.h).  Each function cites the reference lines it follows.
 * Jacobians are written in the reference's own layout: row-major, pose blocks with 7 columns
 * (the 7th is zero), exactly what ceres::CostFunction::Evaluate fills.
 */
#ifndef ISVO_FACTORS_H
#define ISVO_FACTORS_H
#include "isvo_math.h"
#include "../include/isvins_backend.h"

/* ProjectionFactor::Evaluate  src/factor/projection_factor.cpp:24-122.
 * sqrt_info 2x2 row-major; weighted != 0 applies it (Evaluate) else EvaluateOnlyJacobians
 * (:124-196, unweighted).  Any of J_* may be NULL.  J_i,J_j,J_ex are 2x7, J_l is 2x1. */
static inline void isvo_proj_eval(const double *pose_i, const double *pose_j, const double *ex,
                                  double inv_dep_i, const double *pts_i, const double *pts_j,
                                  const double *sqrt_info, int weighted, double *res,
                                  double *J_i, double *J_j, double *J_ex, double *J_l) {
    const double *Pi = pose_i, *Pj = pose_j, *tic = ex;
    quat_t Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j), qic = q_from_pose(ex);
    double pts_camera_i[3] = {pts_i[0] / inv_dep_i, pts_i[1] / inv_dep_i, pts_i[2] / inv_dep_i};
    double pts_imu_i[3], pts_w[3], pts_imu_j[3], pts_camera_j[3], t[3];
    q_rot(qic, pts_camera_i, pts_imu_i);
    for (int k = 0; k < 3; k++) pts_imu_i[k] += tic[k];
    q_rot(Qi, pts_imu_i, pts_w);
    for (int k = 0; k < 3; k++) pts_w[k] += Pi[k];
    for (int k = 0; k < 3; k++) t[k] = pts_w[k] - Pj[k];
    q_rot(q_inv(Qj), t, pts_imu_j);
    for (int k = 0; k < 3; k++) t[k] = pts_imu_j[k] - tic[k];
    q_rot(q_inv(qic), t, pts_camera_j);
    double dep_j = pts_camera_j[2];
    double r0 = pts_camera_j[0] / dep_j - pts_j[0];
    double r1 = pts_camera_j[1] / dep_j - pts_j[1];
    if (weighted) {
        res[0] = sqrt_info[0] * r0 + sqrt_info[1] * r1;
        res[1] = sqrt_info[2] * r0 + sqrt_info[3] * r1;
    } else { res[0] = r0; res[1] = r1; }
    if (!J_i && !J_j && !J_ex && !J_l) return;

    double Ri[9], Rj[9], ric[9], RjT[9], ricT[9];
    q_to_R(Qi, Ri); q_to_R(Qj, Rj); q_to_R(qic, ric);
    m3_t(Rj, RjT); m3_t(ric, ricT);
    double red0[6] = {1. / dep_j, 0, -pts_camera_j[0] / (dep_j * dep_j),
                      0, 1. / dep_j, -pts_camera_j[1] / (dep_j * dep_j)};
    double reduce[6];
    if (weighted) mm(sqrt_info, red0, reduce, 2, 2, 3);
    else memcpy(reduce, red0, sizeof(reduce));
    double A[9]; mm(ricT, RjT, A, 3, 3, 3);            /* ric^T Rj^T */
    if (J_i) {
        double jaco[18], S[9], B[9], C[9];
        skew(pts_imu_i, S);
        for (int k = 0; k < 9; k++) S[k] = -S[k];
        mm(A, Ri, B, 3, 3, 3); mm(B, S, C, 3, 3, 3);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { jaco[r * 6 + c] = A[r * 3 + c]; jaco[r * 6 + 3 + c] = C[r * 3 + c]; }
        double J[12]; mm(reduce, jaco, J, 2, 3, 6);
        for (int r = 0; r < 2; r++) { for (int c = 0; c < 6; c++) J_i[r * 7 + c] = J[r * 6 + c]; J_i[r * 7 + 6] = 0; }
    }
    if (J_j) {
        double jaco[18], S[9], C[9], nRjT[9], D[9];
        for (int k = 0; k < 9; k++) nRjT[k] = -RjT[k];
        mm(ricT, nRjT, D, 3, 3, 3);
        skew(pts_imu_j, S); mm(ricT, S, C, 3, 3, 3);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { jaco[r * 6 + c] = D[r * 3 + c]; jaco[r * 6 + 3 + c] = C[r * 3 + c]; }
        double J[12]; mm(reduce, jaco, J, 2, 3, 6);
        for (int r = 0; r < 2; r++) { for (int c = 0; c < 6; c++) J_j[r * 7 + c] = J[r * 6 + c]; J_j[r * 7 + 6] = 0; }
    }
    if (J_ex) {
        double jaco[18], T1[9], T2[9], tmp_r[9], S[9], S2[9], S3[9], v[3], v2[3], u[3];
        mm(RjT, Ri, T1, 3, 3, 3);
        T1[0] -= 1; T1[4] -= 1; T1[8] -= 1;
        mm(ricT, T1, T2, 3, 3, 3);                      /* ric^T (Rj^T Ri - I) */
        double B[9], C[9];
        mm(A, Ri, B, 3, 3, 3); mm(B, ric, tmp_r, 3, 3, 3);
        skew(pts_camera_i, S); mm(tmp_r, S, C, 3, 3, 3);
        m3v(tmp_r, pts_camera_i, v); skew(v, S2);
        m3v(Ri, tic, u); for (int k = 0; k < 3; k++) u[k] = u[k] + Pi[k] - Pj[k];
        m3v(RjT, u, v2); for (int k = 0; k < 3; k++) v2[k] -= tic[k];
        m3v(ricT, v2, v); skew(v, S3);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
            jaco[r * 6 + c] = T2[r * 3 + c];
            jaco[r * 6 + 3 + c] = -C[r * 3 + c] + S2[r * 3 + c] + S3[r * 3 + c];
        }
        double J[12]; mm(reduce, jaco, J, 2, 3, 6);
        for (int r = 0; r < 2; r++) { for (int c = 0; c < 6; c++) J_ex[r * 7 + c] = J[r * 6 + c]; J_ex[r * 7 + 6] = 0; }
    }
    if (J_l) {
        /* reduce * ric^T * Rj^T * Ri * ric * pts_i * -1.0 / (inv_dep_i * inv_dep_i), left to right */
        double M1[6], M2[6], M3[6], M4[6], v[2];
        mm(reduce, ricT, M1, 2, 3, 3); mm(M1, RjT, M2, 2, 3, 3); mm(M2, Ri, M3, 2, 3, 3); mm(M3, ric, M4, 2, 3, 3);
        mm(M4, pts_i, v, 2, 3, 1);
        J_l[0] = v[0] * -1.0 / (inv_dep_i * inv_dep_i);
        J_l[1] = v[1] * -1.0 / (inv_dep_i * inv_dep_i);
    }
}

/* IntegrationBase::evaluate  include/factor/integration_base.h:160-186 */
static inline void isvo_imu_residual(const isv_imu_t *pre, const double *G, const double *pose_i,
                                     const double *sb_i, const double *pose_j, const double *sb_j,
                                     double *r) {
    const double *Pi = pose_i, *Pj = pose_j, *Vi = sb_i, *Bai = sb_i + 3, *Bgi = sb_i + 6;
    const double *Vj = sb_j, *Baj = sb_j + 3, *Bgj = sb_j + 6;
    quat_t Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j);
    const double *Jm = pre->jacobian;
    double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        dp_dba[a * 3 + b] = Jm[(0 + a) * 15 + 9 + b];
        dp_dbg[a * 3 + b] = Jm[(0 + a) * 15 + 12 + b];
        dq_dbg[a * 3 + b] = Jm[(3 + a) * 15 + 12 + b];
        dv_dba[a * 3 + b] = Jm[(6 + a) * 15 + 9 + b];
        dv_dbg[a * 3 + b] = Jm[(6 + a) * 15 + 12 + b];
    }
    double dba[3], dbg[3], t[3], t2[3], cdp[3], cdv[3];
    for (int k = 0; k < 3; k++) { dba[k] = Bai[k] - pre->linearized_ba[k]; dbg[k] = Bgi[k] - pre->linearized_bg[k]; }
    quat_t dq = {pre->delta_q[3], pre->delta_q[0], pre->delta_q[1], pre->delta_q[2]};
    m3v(dq_dbg, dbg, t);
    quat_t cdq = q_mul(dq, q_delta(t));
    m3v(dv_dba, dba, t); m3v(dv_dbg, dbg, t2);
    for (int k = 0; k < 3; k++) cdv[k] = pre->delta_v[k] + t[k] + t2[k];
    m3v(dp_dba, dba, t); m3v(dp_dbg, dbg, t2);
    for (int k = 0; k < 3; k++) cdp[k] = pre->delta_p[k] + t[k] + t2[k];
    double dt = pre->sum_dt, u[3], o[3];
    quat_t Qii = q_inv(Qi);
    for (int k = 0; k < 3; k++) u[k] = 0.5 * G[k] * dt * dt + Pj[k] - Pi[k] - Vi[k] * dt;
    q_rot(Qii, u, o);
    for (int k = 0; k < 3; k++) r[0 + k] = o[k] - cdp[k];
    quat_t e = q_mul(q_inv(cdq), q_mul(Qii, Qj));
    r[3] = 2 * e.x; r[4] = 2 * e.y; r[5] = 2 * e.z;
    for (int k = 0; k < 3; k++) u[k] = G[k] * dt + Vj[k] - Vi[k];
    q_rot(Qii, u, o);
    for (int k = 0; k < 3; k++) r[6 + k] = o[k] - cdv[k];
    for (int k = 0; k < 3; k++) { r[9 + k] = Baj[k] - Bai[k]; r[12 + k] = Bgj[k] - Bgi[k]; }
}

/* sqrt_info = LLT(covariance.inverse()).matrixL().transpose()  imu_factor.h:44 */
static inline int isvo_imu_sqrt_info(const double *cov, double *sqrt_info) {
    double inv[225], L[225];
    int bad = inv_partial_lu(cov, inv, 15);
    memcpy(L, inv, sizeof(L));
    /* Eigen LLT reads the lower triangle of the (numerically slightly unsymmetric) inverse */
    int info = chol_lower(L, 15);
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) sqrt_info[i * 15 + j] = L[j * 15 + i];
    return bad || info;
}

/* IMUFactor::Evaluate  include/factor/imu_factor.h:23-159 (weighted != 0, Ceres overload)
 * and the void overload :161-265 (weighted == 0: unweighted residual/Jacobians, 15x6 / 15x9).
 * Jacobians here always have 7 / 9 / 7 / 9 columns (7th pose column zero). */
static inline void isvo_imu_eval(const isv_imu_t *pre, const double *G, const double *pose_i,
                                 const double *sb_i, const double *pose_j, const double *sb_j,
                                 const double *sqrt_info /*15x15 or NULL*/, double *res,
                                 double *Jpi, double *Jsi, double *Jpj, double *Jsj) {
    double r[15];
    isvo_imu_residual(pre, G, pose_i, sb_i, pose_j, sb_j, r);
    if (sqrt_info) mm(sqrt_info, r, res, 15, 15, 1); else memcpy(res, r, sizeof(r));
    if (!Jpi && !Jsi && !Jpj && !Jsj) return;
    const double *Pi = pose_i, *Pj = pose_j, *Vi = sb_i, *Bgi = sb_i + 6, *Vj = sb_j;
    quat_t Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j);
    double dt = pre->sum_dt;
    const double *Jm = pre->jacobian;
    double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        dp_dba[a * 3 + b] = Jm[(0 + a) * 15 + 9 + b];
        dp_dbg[a * 3 + b] = Jm[(0 + a) * 15 + 12 + b];
        dq_dbg[a * 3 + b] = Jm[(3 + a) * 15 + 12 + b];
        dv_dba[a * 3 + b] = Jm[(6 + a) * 15 + 9 + b];
        dv_dbg[a * 3 + b] = Jm[(6 + a) * 15 + 12 + b];
    }
    quat_t dq = {pre->delta_q[3], pre->delta_q[0], pre->delta_q[1], pre->delta_q[2]};
    quat_t Qii = q_inv(Qi);
    double RiT[9]; q_to_R(Qii, RiT);                       /* Qi.inverse().toRotationMatrix() */
    double dbg[3], t[3];
    for (int k = 0; k < 3; k++) dbg[k] = Bgi[k] - pre->linearized_bg[k];
    m3v(dq_dbg, dbg, t);
    quat_t cdq = q_mul(dq, q_delta(t));
    double raw[15 * 9], W[15 * 9];
#define BLK(M, ld, r0, c0, B, sgn) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) (M)[((r0) + a) * (ld) + (c0) + b] = (sgn) * (B)[a * 3 + b]
    if (Jpi) {
        memset(raw, 0, sizeof(double) * 15 * 7);
        BLK(raw, 7, 0, 0, RiT, -1.0);
        double u[3], o[3], S[9];
        for (int k = 0; k < 3; k++) u[k] = 0.5 * G[k] * dt * dt + Pj[k] - Pi[k] - Vi[k] * dt;
        q_rot(Qii, u, o); skew(o, S); BLK(raw, 7, 0, 3, S, 1.0);
        double L4[16], R4[16], P4[16], B33[9];
        qleft44(q_mul(q_inv(Qj), Qi), L4); qright44(cdq, R4); mm(L4, R4, P4, 4, 4, 4);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) B33[a * 3 + b] = P4[(a + 1) * 4 + b + 1];
        BLK(raw, 7, 3, 3, B33, -1.0);
        for (int k = 0; k < 3; k++) u[k] = G[k] * dt + Vj[k] - Vi[k];
        q_rot(Qii, u, o); skew(o, S); BLK(raw, 7, 6, 3, S, 1.0);
        if (sqrt_info) { mm(sqrt_info, raw, W, 15, 15, 7); memcpy(Jpi, W, sizeof(double) * 105); }
        else memcpy(Jpi, raw, sizeof(double) * 105);
    }
    if (Jsi) {
        memset(raw, 0, sizeof(double) * 15 * 9);
        double M[9], I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int k = 0; k < 9; k++) M[k] = -RiT[k] * dt;
        BLK(raw, 9, 0, 0, M, 1.0);
        BLK(raw, 9, 0, 3, dp_dba, -1.0);
        BLK(raw, 9, 0, 6, dp_dbg, -1.0);
        double L33[9], T[9];
        qleft33(q_mul(q_mul(q_inv(Qj), Qi), dq), L33);      /* uncorrected delta_q, :105 */
        for (int k = 0; k < 9; k++) L33[k] = -L33[k];
        mm(L33, dq_dbg, T, 3, 3, 3);
        BLK(raw, 9, 3, 6, T, 1.0);
        BLK(raw, 9, 6, 0, RiT, -1.0);
        BLK(raw, 9, 6, 3, dv_dba, -1.0);
        BLK(raw, 9, 6, 6, dv_dbg, -1.0);
        BLK(raw, 9, 9, 3, I3, -1.0);
        BLK(raw, 9, 12, 6, I3, -1.0);
        if (sqrt_info) { mm(sqrt_info, raw, W, 15, 15, 9); memcpy(Jsi, W, sizeof(double) * 135); }
        else memcpy(Jsi, raw, sizeof(double) * 135);
    }
    if (Jpj) {
        memset(raw, 0, sizeof(double) * 15 * 7);
        BLK(raw, 7, 0, 0, RiT, 1.0);
        double L33[9];
        qleft33(q_mul(q_mul(q_inv(cdq), Qii), Qj), L33);
        BLK(raw, 7, 3, 3, L33, 1.0);
        if (sqrt_info) { mm(sqrt_info, raw, W, 15, 15, 7); memcpy(Jpj, W, sizeof(double) * 105); }
        else memcpy(Jpj, raw, sizeof(double) * 105);
    }
    if (Jsj) {
        memset(raw, 0, sizeof(double) * 15 * 9);
        double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        BLK(raw, 9, 6, 0, RiT, 1.0);
        BLK(raw, 9, 9, 3, I3, 1.0);
        BLK(raw, 9, 12, 6, I3, 1.0);
        if (sqrt_info) { mm(sqrt_info, raw, W, 15, 15, 9); memcpy(Jsj, W, sizeof(double) * 135); }
        else memcpy(Jsj, raw, sizeof(double) * 135);
    }
#undef BLK
}

/* IntegrationBase::midPointIntegration + propagate  integration_base.h:54-158.
 * State: the isv_imu_t being built (delta_*, jacobian, covariance, sum_dt) plus acc_0/gyr_0.
 * noise = diag(ACC_N^2, GYR_N^2, ACC_N^2, GYR_N^2, ACC_W^2, GYR_W^2) (x) I3  (:21-27) */
static inline void isvo_preint_init(isv_imu_t *p, const double *ba, const double *bg) {
    memset(p, 0, sizeof(*p));
    p->delta_q[3] = 1.0;
    for (int i = 0; i < 15; i++) p->jacobian[i * 15 + i] = 1.0;
    for (int k = 0; k < 3; k++) { p->linearized_ba[k] = ba[k]; p->linearized_bg[k] = bg[k]; }
}
static inline void isvo_preint_step(isv_imu_t *p, double dt, const double *acc_0, const double *gyr_0,
                                    const double *acc_1, const double *gyr_1, const double *noise4
                                    /* ACC_N, GYR_N, ACC_W, GYR_W */) {
    const double *lba = p->linearized_ba, *lbg = p->linearized_bg;
    quat_t dq = {p->delta_q[3], p->delta_q[0], p->delta_q[1], p->delta_q[2]};
    double a0[3], a1[3], ung[3], un_acc_0[3], un_acc_1[3], un_acc[3];
    for (int k = 0; k < 3; k++) { a0[k] = acc_0[k] - lba[k]; a1[k] = acc_1[k] - lba[k]; ung[k] = 0.5 * (gyr_0[k] + gyr_1[k]) - lbg[k]; }
    q_rot(dq, a0, un_acc_0);
    quat_t inc = {1, ung[0] * dt / 2, ung[1] * dt / 2, ung[2] * dt / 2};
    quat_t rdq = q_mul(dq, inc);
    q_rot(rdq, a1, un_acc_1);
    double rdp[3], rdv[3];
    for (int k = 0; k < 3; k++) {
        un_acc[k] = 0.5 * (un_acc_0[k] + un_acc_1[k]);
        rdp[k] = p->delta_p[k] + p->delta_v[k] * dt + 0.5 * un_acc[k] * dt * dt;
        rdv[k] = p->delta_v[k] + un_acc[k] * dt;
    }
    /* jacobian / covariance propagation :75-126 */
    double R_w_x[9], R_a_0_x[9], R_a_1_x[9], Rd[9], Rr[9];
    skew(ung, R_w_x); skew(a0, R_a_0_x); skew(a1, R_a_1_x);
    q_to_R(dq, Rd); q_to_R(rdq, Rr);
    double F[225], V[15 * 18];
    memset(F, 0, sizeof(F)); memset(V, 0, sizeof(V));
    double ImW[9], T1[9], T2[9], T3[9];
    for (int k = 0; k < 9; k++) ImW[k] = -R_w_x[k] * dt;
    ImW[0] += 1; ImW[4] += 1; ImW[8] += 1;
    mm(Rd, R_a_0_x, T1, 3, 3, 3);                 /* Rd [a0]x */
    mm(Rr, R_a_1_x, T2, 3, 3, 3);                 /* Rr [a1]x */
    mm(T2, ImW, T3, 3, 3, 3);                     /* Rr [a1]x (I - [w]x dt) */
#define FB(r0, c0) F[((r0) + a) * 15 + (c0) + b]
#define VB(r0, c0) V[((r0) + a) * 18 + (c0) + b]
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        double I = (a == b);
        FB(0, 0) = I;
        FB(0, 3) = -0.25 * T1[a * 3 + b] * dt * dt + -0.25 * T3[a * 3 + b] * dt * dt;
        FB(0, 6) = I * dt;
        FB(0, 9) = -0.25 * (Rd[a * 3 + b] + Rr[a * 3 + b]) * dt * dt;
        FB(0, 12) = -0.25 * T2[a * 3 + b] * dt * dt * -dt;
        FB(3, 3) = ImW[a * 3 + b];
        FB(3, 12) = -1.0 * I * dt;
        FB(6, 3) = -0.5 * T1[a * 3 + b] * dt + -0.5 * T3[a * 3 + b] * dt;
        FB(6, 6) = I;
        FB(6, 9) = -0.5 * (Rd[a * 3 + b] + Rr[a * 3 + b]) * dt;
        FB(6, 12) = -0.5 * T2[a * 3 + b] * dt * -dt;
        FB(9, 9) = I;
        FB(12, 12) = I;
        VB(0, 0) = 0.25 * Rd[a * 3 + b] * dt * dt;
        VB(0, 3) = 0.25 * -T2[a * 3 + b] * dt * dt * 0.5 * dt;
        VB(0, 6) = 0.25 * Rr[a * 3 + b] * dt * dt;
        VB(0, 9) = VB(0, 3);
        VB(3, 3) = 0.5 * I * dt;
        VB(3, 9) = 0.5 * I * dt;
        VB(6, 0) = 0.5 * Rd[a * 3 + b] * dt;
        VB(6, 3) = 0.5 * -T2[a * 3 + b] * dt * 0.5 * dt;
        VB(6, 6) = 0.5 * Rr[a * 3 + b] * dt;
        VB(6, 9) = VB(6, 3);
        VB(9, 12) = I * dt;
        VB(12, 15) = I * dt;
    }
#undef FB
#undef VB
    double nd[18];
    for (int k = 0; k < 3; k++) {
        nd[k] = noise4[0] * noise4[0]; nd[3 + k] = noise4[1] * noise4[1];
        nd[6 + k] = noise4[0] * noise4[0]; nd[9 + k] = noise4[1] * noise4[1];
        nd[12 + k] = noise4[2] * noise4[2]; nd[15 + k] = noise4[3] * noise4[3];
    }
    double Jn[225], FC[225], Cn[225], VN[15 * 18], VNV[225];
    mm(F, p->jacobian, Jn, 15, 15, 15);
    mm(F, p->covariance, FC, 15, 15, 15);
    mm_nt(FC, F, Cn, 15, 15, 15);
    for (int i = 0; i < 15; i++) for (int j = 0; j < 18; j++) VN[i * 18 + j] = V[i * 18 + j] * nd[j];
    mm_nt(VN, V, VNV, 15, 18, 15);
    for (int i = 0; i < 225; i++) { p->jacobian[i] = Jn[i]; p->covariance[i] = Cn[i] + VNV[i]; }
    rdq = q_normalized(rdq);                       /* delta_q.normalize()  :153 */
    for (int k = 0; k < 3; k++) { p->delta_p[k] = rdp[k]; p->delta_v[k] = rdv[k]; }
    p->delta_q[0] = rdq.x; p->delta_q[1] = rdq.y; p->delta_q[2] = rdq.z; p->delta_q[3] = rdq.w;
    p->sum_dt += dt;
}

/* SE3PriorFactor::Evaluate  se3_prior_factor.h:21-53 ; J is 6x7 (or NULL).  sqrt_info NULL ->
 * EvaluateOnlyJacobians (:55-71): unweighted. */
static inline void isvo_se3prior_eval(const isv_se3_prior_t *f, const double *sqrt_info,
                                      const double *pose, double *res, double *J) {
    quat_t ri = so3_from_q(q_from_pose(pose)), rp = so3_from_R(f->R);
    quat_t res_r = so3_mul(so3_inv(rp), ri);
    double r[6], lg[3];
    so3_log(res_r, lg);
    for (int k = 0; k < 3; k++) { r[k] = pose[k] - f->t[k]; r[3 + k] = lg[k]; }
    if (sqrt_info) mm(sqrt_info, r, res, 6, 6, 1); else memcpy(res, r, sizeof(r));
    if (!J) return;
    double raw[42], Jr[9];
    memset(raw, 0, sizeof(raw));
    raw[0] = raw[8] = raw[16] = 1.0;
    so3_rjac_inv(lg, Jr);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) raw[(3 + a) * 7 + 3 + b] = Jr[a * 3 + b];
    if (sqrt_info) mm(sqrt_info, raw, J, 6, 6, 7); else memcpy(J, raw, sizeof(raw));
}
/* SE3PriorFactor::update  se3_prior_factor.h:73-81 */
static inline void isvo_se3prior_update(isv_se3_prior_t *f, const double *P_old, const double *R_old,
                                        const double *pose_new) {
    quat_t R0 = so3_from_R(R_old), R1 = so3_from_q(q_from_pose(pose_new));
    double dR[3], E[9], Rn[9];
    so3_log(so3_mul(so3_inv(R1), R0), dR);
    for (int k = 0; k < 3; k++) f->t[k] += pose_new[k] - P_old[k];
    q_to_R(so3_exp(dR), E); mm(f->R, E, Rn, 3, 3, 3);
    memcpy(f->R, Rn, sizeof(Rn));
}

/* Linear9Factor::Evaluate  linear9_factor.h:20-44 ; J 9x9 */
static inline void isvo_linear9_eval(const isv_linear9_t *f, const double *sqrt_info, const double *sb,
                                     double *res, double *J) {
    double r[9];
    for (int k = 0; k < 9; k++) r[k] = sb[k] - f->VB[k];
    if (sqrt_info) mm(sqrt_info, r, res, 9, 9, 1); else memcpy(res, r, sizeof(r));
    if (!J) return;
    if (sqrt_info) memcpy(J, sqrt_info, sizeof(double) * 81);
    else { memset(J, 0, sizeof(double) * 81); for (int k = 0; k < 9; k++) J[k * 9 + k] = 1; }
}
/* Linear9Factor::update  linear9_factor.h:60-68 */
static inline void isvo_linear9_update(isv_linear9_t *f, const double *V, const double *Ba, const double *Bg,
                                       const double *sb_new) {
    double old[9] = {V[0], V[1], V[2], Ba[0], Ba[1], Ba[2], Bg[0], Bg[1], Bg[2]};
    for (int k = 0; k < 9; k++) f->VB[k] += sb_new[k] - old[k];
}

/* RelativePoseFactor::Evaluate  relative_pose_factor.h:27-70 ; Ji, Jj 6x7 */
static inline void isvo_relpose_eval(const isv_relpose_t *f, const double *sqrt_info, const double *pose_i,
                                     const double *pose_j, double *res, double *Ji, double *Jj) {
    const double *Pi = pose_i, *Pj = pose_j;
    quat_t Qi = q_from_pose(pose_i), Qj = q_from_pose(pose_j);
    double Ri[9], Rj[9], RjT[9], RiT[9], d[3], qd[3], M1[9], M2[9], lg[3], r[6];
    q_to_R(Qi, Ri); q_to_R(Qj, Rj); m3_t(Rj, RjT); m3_t(Ri, RiT);
    for (int k = 0; k < 3; k++) d[k] = Pj[k] - Pi[k];
    q_rot(q_inv(Qi), d, qd);
    mm(f->delta_R, RjT, M1, 3, 3, 3); mm(M1, Ri, M2, 3, 3, 3);
    quat_t res_R = so3_from_R(M2);
    so3_log(res_R, lg);
    for (int k = 0; k < 3; k++) { r[k] = f->delta_t[k] - qd[k]; r[3 + k] = lg[k]; }
    if (sqrt_info) mm(sqrt_info, r, res, 6, 6, 1); else memcpy(res, r, sizeof(r));
    if (!Ji && !Jj) return;
    double J[9]; so3_rjac_inv(lg, J);
    if (Ji) {
        double raw[42], S[9]; memset(raw, 0, sizeof(raw));
        skew(qd, S);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            raw[a * 7 + b] = RiT[a * 3 + b];
            raw[a * 7 + 3 + b] = -S[a * 3 + b];
            raw[(3 + a) * 7 + 3 + b] = J[a * 3 + b];
        }
        if (sqrt_info) mm(sqrt_info, raw, Ji, 6, 6, 7); else memcpy(Ji, raw, sizeof(raw));
    }
    if (Jj) {
        double raw[42], nJ[9], T1[9], T2[9]; memset(raw, 0, sizeof(raw));
        for (int k = 0; k < 9; k++) nJ[k] = -J[k];
        mm(nJ, RiT, T1, 3, 3, 3); mm(T1, Rj, T2, 3, 3, 3);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            raw[a * 7 + b] = -RiT[a * 3 + b];
            raw[(3 + a) * 7 + 3 + b] = T2[a * 3 + b];
        }
        if (sqrt_info) mm(sqrt_info, raw, Jj, 6, 6, 7); else memcpy(Jj, raw, sizeof(raw));
    }
}
/* RelativePoseFactor::update (solver overload)  relative_pose_factor.h:103-117 */
static inline void isvo_relpose_update(isv_relpose_t *f, const double *ti, const double *Ri, const double *tj,
                                       const double *Rj, const double *PSi, const double *PSj) {
    quat_t Qi = q_from_pose(PSi), Qj = q_from_pose(PSj);
    double d_tj[3], d_ti[3], A[9], B[9], RiT[9];
    for (int k = 0; k < 3; k++) { d_tj[k] = PSj[k] - tj[k]; d_ti[k] = PSi[k] - ti[k]; }
    /* Sophus::SO3d(Qj.inverse()*Rj): Eigen promotes to a matrix product */
    double Qm[9];
    q_to_R(q_inv(Qj), Qm); mm(Qm, Rj, A, 3, 3, 3); quat_t d_Rj = so3_from_R(A);
    q_to_R(q_inv(Qi), Qm); mm(Qm, Ri, B, 3, 3, 3); quat_t d_Ri = so3_from_R(B);
    double lgi[3], lgj[3], v1[3], v2[3], S[9], v3[3];
    so3_log(d_Ri, lgi); so3_log(d_Rj, lgj);
    m3_t(Ri, RiT);
    m3v(RiT, d_tj, v1); m3v(RiT, d_ti, v2);
    skew(f->delta_t, S); m3v(S, lgi, v3);
    for (int k = 0; k < 3; k++) f->delta_t[k] += v1[k] - v2[k] + v3[k];
    double Ji[9], w[3], E[9], T[9];
    q_to_R(q_mul(q_inv(Qj), Qi), Ji);
    for (int k = 0; k < 9; k++) Ji[k] = -Ji[k];
    m3v(Ji, lgi, w);
    q_to_R(so3_exp(w), E); mm(f->delta_R, E, T, 3, 3, 3); memcpy(f->delta_R, T, sizeof(T));
    q_to_R(so3_exp(lgj), E); mm(f->delta_R, E, T, 3, 3, 3); memcpy(f->delta_R, T, sizeof(T));
}

/* RollPitchFactor::Evaluate  rollpitch_factor.h:26-57 ; J 2x7 */
static inline void isvo_rollpitch_eval(const isv_rollpitch_t *f, const double *sqrt_info, const double *pose,
                                       double *res, double *J) {
    quat_t Ri = so3_from_q(q_from_pose(pose)), Rm = so3_from_R(f->R);
    double nZ[3] = {0, 0, -1.0}, v[3];
    quat_t prod = so3_mul(Rm, so3_inv(Ri));   /* (Rmeas * Ri.inverse()) * nZ, left to right */
    q_rot(prod, nZ, v);
    double r[2] = {v[0], v[1]};
    if (sqrt_info) mm(sqrt_info, r, res, 2, 2, 1); else { res[0] = r[0]; res[1] = r[1]; }
    if (!J) return;
    double S[9], Rmm[9], B[9], raw[14];
    memset(raw, 0, sizeof(raw));
    skew(v, S); q_to_R(Rm, Rmm); mm(S, Rmm, B, 3, 3, 3);
    for (int a = 0; a < 2; a++) for (int b = 0; b < 3; b++) raw[a * 7 + 3 + b] = B[a * 3 + b];
    if (sqrt_info) mm(sqrt_info, raw, J, 2, 2, 7); else memcpy(J, raw, sizeof(raw));
}
/* RollPitchFactor::update  rollpitch_factor.h:78-83 */
static inline void isvo_rollpitch_update(isv_rollpitch_t *f, const double *R_old, const double *pose_new) {
    quat_t R0 = so3_from_R(R_old), R1 = so3_from_q(q_from_pose(pose_new));
    double dR[3], E[9], T[9];
    so3_log(so3_mul(so3_inv(R1), R0), dR);
    q_to_R(so3_exp(dR), E); mm(f->R, E, T, 3, 3, 3); memcpy(f->R, T, sizeof(T));
}

/* YawFactor ctor + EvaluateOnlyJacobians  yaw_factor.h:15-19, 51-65 ; J 1x6 */
static inline void isvo_yaw_jac(const double *pose /* Rz = Qw of this pose */, double *res, double *J6) {
    quat_t Rz = q_from_pose(pose);
    double ex[3] = {1, 0, 0}, yaw_meas[3];
    q_rot(q_inv(Rz), ex, yaw_meas);
    quat_t Ri = so3_from_q(q_from_pose(pose));
    double v[3], Rm[9], S[9], B[9];
    q_rot(Ri, yaw_meas, v);
    if (res) res[0] = v[1];
    q_to_R(Ri, Rm); skew(yaw_meas, S);
    for (int k = 0; k < 9; k++) Rm[k] = -Rm[k];
    mm(Rm, S, B, 3, 3, 3);
    J6[0] = J6[1] = J6[2] = 0;
    for (int b = 0; b < 3; b++) J6[3 + b] = B[1 * 3 + b];
}

/* PoseLocalParameterization::Plus  src/factor/pose_local_parameterization.cpp:3-19 */
static inline void isvo_pose_plus(const double *x, const double *delta, double *xp) {
    quat_t q = {x[6], x[3], x[4], x[5]};
    quat_t r = q_normalized(q_mul(q, q_delta(delta + 3)));
    for (int k = 0; k < 3; k++) xp[k] = x[k] + delta[k];
    xp[3] = r.x; xp[4] = r.y; xp[5] = r.z; xp[6] = r.w;
}
#endif
