// isv_proj_factor.h -- ProjectionFactor::Evaluate geometry for one lane
// (reference src/factor/projection_factor.cpp:24-122), shared by the linearise and marginalisation kernels.
#pragma once
#include "isv_device_math.h"

// Reprojection factor geometry for one lane.  R*/P* come from LDS (staged per wave).
template <bool JAC>
DEV void proj_factor(const double *Ri, const double *Pi, const double *Rj, const double *Pj,
                     const double *ric, const double *tic, const double *sq, double lam,
                     double pix, double piy, double piz, double pjx, double pjy,
                     double &r0, double &r1, double *Ji, double *Jj, double *Jl) {
    double inv = 1.0 / lam;
    double pc[3] = {pix * inv, piy * inv, piz * inv};           // pts_camera_i = pts_i / inv_dep_i
    double pb[3], pw[3], t[3], pbj[3], pcj[3];
    m3v(ric, pc, pb); pb[0] += tic[0]; pb[1] += tic[1]; pb[2] += tic[2];     // pts_imu_i
    m3v(Ri, pb, pw);
    t[0] = pw[0] + Pi[0] - Pj[0]; t[1] = pw[1] + Pi[1] - Pj[1]; t[2] = pw[2] + Pi[2] - Pj[2];
    m3tv(Rj, t, pbj);                                                        // pts_imu_j
    t[0] = pbj[0] - tic[0]; t[1] = pbj[1] - tic[1]; t[2] = pbj[2] - tic[2];
    m3tv(ric, t, pcj);                                                       // pts_camera_j
    double idep = 1.0 / pcj[2];
    double u0 = pcj[0] * idep - pjx, u1 = pcj[1] * idep - pjy;
    r0 = sq[0] * u0 + sq[1] * u1;
    r1 = sq[2] * u0 + sq[3] * u1;
    if (!JAC) return;
    // reduce = sqrt_info * [[1/z, 0, -x/z^2],[0, 1/z, -y/z^2]]
    double a0 = idep, a2 = -pcj[0] * idep * idep, b2 = -pcj[1] * idep * idep;
    double red[6] = {sq[0] * a0, sq[1] * a0, sq[0] * a2 + sq[1] * b2,
                     sq[2] * a0, sq[3] * a0, sq[2] * a2 + sq[3] * b2};
    double A[9], RA[6];
    // A := ric^T * Rj^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[i * 3 + j] = ric[i] * Rj[j * 3] + ric[3 + i] * Rj[j * 3 + 1] + ric[6 + i] * Rj[j * 3 + 2];
    // RA = reduce * A   (2x3)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) RA[i * 3 + j] = red[i * 3] * A[j] + red[i * 3 + 1] * A[3 + j] + red[i * 3 + 2] * A[6 + j];
    // J_pose_i = [RA, -RA Ri [pts_imu_i]x]
    double RB[6];                          // RA * Ri
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) RB[i * 3 + j] = RA[i * 3] * Ri[j] + RA[i * 3 + 1] * Ri[3 + j] + RA[i * 3 + 2] * Ri[6 + j];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        Ji[i * 6 + 0] = RA[i * 3 + 0]; Ji[i * 6 + 1] = RA[i * 3 + 1]; Ji[i * 6 + 2] = RA[i * 3 + 2];
        // -(row) * skew(pb): row*S = [r1*pb2 - r2*pb1, r2*pb0 - r0*pb2, r0*pb1 - r1*pb0]
        double x = RB[i * 3], y = RB[i * 3 + 1], z = RB[i * 3 + 2];
        Ji[i * 6 + 3] = -(y * pb[2] - z * pb[1]);
        Ji[i * 6 + 4] = -(z * pb[0] - x * pb[2]);
        Ji[i * 6 + 5] = -(x * pb[1] - y * pb[0]);
    }
    // J_pose_j = [-RA, reduce ric^T [pts_imu_j]x]
    double RC[6];                          // reduce * ric^T
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) RC[i * 3 + j] = red[i * 3] * ric[j * 3] + red[i * 3 + 1] * ric[j * 3 + 1] + red[i * 3 + 2] * ric[j * 3 + 2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        Jj[i * 6 + 0] = -RA[i * 3 + 0]; Jj[i * 6 + 1] = -RA[i * 3 + 1]; Jj[i * 6 + 2] = -RA[i * 3 + 2];
        double x = RC[i * 3], y = RC[i * 3 + 1], z = RC[i * 3 + 2];
        Jj[i * 6 + 3] = y * pbj[2] - z * pbj[1];
        Jj[i * 6 + 4] = z * pbj[0] - x * pbj[2];
        Jj[i * 6 + 5] = x * pbj[1] - y * pbj[0];
    }
    // J_lambda = reduce ric^T Rj^T Ri ric pts_i * -1/lambda^2 = RB * (ric * pts_i) * -inv^2
    double v[3] = {ric[0] * pix + ric[1] * piy + ric[2] * piz, ric[3] * pix + ric[4] * piy + ric[5] * piz,
                   ric[6] * pix + ric[7] * piy + ric[8] * piz};
    double s = -inv * inv;
    Jl[0] = (RB[0] * v[0] + RB[1] * v[1] + RB[2] * v[2]) * s;
    Jl[1] = (RB[3] * v[0] + RB[4] * v[1] + RB[5] * v[2]) * s;
}


// Residual at x and its directional derivative m = J delta along the tangent step (delta P_i, delta theta_i,
// delta P_j, delta theta_j, delta lambda), WITHOUT forming the Jacobian blocks: the chain rule is applied to
// the step itself (a handful of 3-vector products instead of the 2x6 / 2x6 / 2x1 blocks and their dot
// products).  Same perturbation conventions as proj_factor / PoseLocalParameterization (R <- R Exp(dtheta)).
DEV void proj_residual_dir(const double *Ri, const double *Pi, const double *Rj, const double *Pj,
                           const double *ric, const double *tic, const double *sq, double lam,
                           double pix, double piy, double piz, double pjx, double pjy,
                           const double *di, const double *dj, double dlam,
                           double &r0, double &r1, double &m0, double &m1) {
    const double inv = 1.0 / lam;
    double pc[3] = {pix * inv, piy * inv, piz * inv};
    double pb[3], pw[3], t[3], pbj[3], pcj[3];
    m3v(ric, pc, pb); pb[0] += tic[0]; pb[1] += tic[1]; pb[2] += tic[2];
    m3v(Ri, pb, pw);
    t[0] = pw[0] + Pi[0] - Pj[0]; t[1] = pw[1] + Pi[1] - Pj[1]; t[2] = pw[2] + Pi[2] - Pj[2];
    m3tv(Rj, t, pbj);
    t[0] = pbj[0] - tic[0]; t[1] = pbj[1] - tic[1]; t[2] = pbj[2] - tic[2];
    m3tv(ric, t, pcj);
    const double idep = 1.0 / pcj[2];
    const double u0 = pcj[0] * idep - pjx, u1 = pcj[1] * idep - pjy;
    r0 = sq[0] * u0 + sq[1] * u1;
    r1 = sq[2] * u0 + sq[3] * u1;
    // d pc = -pc dlam / lam;  d pb = ric d pc;  d pw = Ri (d pb + dtheta_i x pb) + dP_i
    const double sl = -dlam * inv;
    double dpc[3] = {pc[0] * sl, pc[1] * sl, pc[2] * sl}, dpb[3], dw[3], dbj[3], dcj[3];
    m3v(ric, dpc, dpb);
    dpb[0] += di[4] * pb[2] - di[5] * pb[1];
    dpb[1] += di[5] * pb[0] - di[3] * pb[2];
    dpb[2] += di[3] * pb[1] - di[4] * pb[0];
    m3v(Ri, dpb, dw);
    dw[0] += di[0] - dj[0]; dw[1] += di[1] - dj[1]; dw[2] += di[2] - dj[2];
    // d pbj = Rj^T dw + pbj x dtheta_j;  d pcj = ric^T d pbj
    m3tv(Rj, dw, dbj);
    dbj[0] += pbj[1] * dj[5] - pbj[2] * dj[4];
    dbj[1] += pbj[2] * dj[3] - pbj[0] * dj[5];
    dbj[2] += pbj[0] * dj[4] - pbj[1] * dj[3];
    m3tv(ric, dbj, dcj);
    // d u = [[1/z, 0, -x/z^2], [0, 1/z, -y/z^2]] d pcj
    const double du0 = (dcj[0] - pcj[0] * idep * dcj[2]) * idep, du1 = (dcj[1] - pcj[1] * idep * dcj[2]) * idep;
    m0 = sq[0] * du0 + sq[1] * du1;
    m1 = sq[2] * du0 + sq[3] * du1;
}

// J_ex of ProjectionFactor::Evaluate (src/factor/projection_factor.cpp:100-111), the block that is only filled when the
// extrinsic is estimated: 2x6 = reduce * [ ric^T (Rj^T Ri - I) | -T [pc]x + [T pc]x + [ric^T (Rj^T (Ri tic + Pi - Pj) - tic)]x ],
// T = ric^T Rj^T Ri ric, pc = pts_camera_i.  Unweighted by the loss (the caller applies the Corrector's scale).
DEV void proj_jac_ex(const double *Ri, const double *Pi, const double *Rj, const double *Pj, const double *ric, const double *tic,
                     const double *sq, double lam, double pix, double piy, double piz, double *Jex) {
    const double inv = 1.0 / lam;
    const double pc[3] = {pix * inv, piy * inv, piz * inv};
    double pb[3], pw[3], t[3], pbj[3], pcj[3];
    m3v(ric, pc, pb); pb[0] += tic[0]; pb[1] += tic[1]; pb[2] += tic[2];
    m3v(Ri, pb, pw);
    t[0] = pw[0] + Pi[0] - Pj[0]; t[1] = pw[1] + Pi[1] - Pj[1]; t[2] = pw[2] + Pi[2] - Pj[2];
    m3tv(Rj, t, pbj);
    t[0] = pbj[0] - tic[0]; t[1] = pbj[1] - tic[1]; t[2] = pbj[2] - tic[2];
    m3tv(ric, t, pcj);
    const double idep = 1.0 / pcj[2];
    const double a0 = idep, a2 = -pcj[0] * idep * idep, b2 = -pcj[1] * idep * idep;
    const double red[6] = {sq[0] * a0, sq[1] * a0, sq[0] * a2 + sq[1] * b2, sq[2] * a0, sq[3] * a0, sq[2] * a2 + sq[3] * b2};
    double M[9], A[9], Tm[9], je[18];
    m3_mul_tn(Rj, Ri, M);                              // Rj^T Ri
    m3_mul_tn(ric, M, A);                              // ric^T Rj^T Ri
    m3_mul(A, ric, Tm);                                // T
    double Tpc[3], v[3], w2[3], u[3];
    m3v(Tm, pc, Tpc);
    m3v(Ri, tic, v); v[0] += Pi[0] - Pj[0]; v[1] += Pi[1] - Pj[1]; v[2] += Pi[2] - Pj[2];
    m3tv(Rj, v, w2); w2[0] -= tic[0]; w2[1] -= tic[1]; w2[2] -= tic[2];
    m3tv(ric, w2, u);
    double Spc[9], S1[9], S2[9], TS[9];
    skew3(pc, Spc); skew3(Tpc, S1); skew3(u, S2); m3_mul(Tm, Spc, TS);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            je[r * 6 + c] = A[r * 3 + c] - ric[c * 3 + r];                        // ric^T (Rj^T Ri - I)
            je[r * 6 + 3 + c] = -TS[r * 3 + c] + S1[r * 3 + c] + S2[r * 3 + c];
        }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) Jex[r * 6 + c] = red[r * 3] * je[c] + red[r * 3 + 1] * je[6 + c] + red[r * 3 + 2] * je[12 + c];
}

// proj_residual_dir with the extrinsic moving as well: (dt_ic, dtheta_ic) = dex, ric <- ric Exp(dtheta_ic)
DEV void proj_residual_dir_ex(const double *Ri, const double *Pi, const double *Rj, const double *Pj,
                              const double *ric, const double *tic, const double *sq, double lam,
                              double pix, double piy, double piz, double pjx, double pjy,
                              const double *di, const double *dj, const double *dex, double dlam,
                              double &r0, double &r1, double &m0, double &m1) {
    const double inv = 1.0 / lam;
    double pc[3] = {pix * inv, piy * inv, piz * inv};
    double pb[3], pw[3], t[3], pbj[3], pcj[3];
    m3v(ric, pc, pb); pb[0] += tic[0]; pb[1] += tic[1]; pb[2] += tic[2];
    m3v(Ri, pb, pw);
    t[0] = pw[0] + Pi[0] - Pj[0]; t[1] = pw[1] + Pi[1] - Pj[1]; t[2] = pw[2] + Pi[2] - Pj[2];
    m3tv(Rj, t, pbj);
    t[0] = pbj[0] - tic[0]; t[1] = pbj[1] - tic[1]; t[2] = pbj[2] - tic[2];
    m3tv(ric, t, pcj);
    const double idep = 1.0 / pcj[2];
    const double u0 = pcj[0] * idep - pjx, u1 = pcj[1] * idep - pjy;
    r0 = sq[0] * u0 + sq[1] * u1;
    r1 = sq[2] * u0 + sq[3] * u1;
    const double sl = -dlam * inv;
    // d pc = -pc dlam / lam;  d pb = ric (d pc + dtheta_ic x pc) + dt_ic + dtheta_i x pb (body frame);  d pw = Ri d pb + dP_i
    double dpc[3] = {pc[0] * sl + dex[4] * pc[2] - dex[5] * pc[1], pc[1] * sl + dex[5] * pc[0] - dex[3] * pc[2], pc[2] * sl + dex[3] * pc[1] - dex[4] * pc[0]};
    double dpb[3], dw[3], dbj[3], dcj[3];
    m3v(ric, dpc, dpb);
    dpb[0] += dex[0] + di[4] * pb[2] - di[5] * pb[1];
    dpb[1] += dex[1] + di[5] * pb[0] - di[3] * pb[2];
    dpb[2] += dex[2] + di[3] * pb[1] - di[4] * pb[0];
    m3v(Ri, dpb, dw);
    dw[0] += di[0] - dj[0]; dw[1] += di[1] - dj[1]; dw[2] += di[2] - dj[2];
    m3tv(Rj, dw, dbj);
    dbj[0] += pbj[1] * dj[5] - pbj[2] * dj[4] - dex[0];
    dbj[1] += pbj[2] * dj[3] - pbj[0] * dj[5] - dex[1];
    dbj[2] += pbj[0] * dj[4] - pbj[1] * dj[3] - dex[2];
    m3tv(ric, dbj, dcj);                               // ric^T (d pbj - dt_ic)
    dcj[0] += pcj[1] * dex[5] - pcj[2] * dex[4];        // - dtheta_ic x pcj
    dcj[1] += pcj[2] * dex[3] - pcj[0] * dex[5];
    dcj[2] += pcj[0] * dex[4] - pcj[1] * dex[3];
    const double du0 = (dcj[0] - pcj[0] * idep * dcj[2]) * idep, du1 = (dcj[1] - pcj[1] * idep * dcj[2]) * idep;
    m0 = sq[0] * du0 + sq[1] * du1;
    m1 = sq[2] * du0 + sq[3] * du1;
}
