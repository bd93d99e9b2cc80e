// isv_imu_factor.h -- raw (unweighted) IMU residual of one factor for one lane.
// IntegrationBase::evaluate  include/factor/integration_base.h:160-186 (called from IMUFactor::Evaluate).
#pragma once
#include "isv_device_types.h"
#include "isv_device_math.h"

DEV void imu_raw_residual(const double *G, const double *rec, const double *pi, const double *pj,
                          const double *si, const double *sj, double *out) {
    Quat Qi = q_from_pose(pi), Qj = q_from_pose(pj);
    Quat Qii = q_inv(Qi);
    const double dt = rec[IMU_DT];
    double dbg[3], dba[3], tt[3], t2[3], cdv[3], cdp[3], u[3], o1[3], o2[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { dbg[k] = si[6 + k] - rec[IMU_LBG + k]; dba[k] = si[3 + k] - rec[IMU_LBA + k]; }
    Quat dq = Quat{rec[IMU_DQ + 3], rec[IMU_DQ], rec[IMU_DQ + 1], rec[IMU_DQ + 2]};
    m3v(rec + IMU_DQ_DBG, dbg, tt);
    Quat cdq = q_mul(dq, q_delta(tt));
    m3v(rec + IMU_DV_DBA, dba, tt); m3v(rec + IMU_DV_DBG, dbg, t2);
#pragma unroll
    for (int k = 0; k < 3; k++) cdv[k] = rec[IMU_DV + k] + tt[k] + t2[k];
    m3v(rec + IMU_DP_DBA, dba, tt); m3v(rec + IMU_DP_DBG, dbg, t2);
#pragma unroll
    for (int k = 0; k < 3; k++) cdp[k] = rec[IMU_DP + k] + tt[k] + t2[k];
#pragma unroll
    for (int k = 0; k < 3; k++) u[k] = 0.5 * G[k] * dt * dt + pj[k] - pi[k] - si[k] * dt;
    q_rot(Qii, u, o1);
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = o1[k] - cdp[k];
    Quat e = q_mul(q_inv(cdq), q_mul(Qii, Qj));
    out[3] = 2 * e.x; out[4] = 2 * e.y; out[5] = 2 * e.z;
#pragma unroll
    for (int k = 0; k < 3; k++) u[k] = G[k] * dt + sj[k] - si[k];
    q_rot(Qii, u, o2);
#pragma unroll
    for (int k = 0; k < 3; k++) out[6 + k] = o2[k] - cdv[k];
#pragma unroll
    for (int k = 0; k < 3; k++) { out[9 + k] = sj[3 + k] - si[3 + k]; out[12 + k] = sj[6 + k] - si[6 + k]; }
}
