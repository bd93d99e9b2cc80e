// isv_device_math.h -- fp64 3x3 / quaternion / SO(3) helpers for the gfx950 kernels.
// Semantics follow the Eigen / Sophus / Utility calls the reference factors make
// (include/utility/utility.h:11-110, include/utility/sophus_utils.hpp:194-236); written for
// registers: everything is passed by value or small local arrays that the compiler keeps in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

#define DEV __device__ __forceinline__
// wave-level ordering of LDS traffic (LDS executes one wavefront's accesses in issue order)
#define ISV_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct Quat { double w, x, y, z; };

DEV Quat q_from_pose(const double *p) { return Quat{p[6], p[3], p[4], p[5]}; }
DEV Quat q_mul(Quat a, Quat b) {
    return Quat{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
DEV Quat q_conj(Quat a) { return Quat{a.w, -a.x, -a.y, -a.z}; }
DEV Quat q_inv(Quat a) {   // Eigen inverse(): conjugate / squaredNorm
    double n2 = a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z;
    return Quat{a.w / n2, -a.x / n2, -a.y / n2, -a.z / n2};
}
DEV Quat q_normalized(Quat a) {
    double n = sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
    return Quat{a.w / n, a.x / n, a.y / n, a.z / n};
}
DEV void q_to_R(Quat q, double *R) {   // Eigen toRotationMatrix
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
DEV void q_rot(Quat q, const double *v, double *o) {   // Eigen _transformVector
    double ux = q.x, uy = q.y, uz = q.z;
    double c0 = uy * v[2] - uz * v[1], c1 = uz * v[0] - ux * v[2], c2 = ux * v[1] - uy * v[0];
    c0 += c0; c1 += c1; c2 += c2;
    double d0 = uy * c2 - uz * c1, d1 = uz * c0 - ux * c2, d2 = ux * c1 - uy * c0;
    o[0] = v[0] + q.w * c0 + d0; o[1] = v[1] + q.w * c1 + d1; o[2] = v[2] + q.w * c2 + d2;
}
DEV Quat q_from_R(const double *m) {   // Eigen matrix -> quaternion (branches spelled out: no indexed locals)
    Quat q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else if (m[0] >= m[4] && m[0] >= m[8]) {             // i = 0, j = 1, k = 2
        t = sqrt(m[0] - m[4] - m[8] + 1.0);
        q.x = 0.5 * t; t = 0.5 / t;
        q.w = (m[7] - m[5]) * t; q.y = (m[3] + m[1]) * t; q.z = (m[6] + m[2]) * t;
    } else if (m[4] > m[0] && m[4] >= m[8]) {              // i = 1, j = 2, k = 0
        t = sqrt(m[4] - m[8] - m[0] + 1.0);
        q.y = 0.5 * t; t = 0.5 / t;
        q.w = (m[2] - m[6]) * t; q.z = (m[7] + m[5]) * t; q.x = (m[1] + m[3]) * t;
    } else {                                               // i = 2, j = 0, k = 1
        t = sqrt(m[8] - m[0] - m[4] + 1.0);
        q.z = 0.5 * t; t = 0.5 / t;
        q.w = (m[3] - m[1]) * t; q.x = (m[2] + m[6]) * t; q.y = (m[5] + m[7]) * t;
    }
    return q;
}
DEV Quat q_delta(const double *th) { return Quat{1.0, th[0] / 2.0, th[1] / 2.0, th[2] / 2.0}; }

DEV void m3_mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
DEV void m3_mul_tn(const double *A, const double *B, double *C) {   // A^T B
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
DEV void m3_mul_nt(const double *A, const double *B, double *C) {   // A B^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
DEV void m3v(const double *A, const double *v, double *o) {
    double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
DEV void m3tv(const double *A, const double *v, double *o) {
    double a = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double b = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double c = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}
DEV void skew3(const double *q, double *S) {
    S[0] = 0; S[1] = -q[2]; S[2] = q[1];
    S[3] = q[2]; S[4] = 0; S[5] = -q[0];
    S[6] = -q[1]; S[7] = q[0]; S[8] = 0;
}
// bottom-right 3x3 of Utility::Qleft / Qright
DEV void qleft33(Quat q, double *M) {
    M[0] = q.w; M[1] = -q.z; M[2] = q.y;
    M[3] = q.z; M[4] = q.w; M[5] = -q.x;
    M[6] = -q.y; M[7] = q.x; M[8] = q.w;
}
DEV void qright33(Quat q, double *M) {
    M[0] = q.w; M[1] = q.z; M[2] = -q.y;
    M[3] = -q.z; M[4] = q.w; M[5] = q.x;
    M[6] = q.y; M[7] = -q.x; M[8] = q.w;
}

// ---- Sophus::SO3d on unit quaternions -------------------------------------------------------
#define ISV_SOPHUS_EPS 1e-10
#define ISV_PI 3.14159265358979323846
DEV Quat so3_mul(Quat a, Quat b) { return q_normalized(q_mul(a, b)); }
DEV void so3_log(Quat q, double *om) {
    double sn = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w, f;
    if (sn < ISV_SOPHUS_EPS * ISV_SOPHUS_EPS) {
        f = 2.0 / w - (2.0 / 3.0) * sn / (w * w * w);
    } else {
        double n = sqrt(sn);
        if (fabs(w) < ISV_SOPHUS_EPS) f = (w > 0 ? ISV_PI : -ISV_PI) / n;
        else f = 2.0 * atan(n / w) / n;
    }
    om[0] = f * q.x; om[1] = f * q.y; om[2] = f * q.z;
}
DEV Quat so3_exp(const double *om) {
    double tsq = om[0] * om[0] + om[1] * om[1] + om[2] * om[2], im, re;
    if (tsq < ISV_SOPHUS_EPS * ISV_SOPHUS_EPS) {
        double t4 = tsq * tsq;
        im = 0.5 - (1.0 / 48.0) * tsq + (1.0 / 3840.0) * t4;
        re = 1.0 - (1.0 / 8.0) * tsq + (1.0 / 384.0) * t4;
    } else {
        double th = sqrt(tsq), h = 0.5 * th;
        im = sin(h) / th; re = cos(h);
    }
    return Quat{re, im * om[0], im * om[1], im * om[2]};
}
DEV void so3_rjac_inv(const double *phi, double *J) {   // sophus_utils.hpp:194-236
    double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    double H[9], H2[9];
    skew3(phi, H); m3_mul(H, H, H2);
    double c;
    if (n2 > ISV_SOPHUS_EPS) {
        double n = sqrt(n2);
        if (n < ISV_PI - 1e-5) c = 1 / n2 - (1 + cos(n)) / (2 * n * sin(n));
        else c = 1.0 / (ISV_PI * ISV_PI);
    } else
        c = 1.0 / 12.0;
#pragma unroll
    for (int i = 0; i < 9; i++) J[i] = H[i] / 2 + H2[i] * c;
    J[0] += 1; J[4] += 1; J[8] += 1;
}
// PoseLocalParameterization::Plus  src/factor/pose_local_parameterization.cpp:3-19
DEV void pose_plus(const double *x, const double *d, double *xp) {
    Quat r = q_normalized(q_mul(Quat{x[6], x[3], x[4], x[5]}, q_delta(d + 3)));
    xp[0] = x[0] + d[0]; xp[1] = x[1] + d[1]; xp[2] = x[2] + d[2];
    xp[3] = r.x; xp[4] = r.y; xp[5] = r.z; xp[6] = r.w;
}
// Utility::R2ypr (degrees) / ypr2R   utility.h:66-110
DEV void R2ypr(const double *R, double *ypr) {
    double y = atan2(R[3], R[0]);
    double p = atan2(-R[6], R[0] * cos(y) + R[3] * sin(y));
    double r = atan2(R[2] * sin(y) - R[5] * cos(y), -R[1] * sin(y) + R[4] * cos(y));
    ypr[0] = y / ISV_PI * 180.0; ypr[1] = p / ISV_PI * 180.0; ypr[2] = r / ISV_PI * 180.0;
}
DEV void ypr2R(const double *ypr, double *R) {
    double y = ypr[0] / 180.0 * ISV_PI, p = ypr[1] / 180.0 * ISV_PI, r = ypr[2] / 180.0 * ISV_PI;
    double Rz[9] = {cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1};
    double Ry[9] = {cos(p), 0., sin(p), 0., 1., 0., -sin(p), 0., cos(p)};
    double Rx[9] = {1., 0., 0., 0., cos(r), -sin(r), 0., sin(r), cos(r)};
    double T[9]; m3_mul(Rz, Ry, T); m3_mul(T, Rx, R);
}
