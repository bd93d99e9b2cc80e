/*
 * isvo_math.h -- small fixed-size linear algebra for the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or executed by the
 * product path (is-vins_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it, and only as the checker.
 *
 * PARITY UNPINNED: the reference cannot be built here (Eigen/Ceres/Sophus absent) and ships
 * no golden vectors; this restatement is pinned by finite-difference checks mirroring the
 * reference's own check() routines and by algebraic invariants (see tests/).
 *
 * Restates the Eigen / Sophus / Utility primitives the reference path calls:
 *   Eigen::Quaterniond (ctor order w,x,y,z; operator*, inverse, toRotationMatrix,
 *     _transformVector, matrix->quaternion)       -- Eigen 3.3 Geometry/Quaternion.h
 *   Utility::{deltaQ,skewSymmetric,Qleft,Qright,R2ypr,ypr2R}  include/utility/utility.h:11-110
 *   Sophus::SO3d::{log,exp}, rightJacobianInvSO3   include/utility/sophus_utils.hpp:194-236
 * All matrices are row-major double arrays.
 */
#ifndef ISVO_MATH_H
#define ISVO_MATH_H
#include <math.h>
#include <string.h>
#include <stdlib.h>

typedef struct { double w, x, y, z; } quat_t;

/* C[m x n] = A[m x k] * B[k x n] */
static inline void mm(const double *A, const double *B, double *C, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int p = 0; p < k; p++) s += A[i * k + p] * B[p * n + j];
            C[i * n + j] = s;
        }
}
/* C[m x n] = A^T (A is k x m) * B[k x n] */
static inline void mm_tn(const double *A, const double *B, double *C, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int p = 0; p < k; p++) s += A[p * m + i] * B[p * n + j];
            C[i * n + j] = s;
        }
}
/* C[m x n] = A[m x k] * B^T (B is n x k) */
static inline void mm_nt(const double *A, const double *B, double *C, int m, int k, int n) {
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int p = 0; p < k; p++) s += A[i * k + p] * B[j * k + p];
            C[i * n + j] = s;
        }
}
static inline void m3_t(const double *A, double *T) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[i * 3 + j] = A[j * 3 + i];
}
static inline void m3v(const double *A, const double *v, double *o) {
    double t0 = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    double t1 = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    double t2 = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static inline void m3tv(const double *A, const double *v, double *o) {
    double t0 = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    double t1 = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    double t2 = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static inline void cross3(const double *a, const double *b, double *o) {
    double t0 = a[1] * b[2] - a[2] * b[1];
    double t1 = a[2] * b[0] - a[0] * b[2];
    double t2 = a[0] * b[1] - a[1] * b[0];
    o[0] = t0; o[1] = t1; o[2] = t2;
}
static inline double dotn(const double *a, const double *b, int n) {
    double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s;
}
/* Utility::skewSymmetric  utility.h:26-34 */
static inline void skew(const double *q, double *S) {
    S[0] = 0;     S[1] = -q[2]; S[2] = q[1];
    S[3] = q[2];  S[4] = 0;     S[5] = -q[0];
    S[6] = -q[1]; S[7] = q[0];  S[8] = 0;
}

/* pose block [px py pz qx qy qz qw] -> Quaterniond(p[6],p[3],p[4],p[5]) (projection_factor.cpp:28) */
static inline quat_t q_from_pose(const double *p) { quat_t q = {p[6], p[3], p[4], p[5]}; return q; }
static inline quat_t q_mul(quat_t a, quat_t b) {
    quat_t r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
static inline quat_t q_conj(quat_t a) { quat_t r = {a.w, -a.x, -a.y, -a.z}; return r; }
/* Eigen QuaternionBase::inverse(): conjugate / squaredNorm */
static inline quat_t q_inv(quat_t a) {
    double n2 = a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z;
    quat_t r = {a.w / n2, -a.x / n2, -a.y / n2, -a.z / n2};
    return r;
}
static inline quat_t q_normalized(quat_t a) {
    double n = sqrt(a.w * a.w + a.x * a.x + a.y * a.y + a.z * a.z);
    quat_t r = {a.w / n, a.x / n, a.y / n, a.z / n};
    return r;
}
/* Eigen QuaternionBase::toRotationMatrix */
static inline void q_to_R(quat_t q, double *R) {
    double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
/* Eigen QuaternionBase::_transformVector: v + 2w (u x v) + 2 u x (u x v) */
static inline void q_rot(quat_t q, const double *v, double *o) {
    double u[3] = {q.x, q.y, q.z}, uv[3], uuv[3];
    cross3(u, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    cross3(u, uv, uuv);
    o[0] = v[0] + q.w * uv[0] + uuv[0];
    o[1] = v[1] + q.w * uv[1] + uuv[1];
    o[2] = v[2] + q.w * uv[2] + uuv[2];
}
/* Eigen quaternion from rotation matrix (QuaternionBase::operator=(MatrixBase)) */
static inline quat_t q_from_R(const double *m) {
    quat_t q; double qq[3];
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 3 + i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
        qq[i] = 0.5 * t; t = 0.5 / t;
        q.w = (m[k * 3 + j] - m[j * 3 + k]) * t;
        qq[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        qq[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
        q.x = qq[0]; q.y = qq[1]; q.z = qq[2];
    }
    return q;
}
/* Utility::deltaQ  utility.h:11-24 (NOT normalised) */
static inline quat_t q_delta(const double *theta) {
    quat_t q = {1.0, theta[0] / 2.0, theta[1] / 2.0, theta[2] / 2.0};
    return q;
}
/* bottom-right 3x3 of Utility::Qleft(q): w I + [v]x   (utility.h:47-55) */
static inline void qleft33(quat_t q, double *M) {
    double v[3] = {q.x, q.y, q.z}; skew(v, M);
    M[0] += q.w; M[4] += q.w; M[8] += q.w;
}
/* bottom-right 3x3 of Utility::Qright(p): w I - [v]x  (utility.h:57-65) */
static inline void qright33(quat_t q, double *M) {
    double v[3] = {q.x, q.y, q.z}; skew(v, M);
    for (int i = 0; i < 9; i++) M[i] = -M[i];
    M[0] += q.w; M[4] += q.w; M[8] += q.w;
}
/* full 4x4 Qleft/Qright in (w,x,y,z) order */
static inline void qleft44(quat_t q, double *M) {
    double B[9]; qleft33(q, B);
    M[0] = q.w; M[1] = -q.x; M[2] = -q.y; M[3] = -q.z;
    M[4] = q.x; M[8] = q.y; M[12] = q.z;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[(i + 1) * 4 + j + 1] = B[i * 3 + j];
}
static inline void qright44(quat_t q, double *M) {
    double B[9]; qright33(q, B);
    M[0] = q.w; M[1] = -q.x; M[2] = -q.y; M[3] = -q.z;
    M[4] = q.x; M[8] = q.y; M[12] = q.z;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[(i + 1) * 4 + j + 1] = B[i * 3 + j];
}
/* Utility::R2ypr (degrees)  utility.h:66-81 */
static inline void R2ypr(const double *R, double *ypr) {
    double n[3] = {R[0], R[3], R[6]}, o[3] = {R[1], R[4], R[7]}, a[3] = {R[2], R[5], R[8]};
    double y = atan2(n[1], n[0]);
    double p = atan2(-n[2], n[0] * cos(y) + n[1] * sin(y));
    double r = atan2(a[0] * sin(y) - a[1] * cos(y), -o[0] * sin(y) + o[1] * cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}
/* Utility::ypr2R (degrees)  utility.h:83-110 */
static inline void ypr2R(const double *ypr, double *R) {
    double y = ypr[0] / 180.0 * M_PI, p = ypr[1] / 180.0 * M_PI, r = ypr[2] / 180.0 * M_PI;
    double Rz[9] = {cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1};
    double Ry[9] = {cos(p), 0., sin(p), 0., 1., 0., -sin(p), 0., cos(p)};
    double Rx[9] = {1., 0., 0., 0., cos(r), -sin(r), 0., sin(r), cos(r)};
    double T[9]; mm(Rz, Ry, T, 3, 3, 3); mm(T, Rx, R, 3, 3, 3);
}

/* ---- Sophus::SO3d restated on unit quaternions ---------------------------------------- */
#define SOPHUS_EPS 1e-10
/* SO3(Quaternion): normalises */
static inline quat_t so3_from_q(quat_t q) { return q_normalized(q); }
/* SO3(Matrix3d): Eigen matrix->quaternion (the orthogonality ENSURE is not restated) */
static inline quat_t so3_from_R(const double *R) { return q_from_R(R); }
/* SO3 * SO3: quaternion product, renormalised by the SO3 constructor */
static inline quat_t so3_mul(quat_t a, quat_t b) { return q_normalized(q_mul(a, b)); }
static inline quat_t so3_inv(quat_t a) { return q_conj(a); }
/* SO3::log (Sophus so3.hpp logAndTheta) */
static inline void so3_log(quat_t q, double *omega) {
    double sn = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w, f;
    if (sn < SOPHUS_EPS * SOPHUS_EPS) {
        double sw = w * w;
        f = 2.0 / w - (2.0 / 3.0) * sn / (w * sw);
    } else {
        double n = sqrt(sn);
        if (fabs(w) < SOPHUS_EPS) f = (w > 0 ? M_PI : -M_PI) / n;
        else f = 2.0 * atan(n / w) / n;
    }
    omega[0] = f * q.x; omega[1] = f * q.y; omega[2] = f * q.z;
}
/* SO3::exp (Sophus so3.hpp expAndTheta) */
static inline quat_t so3_exp(const double *omega) {
    double tsq = omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2], im, re;
    if (tsq < SOPHUS_EPS * SOPHUS_EPS) {
        double t4 = tsq * tsq;
        im = 0.5 - (1.0 / 48.0) * tsq + (1.0 / 3840.0) * t4;
        re = 1.0 - (1.0 / 8.0) * tsq + (1.0 / 384.0) * t4;
    } else {
        double th = sqrt(tsq), h = 0.5 * th;
        im = sin(h) / th; re = cos(h);
    }
    quat_t q = {re, im * omega[0], im * omega[1], im * omega[2]};
    return q;
}
/* Sophus::rightJacobianInvSO3  sophus_utils.hpp:194-236 */
static inline void so3_rjac_inv(const double *phi, double *J) {
    double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    double H[9], H2[9];
    skew(phi, H); mm(H, H, H2, 3, 3, 3);
    for (int i = 0; i < 9; i++) J[i] = H[i] / 2;
    J[0] += 1; J[4] += 1; J[8] += 1;
    double c;
    if (n2 > SOPHUS_EPS) {   /* Sophus::Constants<double>::epsilon() = 1e-10 */
        double n = sqrt(n2);
        if (n < M_PI - 1e-5)  /* epsilonSqrt = sqrt(1e-10) */
            c = 1 / n2 - (1 + cos(n)) / (2 * n * sin(n));
        else
            c = 1.0 / (M_PI * M_PI);
    } else
        c = 1.0 / 12.0;
    for (int i = 0; i < 9; i++) J[i] += H2[i] * c;
}

/* ---- dense helpers (dynamic size, row-major, leading dimension n) --------------------- */
/* Cholesky A = L L^T, lower in place; returns 0 ok, k+1 if pivot k <= 0 (Eigen LLT info) */
static inline int chol_lower(double *A, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d); A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) A[i * n + j] = 0;
    return 0;
}
/* solve L L^T x = b in place */
static inline void chol_solve(const double *L, int n, double *b) {
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}
/* inverse by LU with partial pivoting (Eigen PartialPivLU::inverse); returns 0 ok */
static inline int inv_partial_lu(const double *Ain, double *Inv, int n) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    int *perm = (int *)malloc(sizeof(int) * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) perm[i] = i;
    int bad = 0;
    for (int k = 0; k < n; k++) {
        int p = k; double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (best == 0.0) { bad = 1; continue; }
        if (p != k) {
            for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
            int t = perm[k]; perm[k] = perm[p]; perm[p] = t;
        }
        for (int i = k + 1; i < n; i++) {
            A[i * n + k] /= A[k * n + k];
            double f = A[i * n + k];
            for (int j = k + 1; j < n; j++) A[i * n + j] -= f * A[k * n + j];
        }
    }
    for (int c = 0; c < n; c++) {
        /* solve A x = e_c : P A = L U -> L U x = P e_c */
        double *x = (double *)malloc(sizeof(double) * n);
        for (int i = 0; i < n; i++) x[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= A[i * n + k] * x[k]; x[i] = s; }
        for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= A[i * n + k] * x[k]; x[i] = s / A[i * n + i]; }
        for (int i = 0; i < n; i++) Inv[i * n + c] = x[i];
        free(x);
    }
    free(A); free(perm);
    return bad;
}
/* inverse by LU with FULL pivoting (Eigen FullPivLU::solve(Identity)); rank-deficient pivots
 * (|pivot| <= eps * n * maxpivot) are treated as Eigen does: the corresponding rows of the
 * solution are zero.  Returns the detected rank. */
static inline int inv_full_lu(const double *Ain, double *Inv, int n) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    int *rp = (int *)malloc(sizeof(int) * n), *cp = (int *)malloc(sizeof(int) * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) { rp[i] = i; cp[i] = i; }
    double maxpivot = 0; int nz = n;
    for (int k = 0; k < n; k++) {
        int pi = k, pj = k; double best = 0;
        for (int i = k; i < n; i++) for (int j = k; j < n; j++)
            if (fabs(A[i * n + j]) > best) { best = fabs(A[i * n + j]); pi = i; pj = j; }
        if (best == 0.0) { nz = k; break; }
        if (best > maxpivot) maxpivot = best;
        if (pi != k) { for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[pi * n + j]; A[pi * n + j] = t; } int t = rp[k]; rp[k] = rp[pi]; rp[pi] = t; }
        if (pj != k) { for (int i = 0; i < n; i++) { double t = A[i * n + k]; A[i * n + k] = A[i * n + pj]; A[i * n + pj] = t; } int t = cp[k]; cp[k] = cp[pj]; cp[pj] = t; }
        for (int i = k + 1; i < n; i++) {
            A[i * n + k] /= A[k * n + k];
            double f = A[i * n + k];
            for (int j = k + 1; j < n; j++) A[i * n + j] -= f * A[k * n + j];
        }
    }
    double thr = 2.220446049250313e-16 * n * maxpivot;
    int rank = 0;
    for (int k = 0; k < nz; k++) if (fabs(A[k * n + k]) > thr) rank++;
    double *x = (double *)malloc(sizeof(double) * n);
    for (int c = 0; c < n; c++) {
        for (int i = 0; i < n; i++) x[i] = (rp[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i && k < nz; k++) s -= A[i * n + k] * x[k]; x[i] = s; }
        for (int i = n - 1; i >= 0; i--) {
            if (i >= rank) { x[i] = 0; continue; }
            double s = x[i]; for (int k = i + 1; k < rank; k++) s -= A[i * n + k] * x[k]; x[i] = s / A[i * n + i];
        }
        for (int i = 0; i < n; i++) Inv[cp[i] * n + c] = x[i];
    }
    free(x); free(A); free(rp); free(cp);
    return rank;
}
/* cyclic Jacobi eigen-decomposition of a symmetric matrix: A = V diag(w) V^T, eigenvalues
 * ascending (Eigen SelfAdjointEigenSolver order); V columns are eigenvectors (row-major V) */
static inline void sym_eig(const double *Ain, int n, double *w, double *V) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 100; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) { diag += A[i * n + i] * A[i * n + i]; for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j]; }
        if (off <= 1e-60 || off <= 1e-34 * diag) break;
        for (int p = 0; p < n - 1; p++) for (int q = p + 1; q < n; q++) {
            double apq = A[p * n + q];
            if (apq == 0.0) continue;
            double app = A[p * n + p], aqq = A[q * n + q];
            double theta = (aqq - app) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < n; k++) {
                double akp = A[k * n + p], akq = A[k * n + q];
                A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
            }
            for (int k = 0; k < n; k++) {
                double apk = A[p * n + k], aqk = A[q * n + k];
                A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
            }
            for (int k = 0; k < n; k++) {
                double vkp = V[k * n + p], vkq = V[k * n + q];
                V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
            }
        }
    }
    for (int i = 0; i < n; i++) w[i] = A[i * n + i];
    /* sort ascending */
    for (int i = 0; i < n - 1; i++) {
        int m = i;
        for (int j = i + 1; j < n; j++) if (w[j] < w[m]) m = j;
        if (m != i) {
            double t = w[i]; w[i] = w[m]; w[m] = t;
            for (int k = 0; k < n; k++) { double u = V[k * n + i]; V[k * n + i] = V[k * n + m]; V[k * n + m] = u; }
        }
    }
    free(A);
}
static inline double det_lu(const double *Ain, int n) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    double det = 1;
    for (int k = 0; k < n; k++) {
        int p = k; double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (best == 0.0) { det = 0; break; }
        if (p != k) { for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; } det = -det; }
        det *= A[k * n + k];
        for (int i = k + 1; i < n; i++) { double f = A[i * n + k] / A[k * n + k]; for (int j = k + 1; j < n; j++) A[i * n + j] -= f * A[k * n + j]; }
    }
    free(A);
    return det;
}
/* log |det A| by the same elimination, accumulated in the log domain (a 57 x 57 information matrix with eigenvalues
 * around 1e8 overflows a double determinant) */
static inline double logabsdet_lu(const double *Ain, int n) {
    double *A = (double *)malloc(sizeof(double) * n * n);
    memcpy(A, Ain, sizeof(double) * n * n);
    double ld = 0;
    for (int k = 0; k < n; k++) {
        int p = k; double best = fabs(A[k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
        if (best == 0.0) { ld = -INFINITY; break; }
        if (p != k) for (int j = 0; j < n; j++) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
        ld += log(fabs(A[k * n + k]));
        for (int i = k + 1; i < n; i++) { double f = A[i * n + k] / A[k * n + k]; for (int j = k + 1; j < n; j++) A[i * n + j] -= f * A[k * n + j]; }
    }
    free(A);
    return ld;
}
#endif
