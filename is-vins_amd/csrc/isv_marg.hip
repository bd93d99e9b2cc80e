// isv_marg.hip -- k_marg: Estimator::MargForward (reference src/estimator.cpp:1149-1352) and
// Estimator::MargBackward (:1354-1539) on the device, one wavefront per window (only windows
// uploaded with margin_old != 0 do work).  All matrices are tiny (<= 30x30) and live in LDS; the
// wavefront cooperates lane-per-entry.  Runs after k_finalize: it linearises at the para_* arrays
// (d.pose / d.sb / d.lam, the un-rotated solve output) with the post-update, post-double2vector
// priors, and takes Ri/ti from the rotated Rs[0]/Ps[0] -- exactly the reference's order of events
// (SURVEY.md appendix B.2).
//
// Restatement notes (mathematically identical to the reference, different elimination order):
//   * MargForward eliminates [T0, landmarks] with a fullPivLu inverse of the (L0+6)^2 matrix; the
//     landmark block of that matrix is diagonal, so here the landmarks are eliminated first in closed
//     form (rank-1 downdates of the 12x12 pose block) and T0 by a 6x6 inverse.
//   * Eigen's SelfAdjointEigenSolver / BDCSVD are replaced by cyclic Jacobi iterations.
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"
#include "isv_proj_factor.h"
#include "isv_prior_factor.h"

#define MT 64
#ifdef ISV_STAMP
#define MSTAMP(k) do { if (t == 0) { unsigned long long now_ = wall_clock64(); d.dbg[(size_t)w * 64 + 32 + (k)] += (double)(now_ - t_last); t_last = now_; } } while (0)
#else
#define MSTAMP(k) do {} while (0)
#endif
#define SYNC() __syncthreads()

DEV double m_rsqrt(double x) { double r = __builtin_amdgcn_rsq(x); r = r * (1.5 - 0.5 * x * r * r); r = r * (1.5 - 0.5 * x * r * r); return r; }
DEV double m_rcp(double x) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); r = r * (2.0 - x * r); return r; }

// The arithmetic of one Jacobi rotation, written with explicit fma() so that every instance (the one-wavefront loops, the
// four-wavefront kernel) rounds the same way whatever the compiler would have contracted: their results are bitwise equal.
DEV double jr_rsqrt(double x) { double r = __builtin_amdgcn_rsq(x); const double h = 0.5 * x; r = r * fma(-(h * r), r, 1.5); r = r * fma(-(h * r), r, 1.5); return r; }
DEV double jr_rcp(double x) { double r = __builtin_amdgcn_rcp(x); r = r * fma(-x, r, 2.0); r = r * fma(-x, r, 2.0); return r; }
// Newton-refined rcp / rsq instead of IEEE div / sqrt: the rotation only has to be orthogonal to rounding, which
// c = rsqrt(1 + t^2), s = t c guarantees independently of t's accuracy.  rot = false: the identity.
DEV void jr_params(double app, double aqq, double apq, bool rot, double &c, double &sn) {
    const double theta = ((aqq - app) * 0.5) * jr_rcp(apq);
    const double th2 = fma(theta, theta, 1.0);
    const double tv = (theta >= 0 ? 1.0 : -1.0) * jr_rcp(fma(th2, jr_rsqrt(th2), fabs(theta)));
    const double cc = jr_rsqrt(fma(tv, tv, 1.0));
    c = rot ? cc : 1.0; sn = rot ? tv * cc : 0.0;
}
// the 2x2 block (rows of pair a) x (columns of pair b) of J^T A J: columns first, then rows
DEV void jr_block(double ca, double sa, double cb, double sb, double b00, double b01, double b10, double b11,
                  double &o00, double &o01, double &o10, double &o11) {
    const double c00 = fma(cb, b00, -(sb * b01)), c01 = fma(sb, b00, cb * b01);
    const double c10 = fma(cb, b10, -(sb * b11)), c11 = fma(sb, b10, cb * b11);
    o00 = fma(ca, c00, -(sa * c10)); o01 = fma(ca, c01, -(sa * c11));
    o10 = fma(sa, c00, ca * c10); o11 = fma(sa, c01, ca * c11);
}
DEV void jr_vec(double c, double sn, double vp, double vq, double &op, double &oq) { op = fma(c, vp, -(sn * vq)); oq = fma(sn, vp, c * vq); }

// ---- wave-cooperative dense helpers on LDS matrices (row-major) --------------------------------
DEV void w_mm(const double *A, const double *B, double *C, int m, int k, int n, int t) {
    for (int e = t; e < m * n; e += MT) { const int i = e / n, j = e % n; double s = 0; for (int p = 0; p < k; p++) s += A[i * k + p] * B[p * n + j]; C[e] = s; }
    SYNC();
}
DEV void w_mm_tn(const double *A, const double *B, double *C, int m, int k, int n, int t) {    // A is k x m
    for (int e = t; e < m * n; e += MT) { const int i = e / n, j = e % n; double s = 0; for (int p = 0; p < k; p++) s += A[p * m + i] * B[p * n + j]; C[e] = s; }
    SYNC();
}
DEV void w_mm_nt(const double *A, const double *B, double *C, int m, int k, int n, int t) {    // B is n x k
    for (int e = t; e < m * n; e += MT) { const int i = e / n, j = e % n; double s = 0; for (int p = 0; p < k; p++) s += A[i * k + p] * B[j * k + p]; C[e] = s; }
    SYNC();
}
// inverse by Gauss-Jordan with partial pivoting on the augmented [A | I]; W is n x 2n scratch
DEV void w_inv(const double *A, int n, double *Inv, double *W, int *piv, int t) {
    const int n2 = 2 * n;
    for (int e = t; e < n * n2; e += MT) { const int i = e / n2, j = e % n2; W[e] = (j < n) ? A[i * n + j] : ((j - n) == i ? 1.0 : 0.0); }
    SYNC();
    for (int k = 0; k < n; k++) {
        if (t == 0) { int p = k; double best = fabs(W[k * n2 + k]); for (int i = k + 1; i < n; i++) if (fabs(W[i * n2 + k]) > best) { best = fabs(W[i * n2 + k]); p = i; } piv[0] = p; }
        SYNC();
        const int p = piv[0];
        if (p != k) for (int j = t; j < n2; j += MT) { const double tmp = W[k * n2 + j]; W[k * n2 + j] = W[p * n2 + j]; W[p * n2 + j] = tmp; }
        SYNC();
        const double d = W[k * n2 + k];
        SYNC();
        for (int j = t; j < n2; j += MT) W[k * n2 + j] /= d;
        SYNC();
        for (int e = t; e < n * n2; e += MT) {
            const int i = e / n2, j = e % n2;
            if (i != k && j != k) W[e] -= W[i * n2 + k] * W[k * n2 + j];
        }
        SYNC();
        for (int i = t; i < n; i += MT) if (i != k) W[i * n2 + k] = 0.0;
        SYNC();
    }
    for (int e = t; e < n * n; e += MT) { const int i = e / n, j = e % n; Inv[e] = W[i * n2 + n + j]; }
    SYNC();
}
// U = LLT(M).matrixL().transpose()  (upper triangular), lane-parallel over rows
DEV void w_chol_upper(const double *M, int n, double *U, double *L, int t) {
    for (int e = t; e < n * n; e += MT) L[e] = M[e];
    SYNC();
    for (int j = 0; j < n; j++) {
        if (t == 0) { double dd = L[j * n + j]; for (int k = 0; k < j; k++) dd -= L[j * n + k] * L[j * n + k]; L[j * n + j] = sqrt(dd); }
        SYNC();
        for (int i = j + 1 + t; i < n; i += MT) { double s = L[i * n + j]; for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k]; L[i * n + j] = s / L[j * n + j]; }
        SYNC();
    }
    for (int e = t; e < n * n; e += MT) { const int i = e / n, j = e % n; U[e] = (j >= i) ? L[j * n + i] : 0.0; }
    SYNC();
}
// Jacobi eigen-decomposition with a round-robin (tournament) ordering: the floor(n/2) rotations of a
// round touch disjoint index pairs, so they are computed and applied together (3 barriers per round
// instead of 3 per rotation).  A (destroyed) = V diag(w) V^T.  rot: scratch [3 * 16] doubles + pairs.
template <int NC>                               // NC > 0: compile-time size (the round loops unroll, their LDS reads batch); NC = 0: runtime n
DEV void w_jacobi_t(double *A, int n_rt, double *wv, double *V, double *tmp, int t) {
    const int n = NC > 0 ? NC : n_rt;
    __shared__ double rc[32], rs[32];
    __shared__ int rp[32], rq[32];                 // floor(n / 2) concurrent rotations, n <= 64
    for (int e = t; e < n * n; e += MT) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    SYNC();
    const int m = n + (n & 1), half = m / 2;
    // lane t walks items t, t + 64, ...: the (row, column) split of an item index advances by a fixed step, so the
    // runtime divisions are done once here and not in every round
    const int bq = MT / half, br = MT % half, b_pa0 = t / half, b_pb0 = t % half;      // 2x2 blocks: e = pa * half + pb
    const int vq = MT / n, vr = MT % n, v_pr0 = t / n, v_k0 = t % n;                   // eigenvector items: e = pr * n + k
    bool polish = false;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0, dg = 0;
        for (int e = t; e < n * n; e += MT) { const int i = e / n, j = e % n; if (j > i) off += A[e] * A[e]; else if (i == j) dg += A[e] * A[e]; }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); dg += __shfl_xor(dg, o); }      // one wavefront: butterfly sums, every lane ends with the totals
        const double offs = off, dgs = dg;
        // Stopping rule (round 5, JACOBI_STOP below): the off-diagonal mass at 1e-14 of the diagonal's (squared: 1e-28) fixes the LARGE
        // eigenpairs, but it is measured against the largest eigenvalue: with a spectrum of 8e1 .. 1e9 (MargBackward's 21 x 21 marginal
        // on the EuRoC stand-in) a residual of 1e-5 still couples the smallest KEPT eigenvector to the discarded null space at 1e-5 / 79
        // ~ 1e-7, and that vector carries the largest weight 1 / lambda of the projected covariance -- the recovered roll/pitch factor's
        // information was 1.2e-7 (relative) from the 40-digit result where the oracle's cyclic Jacobi (threshold 1e-34) is at 1e-10
        // (tests/test_marg_third_opinion.py).  One more sweep once the threshold is met squares the residual (Jacobi converges
        // quadratically); no extra sweep when the mass is already below 1e-32 (1e-16 of the diagonal: the level the oracle reaches).
        if (offs <= 1e-60 || offs <= 1e-32 * dgs || polish) break;
        if (offs <= 1e-28 * dgs) polish = true;
        for (int r = 0; r < m - 1; r++) {
            if (t < half) {
                int a = r + t, b = r + m - 1 - t;          // both < 2 (m - 1): one conditional subtraction is the modulo
                if (a >= m - 1) a -= m - 1;
                if (b >= m - 1) b -= m - 1;
                if (t == 0) { a = m - 1; b = r; }
                int p = a < b ? a : b, q = a < b ? b : a;
                double c = 1.0, sn = 0.0;
                if (q < n) {
                    const double apq = A[p * n + q];
                    if (apq != 0.0) jr_params(A[p * n + p], A[q * n + q], apq, true, c, sn);
                } else q = p;                      // odd n: this index sits out the round (identity), but its row and column still see the other rotations
                rp[t] = p; rq[t] = q; rc[t] = c; rs[t] = sn;
            }
            SYNC();
            // A <- J^T A J in ONE pass: the 2x2 block (rows of pair a) x (columns of pair b) only sees the two
            // rotations a and b, and the blocks of a round are disjoint.  V <- V J in the same pass.
            auto block = [&](int pa, int pb) {
                const int p1 = rp[pa], q1 = rq[pa], p2 = rp[pb], q2 = rq[pb];
                const bool ra = p1 != q1, rb = p2 != q2;       // a dummy pair (odd n) holds one real index, identity rotation
                if (!ra && !rb) return;
                const double ca = rc[pa], sa = rs[pa], cb = rc[pb], sb = rs[pb];
                const double b00 = A[p1 * n + p2], b01 = rb ? A[p1 * n + q2] : 0.0;
                const double b10 = ra ? A[q1 * n + p2] : 0.0, b11 = (ra && rb) ? A[q1 * n + q2] : 0.0;
                double o00, o01, o10, o11;
                jr_block(ca, sa, cb, sb, b00, b01, b10, b11, o00, o01, o10, o11);
                A[p1 * n + p2] = o00;
                if (rb) A[p1 * n + q2] = o01;
                if (ra) A[q1 * n + p2] = o10;
                if (ra && rb) A[q1 * n + q2] = o11;
            };
            auto evec = [&](int pr, int k) {             // eigenvector columns
                const int p = rp[pr], q = rq[pr];
                if (p != q) {
                    const double c = rc[pr], sn = rs[pr];
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    jr_vec(c, sn, vkp, vkq, V[k * n + p], V[k * n + q]);
                }
            };
            if constexpr (NC > 0) {                      // constant trip counts: unrolled, the LDS reads of all items issue together
#pragma unroll
                for (int e = t, pa = b_pa0, pb = b_pb0; e < half * half; e += MT, pa += bq, pb += br) { if (pb >= half) { pb -= half; pa++; } block(pa, pb); }
#pragma unroll
                for (int e = t, pr = v_pr0, k = v_k0; e < half * n; e += MT, pr += vq, k += vr) { if (k >= n) { k -= n; pr++; } evec(pr, k); }
            } else {
                for (int e = t, pa = b_pa0, pb = b_pb0; e < half * half; e += MT, pa += bq, pb += br) { if (pb >= half) { pb -= half; pa++; } block(pa, pb); }
                for (int e = t, pr = v_pr0, k = v_k0; e < half * n; e += MT, pr += vq, k += vr) { if (k >= n) { k -= n; pr++; } evec(pr, k); }
            }
            SYNC();
        }
    }
    for (int i = t; i < n; i += MT) wv[i] = A[i * n + i];
    SYNC();
}
// The same rotations in the same order for a compile-time n, arranged for ONE wavefront's latency: a round is a chain of
// dependent LDS round trips and FP64 steps (rotation parameters -> 2x2 blocks -> eigenvector columns), so (i) every item
// is branch-free -- a dummy pair (odd n) is the identity rotation on a doubled index, lanes past the item count work on
// item 0 and store to a sink -- which leaves each phase one basic block the scheduler can interleave, and (ii) the
// parameters of round r + 1 (they need A after round r's blocks, nothing of V) are computed between the same two
// barriers as round r's eigenvector update, double-buffered, so the two chains overlap instead of following each other.
template <int NC>
DEV void w_jacobi_pipe(double *A, double *wv, double *V, int t) {
    constexpr int n = NC, m = n + (n & 1), half = m / 2, NB = half * half, NV = half * n;
    __shared__ double rc[2][32], rs[2][32], sink[2];
    __shared__ int rp[2][32], rq[2][32];
    for (int e = t; e < n * n; e += MT) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    SYNC();
    const int tt = t % half;                          // lanes beyond the first `half` recompute (and re-store) the same parameters
    auto params = [&](int r, int buf) {
        int a = r + tt, b = r + m - 1 - tt;           // both < 2 (m - 1): one conditional subtraction is the modulo
        a = a >= m - 1 ? a - (m - 1) : a;
        b = b >= m - 1 ? b - (m - 1) : b;
        a = tt == 0 ? m - 1 : a; b = tt == 0 ? r : b;
        const int p = a < b ? a : b, q0 = a < b ? b : a;
        const bool real = q0 < n;                     // odd n: the pair holding the phantom index sits out the round
        const int q = real ? q0 : p;
        const double apq = A[p * n + q], aqq = A[q * n + q], app = A[p * n + p];
        double c, sn;
        jr_params(app, aqq, apq, real && apq != 0.0, c, sn);
        rp[buf][tt] = p; rq[buf][tt] = q; rc[buf][tt] = c; rs[buf][tt] = sn;
    };
    bool polish = false;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0, dg = 0;
        for (int e = t; e < n * n; e += MT) { const int i = e / n, j = e % n; if (j > i) off += A[e] * A[e]; else if (i == j) dg += A[e] * A[e]; }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); dg += __shfl_xor(dg, o); }
        if (off <= 1e-60 || off <= 1e-32 * dg || polish) break;      // as w_jacobi_t: one more sweep after the 1e-28 threshold
        if (off <= 1e-28 * dg) polish = true;
        params(0, 0);
        SYNC();
        for (int r = 0; r < m - 1; r++) {
            const int buf = r & 1;
            // A <- J^T A J: the 2x2 block (rows of pair a) x (columns of pair b) only sees the rotations a and b
#pragma unroll
            for (int e0 = 0; e0 < NB; e0 += MT) {
                const int e = e0 + t;
                const bool valid = e < NB;
                const int ee = valid ? e : 0, pa = ee / half, pb = ee % half;
                const int p1 = rp[buf][pa], q1 = rq[buf][pa], p2 = rp[buf][pb], q2 = rq[buf][pb];
                const double ca = rc[buf][pa], sa = rs[buf][pa], cb = rc[buf][pb], sb = rs[buf][pb];
                double *const a00 = valid ? A + p1 * n + p2 : sink, *const a01 = valid ? A + p1 * n + q2 : sink;
                double *const a10 = valid ? A + q1 * n + p2 : sink, *const a11 = valid ? A + q1 * n + q2 : sink;
                const double b00 = *a00, b01 = *a01, b10 = *a10, b11 = *a11;
                double o00, o01, o10, o11;
                jr_block(ca, sa, cb, sb, b00, b01, b10, b11, o00, o01, o10, o11);
                // a dummy pair doubles its index: its two stores hit one address with one value
                *a00 = o00; *a01 = o01; *a10 = o10; *a11 = o11;
            }
            SYNC();
            params(r + 1, buf ^ 1);                        // (after the last round: computed and dropped)
#pragma unroll
            for (int e0 = 0; e0 < NV; e0 += MT) {          // V <- V J
                const int e = e0 + t;
                const bool valid = e < NV;
                const int ee = valid ? e : 0, pr = ee / n, k = ee % n;
                const int p = rp[buf][pr], q = rq[buf][pr];
                const double c = rc[buf][pr], sn = rs[buf][pr];
                double *const vp = valid ? V + k * n + p : sink + 1, *const vq = valid ? V + k * n + q : sink + 1;
                const double vkp = *vp, vkq = *vq;
                jr_vec(c, sn, vkp, vkq, *vp, *vq);
            }
            SYNC();
        }
    }
    for (int i = t; i < n; i += MT) wv[i] = A[i * n + i];
    SYNC();
}
DEV void w_jacobi(double *A, int n, double *wv, double *V, double *tmp, int t) {
    if (n == 21) w_jacobi_pipe<21>(A, wv, V, t);
    else if (n == 6) w_jacobi_pipe<6>(A, wv, V, t);
    else w_jacobi_t<0>(A, n, wv, V, tmp, t);
}
// log(det(A)) of a symmetric positive-definite matrix by unpivoted elimination (lane-parallel)
DEV double w_logdet_spd(const double *A, int n, double *W, int t) {
    for (int e = t; e < n * n; e += MT) W[e] = A[e];
    SYNC();
    double ld = 0;
    for (int k = 0; k < n; k++) {
        const double d = W[k * n + k];
        ld += log(d);
        SYNC();
        for (int e = t; e < n * n; e += MT) { const int i = e / n, j = e % n; if (i > k && j > k) W[e] -= W[i * n + k] * W[k * n + j] / d; }
        SYNC();
    }
    return ld;
}
// Sigma = (Jk U) D^-1 (Jk U)^T over the kept eigenpairs (keep[i] != 0); Jk is rows x n
DEV void w_project_cov(const double *Jk, int rows, int n, const double *V, const double *wv, const int *keep, double *JU, double *Sigma, int t) {
    for (int e = t; e < rows * n; e += MT) { const int a = e / n, k = e % n; double s = 0; for (int c = 0; c < n; c++) s += Jk[a * n + c] * V[c * n + k]; JU[e] = s; }
    SYNC();
    for (int e = t; e < rows * rows; e += MT) {
        const int a = e / rows, b = e % rows; double s = 0;
        for (int k = 0; k < n; k++) if (keep[k]) s += JU[a * n + k] * (1.0 / wv[k]) * JU[b * n + k];
        Sigma[e] = s;
    }
    SYNC();
}
DEV double w_det(const double *A, int n, double *W, int *piv, int t) {     // via LU with partial pivoting (single lane; n <= 21)
    double det = 1;
    for (int e = t; e < n * n; e += MT) W[e] = A[e];
    SYNC();
    if (t == 0) {
        for (int k = 0; k < n; k++) {
            int p = k; double best = fabs(W[k * n + k]);
            for (int i = k + 1; i < n; i++) if (fabs(W[i * n + k]) > best) { best = fabs(W[i * n + k]); p = i; }
            if (best == 0.0) { det = 0; break; }
            if (p != k) { for (int j = 0; j < n; j++) { const double tmp = W[k * n + j]; W[k * n + j] = W[p * n + j]; W[p * n + j] = tmp; } det = -det; }
            det *= W[k * n + k];
            for (int i = k + 1; i < n; i++) { const double f = W[i * n + k] / W[k * n + k]; for (int j = k + 1; j < n; j++) W[i * n + j] -= f * W[k * n + j]; }
        }
        W[0] = det;
    }
    SYNC();
    det = W[0];
    SYNC();
    return det;
}

// clears the marg records (padding included) so downloads are deterministic; valid = margin_old
__global__ void k_marg_clear(DevBatch d) {
    const int w = blockIdx.x, t = threadIdx.x;
    if (ISV_SEQ_IDLE(d, w)) return;        // (a resident sequence without a frame this step keeps the record of its last solve: its slide is still pending)
    double *z = reinterpret_cast<double *>(&d.marg[w]);
    for (int e = t; e < (int)(sizeof(isv_marg_result_t) / sizeof(double)); e += blockDim.x) z[e] = 0.0;
}

// MargForward (+ the pose-graph edge); the record is cleared by k_marg_clear beforehand
// (round 5) MTF = 256 threads: the two phases that walk the window's landmarks -- the Jacobians of the landmarks hosted in frame 0 (a lane per
// landmark of the WINDOW: 2000 of them in BASELINE config 5) and their sums into the 12 x 12 pose blocks -- use four wavefronts, the 12 x 12
// algebra behind them the first (the others have left by then).  Slots and sums in the same order as the one-wavefront kernel: same bits.
// MTF = 64 for batches that fill the GPU (1024 windows: the four-wavefront form takes 350 us against 110 -- the extra wavefronts only add
// barriers where every SIMD already holds windows); chosen per launch from the batch size, the bits are the same.
template <int MTF>
__global__ __launch_bounds__(MTF) void k_marg_fwd(DevBatch d) {
    // (LDS decides how many windows a CU holds next to k_marg_bwd's: 4 + 4 workgroups need <= 40 KB for the pair)
    __shared__ double Lam[144], M1[144], M2[144], Wk[936], Vv[36], wv[8], JU[36];
    __shared__ double sJ[3 * 36];
    __shared__ int keep[8], piv[4], wcnt[4];
    if (MTF == 64 && threadIdx.x >= 1 && threadIdx.x < 4) wcnt[threadIdx.x] = 0;       // (one wavefront: the other three counts of the slot arithmetic are zero)
    const int w = blockIdx.x, t = threadIdx.x;
    isv_marg_result_t &out = d.marg[w];
    if (!d.margin_old[w] || ISV_SEQ_IDLE(d, w)) return;
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64();
#endif
    const int N = d.N, v = d.Nvo;
    const double *pose = d.pose + (size_t)w * N * 7, *sb = d.sb + (size_t)w * N * 9, *ex = d.ex + (size_t)w * 7;
    const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    (void)sb; (void)ex; (void)l0; (void)l1; (void)v;
    // ================= MargForward =================
    // landmarks hosted in frame 0 (forwardProjectiontoSparsify / MargPointIdx, estimator.cpp:1082-1087)
    double *Jw = d.marg_scratch + (size_t)l0 * 26;          // [n0][26] weighted [J_T1(2x6) | J_T0(2x6) | J_l(2)] rows interleaved
    int n0 = 0;
    {
        double Ri[9], Rj[9], ric[9];
        q_to_R(q_from_pose(pose), Ri); q_to_R(q_from_pose(pose + 7), Rj); q_to_R(q_from_pose(ex), ric);
        const double ident[4] = {1, 0, 0, 1};
        for (int base = l0; base < l1; base += MTF) {
            const int l = base + t, lane = t & 63, wvf = t >> 6;      // (MTF = 64: one wavefront, wcnt[1..3] stay zero)
            const bool is0 = l < l1 && d.lm_host[l] == 0;
            const unsigned long long m = __ballot(is0);
            if (lane == 0) wcnt[wvf] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) { const int c = wcnt[q]; total += c; if (q < wvf) before += c; }
            if (is0) {
                const int slot = n0 + before + __popcll(m & ((1ull << lane) - 1));
                const int f = d.lm_f0[l];                     // factor 0 -> 1
                double r0, r1, Ji[12], Jj[12], Jl[2];
                const double *pi3 = d.lm_pts_i + (size_t)l * 3;
                proj_factor<true>(Ri, pose, Rj, pose + 7, ric, ex, ident, d.lam[l], pi3[0], pi3[1], pi3[2],
                                  d.f_pts_j[(size_t)f * 2], d.f_pts_j[(size_t)f * 2 + 1], r0, r1, Ji, Jj, Jl);
                // weight with sqrt_info (infoMatrix = sqrt_info^T sqrt_info, estimator.cpp:1173)
                const double *sq = d.proj_sqrt_info;
                double *o = Jw + (size_t)slot * 26;
                for (int c = 0; c < 6; c++) {
                    o[c] = sq[0] * Jj[c] + sq[1] * Jj[6 + c];        o[13 + c] = sq[2] * Jj[c] + sq[3] * Jj[6 + c];          // T1 = frame 1
                    o[6 + c] = sq[0] * Ji[c] + sq[1] * Ji[6 + c];    o[13 + 6 + c] = sq[2] * Ji[c] + sq[3] * Ji[6 + c];      // T0 = frame 0
                }
                o[12] = sq[0] * Jl[0] + sq[1] * Jl[1]; o[25] = sq[2] * Jl[0] + sq[3] * Jl[1];
            }
            n0 += total;
            __syncthreads();                               // (wcnt is rewritten by the next chunk)
        }
    }
    __threadfence_block();
    SYNC();
    MSTAMP(0);
    // raw pose block Hraw (12x12, order [T1, T0]) in Lam[0..143]; Schur-reduced over the landmarks in M1[0..143]
    {
        // (a thread per entry of the 12 x 12 blocks; the landmarks in slot order: the sums of the one-wavefront kernel)
        constexpr int NE = (144 + MTF - 1) / MTF;            // entries per thread: 1 (MTF = 256) or 3 (MTF = 64)
        double hr[NE], hs[NE];
#pragma unroll
        for (int i = 0; i < NE; i++) { hr[i] = 0; hs[i] = 0; }
        for (int mb = 0; mb < n0; mb += 32) {
            const int cnt = (n0 - mb) < 32 ? (n0 - mb) : 32;
            for (int e = t; e < cnt * 26; e += MTF) Wk[e] = Jw[(size_t)mb * 26 + e];
            SYNC();
            for (int m = 0; m < cnt; m++) {
                const double *o = Wk + m * 26;
                const double dmi = 1.0 / (o[12] * o[12] + o[25] * o[25]);
#pragma unroll
                for (int i = 0; i < NE; i++) {
                    const int e = t + MTF * i;
                    if (e < 144) {
                        const int a = e / 12, b = e % 12;
                        const double h = o[a] * o[b] + o[13 + a] * o[13 + b];
                        const double ba = o[a] * o[12] + o[13 + a] * o[25], bb = o[b] * o[12] + o[13 + b] * o[25];
                        hr[i] += h; hs[i] += h - ba * bb * dmi;
                    }
                }
            }
            SYNC();
        }
#pragma unroll
        for (int i = 0; i < NE; i++) { const int e = t + MTF * i; if (e < 144) { Lam[e] = hr[i]; M1[e] = hs[i]; } }
    }
    SYNC();
    if (t >= MT) return;                                    // the 12 x 12 algebra below: the first wavefront
    MSTAMP(1);
    // pose prior on T0 and relative-pose edge (0,1): unweighted Jacobians, info = S^T S
    if (t == 0) {
        const isv_se3_prior_t &f = d.se3[w];
        Quat rr = so3_mul(q_conj(q_from_R(f.R)), q_normalized(q_from_pose(pose)));
        double lg[3], J3[9];
        so3_log(rr, lg); so3_rjac_inv(lg, J3);
        for (int k = 0; k < 36; k++) sJ[k] = 0;
        sJ[0] = sJ[7] = sJ[14] = 1.0;
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sJ[(3 + a) * 6 + 3 + b] = J3[a * 3 + b];
        double r6[6];
        relpose_jac(d.relpose[(size_t)w * (v - 1)].delta_t, d.relpose[(size_t)w * (v - 1)].delta_R, pose, pose + 7, r6, sJ + 36, sJ + 72);
    }
    SYNC();
    {   // SJ = S * J (6x6) for the se3 prior -> M2[0..35]; add (SJ)^T (SJ) to the T0 block
        w_mm(d.se3[w].sqrt_info, sJ, M2, 6, 6, 6, t);
        for (int e = t; e < 36; e += MT) { const int a = e / 6, b = e % 6; double s = 0; for (int k = 0; k < 6; k++) s += M2[k * 6 + a] * M2[k * 6 + b]; Lam[(6 + a) * 12 + 6 + b] += s; M1[(6 + a) * 12 + 6 + b] += s; }
        SYNC();
        // relpose: J12 = [Jj (T1 cols 0..5) | Ji (T0 cols 6..11)], SJ12 = S * J12 (6x12) in M2
        const double *S = d.relpose[(size_t)w * (v - 1)].sqrt_info;
        for (int e = t; e < 72; e += MT) { const int a = e / 12, c = e % 12; double s = 0; for (int k = 0; k < 6; k++) s += S[a * 6 + k] * (c < 6 ? sJ[72 + k * 6 + c] : sJ[36 + k * 6 + c - 6]); M2[e] = s; }
        SYNC();
        for (int e = t; e < 144; e += MT) { const int a = e / 12, b = e % 12; double s = 0; for (int k = 0; k < 6; k++) s += M2[k * 12 + a] * M2[k * 12 + b]; Lam[e] += s; M1[e] += s; }
        SYNC();
    }
    MSTAMP(2);
    // (i) pose-graph edge (estimator.cpp:1240-1283)
    isv_relpose_t &pg = out.combined.relative_pose;
    if (t == 0) {
        Quat Qi = q_from_pose(pose), Qj = q_from_pose(pose + 7);
        double dd[3] = {pose[7] - pose[0], pose[8] - pose[1], pose[9] - pose[2]};
        q_rot(q_inv(Qi), dd, pg.delta_t);
        q_to_R(q_mul(q_inv(Qi), Qj), pg.delta_R);
        pg.imu_i = 0; pg.imu_j = 1;
        double r6[6];
        relpose_jac(pg.delta_t, pg.delta_R, pose, pose + 7, r6, sJ + 36, sJ + 72);
        // J (6x12) = [jacobians[0] | jacobians[1]] = [d/d pose0 | d/d pose1]  (reference column order, :1253-1255)
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { M2[a * 12 + b] = sJ[36 + a * 6 + b]; M2[a * 12 + 6 + b] = sJ[72 + a * 6 + b]; }
    }
    SYNC();
    {   // pseudo-inverse through the eigen-decomposition of J J^T (6x6), threshold 1e-8 * 12 on singular values
        double *G = Wk, *Gi = Wk + 64, *Jp = Wk + 128;            // Jp: 12 x 6
        w_mm_nt(M2, M2, G, 6, 12, 6, t);
        for (int e = t; e < 36; e += MT) JU[e] = G[e];
        SYNC();
        w_jacobi(JU, 6, wv, Vv, nullptr, t);
        double smax2 = 0; for (int k = 0; k < 6; k++) smax2 = fmax(smax2, wv[k]);
        const double thr = 1e-8 * 12;
        for (int e = t; e < 36; e += MT) {
            const int a = e / 6, b = e % 6; double s = 0;
            for (int k = 0; k < 6; k++) { const double sv = sqrt(fmax(wv[k], 0.0)); if (sv > thr * sqrt(smax2)) s += Vv[a * 6 + k] * Vv[b * 6 + k] / wv[k]; }
            Gi[e] = s;
        }
        SYNC();
        w_mm_tn(M2, Gi, Jp, 12, 6, 6, t);                       // J^T Ginv : 12 x 6
        double *Tm = Wk + 256, *Om = Wk + 400, *cov = Wk + 448;
        w_mm_tn(Jp, Lam, Tm, 6, 12, 12, t);                      // Jp^T Lam_rp : 6 x 12
        w_mm(Tm, Jp, Om, 6, 12, 6, t);
        w_inv(Om, 6, cov, Wk + 600, piv, t);
        w_chol_upper(Om, 6, JU, Wk + 800, t);
        for (int e = t; e < 36; e += MT) { pg.sqrt_info[e] = JU[e]; out.combined.covRel[e] = cov[e]; }
        SYNC();
    }
    if (t == 0) {
        out.combined.has_rollpitch = 0;
        if (d.n_rp[w] > 0 && d.rollpitch[(size_t)w * d.max_rp].index == 0) {
            out.combined.has_rollpitch = 1;
            out.combined.rollpitch = d.rollpitch[(size_t)w * d.max_rp];
            const double *s = out.combined.rollpitch.sqrt_info;
            const double a = s[0] * s[0] + s[2] * s[2], b = s[0] * s[1] + s[2] * s[3], c = s[1] * s[1] + s[3] * s[3], det = a * c - b * b;
            out.combined.covAbs[0] = c / det; out.combined.covAbs[1] = -b / det; out.combined.covAbs[2] = -b / det; out.combined.covAbs[3] = a / det;
        }
        out.combined.distance = sqrt(pg.delta_t[0] * pg.delta_t[0] + pg.delta_t[1] * pg.delta_t[1] + pg.delta_t[2] * pg.delta_t[2]);
        out.combined.ts = d.header0[w];
        for (int k = 0; k < 9; k++) out.combined.Ri[k] = d.Rs[(size_t)w * N * 9 + k];
        for (int k = 0; k < 3; k++) out.combined.ti[k] = d.Ps[(size_t)w * N * 3 + k];
    }
    MSTAMP(3);
    // (ii) new pose prior on T1: eliminate T0 from the landmark-reduced block M1
    {
        double *A66 = Wk, *Ainv = Wk + 64, *Lp = Wk + 128, *cov = Wk + 192, *covi = Wk + 256, *X = Wk + 320, *T = Wk + 384;
        for (int e = t; e < 36; e += MT) A66[e] = M1[(6 + e / 6) * 12 + 6 + e % 6];
        SYNC();
        w_inv(A66, 6, Ainv, Wk + 600, piv, t);
        for (int e = t; e < 36; e += MT) {
            const int a = e / 6, b = e % 6; double s = M1[a * 12 + b];
            for (int p = 0; p < 6; p++) { double tt = 0; for (int q = 0; q < 6; q++) tt += Ainv[p * 6 + q] * M1[b * 12 + 6 + q]; s -= M1[a * 12 + 6 + p] * tt; }
            Lp[e] = s;
        }
        SYNC();
        isv_se3_prior_t &fp = out.forward_pose_prior;
        if (t == 0) {
            for (int k = 0; k < 3; k++) fp.t[k] = pose[7 + k];
            q_to_R(q_from_pose(pose + 7), fp.R);
            fp.index = 0;
            // Jr = SE3PriorFactor(P1,Q1)::EvaluateOnlyJacobians(para_Pose[1])
            Quat rr = so3_mul(q_conj(q_from_R(fp.R)), q_normalized(q_from_pose(pose + 7)));
            double lg[3], J3[9];
            so3_log(rr, lg); so3_rjac_inv(lg, J3);
            for (int k = 0; k < 36; k++) sJ[k] = 0;
            sJ[0] = sJ[7] = sJ[14] = 1.0;
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) sJ[(3 + a) * 6 + 3 + b] = J3[a * 3 + b];
        }
        SYNC();
        // rank of Lp with threshold 1e-16 (FullPivHouseholderQR, estimator.cpp:8,1304): pivots of a full-pivot elimination
        int rank = 6;
        if (t == 0) {
            double Aq[36]; for (int k = 0; k < 36; k++) Aq[k] = Lp[k];
            double pv[6], maxp = 0; rank = 0;
            for (int k = 0; k < 6; k++) pv[k] = 0;
            for (int k = 0; k < 6; k++) {
                int pi = k, pj = k; double best = 0;
                for (int i = k; i < 6; i++) for (int j = k; j < 6; j++) if (fabs(Aq[i * 6 + j]) > best) { best = fabs(Aq[i * 6 + j]); pi = i; pj = j; }
                pv[k] = best; if (best > maxp) maxp = best;
                if (best == 0) break;
                for (int j = 0; j < 6; j++) { const double tmp2 = Aq[k * 6 + j]; Aq[k * 6 + j] = Aq[pi * 6 + j]; Aq[pi * 6 + j] = tmp2; }
                for (int i = 0; i < 6; i++) { const double tmp2 = Aq[i * 6 + k]; Aq[i * 6 + k] = Aq[i * 6 + pj]; Aq[i * 6 + pj] = tmp2; }
                for (int i = k + 1; i < 6; i++) { const double f = Aq[i * 6 + k] / Aq[k * 6 + k]; for (int j = k; j < 6; j++) Aq[i * 6 + j] -= f * Aq[k * 6 + j]; }
            }
            for (int k = 0; k < 6; k++) if (pv[k] > 1e-16 * maxp) rank++;
            piv[1] = rank;
        }
        SYNC();
        rank = piv[1];
        if (rank == 6) {
            w_inv(Lp, 6, cov, Wk + 600, piv, t);
            w_mm(sJ, cov, T, 6, 6, 6, t);
            w_mm_nt(T, sJ, covi, 6, 6, 6, t);
        } else {
            for (int e = t; e < 36; e += MT) JU[e] = Lp[e];
            SYNC();
            w_jacobi(JU, 6, wv, Vv, nullptr, t);
            if (t < 6) keep[t] = wv[t] > d.alpha_cut;
            SYNC();
            w_project_cov(sJ, 6, 6, Vv, wv, keep, JU, covi, t);
        }
        w_inv(covi, 6, X, Wk + 600, piv, t);
        double kld = 0;
        if (rank == 6) {
            double *phi = Wk + 448, *pc = Wk + 512;
            w_mm_tn(sJ, X, T, 6, 6, 6, t);
            w_mm(T, sJ, phi, 6, 6, 6, t);
            w_mm(phi, cov, pc, 6, 6, 6, t);
            const double dphi = w_det(phi, 6, Wk + 900, piv, t), dcov = w_det(cov, 6, Wk + 900, piv, t);
            double a = 0; for (int k = 0; k < 6; k++) a += pc[k * 6 + k];
            kld = 0.5 * (a - log(dphi) - log(dcov) - 6);
        }
        w_chol_upper(X, 6, JU, Wk + 800, t);
        for (int e = t; e < 36; e += MT) fp.sqrt_info[e] = JU[e];
        if (t == 0) { out.forward_kld = kld; out.n_marg_landmarks = n0; out.valid = 1; }
        SYNC();
    }
    MSTAMP(4);
}
template __global__ void k_marg_fwd<64>(DevBatch);
template __global__ void k_marg_fwd<256>(DevBatch);

// ------------------------------------------------------------------------------------------
// The eigen-decomposition of MargBackward's 21 x 21 marginal as a kernel of its own, FOUR wavefronts per window.  One
// wavefront issues ~300 instructions per Jacobi round (11 rotation parameters, 121 2x2 blocks in two passes, 231
// eigenvector items in four); here a round is one item per lane:
//   * wavefronts 0-1: lane e < half^2 owns the 2x2 block (rows of pair e / half) x (columns of pair e % half);
//   * wavefronts 2-3: lane e owns the rotation of pair e / half applied to eigenvector rows 2 (e % half), + 1
//     (the eigenvectors never enter the rotation parameters, so this runs beside the block update);
//   * then lanes 0..half-1 of wavefront 0 compute the next round's parameters from the updated A (double-buffered
//     parameter table), the other wavefronts wait at the barrier: two barriers per round, ~210 instructions per window.
// Rotations, their order and the arithmetic of each item (jr_params / jr_block / jr_vec) are those of w_jacobi_pipe: the
// result is bitwise the same.  In: Lp at ws[0..n^2).  Out: V at ws[896..), eigenvalues at ws[1344..).
template <int NC>
__global__ __launch_bounds__(256) void k_marg_jacobi(DevBatch d) {
    constexpr int n = NC, m = n + (n & 1), half = m / 2, NI = half * half, NN = n * n;
    static_assert(NI <= 128 && half <= 32, "one item per lane of two wavefronts");
    __shared__ double sA[NN + 1], sV[NN + 1], red[8], sink[2], rc[2][32], rs[2][32];
    __shared__ int rp[2][32], rq[2][32];
    const int w = blockIdx.x, t = threadIdx.x;
    if (!d.margin_old[w] || ISV_SEQ_IDLE(d, w)) return;
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64();
#endif
    double *const ws = d.marg_ws + (size_t)w * ISV_MARG_WS;
    for (int e = t; e < NN; e += 256) { sA[e] = ws[e]; sV[e] = (e / n == e % n) ? 1.0 : 0.0; }
    __syncthreads();
    const bool is_block = t < 128;                    // wavefront-uniform
    const int e = is_block ? t : t - 128;
    const bool valid = e < NI;
    const int i0 = valid ? e / half : 0, i1 = valid ? e % half : 0;
    // pair k of round r in the round-robin tournament; the pair holding the phantom index (odd n) doubles its real one
    auto params = [&](int r, int buf) {
        if (t < half) {
            const int k = t;
            int a = r + k, b = r + m - 1 - k;
            a = a >= m - 1 ? a - (m - 1) : a;
            b = b >= m - 1 ? b - (m - 1) : b;
            a = k == 0 ? m - 1 : a; b = k == 0 ? r : b;
            const int p = a < b ? a : b, q0 = a < b ? b : a, q = q0 < n ? q0 : p;
            const double apq = sA[p * n + q], aqq = sA[q * n + q], app = sA[p * n + p];
            double c, sn;
            jr_params(app, aqq, apq, p != q && apq != 0.0, c, sn);
            rp[buf][k] = p; rq[buf][k] = q; rc[buf][k] = c; rs[buf][k] = sn;
        }
    };
    bool polish = false;
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0, dg = 0;
        for (int x = t; x < NN; x += 256) { const int i = x / n, j = x % n; const double v = sA[x]; if (j > i) off += v * v; else if (i == j) dg += v * v; }
        for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); dg += __shfl_xor(dg, o); }
        if ((t & 63) == 0) { red[(t >> 6) * 2] = off; red[(t >> 6) * 2 + 1] = dg; }
        params(0, 0);
        __syncthreads();
        // (the sums of the one-wavefront version run over the same lanes in another order: the test below is a
        // threshold many orders of magnitude wide, not a value that is carried on)
        off = red[0] + red[2] + red[4] + red[6]; dg = red[1] + red[3] + red[5] + red[7];
        if (off <= 1e-60 || off <= 1e-32 * dg || polish) break;      // (w_jacobi_t's rule: one more sweep after the 1e-28 threshold)
        if (off <= 1e-28 * dg) polish = true;
        for (int r = 0; r < m - 1; r++) {
            const int buf = r & 1;
            if (is_block) {
                const int p1 = rp[buf][i0], q1 = rq[buf][i0], p2 = rp[buf][i1], q2 = rq[buf][i1];
                const double ca = rc[buf][i0], sa = rs[buf][i0], cb = rc[buf][i1], sb = rs[buf][i1];
                double *const a00 = valid ? sA + p1 * n + p2 : sink, *const a01 = valid ? sA + p1 * n + q2 : sink;
                double *const a10 = valid ? sA + q1 * n + p2 : sink, *const a11 = valid ? sA + q1 * n + q2 : sink;
                const double b00 = *a00, b01 = *a01, b10 = *a10, b11 = *a11;
                double o00, o01, o10, o11;
                jr_block(ca, sa, cb, sb, b00, b01, b10, b11, o00, o01, o10, o11);
                *a00 = o00; *a01 = o01; *a10 = o10; *a11 = o11;      // a dummy pair doubles its index: two stores, one address, one value
            } else {
                const int p = rp[buf][i0], q = rq[buf][i0];
                const double c = rc[buf][i0], sn = rs[buf][i0];
                const int k0 = 2 * i1, k1 = (2 * i1 + 1 < n) ? 2 * i1 + 1 : k0;       // odd n: the last lane's second row repeats its first
                double *const v0p = valid ? sV + k0 * n + p : sink + 1, *const v0q = valid ? sV + k0 * n + q : sink + 1;
                double *const v1p = valid ? sV + k1 * n + p : sink + 1, *const v1q = valid ? sV + k1 * n + q : sink + 1;
                const double a0 = *v0p, b0 = *v0q, a1 = *v1p, b1 = *v1q;
                double r0p, r0q, r1p, r1q;
                jr_vec(c, sn, a0, b0, r0p, r0q); jr_vec(c, sn, a1, b1, r1p, r1q);
                *v0p = r0p; *v0q = r0q; *v1p = r1p; *v1q = r1q;
            }
            __syncthreads();
            if (r + 1 < m - 1) { params(r + 1, buf ^ 1); __syncthreads(); }
        }
    }
    for (int x = t; x < NN; x += 256) ws[896 + x] = sV[x];
    if (t < n) ws[1344 + t] = sA[t * n + t];
#ifdef ISV_STAMP
    if (t == 0) d.dbg[(size_t)w * 64 + 32 + 6] += (double)(wall_clock64() - t_last);
#endif
}
template __global__ void k_marg_jacobi<21>(DevBatch);

// MargBackward.  In three launches: <0> builds the 21 x 21 marginal Lp and the recovered factors' Jacobian Jr (left in
// d.marg_ws), k_marg_jacobi<21> eigen-decomposes Lp, <1> projects the covariance onto the factors and takes the KLD.
// In one launch: <2>, the eigen-decomposition by the window's one wavefront (w_jacobi_pipe).
template <int PART>
__global__ __launch_bounds__(MT) void k_marg_bwd(DevBatch d) {
    // LDS is what limits residency (one wavefront per window, all windows should be resident at once next to
    // k_marg_fwd's): Lam (30x30) is dead once the 21x21 marginal Lp exists, so the recovered-factor Jacobian Jr
    // and the projection scratch JU live in its space; M1 (Lp) is dead after the eigen-decomposition copy and
    // then holds the block-diagonal information Xall; Vv is dead once (Jr U) is formed and then holds Ak.
    __shared__ double Lam[900], M1[450], M2[450], Wk[464], Vv[441], wv[32];
    __shared__ double sJ[2 * 36];
    __shared__ int keep[32], kpos[32], piv[4];
    double *const Jr = Lam, *const JU = Lam + 441;
    const int w = blockIdx.x, t = threadIdx.x;
    isv_marg_result_t &out = d.marg[w];
    if (!d.margin_old[w] || ISV_SEQ_IDLE(d, w)) return;
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64();
#endif
    const int N = d.N, v = d.Nvo;
    const double *pose = d.pose + (size_t)w * N * 7, *sb = d.sb + (size_t)w * N * 9, *ex = d.ex + (size_t)w * 7;
    const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    (void)sb; (void)ex; (void)l0; (void)l1; (void)v;
    double *const ws = d.marg_ws + (size_t)w * ISV_MARG_WS;
    isv_relpose_t &rp = out.backward_relpose; isv_linear9_t &vb = out.backward_vb; isv_rollpitch_t &gp = out.backward_rollpitch;
    if constexpr (PART != 1) {
    // ================= MargBackward =================
    // order: T1 = frame v (@0), VB1 (@6), T0 = frame v-1 (@15), VB0 (@21)
    for (int e = t; e < 900; e += MT) Lam[e] = 0.0;
    SYNC();
    {
        const double *S = d.lin9[w].sqrt_info;            // VB prior on sb[v-1]: J = I, info = S^T S
        for (int e = t; e < 81; e += MT) { const int a = e / 9, b = e % 9; double s = 0; for (int k = 0; k < 9; k++) s += S[k * 9 + a] * S[k * 9 + b]; Lam[(21 + a) * 30 + 21 + b] += s; }
        SYNC();
        // IMU factor v-1 -> v: weighted Jacobian = strip's J if the strips were taken at this point; they were not
        // (the last accepted point was never re-linearised), so rebuild: raw J through the linearise kernel's layout.
    }
    // raw (unweighted) IMU Jacobian 15x30 in M1 (columns: pose_i 0..5, sb_i 6..14, pose_j 15..20, sb_j 21..29)
    for (int e = t; e < 450; e += MT) M1[e] = 0.0;
    SYNC();
    if (t == 0) {
        const size_t f = (size_t)w * (N - 1) + (v - 1);
        const double *rec = d.imu_in + f * ISV_IMU_IN;
        const double *pi = pose + 7 * (v - 1), *pj = pi + 7, *si = sb + 9 * (v - 1), *sj = si + 9;
        Quat Qi = q_from_pose(pi), Qj = q_from_pose(pj), Qii = q_inv(Qi);
        const double dt = rec[IMU_DT];
        double dbg[3], tt[3], u[3], o1[3], o2[3], RiT[9], S1[9], S2[9], B1[9], B2[9], L[9], Rr[9], T[9];
        for (int k = 0; k < 3; k++) dbg[k] = si[6 + k] - rec[IMU_LBG + k];
        Quat dq = Quat{rec[IMU_DQ + 3], rec[IMU_DQ], rec[IMU_DQ + 1], rec[IMU_DQ + 2]};
        m3v(rec + IMU_DQ_DBG, dbg, tt);
        Quat cdq = q_mul(dq, q_delta(tt));
        q_to_R(Qii, RiT);
        for (int k = 0; k < 3; k++) u[k] = 0.5 * d.G[k] * dt * dt + pj[k] - pi[k] - si[k] * dt;
        q_rot(Qii, u, o1);
        for (int k = 0; k < 3; k++) u[k] = d.G[k] * dt + sj[k] - si[k];
        q_rot(Qii, u, o2);
        skew3(o1, S1); skew3(o2, S2);
        Quat aq = q_mul(q_inv(Qj), Qi);
        qleft33(aq, L); qright33(cdq, Rr); m3_mul(L, Rr, B1);
        const double av[3] = {aq.x, aq.y, aq.z}, bv[3] = {cdq.x, cdq.y, cdq.z};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) B1[r * 3 + c] += av[r] * (-bv[c]);
        qleft33(q_mul(q_mul(q_inv(Qj), Qi), dq), L);
        m3_mul(L, rec + IMU_DQ_DBG, T);
        qleft33(q_mul(q_mul(q_inv(cdq), Qii), Qj), B2);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
            const int ab = a * 3 + b;
            M1[(0 + a) * 30 + 0 + b] = -RiT[ab]; M1[(0 + a) * 30 + 3 + b] = S1[ab]; M1[(3 + a) * 30 + 3 + b] = -B1[ab]; M1[(6 + a) * 30 + 3 + b] = S2[ab];
            M1[(0 + a) * 30 + 6 + b] = -RiT[ab] * dt; M1[(0 + a) * 30 + 9 + b] = -rec[IMU_DP_DBA + ab]; M1[(0 + a) * 30 + 12 + b] = -rec[IMU_DP_DBG + ab];
            M1[(3 + a) * 30 + 12 + b] = -T[ab]; M1[(6 + a) * 30 + 6 + b] = -RiT[ab]; M1[(6 + a) * 30 + 9 + b] = -rec[IMU_DV_DBA + ab];
            M1[(6 + a) * 30 + 12 + b] = -rec[IMU_DV_DBG + ab];
            M1[(9 + a) * 30 + 9 + b] = (a == b) ? -1.0 : 0.0; M1[(12 + a) * 30 + 12 + b] = (a == b) ? -1.0 : 0.0;
            M1[(0 + a) * 30 + 15 + b] = RiT[ab]; M1[(3 + a) * 30 + 18 + b] = B2[ab]; M1[(6 + a) * 30 + 21 + b] = RiT[ab];
            M1[(9 + a) * 30 + 24 + b] = (a == b) ? 1.0 : 0.0; M1[(12 + a) * 30 + 27 + b] = (a == b) ? 1.0 : 0.0;
        }
    }
    SYNC();
    {   // SJ = sqrt_info (15x15) * J (15x30) -> M2; Lam += (SJ)^T (SJ) with the column permutation to [T1 VB1 T0 VB0]
        const double *S = d.imu_sqrt + ((size_t)w * (N - 1) + (v - 1)) * 225;
        for (int e = t; e < 450; e += MT) { const int a = e / 30, c = e % 30; double s = 0; for (int k = 0; k < 15; k++) s += S[a * 15 + k] * M1[k * 30 + c]; M2[e] = s; }
        SYNC();
        for (int e = t; e < 900; e += MT) {
            const int a = e / 30, b = e % 30;            // factor column order: [pose_i sb_i pose_j sb_j] -> Lam index: i-part at 15.., j-part at 0..
            const int ga = (a < 15) ? 15 + a : a - 15, gb = (b < 15) ? 15 + b : b - 15;
            double s = 0; for (int k = 0; k < 15; k++) s += M2[k * 30 + a] * M2[k * 30 + b];
            Lam[ga * 30 + gb] += s;
        }
        SYNC();
    }
    double *Lp = M1;                                     // 21 x 21
    {
        double *A99 = Wk, *Ainv = Wk + 100;
        for (int e = t; e < 81; e += MT) A99[e] = Lam[(21 + e / 9) * 30 + 21 + e % 9];
        SYNC();
        w_inv(A99, 9, Ainv, Wk + 200, piv, t);
        for (int e = t; e < 441; e += MT) {
            const int a = e / 21, b = e % 21; double s = Lam[a * 30 + b];
            for (int p = 0; p < 9; p++) { double tt = 0; for (int q = 0; q < 9; q++) tt += Ainv[p * 9 + q] * Lam[b * 30 + 21 + q]; s -= Lam[a * 30 + 21 + p] * tt; }
            Lp[e] = s;
        }
        SYNC();
    }
    // recovered factors at the current estimate
    for (int e = t; e < 441; e += MT) Jr[e] = 0.0;
    SYNC();
    if (t == 0) {
        const double *PSi = pose + 7 * (v - 1), *PSj = pose + 7 * v;
        Quat Qi = q_from_pose(PSi), Qj = q_from_pose(PSj);
        double dd[3] = {PSj[0] - PSi[0], PSj[1] - PSi[1], PSj[2] - PSi[2]};
        q_rot(q_inv(Qi), dd, rp.delta_t); q_to_R(q_mul(q_inv(Qi), Qj), rp.delta_R);
        rp.imu_i = v - 1; rp.imu_j = v;
        double r6[6];
        relpose_jac(rp.delta_t, rp.delta_R, PSi, PSj, r6, sJ, sJ + 36);
        for (int k = 0; k < 9; k++) vb.VB[k] = sb[9 * v + k];
        vb.index = v;
        q_to_R(Qi, gp.R); gp.index = v - 1;
        // RollPitchFactor(Qw)::EvaluateOnlyJacobians, YawFactor(Qw)::EvaluateOnlyJacobians at pose v-1
        Quat Ri = q_normalized(Qi), Rm = q_from_R(gp.R);
        double nZ[3] = {0, 0, -1.0}, vv[3], S[9], Rmm[9], Bm[9];
        q_rot(so3_mul(Rm, q_conj(Ri)), nZ, vv);
        skew3(vv, S); q_to_R(Rm, Rmm); m3_mul(S, Rmm, Bm);
        double exv[3] = {1, 0, 0}, ym[3], Rr[9], Sy[9], By[9];
        q_rot(q_inv(Qi), exv, ym);
        q_to_R(Ri, Rr); skew3(ym, Sy);
        for (int k = 0; k < 9; k++) Rr[k] = -Rr[k];
        m3_mul(Rr, Sy, By);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { Jr[a * 21 + 15 + b] += sJ[a * 6 + b]; Jr[a * 21 + b] += sJ[36 + a * 6 + b]; }
        for (int a = 0; a < 9; a++) Jr[(6 + a) * 21 + 6 + a] += 1.0;
        for (int a = 0; a < 2; a++) for (int b = 0; b < 3; b++) Jr[(15 + a) * 21 + 18 + b] += Bm[a * 3 + b];
        for (int a = 0; a < 3; a++) Jr[(17 + a) * 21 + 15 + a] += 1.0;
        for (int b = 0; b < 3; b++) Jr[20 * 21 + 18 + b] += By[3 + b];
    }
    SYNC();
    if constexpr (PART == 0) for (int e = t; e < 441; e += MT) { ws[e] = Lp[e]; ws[448 + e] = Jr[e]; }
    MSTAMP(5);
    if constexpr (PART == 2) {                       // one launch: the eigen-decomposition by this wavefront
        for (int e = t; e < 441; e += MT) M2[e] = Lp[e];
        SYNC();
        w_jacobi(M2, 21, wv, Vv, nullptr, t);
        if (t < 21) keep[t] = wv[t] > d.alpha_cut;
        SYNC();
        MSTAMP(6);
    }
    }   // PART != 1
    if constexpr (PART == 1) {
        // eigen-truncate Lp at ALPHA: k_marg_jacobi<21> left the eigenpairs in the workspace
        for (int e = t; e < 441; e += MT) { Jr[e] = ws[448 + e]; Vv[e] = ws[896 + e]; }
        if (t < 21) { const double ev = ws[1344 + t]; wv[t] = ev; keep[t] = ev > d.alpha_cut; }
        SYNC();
    }
    if constexpr (PART != 0) {
    {
        double *Sg = Wk, *Xi = Wk + 100, *Xall = M1;             // Xall: 21x21 block-diagonal information (M1 is free now)
        for (int e = t; e < 441; e += MT) Xall[e] = 0.0;
        SYNC();
        // the five recovered blocks: relpose rows 0..5, VB rows 6..14, roll-pitch 15..16, |position| 17..19, yaw 20
#define RECOVER(R0, NR, DST)                                                                        \
        do {                                                                                        \
            w_project_cov(Jr + (R0) * 21, (NR), 21, Vv, wv, keep, JU, Sg, t);                       \
            w_inv(Sg, (NR), Xi, Wk + 300, piv, t);                                                  \
            for (int e = t; e < (NR) * (NR); e += MT) Xall[((R0) + e / (NR)) * 21 + (R0) + e % (NR)] = Xi[e]; \
            SYNC();                                                                                 \
            if ((DST) != nullptr) {                                                                 \
                w_chol_upper(Xi, (NR), Sg, Wk + 200, t);                                            \
                for (int e = t; e < (NR) * (NR); e += MT) (DST)[e] = Sg[e];                         \
                SYNC();                                                                             \
            }                                                                                       \
        } while (0)
        double *const no_dst = nullptr;
        RECOVER(0, 6, rp.sqrt_info);
        RECOVER(6, 9, vb.sqrt_info);
        RECOVER(15, 2, gp.sqrt_info);
        RECOVER(17, 3, no_dst);
        RECOVER(20, 1, no_dst);
#undef RECOVER
#if defined(MARG_STOP) && MARG_STOP == 3
        return;
#endif
        MSTAMP(7);
        // zero test / KLD (estimator.cpp:1519-1534): A = (Jr U)^T X (Jr U) over the kept eigenpairs vs D
        int rank = 0; for (int k = 0; k < 21; k++) rank += keep[k];
        double *JUa = Wk, *XJU = JU, *A = M2;            // (Jr is dead once JUa exists, the projection scratch JU already is)
        for (int e = t; e < 21 * 21; e += MT) {
            const int a = e / 21, k = e % 21; double s = 0;
            for (int c = 0; c < 21; c++) s += Jr[a * 21 + c] * Vv[c * 21 + k];
            JUa[e] = s;                                   // all 21 columns; kept ones selected below
        }
        SYNC();
        w_mm(Xall, JUa, XJU, 21, 21, 21, t);
        for (int e = t; e < 441; e += MT) { const int a = e / 21, b = e % 21; double s = 0; for (int k = 0; k < 21; k++) s += JUa[k * 21 + a] * XJU[k * 21 + b]; A[e] = s; }
        SYNC();
        // restrict A to kept indices (rank x rank) in Wk+1000.. and evaluate trace / determinants
        double *Ak = Vv;                                  // (Vv was consumed by JUa above)
        if (t < 21) { int c = 0; for (int k = 0; k < t; k++) c += keep[k]; kpos[t] = c; }     // position of a kept index among the kept ones
        SYNC();
        for (int e = t; e < 441; e += MT) {
            const int a = e / 21, b = e % 21;
            if (keep[a] && keep[b]) Ak[kpos[a] * rank + kpos[b]] = A[e];
        }
        SYNC();
        const double ldA = w_logdet_spd(Ak, rank, JU, t);
        if (t == 0) {
            double tr = 0, ldinv = 0; int ia = 0;
            for (int a = 0; a < 21; a++) if (keep[a]) { tr += Ak[ia * rank + ia] / wv[a]; ldinv += log(1.0 / wv[a]); ia++; }
            out.backward_kld = 0.5 * (tr - ldA - ldinv - 21);
        }
        MSTAMP(8);
    }
    }   // PART != 0
}
template __global__ void k_marg_bwd<0>(DevBatch);
template __global__ void k_marg_bwd<1>(DevBatch);
template __global__ void k_marg_bwd<2>(DevBatch);

// ------------------------------------------------------------------------------------------
// Estimator::initFactorGraph, the part after its ceres::Solve (src/estimator.cpp:744-999): build the first prior
// factors from the solved estimate.  One wavefront per window; this runs once per sequence, so the (15 Vo)^2 /
// (6 Vo + 9)^2 matrices simply live in a global scratch (`per_window` doubles per window) and go through the same
// wave-cooperative helpers as the marginalisation kernels.
//   Lambda (15 Vo)^2 from the first Vo-1 IMU factors (unweighted J, omega = sqrt_info^T sqrt_info), order
//   [T0..T_{Vo-1}, VB_{Vo-1}, VB_0..VB_{Vo-2}] (:744-806); Schur out VB_0..VB_{Vo-2} (:810-817); RelativePose (i, i+1),
//   SE3 prior on pose 0, Linear9 on speed/bias Vo-1 at the estimate, unweighted Jacobians stacked into Jr (:821-920);
//   eigen-truncation at ALPHA, Sigma_i = (J_i U) D^-1 (J_i U)^T, sqrt_info = chol(Sigma_i^-1)^T (:927-974); KLD (:976-989).
size_t init_priors_scratch_doubles(int Vo) {
    const size_t n = 15 * (size_t)Vo, rr = 6 * (size_t)Vo + 9, mm = 9 * (size_t)(Vo - 1);
    return n * n + 4 * mm * mm + mm * rr + 9 * rr * rr + 1024;
}
__global__ __launch_bounds__(MT) void k_init_priors(DevBatch d, double *scratch, size_t per_window, double *kld_out) {
    __shared__ double M1[450], M2s[450], wv[64], tmp[128], sJ[72];
    __shared__ int keep[64], piv[4];
    const int w = blockIdx.x, t = threadIdx.x;
    const int N = d.N, V = d.Nvo, n = 15 * V, rr = 6 * V + 9, mm = 9 * (V - 1);
    const double *pose = d.pose + (size_t)w * N * 7, *sb = d.sb + (size_t)w * N * 9;
    double *p = scratch + (size_t)w * per_window;
    double *Lam = p; p += (size_t)n * n;
    double *Wgj = p; p += 2 * (size_t)mm * mm;
    double *Linv = p; p += (size_t)mm * mm;
    double *Lmm = p; p += (size_t)mm * mm;
    double *T = p; p += (size_t)mm * rr;
    double *Lp = p; p += rr * rr;
    double *M2 = p; p += rr * rr;
    double *Vv = p; p += rr * rr;
    double *Jr = p; p += rr * rr;
    double *JU = p; p += rr * rr;
    double *XJU = p; p += rr * rr;
    double *A = p; p += rr * rr;
    double *Xall = p; p += rr * rr;
    double *Ak = p; p += rr * rr;
    double *Wk = p;                                    // 1024 doubles of small scratch
    for (int e = t; e < n * n; e += MT) Lam[e] = 0.0;
    SYNC();
    for (int i = 0; i < V - 1; i++) {
        const size_t f = (size_t)w * (N - 1) + i;
        if (d.imu_skip[f]) continue;                   // uniform over the block
        // raw (unweighted) IMU Jacobian 15x30 (columns: pose_i 0..5, sb_i 6..14, pose_j 15..20, sb_j 21..29), imu_factor.h:161-265
        for (int e = t; e < 450; e += MT) M1[e] = 0.0;
        SYNC();
        if (t == 0) {
            const double *rec = d.imu_in + f * ISV_IMU_IN;
            const double *pi = pose + 7 * i, *pj = pi + 7, *si = sb + 9 * i, *sj = si + 9;
            Quat Qi = q_from_pose(pi), Qj = q_from_pose(pj), Qii = q_inv(Qi);
            const double dt = rec[IMU_DT];
            double dbg[3], tt[3], u[3], o1[3], o2[3], RiT[9], S1[9], S2[9], B1[9], B2[9], L[9], Rr[9], Tm[9];
            for (int k = 0; k < 3; k++) dbg[k] = si[6 + k] - rec[IMU_LBG + k];
            Quat dq = Quat{rec[IMU_DQ + 3], rec[IMU_DQ], rec[IMU_DQ + 1], rec[IMU_DQ + 2]};
            m3v(rec + IMU_DQ_DBG, dbg, tt);
            Quat cdq = q_mul(dq, q_delta(tt));
            q_to_R(Qii, RiT);
            for (int k = 0; k < 3; k++) u[k] = 0.5 * d.G[k] * dt * dt + pj[k] - pi[k] - si[k] * dt;
            q_rot(Qii, u, o1);
            for (int k = 0; k < 3; k++) u[k] = d.G[k] * dt + sj[k] - si[k];
            q_rot(Qii, u, o2);
            skew3(o1, S1); skew3(o2, S2);
            Quat aq = q_mul(q_inv(Qj), Qi);
            qleft33(aq, L); qright33(cdq, Rr); m3_mul(L, Rr, B1);
            const double av[3] = {aq.x, aq.y, aq.z}, bv[3] = {cdq.x, cdq.y, cdq.z};
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) B1[r * 3 + c] += av[r] * (-bv[c]);
            qleft33(q_mul(q_mul(q_inv(Qj), Qi), dq), L);
            m3_mul(L, rec + IMU_DQ_DBG, Tm);
            qleft33(q_mul(q_mul(q_inv(cdq), Qii), Qj), B2);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
                const int ab = a * 3 + b;
                M1[(0 + a) * 30 + 0 + b] = -RiT[ab]; M1[(0 + a) * 30 + 3 + b] = S1[ab]; M1[(3 + a) * 30 + 3 + b] = -B1[ab]; M1[(6 + a) * 30 + 3 + b] = S2[ab];
                M1[(0 + a) * 30 + 6 + b] = -RiT[ab] * dt; M1[(0 + a) * 30 + 9 + b] = -rec[IMU_DP_DBA + ab]; M1[(0 + a) * 30 + 12 + b] = -rec[IMU_DP_DBG + ab];
                M1[(3 + a) * 30 + 12 + b] = -Tm[ab]; M1[(6 + a) * 30 + 6 + b] = -RiT[ab]; M1[(6 + a) * 30 + 9 + b] = -rec[IMU_DV_DBA + ab];
                M1[(6 + a) * 30 + 12 + b] = -rec[IMU_DV_DBG + ab];
                M1[(9 + a) * 30 + 9 + b] = (a == b) ? -1.0 : 0.0; M1[(12 + a) * 30 + 12 + b] = (a == b) ? -1.0 : 0.0;
                M1[(0 + a) * 30 + 15 + b] = RiT[ab]; M1[(3 + a) * 30 + 18 + b] = B2[ab]; M1[(6 + a) * 30 + 21 + b] = RiT[ab];
                M1[(9 + a) * 30 + 24 + b] = (a == b) ? 1.0 : 0.0; M1[(12 + a) * 30 + 27 + b] = (a == b) ? 1.0 : 0.0;
            }
        }
        SYNC();
        const double *S = d.imu_sqrt + f * 225;
        for (int e = t; e < 450; e += MT) { const int a = e / 30, c = e % 30; double s = 0; for (int k = 0; k < 15; k++) s += S[a * 15 + k] * M1[k * 30 + c]; M2s[e] = s; }
        SYNC();
        const int j = i + 1;
        const int si_off = (i == V - 1) ? 6 * V : 6 * V + 9 + 9 * i, sj_off = (j == V - 1) ? 6 * V : 6 * V + 9 + 9 * j;
        for (int e = t; e < 900; e += MT) {
            const int a = e / 30, b = e % 30;
            const int ga = a < 6 ? 6 * i + a : (a < 15 ? si_off + a - 6 : (a < 21 ? 6 * j + a - 15 : sj_off + a - 21));
            const int gb = b < 6 ? 6 * i + b : (b < 15 ? si_off + b - 6 : (b < 21 ? 6 * j + b - 15 : sj_off + b - 21));
            double s = 0; for (int k = 0; k < 15; k++) s += M2s[k * 30 + a] * M2s[k * 30 + b];
            Lam[(size_t)ga * n + gb] += s;
        }
        SYNC();
    }
    // Schur out VB_0..VB_{Vo-2}: Lambda_prior = L_rr - L_rm L_mm^-1 L_rm^T
    for (int e = t; e < mm * mm; e += MT) { const int a = e / mm, b = e % mm; Lmm[e] = Lam[(size_t)(rr + a) * n + rr + b]; }
    SYNC();
    w_inv(Lmm, mm, Linv, Wgj, piv, t);
    for (int e = t; e < mm * rr; e += MT) {
        const int pp = e / rr, b = e % rr; double s = 0;
        for (int q = 0; q < mm; q++) s += Linv[pp * mm + q] * Lam[(size_t)b * n + rr + q];
        T[e] = s;
    }
    SYNC();
    for (int e = t; e < rr * rr; e += MT) {
        const int a = e / rr, b = e % rr; double s = Lam[(size_t)a * n + b];
        for (int q = 0; q < mm; q++) s -= Lam[(size_t)a * n + rr + q] * T[q * rr + b];
        Lp[e] = s;
    }
    for (int e = t; e < rr * rr; e += MT) { Jr[e] = 0.0; Xall[e] = 0.0; }
    SYNC();
    // the recovered factors at the solved estimate (measurements now, information below)
    if (t == 0) {
        for (int i = 0; i < V - 1; i++) {
            isv_relpose_t &f = d.relpose[(size_t)w * (V - 1) + i];
            const double *PSi = pose + 7 * i, *PSj = pose + 7 * (i + 1);
            Quat Qi = q_from_pose(PSi), Qj = q_from_pose(PSj);
            double dd[3] = {PSj[0] - PSi[0], PSj[1] - PSi[1], PSj[2] - PSi[2]}, r6[6];
            q_rot(q_inv(Qi), dd, f.delta_t); q_to_R(q_mul(q_inv(Qi), Qj), f.delta_R);
            f.imu_i = i; f.imu_j = i + 1;
            relpose_jac(f.delta_t, f.delta_R, PSi, PSj, r6, sJ, sJ + 36);
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { Jr[(6 * i + a) * rr + 6 * i + b] += sJ[a * 6 + b]; Jr[(6 * i + a) * rr + 6 * (i + 1) + b] += sJ[36 + a * 6 + b]; }
        }
        isv_se3_prior_t &pp = d.se3[w];
        for (int k = 0; k < 3; k++) pp.t[k] = pose[k];
        q_to_R(q_from_pose(pose), pp.R); pp.index = 0; pp._pad = 0;
        isv_linear9_t &vb = d.lin9[w];
        for (int k = 0; k < 9; k++) vb.VB[k] = sb[9 * (V - 1) + k];
        vb.index = V - 1; vb._pad = 0;
        const int r0 = 6 * (V - 1);
        // SE3PriorFactor::EvaluateOnlyJacobians at its own measurement: residual 0, J = [I 0; 0 Jr^-1(0)] = I; Linear9: I
        for (int a = 0; a < 6; a++) Jr[(r0 + a) * rr + a] += 1.0;
        for (int a = 0; a < 9; a++) Jr[(r0 + 6 + a) * rr + 6 * V + a] += 1.0;
    }
    SYNC();
    // eigen-truncate Lambda_prior at ALPHA
    for (int e = t; e < rr * rr; e += MT) M2[e] = Lp[e];
    SYNC();
    w_jacobi(M2, rr, wv, Vv, tmp, t);
    if (t < rr) keep[t] = wv[t] > d.alpha_cut;
    SYNC();
    {
        double *Sg = Wk, *Xi = Wk + 100;
        int hdim = 0;
        for (int i = 0; i < V + 1; i++) {              // Vo-1 relative poses, the pose prior, the speed/bias prior
            const int rows = i < V ? 6 : 9;
            double *dst = i < V - 1 ? d.relpose[(size_t)w * (V - 1) + i].sqrt_info : (i == V - 1 ? d.se3[w].sqrt_info : d.lin9[w].sqrt_info);
            w_project_cov(Jr + (size_t)hdim * rr, rows, rr, Vv, wv, keep, JU, Sg, t);
            w_inv(Sg, rows, Xi, Wk + 300, piv, t);
            for (int e = t; e < rows * rows; e += MT) Xall[(hdim + e / rows) * rr + hdim + e % rows] = Xi[e];
            SYNC();
            w_chol_upper(Xi, rows, Sg, Wk + 200, t);
            for (int e = t; e < rows * rows; e += MT) dst[e] = Sg[e];
            SYNC();
            hdim += rows;
        }
    }
    // KLD of the recovered factors against the truncated marginal: A = (Jr U)^T X (Jr U) over the kept eigenpairs vs D
    int rank = 0; for (int k = 0; k < rr; k++) rank += keep[k];
    for (int e = t; e < rr * rr; e += MT) {
        const int a = e / rr, k = e % rr; double s = 0;
        for (int c = 0; c < rr; c++) s += Jr[a * rr + c] * Vv[c * rr + k];
        JU[e] = s;
    }
    SYNC();
    w_mm(Xall, JU, XJU, rr, rr, rr, t);
    for (int e = t; e < rr * rr; e += MT) { const int a = e / rr, b = e % rr; double s = 0; for (int k = 0; k < rr; k++) s += JU[k * rr + a] * XJU[k * rr + b]; A[e] = s; }
    SYNC();
    if (t == 0) {
        int ia = 0;
        for (int a = 0; a < rr; a++) if (keep[a]) { int ib = 0; for (int b = 0; b < rr; b++) if (keep[b]) { Ak[ia * rank + ib] = A[a * rr + b]; ib++; } ia++; }
    }
    SYNC();
    const double ldA = w_logdet_spd(Ak, rank, M2, t);
    if (t == 0) {
        double tr = 0, ldinv = 0; int ia = 0;
        for (int a = 0; a < rr; a++) if (keep[a]) { tr += Ak[ia * rank + ia] / wv[a]; ldinv += log(1.0 / wv[a]); ia++; }
        kld_out[w] = 0.5 * (tr - ldA - ldinv - rr);
    }
}
