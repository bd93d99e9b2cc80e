// isv_linearize.hip -- factor linearisation kernels for gfx950 (MI355X).
//
//   k_vector2double    Estimator::vector2double              src/estimator.cpp:474-516
//   k_imu_prep         sqrt_info = LLT(cov^-1).L^T            include/factor/imu_factor.h:44 (once per solve)
//   k_proj_linearize   ProjectionFactor::Evaluate + CauchyLoss corrector
//                                                             src/factor/projection_factor.cpp:24-122
//   k_imu_linearize    IMUFactor::Evaluate                    include/factor/imu_factor.h:23-159
//   k_prior_linearize  SE3Prior/Linear9/RelativePose/RollPitch::Evaluate + corrector
//   k_cost_reduce      0.5 sum rho over the window's residual blocks, fixed order
//
// Design (MI355X): the reprojection kernel is the HBM-bound one (60 B in, 232 B out, ~350 fp64
// FMA per factor = 1.5 flop/B, ridge ~10).  One LANE per factor, one 64-factor tile per
// wavefront; the window's pose blocks are staged once per wave in LDS as rotation matrices; the
// 28-double strip of every factor is transposed through LDS so that the wave stores its
// 64 x 224 B = 14 KiB of strips as 14 fully coalesced 1-KiB (16 B/lane) store instructions.
// IMU / prior factors are matrix shaped (15x15 . 15x30): one wavefront per IMU factor.
#include <hip/hip_runtime.h>
#include "isv_device_types.h"
#include "isv_device_math.h"

// ------------------------------------------------------------------------------------------
__global__ void k_vector2double(DevBatch d) {
    int w = blockIdx.x, i = threadIdx.x;
    if (i < d.N) {
        const double *R = d.Rs + ((size_t)w * d.N + i) * 9;
        Quat q = q_from_R(R);
        double *p = d.pose + ((size_t)w * d.N + i) * 7;
        const double *P = d.Ps + ((size_t)w * d.N + i) * 3;
        p[0] = P[0]; p[1] = P[1]; p[2] = P[2]; p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
        double *s = d.sb + ((size_t)w * d.N + i) * 9;
        for (int k = 0; k < 3; k++) {
            s[k] = d.Vs[((size_t)w * d.N + i) * 3 + k];
            s[3 + k] = d.Bas[((size_t)w * d.N + i) * 3 + k];
            s[6 + k] = d.Bgs[((size_t)w * d.N + i) * 3 + k];
        }
    }
    if (i == 0) {
        Quat q = q_from_R(d.ric + (size_t)w * 9);
        double *e = d.ex + (size_t)w * 7;
        e[0] = d.tic[w * 3]; e[1] = d.tic[w * 3 + 1]; e[2] = d.tic[w * 3 + 2];
        e[3] = q.x; e[4] = q.y; e[5] = q.z; e[6] = q.w;
    }
    // para_Feature = 1 / estimated_depth (getDepthVector, feature_manager.cpp:188-204)
    for (int l = d.lm_off[w] + i; l < d.lm_off[w + 1]; l += blockDim.x) d.lam[l] = 1. / d.depth[l];
}

// ------------------------------------------------------------------------------------------
// sqrt_info of every IMU factor.  The covariance is badly conditioned, so any change of
// operation order moves the result by cond*eps; to stay comparable with the CPU restatement this
// kernel keeps the textbook order (LU with partial pivoting, column-wise solves, Cholesky) and
// forbids FMA contraction.  One wavefront per factor; the work is 15^3 and runs once per solve.
#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void k_imu_prep(DevBatch d, const int32_t *sel) {
    __shared__ double A[225], Inv[225], L[225];
    __shared__ int perm[15];
    int f = sel ? sel[blockIdx.x] : (int)blockIdx.x, t = threadIdx.x;
    if (f < 0) return;                 // (device-resident sequences: only the records of the newest frame are new)
    const double *cov = d.imu_cov + (size_t)f * 225;
    for (int e = t; e < 225; e += 64) A[e] = cov[e];
    if (t < 15) perm[t] = t;
    __syncthreads();
    for (int k = 0; k < 15; k++) {
        if (t == 0) {
            int p = k; double best = fabs(A[k * 15 + k]);
            for (int i = k + 1; i < 15; i++) if (fabs(A[i * 15 + k]) > best) { best = fabs(A[i * 15 + k]); p = i; }
            if (p != k) {
                for (int j = 0; j < 15; j++) { double tmp = A[k * 15 + j]; A[k * 15 + j] = A[p * 15 + j]; A[p * 15 + j] = tmp; }
                int tp = perm[k]; perm[k] = perm[p]; perm[p] = tp;
            }
        }
        __syncthreads();
        if (t > k && t < 15) A[t * 15 + k] /= A[k * 15 + k];
        __syncthreads();
        for (int e = t; e < 225; e += 64) {
            int i = e / 15, j = e % 15;
            if (i > k && j > k) A[e] -= A[i * 15 + k] * A[k * 15 + j];
        }
        __syncthreads();
    }
    if (t < 15) {   // lane c solves column c of the inverse, same order as the restatement
        int c = t; double x[15];
        for (int i = 0; i < 15; i++) x[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < 15; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= A[i * 15 + k] * x[k]; x[i] = s; }
        for (int i = 14; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < 15; k++) s -= A[i * 15 + k] * x[k]; x[i] = s / A[i * 15 + i]; }
        for (int i = 0; i < 15; i++) Inv[i * 15 + c] = x[i];
    }
    __syncthreads();
    for (int e = t; e < 225; e += 64) L[e] = Inv[e];
    __syncthreads();
    for (int j = 0; j < 15; j++) {      // Cholesky, lower, reads the lower triangle (Eigen LLT)
        if (t == 0) {
            double dd = L[j * 15 + j];
            for (int k = 0; k < j; k++) dd -= L[j * 15 + k] * L[j * 15 + k];
            L[j * 15 + j] = sqrt(dd);
        }
        __syncthreads();
        if (t > j && t < 15) {
            double s = L[t * 15 + j];
            for (int k = 0; k < j; k++) s -= L[t * 15 + k] * L[j * 15 + k];
            L[t * 15 + j] = s / L[j * 15 + j];
        }
        __syncthreads();
    }
    double *out = d.imu_sqrt + (size_t)f * 225;
    for (int e = t; e < 225; e += 64) { int i = e / 15, j = e % 15; out[e] = (j >= i) ? L[j * 15 + i] : 0.0; }
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------
#include "isv_proj_factor.h"
#include "isv_prior_factor.h"
#include "isv_imu_factor.h"

#define TILE_LD 15     // LDS row of the half-strip transpose (14 doubles + 1: odd stride, conflict-free)

// LDS per wave: N*12 (R,P per frame) + 12 (ric,tic) [+ 64*TILE_LD doubles in MODE 0]
#include "isv_kernels.h"

// MODE 0: linearise at x (strips + per-factor cost).  MODE 1: cost only at the candidate point
// (cpose/clam), per-factor candidate cost into fcost_out.
template <int MODE>
__global__ __launch_bounds__(256) void k_proj_linearize(DevBatch d, const double *pose_src, const double *lam_src,
                                                         double *fcost_out, int gate) {
    extern __shared__ __align__(16) double lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wv;
    bool live = tile < d.n_tiles;                      // dead waves still reach the barriers
    if (live && gate) {                                // solver schedule: skip windows that do not need this pass
        const SolveState &ss = d.st[d.tile_win[tile]];
        live = ss.termination == ISV_TERM_RUNNING && (gate == 1 ? ss.need_linearize != 0 : ss.step_valid != 0);
    }
    const int N = d.N;
    double *sPose = lds + (size_t)wv * proj_lds_doubles_per_wave(N, MODE);
    double *sEx = sPose + N * 12;
    double *sOut = sEx + 12;
    const int win = live ? d.tile_win[tile] : 0, f0 = live ? d.tile_f0[tile] : 0, n = live ? d.tile_n[tile] : 0;
    // stage this window's pose blocks as rotation matrices (one lane per frame)
    if (!live) {
    } else if (lane < N) {
        const double *p = pose_src + ((size_t)win * N + lane) * 7;
        double R[9]; q_to_R(q_from_pose(p), R);
#pragma unroll
        for (int k = 0; k < 9; k++) sPose[lane * 12 + k] = R[k];
        sPose[lane * 12 + 9] = p[0]; sPose[lane * 12 + 10] = p[1]; sPose[lane * 12 + 11] = p[2];
    } else if (lane == 63) {
        const double *e = d.ex + (size_t)win * 7;
        double R[9]; q_to_R(q_from_pose(e), R);
#pragma unroll
        for (int k = 0; k < 9; k++) sEx[k] = R[k];
        sEx[9] = e[0]; sEx[10] = e[1]; sEx[11] = e[2];
    }
    if (MODE == 1 && live && lane < N) {
        // MODE 1 also needs the poses at x and the tangent step (model cost change, see below)
        const int fr = lane;
        const double *p = d.pose + ((size_t)win * N + fr) * 7;
        double R[9]; q_to_R(q_from_pose(p), R);
        double *o = sOut + fr * 12;
#pragma unroll
        for (int k = 0; k < 9; k++) o[k] = R[k];
        o[9] = p[0]; o[10] = p[1]; o[11] = p[2];
        const double *dp = d.delta_p + (size_t)win * d.np + 15 * fr;
#pragma unroll
        for (int k = 0; k < 6; k++) sOut[N * 12 + fr * 6 + k] = dp[k];
    }
    __syncthreads();

    double r0 = 0, r1 = 0, Ji[12], Jj[12], Jl[2], cost = 0;
    double lm_e = 0, lm_g = 0, lm_wh[6] = {0, 0, 0, 0, 0, 0};     // per-factor pieces of the landmark scalars
    int lm_id = -1;
    const bool active = lane < n;
    if (active) {
        const int f = f0 + lane;
        FactorRec rec = d.f_rec[f];
        const int fi = rec.ij & 255, fj = (rec.ij >> 8) & 255;
        const double lam = lam_src[rec.lm];
        const double *pi3 = d.lm_pts_i + (size_t)rec.lm * 3;
        const double2 pj = *reinterpret_cast<const double2 *>(d.f_pts_j + (size_t)f * 2);
        double ric[9], tic[3], Ri[9], Rj[9], Pi[3], Pj[3];
#pragma unroll
        for (int k = 0; k < 9; k++) { ric[k] = sEx[k]; Ri[k] = sPose[fi * 12 + k]; Rj[k] = sPose[fj * 12 + k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) { tic[k] = sEx[9 + k]; Pi[k] = sPose[fi * 12 + 9 + k]; Pj[k] = sPose[fj * 12 + 9 + k]; }
        proj_factor<MODE == 0>(Ri, Pi, Rj, Pj, ric, tic, d.proj_sqrt_info, lam, pi3[0], pi3[1], pi3[2], pj.x, pj.y,
                               r0, r1, Ji, Jj, Jl);
        // CauchyLoss(1.0): rho = log(1+s), rho' = 1/(1+s), rho'' < 0 -> Corrector scales r and J by sqrt(rho')
        const double s = r0 * r0 + r1 * r1;
        const double sum = 1.0 + s;
        cost = 0.5 * log(sum);
        if (MODE == 0) {
            const double sc = sqrt(fmax(1.0 / sum, 2.2250738585072014e-308));
            r0 *= sc; r1 *= sc;
#pragma unroll
            for (int k = 0; k < 12; k++) { Ji[k] *= sc; Jj[k] *= sc; }
            Jl[0] *= sc; Jl[1] *= sc;
            if (d.est_ex && gate == 0) {               // linearise API with a free extrinsic: J_ex next to the 28-double strip
                double Jex[12];
                proj_jac_ex(Ri, Pi, Rj, Pj, ric, tic, d.proj_sqrt_info, lam, pi3[0], pi3[1], pi3[2], Jex);
#pragma unroll
                for (int k = 0; k < 12; k++) d.strip_ex[(size_t)f * 12 + k] = Jex[k] * sc;
            }
        }
        fcost_out[f] = cost;
        if (MODE == 1) {
            // model cost change piece (J delta)^T (r + J delta / 2) at x: the residual and J delta are
            // re-derived from the poses at x (directional derivative), no strip is read
            const double *sX = sOut, *sD = sOut + N * 12;
            double RiX[9], RjX[9], PiX[3], PjX[3], rx0, rx1, m0, m1;
#pragma unroll
            for (int k = 0; k < 9; k++) { RiX[k] = sX[fi * 12 + k]; RjX[k] = sX[fj * 12 + k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) { PiX[k] = sX[fi * 12 + 9 + k]; PjX[k] = sX[fj * 12 + 9 + k]; }
            proj_residual_dir(RiX, PiX, RjX, PjX, ric, tic, d.proj_sqrt_info, d.lam[rec.lm], pi3[0], pi3[1], pi3[2], pj.x, pj.y,
                              sD + fi * 6, sD + fj * 6, d.delta_l[rec.lm], rx0, rx1, m0, m1);
            const double rp = 1.0 / (1.0 + (rx0 * rx0 + rx1 * rx1));       // CauchyLoss corrector: r, J scaled by sqrt(rho')
            d.fmodel[f] = rp * (m0 * (rx0 + m0 / 2.0) + m1 * (rx1 + m1 / 2.0));
        }
        if (MODE == 0) {
            lm_e = Jl[0] * Jl[0] + Jl[1] * Jl[1]; lm_g = Jl[0] * r0 + Jl[1] * r1;
#pragma unroll
            for (int c2 = 0; c2 < 6; c2++) lm_wh[c2] = Ji[c2] * Jl[0] + Ji[6 + c2] * Jl[1];
            lm_id = rec.lm;
            // w of the observing frame for the Schur sweep: J_pose_j^T J_lambda (6), obs slot f + lm + 1
            // packed per landmark: slot 0 = host frame, slot o = frame host + o (contiguous 6 k doubles); the MFMA
            // downdates expand it to dense panel rows in LDS, the back-substitution reads it as it is
            double *wd = d.W + (size_t)(f + rec.lm + 1) * 6;
#pragma unroll
            for (int c2 = 0; c2 < 6; c2++) wd[c2] = Jj[c2] * Jl[0] + Jj[6 + c2] * Jl[1];
        }
    }
    if (MODE != 0) return;                            // uniform over the block
    if (gate == 1) {
        // Landmark scalars (what SchurEliminator needs per e-block), fused here: a landmark's factors are
        // adjacent lanes of this wavefront (tiles hold whole landmarks), so its first lane sums the pieces
        // of the following lanes in factor order: E = J_l^T J_l, g_l = J_l^T r, host-frame w = sum J_i^T J_l.
        const int prev = __shfl_up(lm_id, 1);
        const bool first = active && (lane == 0 || prev != lm_id);
        const int cnt = first ? d.lm_k[lm_id] - 1 : 0;
        for (int m = 1; m < ISV_MAX_FRAMES; m++) {
            if (!__ballot(first && m < cnt)) break;            // wave-uniform
            const double e_m = __shfl_down(lm_e, m), g_m = __shfl_down(lm_g, m);
            double w_m[6];
#pragma unroll
            for (int c2 = 0; c2 < 6; c2++) w_m[c2] = __shfl_down(lm_wh[c2], m);
            if (first && m < cnt) {
                lm_e += e_m; lm_g += g_m;
#pragma unroll
                for (int c2 = 0; c2 < 6; c2++) lm_wh[c2] += w_m[c2];
            }
        }
        if (first) {
            const SolveState &ss = d.st[win];
            const int l = lm_id;
            double sl;
            if (ss.iteration == 0) { sl = 1.0 / (1.0 + sqrt(lm_e)); d.scale_l[l] = sl; }
            else sl = d.scale_l[l];
            const double Es = sl * sl * lm_e;
            const double Dl2 = fmin(fmax(Es, 1e-6), 1e32);
            const double Dl = sqrt(Dl2);
            d.lm_cg[l] = make_double2(sl * sl / (Es + ss.mu * Dl2), lm_g);
            d.lmE[l] = lm_e; d.lmG[l] = lm_g; d.diag_l[l] = Dl; d.grad_l[l] = sl * lm_g / Dl;
            double *wd = d.W + (size_t)(d.lm_f0[l] + l) * 6;                       // host observation slot
#pragma unroll
            for (int c2 = 0; c2 < 6; c2++) wd[c2] = lm_wh[c2];
        }
    }
    // transpose through this wave's LDS buffer in two halves of 14 columns: lane l owns row l of
    // sOut[64][TILE_LD]; then coalesced 8-B/lane stores of the half rows (the buffer is wave-private,
    // so wave-level ordering is enough)
    double *gout = d.strip + (size_t)f0 * ISV_PROJ_STRIP;
    const int total = n * 14;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        double *row = sOut + lane * TILE_LD;
        if (h == 0) {
            row[0] = r0; row[1] = r1;
#pragma unroll
            for (int k = 0; k < 12; k++) row[2 + k] = Ji[k];
        } else {
#pragma unroll
            for (int k = 0; k < 12; k++) row[k] = Jj[k];
            row[12] = Jl[0]; row[13] = Jl[1];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 14; it++) {
            const int e = it * 64 + lane;
            if (e < total) {
                const int ff = e / 14, c = e - ff * 14;
                gout[ff * ISV_PROJ_STRIP + 14 * h + c] = sOut[ff * TILE_LD + c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
template __global__ void k_proj_linearize<0>(DevBatch, const double *, const double *, double *, int);
template __global__ void k_proj_linearize<1>(DevBatch, const double *, const double *, double *, int);

// ------------------------------------------------------------------------------------------
// FeatureManager::triangulate (src/feature_tracker/feature_manager.cpp:206-258), one lane per landmark that has no
// positive depth yet: the DLT rows of all its views, expressed in the host camera frame, are accumulated as the 4x4
// Gram matrix A^T A (the right singular vectors of A are its eigenvectors); a cyclic Jacobi eigen-solve in
// registers gives the vector of the smallest singular value, depth = v[2] / v[3], clamped to INIT_DEPTH outside
// [0.1, 8] like the reference.
__global__ __launch_bounds__(64) void k_triangulate(DevBatch d) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= d.Ltot) return;
    if (d.depth[l] > 0.0) return;
    int lo = 0, hi = d.B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (d.lm_off[mid] <= l) lo = mid; else hi = mid; }
    const int w = lo, N = d.N, h = d.lm_host[l], k = d.lm_k[l], f0 = d.lm_f0[l];
    const double *Ps = d.Ps + (size_t)w * N * 3, *Rs = d.Rs + (size_t)w * N * 9, *tic = d.tic + (size_t)w * 3, *ric = d.ric + (size_t)w * 9;
    double R0[9], t0[3], tt[3];
    m3_mul(Rs + 9 * h, ric, R0);
    m3v(Rs + 9 * h, tic, tt);
    for (int c = 0; c < 3; c++) t0[c] = Ps[3 * h + c] + tt[c];
    double M[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // upper triangle of A^T A: (0,0) (0,1) (0,2) (0,3) (1,1) (1,2) (1,3) (2,2) (2,3) (3,3)
    for (int o = 0; o < k; o++) {
        const int j = h + o;
        double R1[9], t1[3], dt[3], tr[3], R[9], P[12];
        m3_mul(Rs + 9 * j, ric, R1);
        m3v(Rs + 9 * j, tic, tt);
        for (int c = 0; c < 3; c++) { t1[c] = Ps[3 * j + c] + tt[c]; dt[c] = t1[c] - t0[c]; }
        m3tv(R0, dt, tr);                                // t = R0^T (t1 - t0)
        m3_mul_tn(R0, R1, R);                            // R = R0^T R1
        // P = [R^T | -R^T t]
        double mt[3];
        m3tv(R, tr, mt);
        for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) P[a * 4 + c] = R[c * 3 + a]; P[a * 4 + 3] = -mt[a]; }
        double fx, fy, fz;
        if (o == 0) { fx = d.lm_pts_i[(size_t)l * 3]; fy = d.lm_pts_i[(size_t)l * 3 + 1]; fz = d.lm_pts_i[(size_t)l * 3 + 2]; }
        else { const int f = f0 + o - 1; fx = d.f_pts_j[(size_t)f * 2]; fy = d.f_pts_j[(size_t)f * 2 + 1]; fz = d.f_pts_z[f]; }
        const double nrm = sqrt(fx * fx + fy * fy + fz * fz);
        fx /= nrm; fy /= nrm; fz /= nrm;
        double ra[4], rb[4];
        for (int c = 0; c < 4; c++) { ra[c] = fx * P[8 + c] - fz * P[c]; rb[c] = fy * P[8 + c] - fz * P[4 + c]; }
        int e = 0;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = a; c < 4; c++) M[e++] += ra[a] * ra[c] + rb[a] * rb[c];
    }
    // cyclic Jacobi on the symmetric 4x4 (static indices only: everything stays in registers)
    double A[4][4], V[4][4];
    {
        int e = 0;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = a; c < 4; c++) { A[a][c] = M[e]; A[c][a] = M[e]; e++; }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int c = 0; c < 4; c++) V[a][c] = a == c ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; sweep++) {
        double off = 0, dg = 0;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            dg += A[a][a] * A[a][a];
#pragma unroll
            for (int c = a + 1; c < 4; c++) off += A[a][c] * A[a][c];
        }
        if (off <= 1e-32 * dg || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int q = p + 1; q < 4; q++) {
                const double apq = A[p][q];
                if (apq != 0.0) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                    const double tq = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(tq * tq + 1.0), sn = tq * c;
#pragma unroll
                    for (int r = 0; r < 4; r++) { const double akp = A[r][p], akq = A[r][q]; A[r][p] = c * akp - sn * akq; A[r][q] = sn * akp + c * akq; }
#pragma unroll
                    for (int r = 0; r < 4; r++) { const double apk = A[p][r], aqk = A[q][r]; A[p][r] = c * apk - sn * aqk; A[q][r] = sn * apk + c * aqk; }
#pragma unroll
                    for (int r = 0; r < 4; r++) { const double vkp = V[r][p], vkq = V[r][q]; V[r][p] = c * vkp - sn * vkq; V[r][q] = sn * vkp + c * vkq; }
                }
            }
    }
    double best = A[0][0], v2 = V[2][0], v3 = V[3][0];
#pragma unroll
    for (int c = 1; c < 4; c++) if (A[c][c] < best) { best = A[c][c]; v2 = V[2][c]; v3 = V[3][c]; }
    double dep = v2 / v3;
    if (dep < 0.1 || dep > 8.0) dep = d.init_depth;
    d.depth[l] = dep;
}

// ------------------------------------------------------------------------------------------
// IMU factor: one wavefront per factor.  raw residual (15) and raw Jacobian (15 x 30) are built by
// a few lanes in LDS, then every lane forms rows of sqrt_info * [r | J] (15-term dot products).
// JAC=false: residual only (candidate point), cost into cost_out.
#define IMU_LDS_DOUBLES (225 + 15 * 31 + 16 + 15 * 31 + 2)
template <bool JAC>
DEV void imu_linearize_body(const DevBatch &d, const double *pose_src, const double *sb_src, double *cost_out, int gate,
                            int f, double *lds) {
    double *sS = lds, *sRaw = sS + 226, *sRes = sRaw + 15 * 31 + 1, *sJw = sRes + 16;    // 225 | 465 | 16 | 465
    const int t = threadIdx.x;
    const int N = d.N, w = f / (N - 1), i = f % (N - 1);
    if (gate) {
        const SolveState &ss = d.st[w];
        if (!(ss.termination == ISV_TERM_RUNNING && (gate == 1 ? ss.need_linearize != 0 : ss.step_valid != 0))) return;
    }
    if (d.imu_skip[f]) { if (t == 0) cost_out[f] = 0.0; return; }
    const double *rec = d.imu_in + (size_t)f * ISV_IMU_IN;
    const double *pi = pose_src + ((size_t)w * N + i) * 7, *pj = pi + 7;
    const double *si = sb_src + ((size_t)w * N + i) * 9, *sj = si + 9;
    for (int e = t; e < 225; e += 64) sS[e] = d.imu_sqrt[(size_t)f * 225 + e];
    for (int e = t; e < 15 * 31; e += 64) sRaw[e] = 0.0;
    __syncthreads();
    if (t < (JAC ? 4 : 1)) {
        // lanes 0..3 split the raw factor: 0 = residual (IntegrationBase::evaluate, integration_base.h:160-186),
        // 1 = d/d pose_i, 2 = d/d speedbias_i, 3 = d/d pose_j and speedbias_j (imu_factor.h:66-155).
        Quat Qi = q_from_pose(pi), Qj = q_from_pose(pj);
        Quat Qii = q_inv(Qi);
        const double dt = rec[IMU_DT];
        double dbg[3], tt[3], t2[3];
        for (int k = 0; k < 3; k++) dbg[k] = si[6 + k] - rec[IMU_LBG + k];
        Quat dq = Quat{rec[IMU_DQ + 3], rec[IMU_DQ], rec[IMU_DQ + 1], rec[IMU_DQ + 2]};
        m3v(rec + IMU_DQ_DBG, dbg, tt);
        Quat cdq = q_mul(dq, q_delta(tt));
        if (t == 0) {
            double dba[3], cdv[3], cdp[3], u[3], o1[3], o2[3];
            for (int k = 0; k < 3; k++) dba[k] = si[3 + k] - rec[IMU_LBA + k];
            m3v(rec + IMU_DV_DBA, dba, tt); m3v(rec + IMU_DV_DBG, dbg, t2);
            for (int k = 0; k < 3; k++) cdv[k] = rec[IMU_DV + k] + tt[k] + t2[k];
            m3v(rec + IMU_DP_DBA, dba, tt); m3v(rec + IMU_DP_DBG, dbg, t2);
            for (int k = 0; k < 3; k++) cdp[k] = rec[IMU_DP + k] + tt[k] + t2[k];
            for (int k = 0; k < 3; k++) u[k] = 0.5 * d.G[k] * dt * dt + pj[k] - pi[k] - si[k] * dt;
            q_rot(Qii, u, o1);
            for (int k = 0; k < 3; k++) sRaw[k * 31 + 30] = o1[k] - cdp[k];
            Quat e = q_mul(q_inv(cdq), q_mul(Qii, Qj));
            sRaw[3 * 31 + 30] = 2 * e.x; sRaw[4 * 31 + 30] = 2 * e.y; sRaw[5 * 31 + 30] = 2 * e.z;
            for (int k = 0; k < 3; k++) u[k] = d.G[k] * dt + sj[k] - si[k];
            q_rot(Qii, u, o2);
            for (int k = 0; k < 3; k++) sRaw[(6 + k) * 31 + 30] = o2[k] - cdv[k];
            for (int k = 0; k < 3; k++) { sRaw[(9 + k) * 31 + 30] = sj[3 + k] - si[3 + k]; sRaw[(12 + k) * 31 + 30] = sj[6 + k] - si[6 + k]; }
        } else if (t == 1) {
            // raw Jacobian, tangent columns: pose_i 0..5, sb_i 6..14, pose_j 15..20, sb_j 21..29
            double RiT[9], u[3], o1[3], o2[3], S1[9], S2[9], B1[9], L[9], Rr[9];
            q_to_R(Qii, RiT);
            for (int k = 0; k < 3; k++) u[k] = 0.5 * d.G[k] * dt * dt + pj[k] - pi[k] - si[k] * dt;
            q_rot(Qii, u, o1);
            for (int k = 0; k < 3; k++) u[k] = d.G[k] * dt + sj[k] - si[k];
            q_rot(Qii, u, o2);
            skew3(o1, S1); skew3(o2, S2);
            // -(Qleft(Qj^-1 Qi) Qright(cdq)).bottomRight3x3 : bottom-right of a 4x4 product
            Quat aq = q_mul(q_inv(Qj), Qi);
            qleft33(aq, L); qright33(cdq, Rr); m3_mul(L, Rr, B1);
            const double av[3] = {aq.x, aq.y, aq.z}, bv[3] = {cdq.x, cdq.y, cdq.z};
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) B1[r * 3 + c] += av[r] * (-bv[c]);
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
                const int ab = a * 3 + b;
                sRaw[(0 + a) * 31 + 0 + b] = -RiT[ab];
                sRaw[(0 + a) * 31 + 3 + b] = S1[ab];
                sRaw[(3 + a) * 31 + 3 + b] = -B1[ab];
                sRaw[(6 + a) * 31 + 3 + b] = S2[ab];
            }
        } else if (t == 2) {
            double RiT[9], L[9], T[9];
            q_to_R(Qii, RiT);
            qleft33(q_mul(q_mul(q_inv(Qj), Qi), dq), L);
            m3_mul(L, rec + IMU_DQ_DBG, T);                                 // -> -T at [R, bg]
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
                const int ab = a * 3 + b;
                sRaw[(0 + a) * 31 + 6 + b] = -RiT[ab] * dt;
                sRaw[(0 + a) * 31 + 9 + b] = -rec[IMU_DP_DBA + ab];
                sRaw[(0 + a) * 31 + 12 + b] = -rec[IMU_DP_DBG + ab];
                sRaw[(3 + a) * 31 + 12 + b] = -T[ab];
                sRaw[(6 + a) * 31 + 6 + b] = -RiT[ab];
                sRaw[(6 + a) * 31 + 9 + b] = -rec[IMU_DV_DBA + ab];
                sRaw[(6 + a) * 31 + 12 + b] = -rec[IMU_DV_DBG + ab];
                sRaw[(9 + a) * 31 + 9 + b] = (a == b) ? -1.0 : 0.0;
                sRaw[(12 + a) * 31 + 12 + b] = (a == b) ? -1.0 : 0.0;
            }
        } else {
            double RiT[9], B2[9];
            q_to_R(Qii, RiT);
            qleft33(q_mul(q_mul(q_inv(cdq), Qii), Qj), B2);                 // pose_j [R,R]
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
                const int ab = a * 3 + b;
                sRaw[(0 + a) * 31 + 15 + b] = RiT[ab];
                sRaw[(3 + a) * 31 + 18 + b] = B2[ab];
                sRaw[(6 + a) * 31 + 21 + b] = RiT[ab];
                sRaw[(9 + a) * 31 + 24 + b] = (a == b) ? 1.0 : 0.0;
                sRaw[(12 + a) * 31 + 27 + b] = (a == b) ? 1.0 : 0.0;
            }
        }
    }
    __syncthreads();
    // weighted residual: lane e < 15
    if (t < 15) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += sS[t * 15 + k] * sRaw[k * 31 + 30];
        sRes[t] = s;
        if (JAC) { d.imu_strip[(size_t)f * ISV_IMU_STRIP + t] = s; sJw[t * 31 + 30] = s; }
    }
    if (JAC) {
        // strip layout: [r15 | 15x6 | 15x9 | 15x6 | 15x9] row-major blocks
        double *out = d.imu_strip + (size_t)f * ISV_IMU_STRIP + 15;
        for (int e = t; e < 450; e += 64) {
            int blk, row, col, c;
            if (e < 90) { blk = 0; row = e / 6; col = e % 6; c = col; }
            else if (e < 225) { blk = 1; int q = e - 90; row = q / 9; col = q % 9; c = 6 + col; }
            else if (e < 315) { blk = 2; int q = e - 225; row = q / 6; col = q % 6; c = 15 + col; }
            else { blk = 3; int q = e - 315; row = q / 9; col = q % 9; c = 21 + col; }
            (void)blk;
            double s = 0;
            for (int k = 0; k < 15; k++) s += sS[row * 15 + k] * sRaw[k * 31 + c];
            out[e] = s;
            sJw[row * 31 + c] = s;
        }
    }
    __syncthreads();
    if (JAC) {
        // J^T J (pairs a >= b at a(a+1)/2 + b) and J^T r for k_build_solve: [465 | 30]
        double *H = d.imu_H + (size_t)f * ISV_IMU_H;
        for (int e = t; e < ISV_IMU_H; e += 64) {
            int a, b;
            if (e < 465) { a = 0; while ((a + 1) * (a + 2) / 2 <= e) a++; b = e - a * (a + 1) / 2; }
            else { a = e - 465; b = 30; }
            double s = 0;
            for (int k = 0; k < 15; k++) s += sJw[k * 31 + a] * sJw[k * 31 + b];
            H[e] = s;
        }
    }
    if (t == 0) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += sRes[k] * sRes[k];
        cost_out[f] = 0.5 * s;                          // no loss function on IMU factors (:1050)
    }
}
template <bool JAC>
__global__ __launch_bounds__(64) void k_imu_linearize(DevBatch d, const double *pose_src, const double *sb_src,
                                                       double *cost_out, int gate) {
    __shared__ __align__(16) double lds[IMU_LDS_DOUBLES];
    imu_linearize_body<JAC>(d, pose_src, sb_src, cost_out, gate, blockIdx.x, lds);
}
template __global__ void k_imu_linearize<true>(DevBatch, const double *, const double *, double *, int);
template __global__ void k_imu_linearize<false>(DevBatch, const double *, const double *, double *, int);


// ------------------------------------------------------------------------------------------
// IMU factors at x with Jacobians (solver schedule), two kernels:
//   k_imu_raw     raw residual (15) and raw Jacobian blocks (15 x 30) of 64 factors per workgroup; wavefront p
//                 computes part p (0 residual, 1 d/d pose_i, 2 d/d speedbias_i, 3 d/d pose_j, speedbias_j)
//                 with one FACTOR PER LANE, so no wavefront diverges over the four bodies.  Output (round 4): only the
//                 142 entries that are not structurally 0 / +-1 (the block map of include/factor/imu_factor.h:66-155),
//                 as a COMPACT record of ISV_IMU_RAWC doubles, eight factors interleaved (value j of factor f at
//                 ((f / 8) * ISV_IMU_RAWC + j) * 8 + f % 8): a store instruction of 64 lanes fills 8 whole cache lines.
//                 (The dense [15][32] record it wrote before put every lane's 8 bytes in a cache line of its own, 3840 B
//                 apart: 52 us per launch at N = 11 and 152 us at N = 18, where the 67 MB buffer no longer sat in L2.)
//   k_imu_weight  one workgroup per EIGHT factors: the group's 9 KB of compact values arrive with coalesced loads and stay compact
//                 in LDS; every lane knows which compact value (or 0 / +-1) each of its eight entries of the dense [16][32]
//                 operand [J | r | 0] is; then one wavefront per factor (two factors each),
//                 FP64 MFMA (v_mfma_f64_16x16x4): Jw = sqrt_info * [J | r] (2 tiles x 4 k-steps; sqrt_info goes from global
//                 memory straight into the A operand), then H = Jw^T Jw (3 lower tiles x 4 k-steps) whose row 30 is J^T r --
//                 the A / B operands of these are the Jw accumulator registers themselves (row 4 s + (lane >> 4) of a C tile
//                 is register s of the same lane), no trip through LDS; packed J^T J and cost leave through LDS as coalesced
//                 stores.  Same products and sums as the one-wavefront-per-factor kernel of round 3, bit for bit.
#define ISV_IMU_RAWC 144
// compact record: residual 0..14 | pose_i blocks 16..51 | speedbias_i blocks 52..114 | pose_j / speedbias_j blocks 115..141 (c_imu_block_of below)
// part `part` of factor f (frame i of window w) -> compact value j through store(j, value)
template <class Store>
DEV void imu_raw_part(const DevBatch &d, const double *pose_src, const double *sb_src, int f, int w, int i, int part, Store store) {
    const int N = d.N;
    // (round 4, measured and dropped: the 64 records and states staged through 50 KB of LDS with coalesced loads -- the lane's 25
    //  dependent load -> wait pairs disappear, but the kernel went from 49 to 75 us per launch beside k_lin_gram: its workgroups
    //  then compete with k_lin_gram's for LDS)
    const double *rec = d.imu_in + (size_t)f * ISV_IMU_IN;
    const double *pi = pose_src + ((size_t)w * N + i) * 7, *pj = pi + 7;
    const double *si = sb_src + ((size_t)w * N + i) * 9, *sj = si + 9;
#define RAWC(j, v) store((j), (v))
    Quat Qi = q_from_pose(pi), Qj = q_from_pose(pj);
    Quat Qii = q_inv(Qi);
    const double dt = rec[IMU_DT];
    double dbg[3], tt[3];
    for (int k = 0; k < 3; k++) dbg[k] = si[6 + k] - rec[IMU_LBG + k];
    Quat dq = Quat{rec[IMU_DQ + 3], rec[IMU_DQ], rec[IMU_DQ + 1], rec[IMU_DQ + 2]};
    m3v(rec + IMU_DQ_DBG, dbg, tt);
    Quat cdq = q_mul(dq, q_delta(tt));
    if (part == 0) {
        double r15[15];
        imu_raw_residual(d.G, rec, pi, pj, si, sj, r15);
#pragma unroll
        for (int k = 0; k < 15; k++) RAWC(k, r15[k]);
    } else if (part == 1) {
        // raw Jacobian (imu_factor.h:66-155), tangent columns: pose_i 0..5, sb_i 6..14, pose_j 15..20, sb_j 21..29
        // blocks (row, col): (0,0) (0,3) (3,3) (6,3)
        double RiT[9], u[3], o1[3], o2[3], S1[9], S2[9], B1[9], L[9], Rr[9];
        q_to_R(Qii, RiT);
        for (int k = 0; k < 3; k++) u[k] = 0.5 * d.G[k] * dt * dt + pj[k] - pi[k] - si[k] * dt;
        q_rot(Qii, u, o1);
        for (int k = 0; k < 3; k++) u[k] = d.G[k] * dt + sj[k] - si[k];
        q_rot(Qii, u, o2);
        skew3(o1, S1); skew3(o2, S2);
        Quat aq = q_mul(q_inv(Qj), Qi);
        qleft33(aq, L); qright33(cdq, Rr); m3_mul(L, Rr, B1);
        const double av[3] = {aq.x, aq.y, aq.z}, bv[3] = {cdq.x, cdq.y, cdq.z};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) B1[r * 3 + c] += av[r] * (-bv[c]);
#pragma unroll
        for (int ab = 0; ab < 9; ab++) {
            RAWC(16 + ab, -RiT[ab]);
            RAWC(25 + ab, S1[ab]);
            RAWC(34 + ab, -B1[ab]);
            RAWC(43 + ab, S2[ab]);
        }
    } else if (part == 2) {
        // blocks (0,6) (0,9) (0,12) (3,12) (6,6) (6,9) (6,12); (9,9) and (12,12) are -I (k_imu_weight fills them in)
        double RiT[9], L[9], T[9];
        q_to_R(Qii, RiT);
        qleft33(q_mul(q_mul(q_inv(Qj), Qi), dq), L);
        m3_mul(L, rec + IMU_DQ_DBG, T);
#pragma unroll
        for (int ab = 0; ab < 9; ab++) {
            RAWC(52 + ab, -RiT[ab] * dt);
            RAWC(61 + ab, -rec[IMU_DP_DBA + ab]);
            RAWC(70 + ab, -rec[IMU_DP_DBG + ab]);
            RAWC(79 + ab, -T[ab]);
            RAWC(88 + ab, -RiT[ab]);
            RAWC(97 + ab, -rec[IMU_DV_DBA + ab]);
            RAWC(106 + ab, -rec[IMU_DV_DBG + ab]);
        }
    } else {
        // blocks (0,15) (3,18) (6,21); (9,24) and (12,27) are +I
        double RiT[9], B2[9];
        q_to_R(Qii, RiT);
        qleft33(q_mul(q_mul(q_inv(cdq), Qii), Qj), B2);
#pragma unroll
        for (int ab = 0; ab < 9; ab++) {
            RAWC(115 + ab, RiT[ab]);
            RAWC(124 + ab, B2[ab]);
            RAWC(133 + ab, RiT[ab]);
        }
    }
#undef RAWC
}
__global__ __launch_bounds__(256) void k_imu_raw(DevBatch d, const double *pose_src, const double *sb_src, int gate) {
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int N = d.N, NI = d.B * (N - 1);
    const int f = blockIdx.x * 64 + lane;
    if (f >= NI) return;
    const int w = f / (N - 1), i = f - w * (N - 1);
    if (gate) {
        const SolveState &ss = d.st[w];
        if (!(ss.termination == ISV_TERM_RUNNING && (gate == 1 ? ss.need_linearize != 0 : ss.step_valid != 0))) return;
    }
    if (d.imu_skip[f]) return;
    double *raw = d.imu_raw + (size_t)(f >> 3) * (ISV_IMU_RAWC * 8) + (f & 7);
    imu_raw_part(d, pose_src, sb_src, f, w, i, part, [&](int j, double v) { raw[j * 8] = v; });
}

typedef double double4i __attribute__((ext_vector_type(4)));
DEV double imu_lane_value(double v, int lane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
// where entry (row, col) of the dense [16][32] operand [J | r | 0] comes from: >= 0 compact value j, -1 zero, -2 / -3 the constants -1 / +1
// (3x3 block map of include/factor/imu_factor.h:66-155; block rows: p, q, v, ba, bg; block columns: p_i q_i v_i ba_i bg_i p_j q_j v_j ba_j bg_j)
__constant__ short c_imu_block_of[5][10] = {{16, 25, 52, 61, 70, 115, -1, -1, -1, -1},
                                            {-1, 34, -1, -1, 79, -1, 124, -1, -1, -1},
                                            {-1, 43, 88, 97, 106, -1, -1, 133, -1, -1},
                                            {-1, -1, -1, -2, -1, -1, -1, -1, -3, -1},
                                            {-1, -1, -1, -1, -2, -1, -1, -1, -1, -3}};
DEV int imu_compact_of(int r, int c) {
    if (r >= 15 || c >= 31) return -1;
    if (c == 30) return r;
    const int a = r % 3, cc = c % 3, b0 = c_imu_block_of[r / 3][c / 3];
    if (b0 >= 0) return b0 + a * 3 + cc;
    return (b0 == -1 || a != cc) ? -1 : b0;
}
// one wavefront: IMU factor f from its compact raw values C(j) (j >= 0) and its sqrt_info entries sv[s4] = S[i][4 s4 + kq] (masked to 15 x 15 at use):
// Jw = sqrt_info [J | r], H = Jw^T Jw -> d.imu_H[f], cost_out[f] (and the strips when asked).  src: imu_compact_of of the lane's eight B-operand entries.
template <class CVal>
DEV void imu_weight_factor(const DevBatch &d, int f, const double (&sv)[4], const int (&src)[4][2], CVal C, double *sR, double *cost_out, bool want_strip, int lane) {
    const int i = lane & 15, kq = lane >> 4;
    double *sH = sR;
    double4i a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {                        // Jw = S * raw: output tiles cols 0..15 and 16..31
        const int k = 4 * s4 + kq, j0 = src[s4][0], j1 = src[s4][1];
        const double av = (i < 15 && k < 15) ? sv[s4] : 0.0;
        const double c0 = C(j0 > 0 ? j0 : 0), c1 = C(j1 > 0 ? j1 : 0);
        const double b0 = j0 >= 0 ? c0 : (j0 == -1 ? 0.0 : (j0 == -2 ? -1.0 : 1.0)), b1 = j1 >= 0 ? c1 : (j1 == -1 ? 0.0 : (j1 == -2 ? -1.0 : 1.0));
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b0, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b1, a1, 0, 0, 0);
    }
    double4i h00 = {0, 0, 0, 0}, h10 = {0, 0, 0, 0}, h11 = {0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {                        // H = Jw^T Jw, lower tiles: Jw[4 s4 + kq][i] is register s4 of this lane
        const double x0 = a0[s4], x1 = a1[s4];
        h00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, h00, 0, 0, 0);
        h10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, h10, 0, 0, 0);
        h11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, h11, 0, 0, 0);
    }
    // strip layout: [r15 | 15x6 | 15x9 | 15x6 | 15x9] row-major blocks.  The LDS solver path works from the J^T J
    // blocks alone (k_build_solve_sb, k_dogleg), so the strips are only written for the linearise API (gate 0)
    // and for the generic path: Jw through the wavefront's staging buffer
    if (want_strip) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {                 // C/D layout: col = lane & 15, row = (lane >> 4) + 4 reg
            const int row = kq + 4 * reg;
            sR[row * 32 + i] = a0[reg]; sR[row * 32 + 16 + i] = a1[reg];
        }
        ISV_WSYNC();
        double *out = d.imu_strip + (size_t)f * ISV_IMU_STRIP;
        if (lane < 15) out[lane] = sR[lane * 32 + 30];
        for (int e = lane; e < 450; e += 64) {
            int row, c;
            if (e < 90) { row = e / 6; c = e - 6 * row; }
            else if (e < 225) { const int qq = e - 90; row = qq / 9; c = 6 + (qq - 9 * row); }
            else if (e < 315) { const int qq = e - 225; row = qq / 6; c = 15 + (qq - 6 * row); }
            else { const int qq = e - 315; row = qq / 9; c = 21 + (qq - 9 * row); }
            out[15 + e] = sR[row * 32 + c];
        }
        ISV_WSYNC();
    }
    {
        // 0.5 |Jw[:, 30]|^2, rows in order (no loss function on IMU factors, :1050): row k sits in lane 14 + 16 (k & 3), register k >> 2 of tile 1
        double s = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) { const double v = imu_lane_value(a1[k >> 2], 14 + 16 * (k & 3)); s += v * v; }
        if (lane == 0) cost_out[f] = 0.5 * s;
    }
    // packed J^T J (pairs a >= b at a(a+1)/2 + b, a, b < 30) then J^T r (row 30 of H)
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int row = kq + 4 * reg;
        { const int a = row, b = i; if (b <= a) sH[a * (a + 1) / 2 + b] = h00[reg]; }
        { const int a = 16 + row, b = i; if (a < 30) sH[a * (a + 1) / 2 + b] = h10[reg]; else if (a == 30) sH[465 + b] = h10[reg]; }
        { const int a = 16 + row, b = 16 + i; if (a < 30) { if (b <= a) sH[a * (a + 1) / 2 + b] = h11[reg]; } else if (a == 30 && b < 30) sH[465 + b] = h11[reg]; }
    }
    ISV_WSYNC();
    double *H = d.imu_H + (size_t)f * ISV_IMU_H;
    for (int e = lane; e < ISV_IMU_H; e += 64) H[e] = sH[e];
    ISV_WSYNC();
}
__global__ __launch_bounds__(256) void k_imu_weight(DevBatch d, double *cost_out, int gate) {
    // 25 KB of LDS per eight factors: their compact values (9 KB) and one [16][32] staging buffer per wavefront (Jw for the strips, then the packed J^T J)
    static_assert(ISV_IMU_H <= 16 * 32, "the packed J^T J must fit the staging buffer");
    __shared__ __align__(16) double sC[ISV_IMU_RAWC * 8];
    __shared__ __align__(16) double sA[4 * 512];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int N = d.N, NI = d.B * (N - 1), f0 = blockIdx.x * 8;
    // 0: nothing to do (beyond the batch / gated off), 1: a skipped factor (cost 0), 2: evaluate
    auto state_of = [&](int f) -> int {
        if (f >= NI) return 0;
        if (gate) {
            const SolveState &ss = d.st[f / (N - 1)];
            if (!(ss.termination == ISV_TERM_RUNNING && (gate == 1 ? ss.need_linearize != 0 : ss.step_valid != 0))) return 0;
        }
        return d.imu_skip[f] ? 1 : 2;
    };
    if (!__syncthreads_or(state_of(f0 + (t & 7)) == 2)) {
        if (t < 8 && state_of(f0 + t) == 1) cost_out[f0 + t] = 0.0;
        return;
    }
    const int i = lane & 15, kq = lane >> 4;
    // every global load of the thread in flight together (clamped addresses; the masks are applied at use)
    const double *Rg = d.imu_raw + (size_t)blockIdx.x * (ISV_IMU_RAWC * 8);
    double rc[5], sv[2][4];
#pragma unroll
    for (int k = 0; k < 5; k++) { const int e = t + 256 * k; rc[k] = Rg[e < ISV_IMU_RAWC * 8 ? e : 0]; }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int f = f0 + 2 * wv + q;
        const double *Sg = d.imu_sqrt + (size_t)(f < NI ? f : NI - 1) * 225;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) { const int k = 4 * s4 + kq; sv[q][s4] = Sg[(i < 15 && k < 15) ? i * 15 + k : 0]; }
    }
    // the lane's eight B-operand entries (rows 4 s + kq, columns i and 16 + i of [J | r | 0]) as compact indices
    int src[4][2];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) { src[s4][0] = imu_compact_of(4 * s4 + kq, i); src[s4][1] = imu_compact_of(4 * s4 + kq, 16 + i); }
#pragma unroll
    for (int k = 0; k < 5; k++) { const int e = t + 256 * k; if (e < ISV_IMU_RAWC * 8) sC[e] = rc[k]; }
    __syncthreads();
    const bool want_strip = gate == 0 || !d.lds_T;
    double *sR = sA + wv * 512;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int fl = 2 * wv + q, f = f0 + fl, stf = state_of(f);          // (wave-uniform)
        if (stf == 0) continue;
        if (stf == 1) { if (lane == 0) cost_out[f] = 0.0; continue; }
        imu_weight_factor(d, f, sv[q], src, [&](int j) { return sC[j * 8 + fl]; }, sR, cost_out, want_strip, lane);
    }
}

// ------------------------------------------------------------------------------------------
// k_front (round 5): the IMU and prior factors of the windows of a SMALL batch (one that leaves CUs idle) in ONE launch on the
// solve stream -- blockIdx.y = 0: the window's IMU factors (the raw parts of k_imu_raw, then k_imu_weight's products, the compact
// values staying in LDS); blockIdx.y = 1: its prior factors, eight wavefronts striding the slots.  The same routines and operation
// order as the three kernels it replaces (imu_H, imu_cost, prior_H, prior_strip, prior_cost bit for bit); what it saves is
// their launches, the fork / join events of the side stream (7 + 5..13 us per iteration on the critical stream, measured) and the
// memory round trip of the compact records.
#define ISV_FRONT_LDW 145          // compact values of one factor in LDS (odd stride)
size_t front_lds_bytes(int N, int slots) {
    const size_t imu = ((size_t)(N - 1) * ISV_FRONT_LDW + 8 * 512) * sizeof(double), pr = prior_lds_bytes(slots);
    return imu > pr ? imu : pr;
}
__global__ __launch_bounds__(512) void k_front(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const SolveState &ss = d.st[w];
    if (!(ss.termination == ISV_TERM_RUNNING && ss.need_linearize != 0)) return;
    const int N = d.N;
    if (blockIdx.y == 1) {
        prior_linearize_body<true, true>(d, d.pose, d.sb, d.prior_cost, 0, w, lds, wv, 8);      // eight wavefronts stride the slots
        return;
    }
    double *sCw = lds, *sA = lds + (size_t)(N - 1) * ISV_FRONT_LDW;
    const int f0 = w * (N - 1);
    if (wv < 4 && lane < N - 1 && !d.imu_skip[f0 + lane]) {
        double *row = sCw + lane * ISV_FRONT_LDW;
        imu_raw_part(d, d.pose, d.sb, f0 + lane, w, lane, wv, [&](int j, double v) { row[j] = v; });
    }
    const int i = lane & 15, kq = lane >> 4;
    int src[4][2];
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) { src[s4][0] = imu_compact_of(4 * s4 + kq, i); src[s4][1] = imu_compact_of(4 * s4 + kq, 16 + i); }
    // sqrt_info of this wavefront's factors (wv, wv + 8, ...; N <= 32): in flight across the barrier
    double sv[4][4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int fw = wv + 8 * q, f = f0 + (fw < N - 1 ? fw : N - 2);
        const double *Sg = d.imu_sqrt + (size_t)f * 225;
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) { const int k = 4 * s4 + kq; sv[q][s4] = Sg[(i < 15 && k < 15) ? i * 15 + k : 0]; }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int fw = wv + 8 * q, f = f0 + fw;
        if (fw >= N - 1) break;
        if (d.imu_skip[f]) { if (lane == 0) d.imu_cost[f] = 0.0; continue; }
        const double *row = sCw + fw * ISV_FRONT_LDW;
        imu_weight_factor(d, f, sv[q], src, [&](int j) { return row[j]; }, sA + wv * 512, d.imu_cost, false, lane);
    }
}

template <bool JAC>
__global__ __launch_bounds__(64) void k_prior_linearize(DevBatch d, const double *pose_src, const double *sb_src, double *cost_out, int gate) {
    extern __shared__ __align__(16) double lds[];
    prior_linearize_body<JAC, false>(d, pose_src, sb_src, cost_out, gate, blockIdx.x, lds);
}
template __global__ void k_prior_linearize<true>(DevBatch, const double *, const double *, double *, int);
template __global__ void k_prior_linearize<false>(DevBatch, const double *, const double *, double *, int);

// ------------------------------------------------------------------------------------------
// model cost change pieces: (J delta)^T (r + J delta / 2) per residual block, from the strips at x
// (the reprojection factors' part is fused into k_proj_linearize<1>)
DEV void model_imu_prior_body(const DevBatch &d, int bid, double *sm) {
    const int w = bid / (d.N - 1 + 1), q = bid % (d.N - 1 + 1), t = threadIdx.x;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.step_valid) return;
    const double *dp = d.delta_p + (size_t)w * d.np;
    if (q < d.N - 1) {
        const size_t f = (size_t)w * (d.N - 1) + q;
        if (d.imu_skip[f]) { if (t == 0) d.imu_model[f] = 0.0; return; }
        const double *s = d.imu_strip + f * ISV_IMU_STRIP;
        if (t < 15) {
            double m = 0;
            const double *dd = dp + 15 * q;            // 30 contiguous tangent entries
            for (int c = 0; c < 6; c++) m += s[15 + t * 6 + c] * dd[c];
            for (int c = 0; c < 9; c++) m += s[105 + t * 9 + c] * dd[6 + c];
            for (int c = 0; c < 6; c++) m += s[240 + t * 6 + c] * dd[15 + c];
            for (int c = 0; c < 9; c++) m += s[330 + t * 9 + c] * dd[21 + c];
            sm[t] = m * (s[t] + m / 2.0);
        }
        __syncthreads();
        if (t == 0) { double a = 0; for (int k = 0; k < 15; k++) a += sm[k]; d.imu_model[f] = a; }
    } else {
        // priors of this window: lane per slot
        const int slots = d.n_prior_slots;
        if (t < slots) {
            const double *ps = d.prior_strip + (size_t)w * d.prior_strip_sz;
            double acc = 0;
            if (t == 0) {
                for (int r = 0; r < 6; r++) { double m = 0; for (int c = 0; c < 6; c++) m += ps[PR_SE3 + 6 + r * 6 + c] * dp[c]; acc += m * (ps[PR_SE3 + r] + m / 2.0); }
            } else if (t == 1) {
                const double *dd = dp + 15 * (d.Nvo - 1) + 6;
                for (int r = 0; r < 9; r++) { double m = 0; for (int c = 0; c < 9; c++) m += ps[PR_LIN9 + 9 + r * 9 + c] * dd[c]; acc += m * (ps[PR_LIN9 + r] + m / 2.0); }
            } else if (t < 1 + d.Nvo) {
                const int k = t - 2; const double *o = ps + PR_REL0 + PR_REL_SZ * k;
                for (int r = 0; r < 6; r++) {
                    double m = 0;
                    for (int c = 0; c < 6; c++) m += o[6 + r * 6 + c] * dp[15 * k + c] + o[42 + r * 6 + c] * dp[15 * (k + 1) + c];
                    acc += m * (o[r] + m / 2.0);
                }
            } else {
                const int mm = t - 1 - d.Nvo;
                if (mm < d.n_rp[w]) {
                    const double *o = ps + PR_REL0 + PR_REL_SZ * (d.Nvo - 1) + PR_RP_SZ * mm;
                    const int idx = d.rollpitch[(size_t)w * d.max_rp + mm].index;
                    for (int r = 0; r < 2; r++) { double m = 0; for (int c = 0; c < 6; c++) m += o[2 + r * 6 + c] * dp[15 * idx + c]; acc += m * (o[r] + m / 2.0); }
                }
            }
            d.prior_model[(size_t)w * slots + t] = acc;
        }
    }
}

__global__ __launch_bounds__(64) void k_model_imu_prior(DevBatch d) {
    __shared__ double sm[16];
    model_imu_prior_body(d, blockIdx.x, sm);
}

// ------------------------------------------------------------------------------------------
// cost of one window = sum over its residual blocks, fixed-shape tree (bitwise reproducible):
// per-thread strided partial sums, then a 256-wide LDS tree.
__global__ __launch_bounds__(256) void k_cost_reduce(DevBatch d, const double *fcost, const double *imu_cost,
                                                     const double *prior_cost, double *out, int gate) {
    __shared__ double red[256];
    const int w = blockIdx.x, t = threadIdx.x;
    if (gate) {
        const SolveState &ss = d.st[w];
        if (!(ss.termination == ISV_TERM_RUNNING && ss.need_linearize != 0)) return;
    }
    double s = 0;
    for (int f = d.f_off[w] + t; f < d.f_off[w + 1]; f += 256) s += fcost[f];
    for (int i = t; i < d.N - 1; i += 256) s += imu_cost[(size_t)w * (d.N - 1) + i];
    for (int i = t; i < d.n_prior_slots; i += 256) s += prior_cost[(size_t)w * d.n_prior_slots + i];
    red[t] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (t < off) red[t] += red[t + off];
        __syncthreads();
    }
    if (t == 0) out[w] = red[0];
}
