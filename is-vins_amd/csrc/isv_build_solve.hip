// isv_build_solve.hip -- k_build_solve: normal equations + landmark Schur complement + dense
// Cholesky of the reduced system, one workgroup per window.
//
// Replaces, for the problem problemSolve() builds (reference src/estimator.cpp:1022-1128), what
// Ceres-Solver 2.0.0 (external dependency) does inside DoglegStrategy::ComputeStep for
// linear_solver_type = DENSE_SCHUR: SchurEliminator::Eliminate -> dense Cholesky of the reduced
// camera matrix; Jacobi scaling (trust_region_minimizer.cc) and the dogleg's mu-regularised
// Gauss-Newton retry loop (dogleg_strategy.cc ComputeGaussNewtonStep) are folded in.
//
// Data layout: the reduced matrix T (15N x 15N, symmetric) is kept as the lower block triangle of
// 15x15 blocks (one block row per frame), N(N+1)/2 * 225 doubles -- 118.8 KB for N = 11, resident
// in LDS for the whole build + factorisation; larger windows (reference N = 18, stress N = 20) use
// the same code on an L2-resident global scratch.  Reprojection strips are streamed through LDS in
// chunks of <= 64 factors (whole landmarks).  Accumulation is OWNER-COMPUTES: wavefront `a` owns
// block column `a` (frame a), so every T entry is summed by one lane in landmark order -- bitwise
// reproducible, no atomics.
//
// Scaling algebra: with Jacobi scaling Sc and LM diagonal mu D^2, Ceres solves in scaled space
//   (Sc H Sc + mu D^2) y = Sc g.   Writing c_l = s_l^2 / (s_l^2 E_l + mu D_l^2) for landmark l, the
// Schur complement is  Sc_p [H_pp - sum_l c_l w_l w_l^T] Sc_p + mu D_p^2, so the sweep accumulates
// the UNSCALED bracket T and the pose scales are applied afterwards (they are only known after the
// first sweep: s = 1/(1+sqrt(diag H)) at iteration 0).
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"

#define BS_THREADS 512
#define BS_WAVES 8
#define CH 64                      // factors per staged chunk

DEV int tblk(int I, int J) { return (I * (I + 1) / 2 + J) * 225; }
// element (gi, gj) with gi >= gj of the block-packed lower triangle
DEV int tidx(int gi, int gj) { return tblk(gi / 15, gj / 15) + (gi % 15) * 15 + (gj % 15); }

#ifdef ISV_STAMP
#define STAMP(k) do { if (t == 0) d.dbg[(size_t)w * 64 + (k)] += (double)(wall_clock64() - t_last); if (t == 0) t_last = wall_clock64(); } while (0)
#else
#define STAMP(k) do {} while (0)
#endif
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

// accumulate J^T J (lower triangle) and J^T r of a staged dense Jacobian Jd[dim][ld] whose column c
// maps to global tangent index col[c]; entries are distributed over the block's threads.
DEV void accum_dense(double *T, double *g, double *hdiag, const double *Jd, const double *r, int dim, int ncol, int ld,
                     const int *col, int t) {
    const int npairs = ncol * (ncol + 1) / 2;
    for (int e = t; e < npairs + ncol; e += BS_THREADS) {
        if (e < npairs) {
            int a = 0;
            while ((a + 1) * (a + 2) / 2 <= e) a++;
            const int b = e - a * (a + 1) / 2;                 // a >= b
            double s = 0;
            for (int k = 0; k < dim; k++) s += Jd[k * ld + a] * Jd[k * ld + b];
            int ga = col[a], gb = col[b];
            if (ga < gb) { int tmp = ga; ga = gb; gb = tmp; }
            T[tidx(ga, gb)] += s;
            if (a == b) hdiag[ga] += s;
        } else {
            const int a = e - npairs;
            double s = 0;
            for (int k = 0; k < dim; k++) s += Jd[k * ld + a] * r[k];
            g[col[a]] += s;
        }
    }
}

template <bool LDS_T>
__global__ __launch_bounds__(BS_THREADS) void k_build_solve(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = d.N, n = 15 * N, nblkT = N * (N + 1) / 2 * 225;
    double *T = LDS_T ? lds : d.Tglob + (size_t)w * nblkT;
    double *p = LDS_T ? lds + nblkT : lds;
    double *g = p; p += n;            // unscaled gradient J^T r (pose part)
    double *bs = p; p += n;           // Schur rhs correction  -sum_l c_l w_l g_l
    double *hdiag = p; p += n;        // diag(H_pp) before elimination
    double *sc = p; p += n;           // Jacobi scaling
    double *D = p; p += n;            // dogleg diagonal
    double *y = p; p += n;            // rhs / solution
    double *u = p; p += n;            // unscaled Cauchy direction
    double *red = p; p += BS_THREADS; // reduction scratch
    double *sS = p; p += CH * 28;     // staged strips (CH*28 = 1792 >= 15*31 for an IMU factor)
    double *Wc = p; p += 2 * CH * 6;  // per-observation w vectors of the chunk
    double *cC = p; p += CH;          // c_l per landmark of the chunk
    double *cG = p; p += CH;          // g_l
    double *Linv = p; p += 225;       // inverse of the current diagonal Cholesky block
    int *ip = (int *)p;
    int *colmap = ip; ip += 32;
    int *chunk = ip; ip += 8;         // l_begin, l_end, f_begin, nf
    int *flag = ip; ip += 4;          // cholesky failure

    const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    const int iteration = st.iteration;
    double mu = st.mu;
    int ls_fail = 0, assembled = 0;
    double gmax_l = 0.0;

#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64();
#endif
    for (;;) {
        if (!(mu < 1.0)) { ls_fail = 1; break; }      // while (mu_ < max_mu_) of ComputeGaussNewtonStep
        assembled = 1;
        for (int e = t; e < nblkT; e += BS_THREADS) T[e] = 0.0;
        for (int e = t; e < n; e += BS_THREADS) { g[e] = 0.0; bs[e] = 0.0; hdiag[e] = 0.0; }
        if (t == 0) flag[0] = 0;
        __syncthreads();
        STAMP(0);
        // ---- P1: IMU factors (no loss) and prior factors, one after the other --------------
        for (int i = 0; i < N - 1; i++) {
            const size_t f = (size_t)w * (N - 1) + i;
            if (d.imu_skip[f]) continue;
            const double *s = d.imu_strip + f * ISV_IMU_STRIP;
            // stage as dense [15][31]: 30 Jacobian columns + residual
            for (int e = t; e < 465; e += BS_THREADS) {
                if (e < 15) sS[e * 31 + 30] = s[e];
                else {
                    const int q = e - 15; int row, c;
                    if (q < 90) { row = q / 6; c = q % 6; }
                    else if (q < 225) { row = (q - 90) / 9; c = 6 + (q - 90) % 9; }
                    else if (q < 315) { row = (q - 225) / 6; c = 15 + (q - 225) % 6; }
                    else { row = (q - 315) / 9; c = 21 + (q - 315) % 9; }
                    sS[row * 31 + c] = s[e];
                }
            }
            if (t < 30) colmap[t] = 15 * i + t;
            if (t >= 32 && t < 47) red[t - 32] = s[t - 32];
            __syncthreads();
            accum_dense(T, g, hdiag, sS, red, 15, 30, 31, colmap, t);
            __syncthreads();
        }
        {
            const double *ps = d.prior_strip + (size_t)w * d.prior_strip_sz;
            const int nprior = 2 + (d.Nvo - 1) + d.n_rp[w];
            for (int q = 0; q < nprior; q++) {
                int dim, ncol, off;
                if (q == 0) { dim = 6; ncol = 6; off = PR_SE3; if (t < 6) colmap[t] = t; }
                else if (q == 1) { dim = 9; ncol = 9; off = PR_LIN9; if (t < 9) colmap[t] = 15 * (d.Nvo - 1) + 6 + t; }
                else if (q < 1 + d.Nvo) { const int k = q - 2; dim = 6; ncol = 12; off = PR_REL0 + PR_REL_SZ * k; if (t < 12) colmap[t] = 15 * (k + t / 6) + t % 6; }
                else { const int m = q - 1 - d.Nvo; dim = 2; ncol = 6; off = PR_REL0 + PR_REL_SZ * (d.Nvo - 1) + PR_RP_SZ * m;
                       if (t < 6) colmap[t] = 15 * d.rollpitch[(size_t)w * d.max_rp + m].index + t; }
                // strip: r[dim] then row-major blocks of 6 (or 9) columns; stage dense [dim][ncol]
                const int nblocks = (ncol == 12) ? 2 : 1, bw = ncol / nblocks;
                for (int e = t; e < dim * ncol; e += BS_THREADS) {
                    const int bI = e / (dim * bw), rem = e % (dim * bw), row = rem / bw, c = rem % bw;
                    sS[row * ncol + bI * bw + c] = ps[off + dim + e];
                }
                if (t >= 32 && t < 32 + dim) red[t - 32] = ps[off + t - 32];
                __syncthreads();
                accum_dense(T, g, hdiag, sS, red, dim, ncol, ncol, colmap, t);
                __syncthreads();
            }
        }
        STAMP(1);
        // ---- P2: reprojection factors, chunks of whole landmarks -----------------------------
        int lb = l0;
        while (lb < l1) {
            if (t == 0) {
                int le = lb, nf = 0;
                while (le < l1 && nf + d.lm_k[le] - 1 <= CH) { nf += d.lm_k[le] - 1; le++; }
                chunk[0] = lb; chunk[1] = le; chunk[2] = d.lm_f0[lb]; chunk[3] = nf;
            }
            __syncthreads();
            const int le = chunk[1], fb = chunk[2], nf = chunk[3], nlm = le - lb;
            {   // coalesced stage of nf x 28 doubles
                const double *src = d.strip + (size_t)fb * ISV_PROJ_STRIP;
                for (int e = t; e < nf * ISV_PROJ_STRIP; e += BS_THREADS) sS[e] = src[e];
            }
            __syncthreads();
            STAMP(8);
            // (a) per-landmark scalars + host-frame w, per-factor w
            for (int li = t; li < nlm; li += BS_THREADS) {
                const int l = lb + li, k = d.lm_k[l], fo = d.lm_f0[l] - fb;
                double E = 0, gl = 0, wh[6] = {0, 0, 0, 0, 0, 0};
                for (int m = 0; m < k - 1; m++) {
                    const double *s = sS + (fo + m) * ISV_PROJ_STRIP;
                    const double j0 = s[26], j1 = s[27];
                    E += j0 * j0 + j1 * j1; gl += j0 * s[0] + j1 * s[1];
#pragma unroll
                    for (int c = 0; c < 6; c++) wh[c] += s[2 + c] * j0 + s[8 + c] * j1;
                }
                double sl;
                if (iteration == 0) { sl = 1.0 / (1.0 + sqrt(E)); d.scale_l[l] = sl; }
                else sl = d.scale_l[l];
                const double Es = sl * sl * E;
                const double Dl2 = fmin(fmax(Es, 1e-6), 1e32);
                const double Dl = sqrt(Dl2);
                cC[li] = sl * sl / (Es + mu * Dl2);
                cG[li] = gl;
                d.lmE[l] = E; d.lmG[l] = gl; d.diag_l[l] = Dl; d.grad_l[l] = sl * gl / Dl;
                gmax_l = fmax(gmax_l, fabs(gl));
                double *wo = Wc + (size_t)(fo + li) * 6;
#pragma unroll
                for (int c = 0; c < 6; c++) wo[c] = wh[c];
            }
            for (int ff = t; ff < nf; ff += BS_THREADS) {
                const int l = d.f_rec[fb + ff].lm, li = l - lb, fo = d.lm_f0[l] - fb, m = ff - fo;
                const double *s = sS + ff * ISV_PROJ_STRIP;
                const double j0 = s[26], j1 = s[27];
                double *wo = Wc + (size_t)(fo + li + m + 1) * 6;
#pragma unroll
                for (int c = 0; c < 6; c++) wo[c] = s[14 + c] * j0 + s[20 + c] * j1;
            }
            __syncthreads();
            STAMP(9);
            // (b) owner-computes: wave wv owns block columns a = wv, wv + 8, ...
            for (int a = wv; a < N; a += BS_WAVES) {
                for (int li = 0; li < nlm; li++) {
                    const int l = lb + li, h = d.lm_host[l], k = d.lm_k[l];
                    if (a < h || a >= h + k) continue;
                    const int fo = d.lm_f0[l] - fb, ob = fo + li;
                    const double cl = cC[li];
                    const int nb = h + k - a;
                    const double *wa = Wc + (size_t)(ob + (a - h)) * 6;
                    const double *sa = sS + (fo + (a - h) - 1) * ISV_PROJ_STRIP;     // factor observing frame a (a > h)
                    for (int idx = lane; idx < 36 * nb; idx += 64) {
                        const int bo = idx / 36, r = (idx % 36) / 6, c = idx % 6, b = a + bo;
                        if (bo == 0 && r < c) continue;
                        const double *wb = Wc + (size_t)(ob + (b - h)) * 6;
                        double direct = 0;
                        if (a == h) {
                            if (b == h) {
                                for (int m = 0; m < k - 1; m++) {
                                    const double *s = sS + (fo + m) * ISV_PROJ_STRIP;
                                    direct += s[2 + r] * s[2 + c] + s[8 + r] * s[8 + c];
                                }
                            } else {
                                const double *s = sS + (fo + (b - h) - 1) * ISV_PROJ_STRIP;
                                direct = s[14 + r] * s[2 + c] + s[20 + r] * s[8 + c];      // Jj^T Ji
                            }
                        } else if (b == a) {
                            direct = sa[14 + r] * sa[14 + c] + sa[20 + r] * sa[20 + c];    // Jj^T Jj
                        }
                        T[tblk(b, a) + r * 15 + c] += direct - cl * wb[r] * wa[c];
                        if (bo == 0 && r == c) hdiag[15 * a + r] += direct;
                    }
                    if (lane < 6) {
                        double gs = 0;
                        if (a == h) {
                            for (int m = 0; m < k - 1; m++) {
                                const double *s = sS + (fo + m) * ISV_PROJ_STRIP;
                                gs += s[2 + lane] * s[0] + s[8 + lane] * s[1];
                            }
                        } else gs = sa[14 + lane] * sa[0] + sa[20 + lane] * sa[1];
                        g[15 * a + lane] += gs;
                        bs[15 * a + lane] -= cl * wa[lane] * cG[li];
                    }
                }
            }
            __syncthreads();
            STAMP(10);
            lb = le;
        }
        STAMP(2);
        // ---- P3: scaling, Cauchy data, reduced system -----------------------------------------
        for (int e = t; e < n; e += BS_THREADS) {
            double s;
            if (iteration == 0) { s = 1.0 / (1.0 + sqrt(hdiag[e])); d.scale_p[(size_t)w * n + e] = s; }
            else s = d.scale_p[(size_t)w * n + e];
            const double D2 = fmin(fmax(s * s * hdiag[e], 1e-6), 1e32);
            sc[e] = s; D[e] = sqrt(D2);
            d.diag_p[(size_t)w * n + e] = D[e];
            d.grad_p[(size_t)w * n + e] = s * g[e] / D[e];
            u[e] = s * s * g[e] / D2;
            d.up[(size_t)w * n + e] = u[e];
        }
        __syncthreads();
        // qT = u^T T u  (T still unscaled; symmetric access into the lower block triangle)
        {
            double acc = 0;
            for (int i = t; i < n; i += BS_THREADS) {
                double v = 0;
                for (int j = 0; j < n; j++) v += (j <= i ? T[tidx(i, j)] : T[tidx(j, i)]) * u[j];
                acc += u[i] * v;
            }
            red[t] = acc;
            __syncthreads();
            for (int off = BS_THREADS / 2; off > 0; off >>= 1) { if (t < off) red[t] += red[t + off]; __syncthreads(); }
            if (t == 0) st.qT = red[0];
            __syncthreads();
        }
        STAMP(3);
        // scale in place, add the LM diagonal, form the rhs
        for (int e = t; e < nblkT; e += BS_THREADS) {
            // decode block-packed index
            const int bq = e / 225, rc = e % 225, r = rc / 15, c = rc % 15;
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= bq) I++;
            const int J = bq - I * (I + 1) / 2;
            const int gi = 15 * I + r, gj = 15 * J + c;
            double v = T[e] * sc[gi] * sc[gj];
            if (gi == gj) v += mu * D[gi] * D[gi];
            T[e] = v;
        }
        for (int e = t; e < n; e += BS_THREADS) y[e] = sc[e] * (g[e] + bs[e]);
        __syncthreads();
        STAMP(4);
        // ---- blocked right-looking Cholesky (block = one frame, 15) -----------------------------
        for (int J = 0; J < N; J++) {
            double *Dj = T + tblk(J, J);
            if (wv == 0) {
                // lanes 0..14: row i of the diagonal block in registers; left-looking by columns
                double rowv[15];
                if (lane < 15) {
#pragma unroll
                    for (int k = 0; k < 15; k++) rowv[k] = Dj[lane * 15 + k];
                }
                for (int j = 0; j < 15; j++) {
                    if (lane < 15 && lane >= j) {
                        double s = rowv[j];
                        for (int k = 0; k < j; k++) s -= rowv[k] * Dj[j * 15 + k];   // row j finalised for k < j
                        if (lane == j) {
                            if (!(s > 0.0)) flag[0] = 1;
                            s = sqrt(s);
                            Dj[j * 15 + j] = s;
                        }
                        rowv[j] = s;     // lane j: L_jj ; lanes > j: numerator
                    }
                    WAVE_SYNC();
                    if (lane < 15 && lane > j) {
                        rowv[j] = rowv[j] / Dj[j * 15 + j];
                        Dj[lane * 15 + j] = rowv[j];
                    }
                    WAVE_SYNC();
                }
                // inverse of the lower-triangular block: lane c solves L x = e_c
                if (lane < 15) {
                    double x[15];
#pragma unroll
                    for (int i = 0; i < 15; i++) x[i] = 0.0;
                    for (int i = 0; i < 15; i++) {
                        double s = (i == lane) ? 1.0 : 0.0;
                        for (int k = 0; k < i; k++) s -= Dj[i * 15 + k] * x[k];
                        x[i] = s / Dj[i * 15 + i];
                    }
#pragma unroll
                    for (int i = 0; i < 15; i++) Linv[i * 15 + lane] = x[i];
                }
            }
            __syncthreads();
            if (flag[0]) break;
            // panel: L[I,J] = A[I,J] * Linv^T; one thread per row (reads and rewrites its own row only)
            const int prow = (N - J - 1) * 15;
            for (int rr = t; rr < prow; rr += BS_THREADS) {
                const int I = J + 1 + rr / 15, r = rr % 15;
                double *A = T + tblk(I, J) + r * 15;
                double av[15], ov[15];
#pragma unroll
                for (int k = 0; k < 15; k++) av[k] = A[k];
#pragma unroll
                for (int c = 0; c < 15; c++) {
                    double s2 = 0;
#pragma unroll
                    for (int k = 0; k <= c; k++) s2 += av[k] * Linv[c * 15 + k];
                    ov[c] = s2;
                }
#pragma unroll
                for (int c = 0; c < 15; c++) A[c] = ov[c];
            }
            __syncthreads();
            // trailing update: T[I,K] -= L[I,J] L[K,J]^T for I >= K > J, 3x3 register tiles
            const int m = N - J - 1, nb = m * (m + 1) / 2;
            for (int tile = t; tile < nb * 25; tile += BS_THREADS) {
                const int q = tile / 25, tt = tile % 25, tr = (tt / 5) * 3, tc = (tt % 5) * 3;
                int ii = 0;
                while ((ii + 1) * (ii + 2) / 2 <= q) ii++;
                const int kk = q - ii * (ii + 1) / 2;
                const int I = J + 1 + ii, K = J + 1 + kk;
                const double *LI = T + tblk(I, J) + tr * 15, *LK = T + tblk(K, J) + tc * 15;
                double a00 = 0, a01 = 0, a02 = 0, a10 = 0, a11 = 0, a12 = 0, a20 = 0, a21 = 0, a22 = 0;
#pragma unroll
                for (int k = 0; k < 15; k++) {
                    const double x0 = LI[k], x1 = LI[15 + k], x2 = LI[30 + k];
                    const double z0 = LK[k], z1 = LK[15 + k], z2 = LK[30 + k];
                    a00 += x0 * z0; a01 += x0 * z1; a02 += x0 * z2;
                    a10 += x1 * z0; a11 += x1 * z1; a12 += x1 * z2;
                    a20 += x2 * z0; a21 += x2 * z1; a22 += x2 * z2;
                }
                double *C = T + tblk(I, K) + tr * 15 + tc;
                C[0] -= a00; C[1] -= a01; C[2] -= a02;
                C[15] -= a10; C[16] -= a11; C[17] -= a12;
                C[30] -= a20; C[31] -= a21; C[32] -= a22;
            }
            __syncthreads();
        }
        if (flag[0]) {
            // LINEAR_SOLVER_FAILURE: mu *= 10 and retry while mu < max_mu (dogleg_strategy.cc)
            mu *= 10.0;
            __syncthreads();
            continue;
        }
        STAMP(5);
        // ---- solve L L^T y = rhs, blocked with the diagonal-block inverses recomputed per block ---
        for (int J = 0; J < N; J++) {                 // forward
            const double *Dj = T + tblk(J, J);
            if (wv == 0) {
                if (lane == 0) {
                    for (int i = 0; i < 15; i++) {
                        double s = y[15 * J + i];
                        for (int k = 0; k < i; k++) s -= Dj[i * 15 + k] * y[15 * J + k];
                        y[15 * J + i] = s / Dj[i * 15 + i];
                    }
                }
            }
            __syncthreads();
            for (int rr = t; rr < (N - J - 1) * 15; rr += BS_THREADS) {
                const int I = J + 1 + rr / 15, r = rr % 15;
                const double *A = T + tblk(I, J) + r * 15;
                double s = 0;
#pragma unroll
                for (int k = 0; k < 15; k++) s += A[k] * y[15 * J + k];
                y[15 * I + r] -= s;
            }
            __syncthreads();
        }
        for (int J = N - 1; J >= 0; J--) {            // backward
            const double *Dj = T + tblk(J, J);
            // y_J -= sum_{I>J} L[I,J]^T y_I : 15 outputs, each a sum over (N-J-1)*15 terms
            if (t < 15) {
                double s = 0;
                for (int I = J + 1; I < N; I++) {
                    const double *A = T + tblk(I, J);
                    for (int r = 0; r < 15; r++) s += A[r * 15 + t] * y[15 * I + r];
                }
                red[t] = s;
            }
            __syncthreads();
            if (t == 0) {
                for (int i = 14; i >= 0; i--) {
                    double s = y[15 * J + i] - red[i];
                    for (int k = i + 1; k < 15; k++) s -= Dj[k * 15 + i] * y[15 * J + k];
                    y[15 * J + i] = s / Dj[i * 15 + i];
                }
            }
            __syncthreads();
        }
        STAMP(6);
        break;
    }
    // ---- outputs --------------------------------------------------------------------------------
    if (!ls_fail) {
        for (int e = t; e < n; e += BS_THREADS) {
            d.zp[(size_t)w * n + e] = sc[e] * y[e];          // unscaled Gauss-Newton solution (pose part), sign +
            d.gn_p[(size_t)w * n + e] = -D[e] * y[e];        // gauss_newton_step_ *= -diagonal_
        }
    }
    // gradient_max_norm = |x - Plus(x, -g)|_inf  (trust_region_minimizer.cc)
    {
        double m = gmax_l;
        for (int i = t; i < N; i += BS_THREADS) {
            const double *x = d.pose + ((size_t)w * N + i) * 7;
            double ng[6], xp[7];
            for (int k = 0; k < 6; k++) ng[k] = -g[15 * i + k];
            pose_plus(x, ng, xp);
            for (int k = 0; k < 7; k++) m = fmax(m, fabs(x[k] - xp[k]));
            for (int k = 0; k < 9; k++) m = fmax(m, fabs(g[15 * i + 6 + k]));
        }
        red[t] = m;
        __syncthreads();
        for (int off = BS_THREADS / 2; off > 0; off >>= 1) { if (t < off) red[t] = fmax(red[t], red[t + off]); __syncthreads(); }
    }
    if (t == 0) {
        // (mu already at max_mu on entry: nothing was assembled, x has not moved, the gradient is the previous one)
        if (assembled) st.gmax = red[0];
        st.mu = mu;
        st.ls_fail = ls_fail;
        st.need_linearize = 0;
        st.fresh = 1;
        st.x_cost = d.cost[w];
        if (iteration == 0) {
            st.initial_cost = d.cost[w];
            d.trace_cost[(size_t)w * ISV_MAX_TRACE] = d.cost[w];
            d.trace_radius[(size_t)w * ISV_MAX_TRACE] = st.radius;
        }
        if (st.gmax <= 1e-10) st.termination = ISV_TERM_GRADIENT_TOL;
    }
}
template __global__ void k_build_solve<true>(DevBatch);
template __global__ void k_build_solve<false>(DevBatch);

size_t build_solve_lds_bytes(int N, bool lds_T) {
    const size_t n = 15 * (size_t)N, nblkT = (size_t)N * (N + 1) / 2 * 225;
    size_t dbl = (lds_T ? nblkT : 0) + 7 * n + BS_THREADS + CH * 28 + 2 * CH * 6 + 2 * CH + 225;
    return dbl * sizeof(double) + 64 * sizeof(int);
}
