// isv_build_solve_st.hip -- k_build_solve_st: the DENSE_SCHUR linear solve of one trust-region iteration (reference
// src/estimator.cpp:1119-1128; same mathematics, inputs and outputs as k_build_solve_sb, isv_build_solve_sb.hip), re-phased
// so that a window needs <= 40 KB of LDS at N = 11 (80 KB at N = 18) and 256 threads: FOUR windows share a compute unit
// (two at N = 18) instead of two (one).  The solve is a chain of dependent 9x9 / 6x6 pivots -- latency bound, VALUBusy 26 % --
// so what a CU gains is the overlap of more windows' serial sections: a 1024-window launch is ONE round of four co-resident
// workgroups instead of two rounds of two.
//
// What changed against k_build_solve_sb (VERDICT r3 task 2 / DESIGN r3 10.1):
//   * the speed/bias blocks are never all in LDS.  A chain node's D_i (9x9), C_i (coupling to its parent) and the three pose
//     x speed/bias blocks the IMU factors give it are formed ON THE WAY IN from the packed J^T J records (imu_H, the
//     Linear9 prior), already Jacobi-scaled, one node ahead of the factorisation, into small rings (4 D slots, 3 C slots per
//     chain); the fill Y_i lives in ONE buffer per chain that is transformed in place node after node
//     (Y_parent = init - Y_i' C_i'^T row by row);
//   * the pose-system downdate Spp -= Y_i' Y_i'^T is applied as node i LEAVES the pipeline (it was a phase of its own after
//     both chains, which is why every Y_i had to stay resident);
//   * L_i^-1 and C_i' go to a global scratch (d.st_ws) as they become final and come back for the back-substitution; Y_i' does
//     not (round 5, ST_NO_YSPILL): Y_i'^T x_pose is rebuilt by a recurrence along the chains from the nodes' own blocks;
//   * every global load of the assembly is unconditional (clamped address, multiplied by a 0/1 mask): written as
//     `if (has) v += H[i]` the compiler puts each load in a branch of its own behind an s_waitcnt vmcnt(0).
// Summation orders differ from k_build_solve_sb in the last bits (Spp is downdated node by node; u^T T u is summed per
// source record), so a handle uses ONE of the two kernels for all its launches (isv_solver.hip), never a mix.
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"

// threads per window: 256 (4 wavefronts; four workgroups per CU) for N <= 11, 512 (8 wavefronts; two per CU) for long windows
#define ST_THREADS(BIG) ((BIG) ? 512 : 256)
#ifdef ISV_STAMP
#define STSTAMP(k) do { if (t_outer == 0) { unsigned long long now_ = wall_clock64(); st_acc[k] += now_ - t_last; t_last = now_; } } while (0)
#else
#define STSTAMP(k) do {} while (0)
#endif
#define RCH_ST 16                  // landmarks per staged chunk of the retry correction
// ST_NO_YSPILL (round 5): the back-substitution does not read the nodes' Y_i' back from the global scratch: v_i = Y_i'^T x_pose
// follows from the chain recurrence v_p = L_p^-1 (E_p^T x_pose - C_j' v_j) (E_p: the node's own three blocks, j: its child), so a
// short window no longer spills Y_i' at all (22 KB written + 22 KB read per window at N = 11; long windows keep the spill for
// their tile wavefronts).  0: the round-4 form (Y_i' column-major in the scratch, one sweep Y_i'^T x_pose), kept for A/B builds.
#ifndef ST_NO_YSPILL
#define ST_NO_YSPILL 1
#endif

DEV int st_sblk(int I, int J, int N) { return (J * N - J * (J - 1) / 2 + (I - J)) * 36; }   // I >= J
DEV int st_pair(int a, int b) { return a * (a + 1) / 2 + b; }      // a >= b
DEV int st_nlo(int i, int M) { return i > M ? i - 1 : 0; }
DEV int st_nhi(int i, int M, int N) { return i < M ? i + 1 : N - 1; }

DEV double st_readlane(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
DEV double st_rsqrt(double x) {    // 1/sqrt(x) to ~1 ulp: hardware estimate + two Newton steps
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
#define ST_WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
// a block barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL store of the wavefront (its
// release fence covers all address spaces), i.e. for the acknowledgement of the chain pipeline's spills -- a memory round trip per
// chain node on the critical path.  Used where no wavefront reads global data another one wrote since the last full barrier.
#define ST_LDS_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)
#define ST_IDS() int t = t_outer; asm volatile("" : "+v"(t)); const int lane = t & 63; (void)lane

// Factor the BS x BS SPD block at A (row-major, lower part valid) in registers and overwrite it with the INVERSE of its
// Cholesky factor (lower, upper part zeroed).  One wavefront.  (k_build_solve_sb's routine.)
template <int BS>
DEV bool st_chol_inv(double *A, int lane) {
    double row[BS], dinv[BS], x[BS];
#pragma unroll
    for (int k = 0; k < BS; k++) row[k] = (lane < BS) ? A[lane * BS + k] : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < BS; j++) {
        double s = row[j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= row[k] * st_readlane(row[k], j);
        const double sj = st_readlane(s, j);               // pivot
        if (!(sj > 0.0)) bad = true;
        dinv[j] = st_rsqrt(sj);                            // 1 / L_jj (wave-uniform)
        row[j] = (lane == j) ? sj * dinv[j] : s * dinv[j];
    }
#pragma unroll
    for (int i = 0; i < BS; i++) {                          // lane c solves L x = e_c
        double s = (lane == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s -= st_readlane(row[k], i) * x[k];
        x[i] = s * dinv[i];
    }
    ST_WSYNC();
    if (lane < BS) {
#pragma unroll
        for (int k = 0; k < BS; k++) A[k * BS + lane] = x[k];   // x[k] = Linv[k][lane], zero for k < lane
    }
    ST_WSYNC();
    return bad;
}

// ---- sizes ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline int st_ytot(int N) {
    const int M = N / 2;
    int o = 0;
    for (int i = 0; i < N; i++) o += ((i < M ? i + 1 : N - 1) - (i > M ? i - 1 : 0) + 1) * 54;
    return o;
}
// doubles of the chain work area: Y buffers (N + 1 block slots), per chain 4 D slots + 3 C slots; never smaller than what
// the early phases stage there (prior blocks + u; the retry correction's dense w rows)
__host__ __device__ inline int st_work_doubles(int N, int prior_H_sz) {
    int wk = (N + 1) * 54 + 2 * 7 * 81;
    const int early = ((prior_H_sz + 1) & ~1) + 15 * N;
    const int retry = RCH_ST * (6 * N + 1) + 2 * RCH_ST;
    if (N <= 11 && 162 * N > wk) wk = 162 * N;           // short windows: the back-substitution stages every L_i^-1 and C_i' there
    if (early > wk) wk = early;
    if (retry > wk) wk = retry;
    return (wk + 1) & ~1;
}
// the global scratch of a window: L_i^-1 | C_i' | Y_i' (long windows, whose tile wavefronts read it; every window in the
// ST_NO_YSPILL = 0 form) | the scaled init blocks of every node
__host__ __device__ inline int st_ws_y(int N) { return (!ST_NO_YSPILL || N > 11) ? st_ytot(N) : 0; }
__host__ __device__ inline size_t st_ws_doubles(int N) { return (size_t)324 * N + (size_t)st_ws_y(N); }
size_t build_solve_st_ws_doubles(int N) { return st_ws_doubles(N); }
size_t build_solve_st_bytes(int N, int prior_H_sz) {
    const size_t n = 15 * (size_t)N, nS = (size_t)N * (N + 1) / 2 * 36;
    //      g, y, D, sc   red   yo + skip (ints)   flag   blkIJ (ushort)              triAB (bytes)   Spp   work
    return (4 * n + 32 + 32 + 2 + ((size_t)N * (N + 1) / 2 + 3) / 4 + 1 + 20 + nS + (size_t)st_work_doubles(N, prior_H_sz) + 2) * sizeof(double);
}

// BIG = false: 256 threads, four workgroups per CU; BIG = true: 512 threads, two per CU (long windows, N <= 18): 16 wavefronts per CU
// and <= 128 VGPRs either way.
// NC: compile-time window length (0 = d.N).
template <bool BIG, int NC>
__global__ __launch_bounds__(ST_THREADS(BIG), 4) void k_build_solve_st(DevBatch d) {
    constexpr int LT = ST_THREADS(BIG), NW = LT / 64;
    extern __shared__ __align__(16) double lds[];
    const int w = blockIdx.x, t_outer = threadIdx.x;
    SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = NC ? NC : d.N, n = 15 * N, M = N / 2, n6 = 6 * N, nS = N * (N + 1) / 2 * 36;
    double *p = lds;
    double *g = p; p += n;
    double *y = p; p += n;                    // bs until y = s (g + bs) is formed
    double *D = p; p += n;                    // hdiag until the LM diagonal is formed
    double *sc = p; p += n;                   // Jacobi scales
    double *red = p; p += 32;
    int *yo = (int *)p;                       // yo[0..N]: offsets of the nodes' Y' in the global scratch (N <= 31)
    int *skipL = yo + 32; p += 32;            // imu_skip flags of this window (N - 1 <= 31)
    int *flag = (int *)p; p += 2;
    unsigned short *blkIJ = (unsigned short *)p; p += (N * (N + 1) / 2 + 3) / 4 + 1;   // (I | J << 8) of packed pose block q
    unsigned char *triAB = (unsigned char *)p; p += 20;   // (row, col) of triangular pair index e < 78
    double *Spp = p; p += nS;                 // pose-pose, packed lower block triangle of 6x6 blocks (Tvis layout)
    double *work = p;                         // chain buffers; early phases: prior blocks + u, retry staging
    double *Ybuf = work;                      // [(N + 1) slots][6][9]: forward chain slot = pose, backward chain slot = pose + 1
    double *Dring = work + (N + 1) * 54;      // [2 chains][4][81]
    double *Cring = Dring + 2 * 4 * 81;       // [2 chains][3][81]
    double *stageP = work;                    // early: the window's prior J^T J record
    double *u = work + ((d.prior_H_sz + 1) & ~1);      // early: Cauchy direction
    const int wv = t_outer >> 6;
    {
        const int t = t_outer;
        if (t == 0) {
            int o = 0;
            for (int i = 0; i < N; i++) { yo[i] = o; o += (st_nhi(i, M, N) - st_nlo(i, M) + 1) * 54; }
            yo[N] = o; flag[0] = 0;
        }
        if (t < N - 1) skipL[t] = d.imu_skip[(size_t)w * (N - 1) + t];
        for (int q = t; q < N * (N + 1) / 2; q += LT) {
            int ca = 0;
            while (ca + 1 < N && (ca + 1) * N - (ca + 1) * ca / 2 <= q) ca++;
            blkIJ[q] = (unsigned short)((ca + (q - (ca * N - ca * (ca - 1) / 2))) | (ca << 8));
        }
        if (t < 78) { int a = 0; while ((a + 1) * (a + 2) / 2 <= t) a++; triAB[2 * t] = (unsigned char)a; triAB[2 * t + 1] = (unsigned char)(t - a * (a + 1) / 2); }
    }
    __syncthreads();
    // cost of the window at x = sum over its residual blocks (fixed-shape strided partials + tree at the end)
    double cpart = 0;
    {
        const int t = t_outer;
        const int f0w = d.f_off[w], f1w = d.f_off[w + 1];
        for (int f0 = f0w + t; f0 < f1w; f0 += 4 * LT) {
            double c4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int f = f0 + LT * q < f1w ? f0 + LT * q : f1w - 1; c4[q] = d.fcost[f]; }
#pragma unroll
            for (int q = 0; q < 4; q++) if (f0 + LT * q < f1w) cpart += c4[q];
        }
        for (int i = t; i < N - 1; i += LT) cpart += d.imu_cost[(size_t)w * (N - 1) + i];
        for (int i = t; i < d.n_prior_slots; i += LT) cpart += d.prior_cost[(size_t)w * d.n_prior_slots + i];
    }
    const int iteration = st.iteration;
    double mu = st.mu;
    int ls_fail = 0, attempt = 0, assembled = 0;
    double gmax_l = 0.0;
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64(), st_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    const double *V = d.Tvis + (size_t)w * d.tvis_sz;
    const double *H = d.imu_H + (size_t)w * (N - 1) * ISV_IMU_H;
    const double *PH = d.prior_H + (size_t)w * d.prior_H_sz;
    double *ws = d.st_ws + (size_t)w * st_ws_doubles(N);
    double *gDinv = ws, *gC = ws + 81 * N, *gY = ws + 162 * N, *gYi = gY + st_ws_y(N);

    // ---- the raw (unscaled) blocks of chain node i from the packed records; every load unconditional, masked ---------
    // IMU factor f = frames (f, f + 1); local tangent order pose_f (0..5) sb_f (6..14) pose_f+1 (15..20) sb_f+1 (21..29)
    auto hasA = [&](int i) { return (i >= 1 && !skipL[i - 1]) ? 1.0 : 0.0; };          // factor i - 1 exists
    auto hasB = [&](int i) { return (i <= N - 2 && !skipL[i]) ? 1.0 : 0.0; };          // factor i exists
    auto HAp = [&](int i) { return H + (size_t)(i >= 1 ? i - 1 : 0) * ISV_IMU_H; };
    auto HBp = [&](int i) { return H + (size_t)(i <= N - 2 ? i : N - 2) * ISV_IMU_H; };
    auto rawD = [&](int i, int r, int c) -> double {       // sb_i x sb_i, r >= c
        double v = HAp(i)[st_pair(21 + r, 21 + c)] * hasA(i);
        v += HBp(i)[st_pair(6 + r, 6 + c)] * hasB(i);
        const double pl = PH[PH_LIN9 + st_pair(r, c)];
        if (i == d.Nvo - 1) v += pl;                       // Linear9Factor on the newest visual-odometry frame's speed/bias block
        if (d.est_ex && i == N - 1) v = r == c ? 1.0 : 0.0;   // the extrinsic's pseudo-frame: dummy unit block
        return v;
    };
    auto rawC = [&](int i, int a, int b) -> double {       // coupling of node i (column b) to its parent (row a)
        if (i < M) return H[(size_t)i * ISV_IMU_H + st_pair(21 + a, 6 + b)] * (skipL[i] ? 0.0 : 1.0);               // parent i + 1: factor i
        return H[(size_t)(i - 1) * ISV_IMU_H + st_pair(21 + b, 6 + a)] * (skipL[i - 1] ? 0.0 : 1.0);              // parent i - 1: factor i - 1
    };
    auto rawY = [&](int i, int pz, int r, int c) -> double {      // pose_pz (row r) x sb_i (column c), pz in {i - 1, i, i + 1}
        if (pz == i) {
            double v = HAp(i)[st_pair(21 + c, 15 + r)] * hasA(i);
            v += HBp(i)[st_pair(6 + c, r)] * hasB(i);
            return v;
        }
        if (pz == i + 1) return H[(size_t)(i <= N - 2 ? i : N - 2) * ISV_IMU_H + st_pair(15 + r, 6 + c)] * hasB(i);
        return H[(size_t)(i >= 1 ? i - 1 : 0) * ISV_IMU_H + st_pair(21 + c, r)] * hasA(i);
    };

    for (;;) {
        if (!(mu < 1.0)) { ls_fail = 1; break; }
        assembled = 1;
        int t = t_outer;
        asm volatile("" : "+v"(t));
        // ---- pose blocks, gradient, diagonal: reprojection part (k_lin_gram / k_rank1_mfma) + IMU + priors --------------
        for (int e0 = t; e0 < nS; e0 += 4 * LT) {
            double v4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const int e = e0 + k * LT; v4[k] = V[e < nS ? e : nS - 1]; }
#pragma unroll
            for (int k = 0; k < 4; k++) { const int e = e0 + k * LT; if (e < nS) Spp[e] = v4[k]; }
        }
        for (int e = t; e < n; e += LT) {
            const int I = e / 15, r = e - 15 * I;
            // gradient of the IMU factors; pose part of the reprojection factors' gradient, diagonal and reduced rhs
            double vg = HAp(I)[465 + 15 + r] * hasA(I);
            vg += HBp(I)[465 + r] * hasB(I);
            const int ev = r < 6 ? 6 * I + r : 0;
            const double v0 = V[nS + ev], v1 = V[nS + n6 + ev], v2 = V[nS + 2 * n6 + ev];
            // diagonal of the IMU factors' J^T J
            double vd = HAp(I)[st_pair(15 + r, 15 + r)] * hasA(I);
            vd += HBp(I)[st_pair(r, r)] * hasB(I);
            if (r < 6) { D[e] = v0 + vd; g[e] = v1 + vg; y[e] = v2; }
            else { D[e] = vd; g[e] = vg; y[e] = 0.0; }
        }
        for (int e = t; e < d.prior_H_sz; e += LT) stageP[e] = PH[e];
        if (attempt == 0) {
            for (int l = l0 + t; l < l1; l += LT) gmax_l = fmax(gmax_l, fabs(d.lmG[l]));
        }
        __syncthreads();
        if (attempt > 0) {
            // ---- mu retry: T -= sum_l (c_l(mu) - c_l(mu0)) w_l w_l^T, bs -= sum_l dc_l g_l w_l (rare: never on well-posed windows)
            double *wS = u + n;                                 // (behind the prior blocks and u: work area >= retry staging, st_work_doubles)
            wS = work + ((d.prior_H_sz + 1) & ~1);              // the prior blocks stay; u is not live yet
            const int wld = n6 + 1;
            double *dC = wS + RCH_ST * wld, *dG = dC + RCH_ST;
            const bool fits = ((d.prior_H_sz + 1) & ~1) + RCH_ST * wld + 2 * RCH_ST <= st_work_doubles(N, d.prior_H_sz);
            // (when the staging does not fit behind the prior blocks it overlays them and they are re-staged afterwards)
            if (!fits) { wS = work; dC = wS + RCH_ST * wld; dG = dC + RCH_ST; }
            for (int lb = l0; lb < l1; lb += RCH_ST) {
                const int cnt = (l1 - lb) < RCH_ST ? (l1 - lb) : RCH_ST;
                __syncthreads();
                for (int e = t; e < cnt * n6; e += LT) {         // expand the packed w vectors to dense rows
                    const int r = e / n6, c = e - r * n6;
                    const unsigned m0 = d.lm_meta[lb + r];
                    const int h6 = 6 * (int)(m0 & 255), k6 = 6 * (int)((m0 >> 8) & 255);
                    wS[r * wld + c] = (c >= h6 && c < h6 + k6) ? d.W[(size_t)(d.f_off[w] + (int)(m0 >> 16) + lb + r) * 6 + (c - h6)]
                                    : ((d.est_ex && c >= 6 * d.Nr) ? d.Wex[(size_t)(lb + r) * 6 + (c - 6 * d.Nr)] : 0.0);
                }
                if (t < cnt) {
                    const int l = lb + t;
                    const double2 cg = d.lm_cg[l];
                    const double sl = d.scale_l[l], Es = sl * sl * d.lmE[l];
                    const double Dl2 = fmin(fmax(Es, 1e-6), 1e32);
                    const double dc = sl * sl / (Es + mu * Dl2) - cg.x;
                    dC[t] = dc; dG[t] = dc * cg.y;
                }
                __syncthreads();
                for (int e = t; e < nS; e += LT) {               // owner-computes: entry e of the packed pose blocks
                    const int q = e / 36, rc = e - 36 * q, r = rc / 6, c = rc - 6 * r;
                    const int ij = blkIJ[q], ra = 6 * (ij & 255) + r, cb = 6 * (ij >> 8) + c;
                    double acc = 0;
                    for (int l = 0; l < cnt; l++) acc += dC[l] * wS[l * wld + ra] * wS[l * wld + cb];
                    Spp[e] -= acc;
                }
                if (t < n6) {
                    double accb = 0;
                    for (int l = 0; l < cnt; l++) accb += dG[l] * wS[l * wld + t];
                    y[15 * (t / 6) + t % 6] -= accb;
                }
            }
            __syncthreads();
            if (!fits) { for (int e = t; e < d.prior_H_sz; e += LT) stageP[e] = PH[e]; }
            __syncthreads();
        }
        STSTAMP(0);
        // IMU factors' pose-pose blocks (owner of the item adds into Spp; the items are disjoint entries)
        for (int e = t; e < N * 21 + (N - 1) * 36; e += LT) {
            if (e < N * 21) {
                const int I = e / 21, q = e - 21 * I, r = triAB[2 * q], c = triAB[2 * q + 1];
                double v = HAp(I)[st_pair(15 + r, 15 + c)] * hasA(I);
                v += HBp(I)[st_pair(r, c)] * hasB(I);
                Spp[st_sblk(I, I, N) + r * 6 + c] += v;
            } else {
                const int q0 = e - N * 21, I = q0 / 36, q = q0 - 36 * I, r = q / 6, c = q - 6 * r;
                Spp[st_sblk(I + 1, I, N) + q] += H[(size_t)I * ISV_IMU_H + st_pair(15 + r, c)] * (skipL[I] ? 0.0 : 1.0);
            }
        }
        __syncthreads();
        STSTAMP(1);
        if (d.est_ex && t < 9) D[15 * (N - 1) + 6 + t] = 1.0;    // the pseudo-frame's dummy speed/bias block: unit diagonal
        // ---- prior factors (precomputed J^T J, staged): three conflict-free phases (k_build_solve_sb's scheme); the Linear9
        //      block's speed/bias part only feeds the diagonal and the gradient here -- its matrix entries join D_i on the way in
        {
            auto prior_entry = [&](int q, int e) {
                int ncol, off, c0, c1 = 0;
                if (q == 0) { ncol = 6; off = PH_SE3; c0 = 0; }
                else if (q == 1) { ncol = 9; off = PH_LIN9; c0 = 15 * (d.Nvo - 1) + 6; }
                else if (q < 1 + d.Nvo) { const int k = q - 2; ncol = 12; off = PH_REL0 + PH_REL_SZ * k; c0 = 15 * k; c1 = 15 * (k + 1); }
                else { const int m = q - 1 - d.Nvo; ncol = 6; off = PH_REL0 + PH_REL_SZ * (d.Nvo - 1) + PH_RP_SZ * m; c0 = 15 * d.rollpitch[(size_t)w * d.max_rp + m].index; }
                const int np2 = ncol * (ncol + 1) / 2;
                if (e >= np2 + ncol) return;
                const double v = stageP[off + e];
                if (e < np2) {
                    const int aa = triAB[2 * e], bb = triAB[2 * e + 1];
                    const int ga = (aa < 6 || ncol != 12) ? c0 + aa : c1 + aa - 6;
                    const int gb = (bb < 6 || ncol != 12) ? c0 + bb : c1 + bb - 6;
                    const int Ia = ga / 15, ra = ga - 15 * Ia, Ib = gb / 15, rb = gb - 15 * Ib;
                    if (ra < 6) Spp[st_sblk(Ia, Ib, N) + ra * 6 + rb] += v;
                    if (aa == bb) D[ga] += v;
                } else {
                    const int aa = e - np2;
                    const int ga = (aa < 6 || ncol != 12) ? c0 + aa : c1 + aa - 6;
                    g[ga] += v;
                }
            };
            const int nrel = d.Nvo - 1, nrp = d.n_rp[w];
            for (int e = t; e < 90 * (1 + (nrel + 1) / 2); e += LT) { const int sl = e / 90; prior_entry(sl == 0 ? 1 : 2 + 2 * (sl - 1), e - 90 * sl); }
            __syncthreads();
            for (int e = t; e < 90 * (1 + nrel / 2); e += LT) { const int sl = e / 90; prior_entry(sl == 0 ? 0 : 2 + 2 * (sl - 1) + 1, e - 90 * sl); }
            __syncthreads();
            if (t < 27) for (int m = 0; m < nrp; m++) prior_entry(1 + d.Nvo + m, t);
            __syncthreads();
        }
        STSTAMP(2);
        // ---- Jacobi scaling, LM diagonal, Cauchy data --------------------------------------------------------------------
        for (int e = t; e < n; e += LT) {
            const double hd = D[e];
            double s;
            if (iteration == 0) { s = 1.0 / (1.0 + sqrt(hd)); d.scale_p[(size_t)w * n + e] = s; }
            else s = d.scale_p[(size_t)w * n + e];
            const double D2 = fmin(fmax(s * s * hd, 1e-6), 1e32);
            const double De = sqrt(D2);
            sc[e] = s; D[e] = De;
            d.diag_p[(size_t)w * n + e] = De;
            d.grad_p[(size_t)w * n + e] = s * g[e] / De;
            const double ue = s * s * g[e] / D2;
            u[e] = ue;
            d.up[(size_t)w * n + e] = ue;
            y[e] = s * (g[e] + y[e]);
        }
        __syncthreads();
        {   // qT = u^T T u on the unscaled blocks: the pose blocks from LDS (scaled in place afterwards), everything that involves a
            // speed/bias block straight from the packed records (those entries are never assembled unscaled)
            double accq = 0;
            for (int e = t; e < nS; e += LT) {
                const int q = e / 36, rc = e - 36 * q, r = rc / 6, c = rc - 6 * r;
                const int ij = blkIJ[q], I = ij & 255, J = ij >> 8;
                if (I == J && r < c) continue;
                const int gi = 15 * I + r, gj = 15 * J + c;
                const double v = Spp[e];
                accq += (gi == gj ? 1.0 : 2.0) * v * u[gi] * u[gj];
                double sv = v * sc[gi] * sc[gj];
                if (gi == gj) sv += mu * D[gi] * D[gi];
                Spp[e] = sv;
            }
            // every block that involves a speed/bias block: formed ONCE here by all threads from the packed records -- D_i (lower),
            // C_i, the three pose x speed/bias blocks of node i -- scaled, with the LM diagonal, into the global scratch from which
            // the chain pipeline streams them (plain coalesced copies there: no index arithmetic, no dependent loads on the chain);
            // the unscaled value's term of u^T T u rides along.  A thread owns entry q of EVERY node (its index arithmetic is node
            // independent) and keeps four nodes' loads in flight.
            {
                const int q = t;
                if (q < 81) {
                    const int r = q / 9, c = q - 9 * r;
                    const bool lower = r >= c;
                    for (int i0 = 0; i0 < N; i0 += 4) {
                        double raw[4];
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) { const int i = i0 + k4 < N ? i0 + k4 : N - 1; raw[k4] = rawD(i, lower ? r : c, lower ? c : r); }
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) {
                            const int i = i0 + k4;
                            if (i < N) {
                                double sv = 0.0;
                                if (lower) {
                                    const int gi = 15 * i + 6 + r, gj = 15 * i + 6 + c;
                                    accq += (r == c ? 1.0 : 2.0) * raw[k4] * u[gi] * u[gj];
                                    sv = raw[k4] * sc[gi] * sc[gj];
                                    if (r == c) sv += mu * D[gi] * D[gi];
                                }
                                gDinv[i * 81 + q] = sv;
                            }
                        }
                    }
                } else if (q < 162) {
                    const int a = (q - 81) / 9, b = (q - 81) - 9 * a;
                    for (int i0 = 0; i0 < N; i0 += 4) {
                        double raw[4];
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) { const int i = i0 + k4 < N ? i0 + k4 : N - 1; raw[k4] = rawC(i, a, b); }     // (i == M: a valid load, unused)
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) {
                            const int i = i0 + k4;
                            if (i < N) {
                                double sv = 0.0;
                                if (i != M) {
                                    const int pp = i < M ? i + 1 : i - 1;
                                    const int gi = 15 * pp + 6 + a, gj = 15 * i + 6 + b;
                                    accq += 2.0 * raw[k4] * u[gi] * u[gj];
                                    sv = raw[k4] * sc[gi] * sc[gj];
                                }
                                gC[i * 81 + (q - 81)] = sv;
                            }
                        }
                    }
                } else if (q < 324) {                           // (512-thread instantiation: the threads beyond the 324 entries idle here)
                    const int b3 = (q - 162) / 54, rc = (q - 162) - 54 * b3, r = rc / 9, c = rc - 9 * r;
                    for (int i0 = 0; i0 < N; i0 += 4) {
                        double raw[4];
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) { const int i = i0 + k4 < N ? i0 + k4 : N - 1; raw[k4] = rawY(i, i - 1 + b3, r, c); }     // (a block outside the window: clamped record, masked to zero)
#pragma unroll
                        for (int k4 = 0; k4 < 4; k4++) {
                            const int i = i0 + k4, pz = i - 1 + b3;
                            if (i < N) {
                                double sv = 0.0;
                                if (pz >= 0 && pz <= N - 1) {
                                    const int gi = 15 * pz + r, gj = 15 * i + 6 + c;
                                    accq += 2.0 * raw[k4] * u[gi] * u[gj];
                                    sv = raw[k4] * sc[gi] * sc[gj];
                                }
                                gYi[(size_t)i * 162 + (q - 162)] = sv;
                            }
                        }
                    }
                }
            }
            if constexpr (LT < 324) {   // entries 256 .. 323 of every node (the tail of its pose x speed/bias blocks), item-wise over all threads
                constexpr int TQ = 324 - LT;
                const int nitems = N * TQ;
                for (int it0 = t; it0 < nitems; it0 += 4 * LT) {
                    double raw[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; k4++) {
                        const int it = it0 + k4 * LT < nitems ? it0 + k4 * LT : nitems - 1;
                        const int i = it / TQ, q = LT + (it - TQ * i), b3 = (q - 162) / 54, rc = (q - 162) - 54 * b3, r = rc / 9, c = rc - 9 * r;
                        raw[k4] = rawY(i, i - 1 + b3, r, c);
                    }
#pragma unroll
                    for (int k4 = 0; k4 < 4; k4++) {
                        const int it = it0 + k4 * LT;
                        if (it < nitems) {
                            const int i = it / TQ, q = LT + (it - TQ * i), b3 = (q - 162) / 54, rc = (q - 162) - 54 * b3, r = rc / 9, c = rc - 9 * r, pz = i - 1 + b3;
                            double sv = 0.0;
                            if (pz >= 0 && pz <= N - 1) {
                                const int gi = 15 * pz + r, gj = 15 * i + 6 + c;
                                accq += 2.0 * raw[k4] * u[gi] * u[gj];
                                sv = raw[k4] * sc[gi] * sc[gj];
                            }
                            gYi[(size_t)i * 162 + (q - 162)] = sv;
                        }
                    }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) accq += __shfl_xor(accq, off);
            if ((t & 63) == 0) red[t >> 6] = accq;
            __syncthreads();
            if (t == 0) st.qT = BIG ? ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])) : (red[0] + red[1]) + (red[2] + red[3]);
        }
        __syncthreads();                                           // u and the staged priors are dead: the work area becomes the chain buffers
        STSTAMP(3);
        // Spp -= Y' Y'^T of a panel as FP64 MFMA tiles (v_mfma_f64_16x16x4: D += A^T B over four k): the operand is a row-major panel
        // [pose rows][k columns] in LDS; a wavefront takes the 16 x 16 tiles tile0, tile0 + tstep, .. of the lower triangle; every
        // lower entry of the covered pose blocks is owned by one lane of one tile (no atomics; per entry: k ascending, ONE
        // subtraction).  skipPM: leave out pose block (P0, P0) (the backward child of M: that block is the forward child's in the
        // same step; applied in the join).  As scalar 2 x 3 sub-blocks this cost the lag wavefront 4 us per node.
        typedef double st_double4 __attribute__((ext_vector_type(4)));
        auto yyt_tiles = [&](auto rowoff, const double *Yb, int R, int K, int P0, int tile0, int tstep, bool skipPM, int lane) {
            const int T = (R + 15) >> 4, ntile = T * (T + 1) / 2, ksteps = (K + 3) >> 2;
            const int i16 = lane & 15, kq = lane >> 4;
            for (int tile = tile0; tile < ntile; tile += tstep) {
                int TI = 0;
                while ((TI + 1) * (TI + 2) / 2 <= tile) TI++;
                const int TJ = tile - TI * (TI + 1) / 2;
                const int ra = 16 * TI + i16, rb = 16 * TJ + i16;
                const double ma = ra < R ? 1.0 : 0.0, mb = rb < R ? 1.0 : 0.0;
                const int oa = rowoff(ra < R ? ra : R - 1), ob = rowoff(rb < R ? rb : R - 1);
                st_double4 acc = {0, 0, 0, 0};
                for (int s4 = 0; s4 < ksteps; s4++) {
                    const int k = 4 * s4 + kq, kc = k < K ? k : K - 1;
                    const double mk = k < K ? 1.0 : 0.0;
                    const double av = Yb[oa + kc] * (ma * mk), bv = Yb[ob + kc] * (mb * mk);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {          // C/D layout: col = lane & 15, row = (lane >> 4) + 4 reg
                    const int Rr = 16 * TI + kq + 4 * reg, Cc = 16 * TJ + i16;
                    if (Rr < R && Cc <= Rr && !(skipPM && Rr < 6)) {
                        const int pI = Rr / 6, r = Rr - 6 * pI, pJ = Cc / 6, c = Cc - 6 * pJ;
                        Spp[st_sblk(P0 + pI, P0 + pJ, N) + r * 6 + c] -= acc[reg];
                    }
                }
            }
        };
        // ---- speed/bias chains -------------------------------------------------------------------------------------------
        // wavefront 0 / 1: the critical path of the forward (0 .. M-1) / backward (N-1 .. M+1) chain -- factor D_i, C_i' = C_i L^-T,
        // z_i, downdate of the parent's D and rhs.  Wavefronts 2 / 3 follow ONE NODE BEHIND with the bulk: Y_i' = Y_i L^-T (spilled to
        // the global scratch), Spp -= Y_i' Y_i'^T, y_pose -= Y_i' z_i, the parent's fill formed in place; the blocks the chain needs two
        // nodes ahead (D, C, the parent's own pose x speed/bias blocks) are REQUESTED at the top of the step and land in LDS at its
        // end: their memory latency runs behind the LDS work.  One block barrier per node.
        {
            ST_IDS();
            const int cntF = M, cntB = N - 1 - M;
            auto Dslot = [&](int ch, int i) { return Dring + (ch * 4 + (i & 3)) * 81; };
            auto Cslot = [&](int ch, int i) { return Cring + (ch * 3 + (i % 3)) * 81; };
            // first nodes and the first ring entries
            for (int e = t; e < 108; e += LT) {
                const int b = e / 54, rc = e - 54 * b;
                if (cntF >= 1) Ybuf[b * 54 + rc] = gYi[0 * 162 + (1 + b) * 54 + rc];                              // Y_0: poses 0, 1 (blocks 1, 2 of node 0)
                if (cntB >= 1) Ybuf[(N - 2 + b + 1) * 54 + rc] = gYi[(size_t)(N - 1) * 162 + b * 54 + rc];        // Y_N-1: poses N-2, N-1 (blocks 0, 1)
            }
            for (int e = t; e < 81 * 5; e += LT) {
                const int wh = e / 81, q = e - 81 * wh;
                if (wh == 0) { if (cntF >= 1) Dslot(0, 0)[q] = gDinv[q]; }
                else if (wh == 1) { if (cntF >= 1) Cslot(0, 0)[q] = gC[q]; }
                else if (wh == 2) Dslot(0, 1 <= M ? 1 : M)[q] = gDinv[(1 <= M ? 1 : M) * 81 + q];                 // (D_M when there is no forward chain)
                else if (wh == 3) { if (cntB >= 1) Dslot(1, N - 1)[q] = gDinv[(N - 1) * 81 + q]; }
                else { if (cntB >= 1) Cslot(1, N - 1)[q] = gC[(N - 1) * 81 + q]; }
            }
            if (cntB >= 2) { for (int e = t; e < 81; e += LT) Dslot(1, N - 2)[e] = gDinv[(N - 2) * 81 + e]; }
            __syncthreads();
            auto crit = [&](int ch, int i) {
                double *Di = Dslot(ch, i), *Ci = Cslot(ch, i);
                if (st_chol_inv<9>(Di, lane)) { if (lane == 0) flag[0] = 1; return; }
                if (lane < 10) {                                   // C_i' = C_i L_i^-T (9 rows), z_i^T = y_i^T L_i^-T
                    double *ptr = lane < 9 ? Ci + lane * 9 : y + 15 * i + 6;
                    double v[9], o[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = ptr[k];
#pragma unroll
                    for (int c = 0; c < 9; c++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k <= c; k++) s += v[k] * Di[c * 9 + k];
                        o[c] = s;
                    }
#pragma unroll
                    for (int k = 0; k < 9; k++) ptr[k] = o[k];
                }
                ST_WSYNC();
                const int pp = ch == 0 ? i + 1 : i - 1;
                if (pp == M) return;                               // both chains end in M: applied after the loop
                double *Dp = Dslot(ch, pp);
                if (lane < 45) {                                   // D_p -= C_i' C_i'^T
                    const int r = triAB[2 * lane], c = triAB[2 * lane + 1];
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += Ci[r * 9 + k] * Ci[c * 9 + k];
                    Dp[r * 9 + c] -= s;
                } else if (lane < 54) {                            // y_p -= C_i' z_i
                    const int r = lane - 45;
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += Ci[r * 9 + k] * y[15 * i + 6 + k];
                    y[15 * pp + 6 + r] -= s;
                }
                ST_WSYNC();
            };
            // Y-work of node j by ONE wavefront (nothing stays in registers across its stages).  A child of M (last = true) forms no
            // parent; the BACKWARD child also leaves pose block (M, M) and pose M's rhs to the join: the forward child updates them in
            // the same step.
            auto ywork = [&](int ch, int j, bool last) {
                const int lo = st_nlo(j, M), hi = st_nhi(j, M, N), nr = hi - lo + 1, s0 = lo + ch;       // first slot
                const double *Lj = Dslot(ch, j), *Cj = Cslot(ch, j);
                double *Yg = gY + yo[j]; (void)Yg;
                // spill L_j^-1 and C_j' (final since the critical path left node j)
                for (int e = lane; e < 162; e += 64) { if (e < 81) gDinv[j * 81 + e] = Lj[e]; else gC[j * 81 + (e - 81)] = Cj[e - 81]; }
                for (int rho = lane; rho < 6 * nr; rho += 64) {
                    double *ptr = Ybuf + (s0 * 6 + rho) * 9;
                    double v[9], o[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = ptr[k];
#pragma unroll
                    for (int c = 0; c < 9; c++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k <= c; k++) s += v[k] * Lj[c * 9 + k];
                        o[c] = s;
                    }
#pragma unroll
                    for (int k = 0; k < 9; k++) ptr[k] = o[k];
                    if constexpr (BIG || !ST_NO_YSPILL) {              // (long windows: the tile wavefronts read it; ST_NO_YSPILL = 0: the back-substitution)
#pragma unroll
                        for (int k = 0; k < 9; k++) Yg[k * (6 * nr) + rho] = o[k];                       // global: column-major per node
                    }
                }
                ST_WSYNC();
                {   // pose rhs -= Y_j' z_j, then the pose blocks' downdate
                    const bool deferM = last && ch == 1;             // (lo == M there: pose M's rhs rows are the forward child's in this step)
                    for (int rr = lane; rr < 6 * nr; rr += 64) {
                        const int a = rr / 6, r = rr - 6 * a;
                        if (deferM && a == 0) continue;
                        const double *Yr = Ybuf + ((s0 + a) * 6 + r) * 9, *zi = y + 15 * j + 6;
                        double s = 0;
#pragma unroll
                        for (int kk = 0; kk < 9; kk++) s += Yr[kk] * zi[kk];
                        y[15 * (lo + a) + r] -= s;
                    }
                    // Spp -= Y_j' Y_j'^T: this node's rows of Ybuf are the panel (9 columns)
                    // (long windows: wavefronts 4 .. 7 do this two nodes behind, from the spilled Y' -- ten tiles of a 60-row node on this
                    //  wavefront made IT the chain's critical path: 7 us per node against 3.5 us for the factorisation)
                    if constexpr (!BIG) yyt_tiles([&](int row) { return (s0 * 6 + row) * 9; }, Ybuf, 6 * nr, 9, lo, 0, 1, deferM, lane);
                }
                if (last) return;
                ST_WSYNC();
                for (int rho = lane; rho < 6 * nr; rho += 64) {      // parent row rho: - Y'[rho] C_j'^T, in place
                    double *ptr = Ybuf + (s0 * 6 + rho) * 9;
                    double o[9], q9v[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) o[k] = ptr[k];
#pragma unroll
                    for (int c = 0; c < 9; c++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 9; k++) s += o[k] * Cj[c * 9 + k];
                        q9v[c] = -s;
                    }
#pragma unroll
                    for (int c = 0; c < 9; c++) ptr[c] = q9v[c];
                }
                ST_WSYNC();
            };
            const int steps = (cntF > cntB ? cntF : cntB) + 1;
            for (int k = 0; k < steps; k++) {
                if (wv < 2) {
                    const int cnt = wv == 0 ? cntF : cntB;
                    if (k < cnt) crit(wv, wv == 0 ? k : N - 1 - k);
                } else if (wv < 4) {
                    const int ch = wv - 2, cnt = ch == 0 ? cntF : cntB, kk = k - 1;
                    // requests: D two nodes ahead of the critical path (k + 2; the forward ring also carries D_M), the coupling of the next
                    // node (k + 1), and the own blocks of the node whose fill this step forms (the parent of kk)
                    const int iD = ch == 0 ? k + 2 : N - 1 - (k + 2), iC = ch == 0 ? k + 1 : N - 1 - (k + 1);
                    const bool needD = ch == 0 ? (k + 2 <= M && cntF >= 1) : (k + 2 < cntB);
                    const bool needC = k + 1 < cnt;
                    const bool work_j = kk >= 0 && kk < cnt, last = kk == cnt - 1;
                    const int j = ch == 0 ? kk : N - 1 - kk, pp = ch == 0 ? j + 1 : j - 1;
                    const bool needI = work_j && !last;
                    double pD[2], pC[2], pI[3];
                    {
                        const int iDc = needD ? iD : M, iCc = needC ? iC : (M >= 1 ? M - 1 : 0), ppc = needI ? pp : M;
#pragma unroll
                        for (int q = 0; q < 2; q++) { const int e = lane + 64 * q < 81 ? lane + 64 * q : 80; pD[q] = gDinv[iDc * 81 + e]; pC[q] = gC[iCc * 81 + e]; }
#pragma unroll
                        for (int q = 0; q < 3; q++) { const int e = lane + 64 * q < 162 ? lane + 64 * q : 161; pI[q] = gYi[(size_t)ppc * 162 + e]; }
                    }
                    if (work_j) ywork(ch, j, last);
                    if (needI) {                                     // the parent's own three blocks (poses pp - 1, pp, pp + 1); one of them is a new slot
#pragma unroll
                        for (int q = 0; q < 3; q++) {
                            const int e = lane + 64 * q;
                            if (e < 162) {
                                const int b = e / 54, rc = e - 54 * b, pz = pp - 1 + b;
                                if (pz >= 0 && pz <= N - 1) {
                                    double *dst = Ybuf + (pz + ch) * 54 + rc;
                                    const bool fresh = ch == 0 ? pz == pp + 1 : pz == pp - 1;       // the slot this node adds to the chain's fill
                                    if (fresh) *dst = pI[q]; else *dst += pI[q];
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int e = lane + 64 * q;
                        if (e < 81) { if (needD) Dslot(ch, iD)[e] = pD[q]; if (needC) Cslot(ch, iC)[e] = pC[q]; }
                    }
                }
                if constexpr (BIG) {
                    if (wv >= 4) {
                        // Spp -= Y_j' Y_j'^T of the node the lag wavefront finished in the PREVIOUS step (its Y' is in the global scratch,
                        // column-major: an operand is a coalesced read of 16 consecutive rows); two wavefronts per chain split the tiles.
                        // The children of M are left to the join (both touch pose block (M, M)).
                        const int ch = (wv - 4) >> 1, half = (wv - 4) & 1, cnt = ch == 0 ? cntF : cntB, kk = k - 2;
                        if (kk >= 0 && kk < cnt - 1) {
                            const int j = ch == 0 ? kk : N - 1 - kk;
                            const int lo = st_nlo(j, M), nr = st_nhi(j, M, N) - lo + 1, R = 6 * nr;
                            const double *Yg = gY + yo[j];
                            const int T = (R + 15) >> 4, ntile = T * (T + 1) / 2;
                            const int i16 = lane & 15, kq = lane >> 4;
                            for (int tile = half; tile < ntile; tile += 2) {
                                int TI = 0;
                                while ((TI + 1) * (TI + 2) / 2 <= tile) TI++;
                                const int TJ = tile - TI * (TI + 1) / 2;
                                const int ra = 16 * TI + i16, rb = 16 * TJ + i16;
                                double av[3], bv[3];
#pragma unroll
                                for (int s4 = 0; s4 < 3; s4++) {          // (clamped, unconditional loads; masked by multiplication)
                                    const int kcol = 4 * s4 + kq, kc = kcol < 9 ? kcol : 8;
                                    av[s4] = Yg[kc * R + (ra < R ? ra : R - 1)]; bv[s4] = Yg[kc * R + (rb < R ? rb : R - 1)];
                                }
                                st_double4 acc = {0, 0, 0, 0};
#pragma unroll
                                for (int s4 = 0; s4 < 3; s4++) {
                                    const double mk = 4 * s4 + kq < 9 ? 1.0 : 0.0;
                                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4] * (ra < R ? mk : 0.0), bv[s4] * (rb < R ? mk : 0.0), acc, 0, 0, 0);
                                }
#pragma unroll
                                for (int reg = 0; reg < 4; reg++) {
                                    const int Rr = 16 * TI + kq + 4 * reg, Cc = 16 * TJ + i16;
                                    if (Rr < R && Cc <= Rr) {
                                        const int pI = Rr / 6, r = Rr - 6 * pI, pJ = Cc / 6, c = Cc - 6 * pJ;
                                        Spp[st_sblk(lo + pI, lo + pJ, N) + r * 6 + c] -= acc[reg];
                                    }
                                }
                            }
                        }
                    }
                }
                // (short windows: inside the loop nobody reads global data written inside it -- the spills are read after the loop, the
                //  prefetched blocks were written before it -- so the per-node barrier orders LDS only; long windows: the tile wavefronts
                //  read the previous step's spills, a full barrier)
                if constexpr (BIG) __syncthreads(); else ST_LDS_BARRIER();
                if (flag[0]) break;
            }
            __syncthreads();
            if constexpr (BIG) {
                // (every non-child node got its tiles inside the loop: node cnt - 2 in step cnt.)  The children of M: straight from Ybuf,
                // where their Y' still lie, with all wavefronts; forward child, then the backward one (it leaves block (M, M) to the join)
                if (!flag[0]) {
                    for (int ch = 0; ch < 2; ch++) {
                        const int cnt = ch == 0 ? cntF : cntB;
                        if (cnt >= 1) {
                            const int j = ch == 0 ? M - 1 : M + 1, lo = st_nlo(j, M), nr = st_nhi(j, M, N) - lo + 1, s0 = lo + ch;
                            yyt_tiles([&](int row) { return (s0 * 6 + row) * 9; }, Ybuf, 6 * nr, 9, lo, wv, NW, ch == 1, lane);
                        }
                        __syncthreads();
                    }
                }
            }
        }
        STSTAMP(4);
        if (!flag[0]) {
            // ---- the join: what the backward child left of pose block (M, M), then node M -----------------------------------
            ST_IDS();
            auto Dslot = [&](int ch, int i) { return Dring + (ch * 4 + (i & 3)) * 81; };
            auto Cslot = [&](int ch, int i) { return Cring + (ch * 3 + (i % 3)) * 81; };
            auto slotM = [&](int pz) { return pz <= M ? pz : pz + 1; };
            const bool hasF = M >= 1, hasB2 = M + 1 <= N - 1;
            double *DM = Dslot(0, M);
            if (wv == 3 && hasB2) {
                // the backward child's Y' rows of pose M (slot M + 1): pose block (M, M) and pose M's rhs
                const double *YM = Ybuf + (M + 1) * 54, *zi = y + 15 * (M + 1) + 6;
                if (lane < 36) {
                    const int r = lane / 6, c = lane - 6 * r;
                    if (r >= c) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 9; k++) s += YM[r * 9 + k] * YM[c * 9 + k];
                        Spp[st_sblk(M, M, N) + r * 6 + c] -= s;
                    }
                } else if (lane < 42) {
                    const int r = lane - 36;
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += YM[r * 9 + k] * zi[k];
                    y[15 * M + r] -= s;
                }
            } else if (wv == 0) {                                // D_M -= C_a' C_a'^T + C_b' C_b'^T, y_M -= C_a' z_a + C_b' z_b (a = M - 1 first)
                for (int side = 0; side < 2; side++) {
                    if (side == 0 ? !hasF : !hasB2) continue;
                    const int j = side == 0 ? M - 1 : M + 1;
                    const double *Cj = Cslot(side, j);
                    if (lane < 45) {
                        const int r = triAB[2 * lane], c = triAB[2 * lane + 1];
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 9; k++) s += Cj[r * 9 + k] * Cj[c * 9 + k];
                        DM[r * 9 + c] -= s;
                    } else if (lane < 54) {
                        const int r = lane - 45;
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 9; k++) s += Cj[r * 9 + k] * y[15 * j + 6 + k];
                        y[15 * M + 6 + r] -= s;
                    }
                }
                ST_WSYNC();
                if (st_chol_inv<9>(DM, lane)) { if (lane == 0) flag[0] = 1; }
                else if (lane == 0) {                            // z_M
                    double v[9], o[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = y[15 * M + 6 + k];
#pragma unroll
                    for (int c = 0; c < 9; c++) { double s = 0; for (int k = 0; k <= c; k++) s += v[k] * DM[c * 9 + k]; o[c] = s; }
#pragma unroll
                    for (int k = 0; k < 9; k++) y[15 * M + 6 + k] = o[k];
                }
            }
            // (the rows of Y_M below read the children's Y' rows and C' -- not D_M, not pose block (M, M): no barrier needed before them)
            // Y_M rows in the join layout (slotM): pose < M from the forward child, pose > M from the backward child, pose M from
            // both; plus the node's own three blocks.  Row by row in place (a row of Y_M depends on the same rows of the children).
            for (int rho = t; rho < 6 * N; rho += LT) {
                const int pz = rho / 6, r = rho - 6 * pz;
                double acc[9];
#pragma unroll
                for (int c = 0; c < 9; c++) acc[c] = 0.0;
                for (int side = 0; side < 2; side++) {
                    if (side == 0 ? !(hasF && pz <= M) : !(hasB2 && pz >= M)) continue;
                    const int j = side == 0 ? M - 1 : M + 1;
                    const double *Cj = Cslot(side, j), *Yr = Ybuf + ((pz + side) * 6 + r) * 9;
                    double o[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) o[k] = Yr[k];
#pragma unroll
                    for (int c = 0; c < 9; c++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 9; k++) s += o[k] * Cj[c * 9 + k];
                        acc[c] -= s;
                    }
                }
                if (pz >= M - 1 && pz <= M + 1) {
                    const double *Yi = gYi + (size_t)M * 162 + (pz - (M - 1)) * 54 + r * 9;
#pragma unroll
                    for (int c = 0; c < 9; c++) acc[c] += Yi[c];
                }
                // (pose M reads slots M and M + 1 and writes slot M; every other pose reads and writes its own slot; the deferred
                //  (M, M) update above reads slot M + 1, which pose M's row only reads as well)
                double *dst = Ybuf + (slotM(pz) * 6 + r) * 9;
#pragma unroll
                for (int c = 0; c < 9; c++) dst[c] = acc[c];
            }
            __syncthreads();
            if (!flag[0]) {
                // Y_M' = Y_M L_M^-T, spill; then Spp -= Y_M' Y_M'^T over ALL pose blocks, pose rhs -= Y_M' z_M
                double *Yg = gY + yo[M]; (void)Yg;
                for (int e = t; e < 81; e += LT) gDinv[M * 81 + e] = DM[e];
                for (int rho = t; rho < 6 * N; rho += LT) {
                    const int pz = rho / 6, r = rho - 6 * pz;
                    double *ptr = Ybuf + (slotM(pz) * 6 + r) * 9;
                    double v[9], o[9];
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = ptr[k];
#pragma unroll
                    for (int c = 0; c < 9; c++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k <= c; k++) s += v[k] * DM[c * 9 + k];
                        o[c] = s;
                    }
#pragma unroll
                    for (int k = 0; k < 9; k++) ptr[k] = o[k];
#if !ST_NO_YSPILL
#pragma unroll
                    for (int k = 0; k < 9; k++) Yg[k * (6 * N) + rho] = o[k];
#endif
                }
                __syncthreads();
                // Spp -= sum_i Y_i' Y_i'^T as FP64 MFMA tiles (v_mfma_f64_16x16x4: D += A^T B over four k): the operand is a dense
                // row-major panel [rows = the pose rows a set of nodes covers][k = the nodes' 9 columns side by side, zero where a node
                // does not reach a pose]; one wavefront per 16 x 16 tile of the lower triangle; every lower entry of the covered pose
                // blocks is owned by one lane of one tile (no atomics; per entry: nodes ascending, k ascending, ONE subtraction per
                // panel).  Node M's panel is its fill as it lies in Ybuf; the chain nodes' Y' come back from the global scratch in
                // panels that fit the (then dead) chain buffers.  As scalar 2 x 3 sub-blocks on 256 threads this took 22 us per window.
                // node M: rows in the join layout (slotM), 9 columns; its pose rhs first (rows of Y_M' times z_M)
                for (int rr = t; rr < 6 * N; rr += LT) {
                    const int a = rr / 6, r = rr - 6 * a;
                    const double *Yr = Ybuf + (slotM(a) * 6 + r) * 9, *zi = y + 15 * M + 6;
                    double s = 0;
#pragma unroll
                    for (int kk = 0; kk < 9; kk++) s += Yr[kk] * zi[kk];
                    y[15 * a + r] -= s;
                }
                yyt_tiles([&](int row) { const int pz = row / 6; return (slotM(pz) * 6 + (row - 6 * pz)) * 9; }, Ybuf, 6 * N, 9, 0, wv, NW, false, lane);
                __syncthreads();
            }
        }
        STSTAMP(5);
        if (!flag[0]) {
            // ---- blocked Cholesky of the pose system (6x6 blocks; diagonal blocks hold L_JJ^-1 afterwards), look-ahead on wavefront 0
            ST_IDS();
            if (wv == 0 && st_chol_inv<6>(Spp + st_sblk(0, 0, N), lane)) { if (lane == 0) flag[0] = 1; }
            __syncthreads();
            for (int J = 0; J < N; J++) {
                if (flag[0]) break;
                const int m = N - J - 1;
                const double *Li = Spp + st_sblk(J, J, N);
                for (int rr = t; rr < m * 6 + 1; rr += LT) {    // panel rows: X = A L_JJ^-T; last row = rhs (z_J)
                    double *A = rr < m * 6 ? Spp + st_sblk(J + 1, J, N) + rr * 6 : y + 15 * J;   // blocks (J+1.., J) are contiguous
                    double v[6], o[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) v[k] = A[k];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k <= c; k++) s += v[k] * Li[c * 6 + k];
                        o[c] = s;
                    }
#pragma unroll
                    for (int k = 0; k < 6; k++) A[k] = o[k];
                }
                __syncthreads();
                if (m == 0) break;
                const int e0 = st_sblk(J + 1, J + 1, N), cntT = m * (m + 1) / 2 * 36;
                const double *X = Spp + st_sblk(J + 1, J, N);       // X_I at (I - J - 1) * 36
                auto trailing_entry = [&](int e) {
                    if (e >= cntT) {                             // rhs rows: y_I -= X_I z_J
                        const int rr = e - cntT;
                        const double *XI = X + rr * 6;
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 6; k++) s += XI[k] * y[15 * J + k];
                        y[15 * (J + 1 + rr / 6) + rr % 6] -= s;
                        return;
                    }
                    const int q = e / 36, rc = e - 36 * q, r = rc / 6, c = rc - 6 * r;
                    const int ij = blkIJ[e0 / 36 + q];
                    const int ia = (ij & 255) - (J + 1), ca = (ij >> 8) - (J + 1);
                    if (ia == ca && r < c) return;
                    const double *XI = X + ia * 36 + r * 6, *XK = X + ca * 36 + c * 6;
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) s += XI[k] * XK[k];
                    Spp[e0 + e] -= s;
                };
                if (wv == 0) {
                    if (lane < 36) trailing_entry(lane);         // block (J+1, J+1) is the first of the trailing storage
                    ST_WSYNC();
                    if (st_chol_inv<6>(Spp + e0, lane)) { if (lane == 0) flag[0] = 1; }
                } else {
                    const int nit = (m * (m + 1) / 2 - 1) * 6;         // 2 x 3 sub-blocks of the trailing blocks after the first one
                    for (int it = t - 64; it < nit + m * 6; it += LT - 64) {
                        if (it >= nit) { trailing_entry(cntT + (it - nit)); continue; }      // rhs rows
                        const int q = 1 + it / 6, sbk = it - 6 * (q - 1), r0 = 2 * (sbk >> 1), c0 = 3 * (sbk & 1);
                        const int ij = blkIJ[e0 / 36 + q];
                        const int ia = (ij & 255) - (J + 1), ca = (ij >> 8) - (J + 1);
                        const bool dg = ia == ca;
                        if (dg && r0 + 1 < c0) continue;
                        const double *XI = X + ia * 36 + r0 * 6, *XK = X + ca * 36 + c0 * 6;
                        double s00 = 0, s01 = 0, s02 = 0, s10 = 0, s11 = 0, s12 = 0;
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            const double a0 = XI[k], a1 = XI[6 + k], b0 = XK[k], b1 = XK[6 + k], b2 = XK[12 + k];
                            s00 += a0 * b0; s01 += a0 * b1; s02 += a0 * b2;
                            s10 += a1 * b0; s11 += a1 * b1; s12 += a1 * b2;
                        }
                        double *Bq = Spp + e0 + q * 36 + r0 * 6 + c0;
                        if (!dg || r0 >= c0) Bq[0] -= s00;
                        if (!dg || r0 >= c0 + 1) Bq[1] -= s01;
                        if (!dg || r0 >= c0 + 2) Bq[2] -= s02;
                        if (!dg || r0 + 1 >= c0) Bq[6] -= s10;
                        if (!dg || r0 + 1 >= c0 + 1) Bq[7] -= s11;
                        if (!dg || r0 + 1 >= c0 + 2) Bq[8] -= s12;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        STSTAMP(6);
        if (flag[0] || attempt < d.force_retry) {
            mu *= 10.0; attempt++;
            __syncthreads();
            if (t_outer == 0) flag[0] = 0;
            __syncthreads();
            continue;
        }
        // ---- backward substitution (the forward one rode along with the factorisations) ------------------------------------
        // pose block: one wavefront, right-hand side in REGISTERS (k_build_solve_sb's routine)
        if (wv == 0) {
            ST_IDS();
            const int q6 = lane / 6, c6 = lane - 6 * q6;
            const int fA = q6, fB = 10 + q6;                    // my frame in yp0 / yp1
            const bool hasPA = q6 < 10 && fA < N, hasPB = q6 < 10 && fB < N;
            double yp0 = hasPA ? y[15 * fA + c6] : 0.0, yp1 = hasPB ? y[15 * fB + c6] : 0.0;
            for (int J = N - 1; J >= 0; J--) {
                const int b = 6 * (J % 10);
                double v[6];
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = J < 10 ? st_readlane(yp0, b + k) : st_readlane(yp1, b + k);
                const double *Lc = Spp + st_sblk(J, J, N) + c6;    // column c6 of L_JJ^-1
                double x = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) x += Lc[k * 6] * v[k];
                if (J < 10) { if (hasPA && fA == J) yp0 = x; } else { if (hasPB && fB == J) yp1 = x; }
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = J < 10 ? st_readlane(yp0, b + k) : st_readlane(yp1, b + k);
                if (hasPA && fA < J) {
                    const double *Lb = Spp + st_sblk(J, fA, N) + c6;
#pragma unroll
                    for (int k = 0; k < 6; k++) yp0 -= Lb[k * 6] * v[k];
                }
                if (N > 10 && hasPB && fB < J) {
                    const double *Lb = Spp + st_sblk(J, fB, N) + c6;
#pragma unroll
                    for (int k = 0; k < 6; k++) yp1 -= Lb[k * 6] * v[k];
                }
            }
            if (hasPA) y[15 * fA + c6] = yp0;
            if (hasPB) y[15 * fB + c6] = yp1;
        }
        __syncthreads();
#if ST_NO_YSPILL
        // chain rhs -= v_i, v_i = Y_i'^T x_pose, WITHOUT the nodes' Y' (round 5: they are not spilled): Y_p = E_p - Y_j' C_j'^T (E_p: the
        // node's own three pose x speed/bias blocks, j: its child; two children for the middle node), so
        //     v_p = L_p^-1 Y_p^T x_pose = L_p^-1 (E_p^T x_pose - C_j' v_j),
        // one pass along each chain in elimination order (wavefronts 0 / 1), then x_i = L_i^-T (z_i - v_i - C_i'^T x_parent) in reverse
        // order as before.  e_p = E_p^T x_pose for all nodes first, by all threads (an item = (speed/bias row, pose block), from the
        // prescaled init blocks in the scratch), folded in fixed order.
        {
            // every node's C_i' and L_i^-1 come back from the global scratch with ONE coalesced sweep -- into the (dead) chain buffers
            // (162 N doubles fit them up to N = 11: st_work_doubles), or, for long windows, over the pose blocks: the pose system's
            // factor is dead once x_pose is known -- then the nodes run out of LDS
            ST_IDS();
            double *stg = BIG ? Spp : work;
            double *sDi = stg, *sC = stg + 81 * N;
            double *ev = BIG ? Spp + 162 * N : Spp, *vv = ev + 9 * N, *pt = vv + 9 * N;       // e_i, v_i [9 N], the fold's parts [27 N]: over the dead pose blocks
            for (int e = t; e < 162 * N; e += LT) stg[e] = ws[e];          // gDinv | gC are contiguous in the scratch
            for (int it = t; it < 27 * N; it += LT) {
                const int o = it / 3, b3 = it - 3 * o, i = o / 9, c = o - 9 * i, pz = i - 1 + b3;
                const double *xa = y + 15 * (pz < 0 ? 0 : pz > N - 1 ? N - 1 : pz);
                const double *Ei = gYi + (size_t)i * 162 + b3 * 54 + c;               // (zeros where the pose lies outside the window)
                double ya[6];
#pragma unroll
                for (int r = 0; r < 6; r++) ya[r] = Ei[r * 9];
                double s = 0;
#pragma unroll
                for (int r = 0; r < 6; r++) s += ya[r] * xa[r];
                pt[it] = s;
            }
            __syncthreads();
            for (int o = t; o < 9 * N; o += LT) ev[o] = (pt[3 * o] + pt[3 * o + 1]) + pt[3 * o + 2];
            __syncthreads();
            const int cl = lane < 9 ? lane : 0;
            auto node_fwd = [&](int i, int c1, int c2) {            // v_i from its children's (c1, c2; -1: none); the rhs takes -v_i
                double wr = ev[9 * i + cl];
                if (c1 >= 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) wr -= sC[c1 * 81 + cl * 9 + k] * vv[9 * c1 + k];
                }
                if (c2 >= 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) wr -= sC[c2 * 81 + cl * 9 + k] * vv[9 * c2 + k];
                }
                double v = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) v += sDi[i * 81 + cl * 9 + k] * st_readlane(wr, k);      // (L^-1 is lower: the upper part is stored as zeros)
                ST_WSYNC();
                if (lane < 9) { vv[9 * i + lane] = v; y[15 * i + 6 + lane] -= v; }
                ST_WSYNC();
            };
            auto node_bwd = [&](int i, int pp) {
                double sv = y[15 * i + 6 + cl];
                if (pp >= 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) sv -= sC[i * 81 + k * 9 + cl] * y[15 * pp + 6 + k];
                }
                double x = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) x += sDi[i * 81 + k * 9 + cl] * st_readlane(sv, k);
                ST_WSYNC();
                if (lane < 9) y[15 * i + 6 + lane] = x;
                ST_WSYNC();
            };
            // (run-time N: ONE loop for both chains with per-lane selects, as in the elimination.  Written as two loops under
            //  `if (wv == 0) .. else if (wv == 1)`, or as one loop on a readfirstlane'd wavefront index, the run-time-N instantiation gave
            //  wrong v_i on gfx950 / ROCm 7.2 (scripts/st_probe.py: every N <= 11 through <false, 0>) while the compile-time-N ones,
            //  where both loops unroll to straight-line code, are right and 8 us per launch faster than the one-loop form: each
            //  instantiation gets the form that is verified for it -- tests/test_gpu_branches.py covers N = 6, 9, 11, 16, 18 against
            //  the oracle and N = 11 through both instantiations)
            if constexpr (NC != 0) {
                if (wv == 0) { for (int i = 0; i < M; i++) node_fwd(i, i - 1, -1); }
                else if (wv == 1) { for (int i = N - 1; i > M; i--) node_fwd(i, i + 1 <= N - 1 ? i + 1 : -1, -1); }
            } else if (wv < 2) {
                const int cnt = wv == 0 ? M : N - 1 - M;
                for (int k = 0; k < cnt; k++) { const int i = wv == 0 ? k : N - 1 - k; node_fwd(i, k > 0 ? (wv == 0 ? i - 1 : i + 1) : -1, -1); }
            }
            __syncthreads();
            if (wv == 0) { node_fwd(M, M - 1, M + 1 <= N - 1 ? M + 1 : -1); node_bwd(M, -1); }
            __syncthreads();
            if (wv == 0) { for (int i = M - 1; i >= 0; i--) node_bwd(i, i + 1); }
            else if (wv == 1) { for (int i = M + 1; i <= N - 1; i++) node_bwd(i, i - 1); }
        }
#else
        // chain rhs -= Y_i'^T x_pose from the global scratch (column-major per node: a thread reads contiguous rows);
        // thread = (speed/bias row, quarter of the pose blocks), folded in fixed order through the work area
        {
            ST_IDS();
            double *part = work;                                   // [36 N] (the chain buffers are dead)
            for (int tq = t; tq < 36 * N; tq += LT) {
                const int o = tq >> 2, prt = tq & 3, i = o / 9, c = o - 9 * i, lo = st_nlo(i, M), nr = st_nhi(i, M, N) - lo + 1;
                const double *Yc = gY + yo[i] + c * (6 * nr);
                double s = 0;
                for (int a = prt; a < nr; a += 4) {
                    const double *xa = y + 15 * (lo + a);
                    double yv[6];
#pragma unroll
                    for (int r = 0; r < 6; r++) yv[r] = Yc[a * 6 + r];
#pragma unroll
                    for (int r = 0; r < 6; r++) s += yv[r] * xa[r];
                }
                part[tq] = s;
            }
            __syncthreads();
            for (int tq = t; tq < 9 * N; tq += LT) {
                const int i = tq / 9, c = tq - 9 * i;
                y[15 * i + 6 + c] -= (part[4 * tq] + part[4 * tq + 1]) + (part[4 * tq + 2] + part[4 * tq + 3]);
            }
        }
        __syncthreads();
        // chains, reverse elimination order: x_i = L_i^-T (z_i - C_i'^T x_parent).  Wavefront 0: M, then the forward chain downwards;
        // wavefront 1 (after M): the backward chain upwards.
        {
            // every node's C_i' and L_i^-1 come back from the global scratch with ONE coalesced sweep -- into the (dead) chain buffers
            // (162 N doubles fit them up to N = 11: st_work_doubles), or, for long windows, over the pose blocks: the pose system's
            // factor is dead once x_pose is known -- then the nodes run out of LDS
            ST_IDS();
            double *stg = BIG ? Spp : work;
            double *sDi = stg, *sC = stg + 81 * N;
            for (int e = t; e < 162 * N; e += LT) stg[e] = ws[e];          // gDinv | gC are contiguous in the scratch
            __syncthreads();
            const int cl = lane < 9 ? lane : 0;
            auto node_bwd = [&](int i, int pp) {
                double sv = y[15 * i + 6 + cl];
                if (pp >= 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) sv -= sC[i * 81 + k * 9 + cl] * y[15 * pp + 6 + k];
                }
                double x = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) x += sDi[i * 81 + k * 9 + cl] * st_readlane(sv, k);
                ST_WSYNC();
                if (lane < 9) y[15 * i + 6 + lane] = x;
                ST_WSYNC();
            };
            if (wv == 0) node_bwd(M, -1);
            __syncthreads();
            if (wv == 0) { for (int i = M - 1; i >= 0; i--) node_bwd(i, i + 1); }
            else if (wv == 1) { for (int i = M + 1; i <= N - 1; i++) node_bwd(i, i - 1); }
        }
#endif
        __syncthreads();
        STSTAMP(7);
        break;
    }
    const int t = t_outer, lane = t & 63;
    if (!ls_fail) {
        for (int e = t; e < n; e += LT) {
            d.zp[(size_t)w * n + e] = sc[e] * y[e];
            d.gn_p[(size_t)w * n + e] = -D[e] * y[e];
        }
    }
    double cost_w = 0.0;
    {
        double m = gmax_l;
        for (int i = t; i < N; i += LT) {
            const double *x = d.pose + ((size_t)w * N + i) * 7;
            double ng[6], xp[7];
            for (int k = 0; k < 6; k++) ng[k] = -g[15 * i + k];
            pose_plus(x, ng, xp);
            for (int k = 0; k < 7; k++) m = fmax(m, fabs(x[k] - xp[k]));
            for (int k = 0; k < 9; k++) m = fmax(m, fabs(g[15 * i + 6 + k]));
        }
        __syncthreads();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { cpart += __shfl_xor(cpart, off); m = fmax(m, __shfl_xor(m, off)); }
        if (lane == 0) { red[wv] = cpart; red[8 + wv] = m; }
        __syncthreads();
        cost_w = BIG ? ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])) : (red[0] + red[1]) + (red[2] + red[3]);
        if (t == 0) {
            double mm = red[8];
            for (int k = 1; k < NW; k++) mm = fmax(mm, red[8 + k]);
            d.cost[w] = cost_w;
            red[16] = mm;
        }
        __syncthreads();
    }
    if (t == 0) {
        atomicAdd(&d.act[iteration], 1);
        // (mu already at max_mu on entry: nothing was assembled, x has not moved, the gradient is the previous one)
        if (assembled) st.gmax = red[16];
        st.mu = mu;
        st.ls_fail = ls_fail;
        st.need_linearize = 0;
        st.fresh = 1;
        st.x_cost = cost_w;
        if (iteration == 0) {
            st.initial_cost = cost_w;
            d.trace_cost[(size_t)w * ISV_MAX_TRACE] = cost_w;
            d.trace_radius[(size_t)w * ISV_MAX_TRACE] = st.radius;
        }
        if (st.gmax <= 1e-10) st.termination = ISV_TERM_GRADIENT_TOL;
    }
#ifdef ISV_STAMP
    STSTAMP(8);
    if (t == 0) for (int k = 0; k < 9; k++) d.dbg[(size_t)w * 64 + k] += (double)st_acc[k];
#endif
}

template __global__ void k_build_solve_st<false, 0>(DevBatch);
template __global__ void k_build_solve_st<false, 11>(DevBatch);
template __global__ void k_build_solve_st<true, 0>(DevBatch);
template __global__ void k_build_solve_st<true, 18>(DevBatch);      // the reference's own window length (ALL_BUF_SIZE 18)
