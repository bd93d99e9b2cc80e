// isv_build_solve_sb.hip -- k_build_solve_sb: structure-aware reduced-system solve (the DENSE_SCHUR
// linear solve of one trust-region iteration, after the landmarks were eliminated by k_sweep /
// k_rank1_mfma).  Same mathematics and outputs as k_build_solve (isv_build_solve.hip header), but the
// 15N x 15N reduced camera matrix is never formed densely:
//
//   * unknowns are split into the 6-dof pose blocks (dense among themselves: every frame pair shares
//     landmarks) and the 9-dof speed/bias blocks, which only couple along the IMU chain
//     (sb_i -- sb_i+1, sb_i -- pose_i-1..i+1).  The speed/bias blocks are eliminated FIRST, from both
//     ends of the window towards the middle frame M = N/2 (two independent chains, one wavefront each,
//     no block barriers), which bounds the fill to sb_i x pose[0..i+1] (i < M), sb_i x pose[i-1..N-1]
//     (i > M), sb_M x all poses.  The remaining 6N x 6N pose system gets a blocked Cholesky (6x6
//     blocks, diagonal blocks factored and inverted in registers by one wavefront).
//   * LDS per window: 6x6 pose blocks (packed lower block triangle, same layout as Tvis) + 9x9
//     chain blocks + the pose/speed-bias fill = 68.6 KB for N = 11 instead of 118.8 KB for the dense
//     lower triangle, so TWO windows are resident per CU (160 KB LDS) and hide each other's
//     barrier / latency stalls; ~4x fewer flops than the dense factorisation as well.
//   * triangular solves run on one wavefront with wave-level LDS ordering only (no block barriers in
//     the 2 x (N + N) dependent steps).
//   * a failed factorisation (mu too small) retries with mu x 10 like ceres' dogleg strategy
//     (dogleg_strategy.cc ComputeGaussNewtonStep); the landmark part is then corrected in place:
//     T' = T - sum_l (c_l(mu') - c_l(mu)) w_l w_l^T from the w vectors.
//
// Outputs (unchanged): zp, gn_p, up, diag_p, grad_p, scale_p, st.qT / gmax / mu / ls_fail / flags.
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"
#include "isv_lin_gram.h"
#include "isv_rank1.h"
#include "isv_dogleg.h"

#ifdef ISV_STAMP
#define STAMP(k) do { if (t == 0) { unsigned long long now_ = wall_clock64(); d.dbg[(size_t)w * 64 + (k) + (MODE == 1 ? 32 : 0)] += (double)(now_ - t_last); t_last = now_; } } while (0)     // (the chain kernel of the split solve: slots 32..37)
#else
#define STAMP(k) do {} while (0)
#endif
#define LS 512                     // threads (8 wavefronts); two workgroups per CU
__host__ __device__ inline int sb_nred(int N, int prior_H_sz) {   // doubles in the shared reduction / staging buffer
    int m = LS;
    if (prior_H_sz > m) m = prior_H_sz;
    if (36 * N > m) m = 36 * N;
    if (30 * N > m) m = 30 * N;
    return (m + 1) & ~1;
}
// a fresh, opaque copy of the thread id per phase: index arithmetic is then recomputed where it is used instead of being
// shared across phases by CSE -- the shared values live through the whole kernel and spill under the 128-VGPR cap, and
// every reload is a scratch (global memory) round trip on the serial path
#define PHASE_IDS() int t = t_outer; asm volatile("" : "+v"(t)); const int lane = t & 63; (void)lane
#define RCH 32                     // landmarks per staged chunk of the retry correction

DEV int sblk(int I, int J, int N) { return (J * N - J * (J - 1) / 2 + (I - J)) * 36; }   // I >= J
// the non-visual pose blocks are block-tridiagonal (IMU factors and relative-pose priors couple neighbours, the other priors one pose):
// the chain kernel of the split solve (MODE 1) keeps them as N diagonal blocks followed by N - 1 sub-diagonal ones
DEV int sbt(int I, int J, int N) { return (I == J ? I : N + J) * 36; }
DEV int pairidx2(int a, int b) { return a * (a + 1) / 2 + b; }      // a >= b
DEV int nlo(int i, int M) { return i > M ? i - 1 : 0; }
DEV int nhi(int i, int M, int N) { return i < M ? i + 1 : N - 1; }
DEV int npar(int i, int M) { return i < M ? i + 1 : (i > M ? i - 1 : -1); }

DEV double readlane_d2(double v, int lane) {
    union { double d; int i[2]; } u; u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.d;
}
DEV double rsqrt_nr2(double x) {   // 1/sqrt(x) to ~1 ulp: hardware estimate + two Newton steps
    double r = __builtin_amdgcn_rsq(x);
    r = r * (1.5 - 0.5 * x * r * r);
    r = r * (1.5 - 0.5 * x * r * r);
    return r;
}
// wave-level ordering of LDS traffic (LDS executes one wavefront's accesses in issue order)
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Factor the BS x BS SPD block at A (row-major, leading dimension BS, lower part valid) in registers and
// overwrite it with the INVERSE of its Cholesky factor (lower, upper part zeroed).  One wavefront.
template <int BS>
DEV bool chol_inv_block(double *A, int lane) {
    double row[BS], dinv[BS], x[BS];
#pragma unroll
    for (int k = 0; k < BS; k++) row[k] = (lane < BS) ? A[lane * BS + k] : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < BS; j++) {
        double s = row[j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= row[k] * readlane_d2(row[k], j);
        const double sj = readlane_d2(s, j);               // pivot
        if (!(sj > 0.0)) bad = true;
        dinv[j] = rsqrt_nr2(sj);                            // 1 / L_jj (wave-uniform)
        row[j] = (lane == j) ? sj * dinv[j] : s * dinv[j];
    }
#pragma unroll
    for (int i = 0; i < BS; i++) {                          // lane c solves L x = e_c
        double s = (lane == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s -= readlane_d2(row[k], i) * x[k];
        x[i] = s * dinv[i];
    }
    WSYNC();
    if (lane < BS) {
#pragma unroll
        for (int k = 0; k < BS; k++) A[k * BS + lane] = x[k];   // x[k] = Linv[k][lane], zero for k < lane
    }
    WSYNC();
    return bad;
}

// BIG = false: N <= 11, two workgroups per CU (<= 128 VGPRs), register prefetch of the assembly, chain back-
//               substitution with all nodes of a chain in one register;
// BIG = true:  N <= 20 (one workgroup per CU: the system needs up to ~160 KB of LDS), plain loops instead of the
//               fixed-size register stages.
// NC: the window length as a compile-time constant (0 = read d.N): the LDS layout, every loop bound and the index
//     divisions fold, which is worth registers and integer work in every phase; instantiated for the benchmark's 11 and the
//     reference's 18 frames.
// MODE (round 5): 0 = the whole solve in one launch (above).  Handles whose batches leave CUs idle (max_batch <= CUs) split it:
//   MODE 1 (k "chain", SIDE stream, after the IMU / prior kernels, BESIDE k_lin_gram / k_rank1_mfma): everything that does not depend on
//          the reprojection factors -- the speed/bias blocks couple only through IMU factors and the Linear9 prior, and the landmark
//          elimination fills pose x pose blocks only -- i.e. the assembly of the non-visual blocks, the Jacobi scaling of the speed/bias
//          columns, both elimination chains, node M, Y Y^T and Y z.  The POSE rows' Jacobi scales depend on the visual diagonal (first
//          iteration), so the pose rows of Y stay unscaled here: (S_p Y) L^-T = S_p (Y L^-T), row scaling commutes with every chain operation.
//          Leaves in d.cs_ws: S_nv (non-visual pose blocks) | S_nv - Y'Y'^T | L_i^-1, C_i', Y_i' | g, hdiag, y (z_i; pose rows: -Y' z) | t_Y | q_ss | flag.
//   MODE 2 (k "pose", main stream, after k_rank1_mfma and the join): T_pp = s s (V + S_nv - Y'Y'^T) + mu D^2, pose Cholesky, the back-substitutions
//          and the outputs.  A failed factorisation on either side (mu retry) falls back to the one-launch path inside this kernel.
// The split sums the same products in a different order than MODE 0 (the row scaling is applied after the chain instead of before): results agree to
// rounding; which one a handle runs is decided by its max_batch (SolverHost::chain_split), never by the uploaded batch.
__host__ __device__ inline size_t sb_cs_doubles(int N) {
    const int M = N / 2;
    size_t ytot = 0;
    for (int i = 0; i < N; i++) ytot += (size_t)((i < M ? i + 1 : N - 1) - (i > M ? i - 1 : 0) + 1) * 54;
    return (2 * (size_t)N - 1) * 36 + (size_t)N * (N + 1) / 2 * 36 + 162 * (size_t)N + ytot + 3 * 15 * (size_t)N + 6 * (size_t)N + 8;
}
template <bool BIG, int NC, int MODE>
DEV void build_solve_sb_body(const DevBatch &d, double *lds) {
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
    const int N = NC ? NC : d.N, n = 15 * N, M = N / 2, n6 = 6 * N, nS = N * (N + 1) / 2 * 36;
    double *p = lds;
    double *g = p; p += n;
    double *bs = p; p += n;
    double *hdiag = p; p += n;
    // lifetimes: bs is consumed when y = s (g + bs) is formed, hdiag when the LM diagonal D is; u (Cauchy
    // direction) and the Jacobi scales only live between the scaling loop and the u^T T u reduction: shared storage
    double *D = hdiag;
    double *y = bs;
    const int nred = sb_nred(N, d.prior_H_sz);     // reductions; also stages the prior blocks and the chain gather partials
    double *red = p; p += nred;
    double *u = red;                         // (2 n <= nred)
    double *sc = red + n;                    // Jacobi scales while the blocks are scaled; the outputs re-read them from HBM
    int *yo = (int *)p; p += 16;             // yo[0..N]: offsets of the fill blocks of each chain node (N <= 31)
    int *skipL = (int *)p; p += 16;          // imu_skip flags of this window
    int *flag = (int *)p; p += 2;
    // (I | J << 8) of packed pose block q: one LDS lookup instead of a search loop per matrix entry
    unsigned short *blkIJ = (unsigned short *)p; p += (N * (N + 1) / 2 + 3) / 4 + 1;
    unsigned char *triAB = (unsigned char *)p; p += 20;   // (row, col) of triangular pair index e < 78 (12 x 12 lower): triAB[2e], triAB[2e+1]
    unsigned char *yNode = (unsigned char *)p; p += 24;   // chain node of fill block b (ytot / 54 <= 139 blocks for N <= 20)
    double *Spp = p; p += nS;                // pose-pose, packed lower block triangle of 6x6 blocks (Tvis layout)
    double *Dss = p; p += N * 81;            // speed/bias diagonal blocks -> inverse Cholesky factors
    double *Css = p; p += N * 81;            // coupling of node i to its parent: rows parent, cols i
    double *Ysb = p;                         // pose x speed/bias blocks incl. fill: node i, poses nlo..nhi, [6][9]

    if (t == 0) {
        int o = 0;
        for (int i = 0; i < N; i++) { yo[i] = o; o += (nhi(i, M, N) - nlo(i, M) + 1) * 54; }
        yo[N] = o; flag[0] = 0;
    }
    if (t < N - 1) skipL[t] = d.imu_skip[(size_t)w * (N - 1) + t];
    for (int q = t; q < N * (N + 1) / 2; q += LS) {
        int ca = 0;
        while (ca + 1 < N && (ca + 1) * N - (ca + 1) * ca / 2 <= q) ca++;
        blkIJ[q] = (unsigned short)((ca + (q - (ca * N - ca * (ca - 1) / 2))) | (ca << 8));
    }
    if (t < 78) { int a = 0; while ((a + 1) * (a + 2) / 2 <= t) a++; triAB[2 * t] = (unsigned char)a; triAB[2 * t + 1] = (unsigned char)(t - a * (a + 1) / 2); }
    __syncthreads();
    for (int b = t; b < yo[N] / 54; b += LS) { int i = 0; while (yo[i + 1] <= 54 * b) i++; yNode[b] = (unsigned char)i; }
    __syncthreads();
    const int ytot = yo[N];
    // cost of the window at x = sum over its residual blocks (fixed-shape strided partials + tree below)
    double cpart = 0;
    if (MODE != 1) {
        const int f0w = d.f_off[w], f1w = d.f_off[w + 1];
        // (four trips' loads in flight, masked adds in the same order: a 30 000-factor window waited 59 memory latencies here)
        for (int f0 = f0w + t; f0 < f1w; f0 += 4 * LS) {
            double c4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int f = f0 + LS * u < f1w ? f0 + LS * u : f1w - 1; c4[u] = d.fcost[f]; }
#pragma unroll
            for (int u = 0; u < 4; u++) if (f0 + LS * u < f1w) cpart += c4[u];
        }
        for (int i = t; i < N - 1; i += LS) cpart += d.imu_cost[(size_t)w * (N - 1) + i];
        for (int i = t; i < d.n_prior_slots; i += LS) cpart += d.prior_cost[(size_t)w * d.n_prior_slots + i];
    }
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64();
#endif
    const int iteration = st.iteration;
    double mu = st.mu;
    int ls_fail = 0, attempt = 0, assembled = 0;
    double gmax_l = 0.0;
    const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    const double *V = d.Tvis + (size_t)w * d.tvis_sz;

    // decode a packed Spp entry index into (I, J, r, c)
    auto spp_decode = [&](int e, int &I, int &J, int &r, int &c) {
        const int q = e / 36, rc = e - 36 * q;
        r = rc / 6; c = rc - 6 * r;
        const int ij = blkIJ[q];
        I = ij & 255; J = ij >> 8;
    };

    const int t_outer = t;
    // the split solve's hand-over (MODE 1 writes, MODE 2 reads): see the kernel's header comment
    const int tailsz = 162 * N + ytot;
    double *CS = MODE ? d.cs_ws + (size_t)w * sb_cs_doubles(N) : nullptr;
    double *CS_snv = CS, *CS_dlt = CS + (2 * N - 1) * 36, *CS_tail = CS_dlt + nS, *CS_g = CS_tail + tailsz, *CS_hd = CS_g + n, *CS_y = CS_hd + n, *CS_tY = CS_y + n, *CS_sc = CS_tY + n6;
    if (MODE == 1 && !(mu < 1.0)) return;               // (MODE 2 reports the failure)
    for (;;) {
        if (!(mu < 1.0)) { ls_fail = 1; break; }
        assembled = 1;
        // MODE 2, first attempt: the chains were eliminated by the MODE 1 launch; a failure there (its flag) is this attempt's failure
        const bool pre = MODE == 2 && attempt == 0;
        if (pre && CS_sc[1] != 0.0) { mu *= 10.0; attempt++; continue; }
        if (pre) {
            PHASE_IDS();
            // ---- (1) vectors: visual diagonal / gradient / reduced rhs + the non-visual parts and the chain's right-hand sides
            double vh = 0, vg = 0, vy = 0, vs = 0, vd = 0, vT0 = 0, vT1 = 0, vT2 = 0, vtY = 0;
            const bool sbrow = t < n && (t % 15) >= 6;
            if (t < n) {
                vh = CS_hd[t]; vg = CS_g[t]; vy = CS_y[t];
                if (sbrow || iteration != 0) vs = d.scale_p[(size_t)w * n + t];
                if (sbrow) vd = d.diag_p[(size_t)w * n + t];
            }
            if (t < n6) { vT0 = V[nS + t]; vT1 = V[nS + n6 + t]; vT2 = V[nS + 2 * n6 + t]; vtY = CS_tY[t]; }
            const double q_ss = CS_sc[0];
            for (int l = l0 + t; l < l1; l += LS) gmax_l = fmax(gmax_l, fabs(d.lmG[l]));
            for (int e = t; e < n; e += LS) bs[e] = 0.0;
            if (t < n) { g[t] = vg; hdiag[t] = vh; }
            __syncthreads();
            if (t < n6) {
                const int fa = t / 6, r = t - 6 * fa;
                hdiag[15 * fa + r] += vT0; g[15 * fa + r] += vT1; bs[15 * fa + r] = vT2;
            }
            __syncthreads();
            // ---- (2) Jacobi scaling of the pose rows (the speed/bias rows were scaled by the chain kernel)
            if (t < n) {
                const int e = t;
                if (sbrow) { sc[e] = vs; u[e] = 0.0; D[e] = vd; y[e] = vy; }     // (D aliases hdiag, y aliases bs: own entry only)
                else {
                    double s_;
                    if (iteration == 0) { s_ = 1.0 / (1.0 + sqrt(hdiag[e])); d.scale_p[(size_t)w * n + e] = s_; }
                    else s_ = vs;
                    const double D2 = fmin(fmax(s_ * s_ * hdiag[e], 1e-6), 1e32);
                    const double ge = g[e], be = bs[e];
                    sc[e] = s_; D[e] = sqrt(D2);
                    d.diag_p[(size_t)w * n + e] = D[e];
                    d.grad_p[(size_t)w * n + e] = s_ * ge / D[e];
                    u[e] = s_ * s_ * ge / D2;
                    d.up[(size_t)w * n + e] = u[e];
                    y[e] = s_ * (ge + be) + s_ * vy;
                }
            }
            __syncthreads();
            // ---- (3) the pose system and the chain factors; u^T T u over the unscaled V + S_nv
            double accq = 0;
            constexpr int KD = 8;                                  // loads in flight per array and trip
            for (int e0 = t; e0 < nS; e0 += KD * LS) {
                double a8[KD], b8[KD], c8[KD];
#pragma unroll
                for (int k = 0; k < KD; k++) {
                    const int e = e0 + k * LS < nS ? e0 + k * LS : nS - 1;
                    const int ij = blkIJ[e / 36], I = ij & 255, J = ij >> 8;
                    a8[k] = V[e]; c8[k] = CS_dlt[e];
                    b8[k] = CS_snv[(I - J <= 1 ? sbt(I, J, N) : 0) + (e % 36)];       // (clamped: masked below)
                }
#pragma unroll
                for (int k = 0; k < KD; k++) {
                    const int e = e0 + k * LS;
                    if (e < nS) {
                        int I, J, r, c;
                        spp_decode(e, I, J, r, c);
                        const int gi = 15 * I + r, gj = 15 * J + c;
                        const double tpp = a8[k] + (I - J <= 1 ? b8[k] : 0.0);           // unscaled T_pp entry: visual + non-visual
                        double sv = (tpp + c8[k]) * sc[gi] * sc[gj];
                        if (!(I == J && r < c)) {
                            accq += (gi == gj ? 1.0 : 2.0) * tpp * u[gi] * u[gj];
                            if (gi == gj) sv += mu * D[gi] * D[gi];
                        }
                        Spp[e] = sv;
                    }
                }
            }
            if (t < n6) accq += 2.0 * u[15 * (t / 6) + t % 6] * vtY;
            for (int e0 = t; e0 < tailsz; e0 += KD * LS) {
                double a8[KD];
#pragma unroll
                for (int k = 0; k < KD; k++) { const int e = e0 + k * LS < tailsz ? e0 + k * LS : tailsz - 1; a8[k] = CS_tail[e]; }
#pragma unroll
                for (int k = 0; k < KD; k++) {
                    const int e = e0 + k * LS;
                    if (e < tailsz) {
                        double v = a8[k];
                        if (e >= 162 * N) {                     // a row of Y_i': the pose row's Jacobi scale
                            const int q = e - 162 * N, i = yNode[q / 54], qq = q - yo[i], ai = qq / 54, r = (qq - 54 * ai) / 9;
                            v *= sc[15 * (nlo(i, M) + ai) + r];
                        }
                        Dss[e] = v;
                    }
                }
            }
            __syncthreads();                               // every thread is done with u (it lives in red)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) accq += __shfl_xor(accq, off);
            if (lane == 0) red[wv] = accq;
            __syncthreads();
            if (t == 0) st.qT = (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) + q_ss;
            __syncthreads();
            STAMP(0);
        } else {
        // the retry loop almost never iterates: keep the compiler from hoisting per-thread index math out of
        // it (the hoisted values stay live across the whole body and spill)
        int t = t_outer;
        asm volatile("" : "+v"(t));
        // ---- issue every global load of the assembly up front (one memory latency instead of five) ----
        // (round 4, measured and dropped: the IMU fetches below as unconditional, clamped, mask-multiplied loads -- the form that pays in
        //  k_build_solve_st and k_imu_weight -- made THIS kernel slower: one 18-frame window 2.64 -> 2.78 ms, 256 of them 3.28 -> 3.38 ms)
        const double *H = d.imu_H + (size_t)w * (N - 1) * ISV_IMU_H;
        const double *PH = d.prior_H + (size_t)w * d.prior_H_sz;
        // (BIG: one workgroup per CU = 256 VGPRs per lane, room for the IMU blocks of up to 20 frames: 2400 / 512 and 4275 / 512 entries per thread)
        constexpr int KF = BIG ? 5 : 3, KX = BIG ? 9 : 5;
        double vS[5], vT[3] = {0, 0, 0}, vF[KF], vX[KX], vG = 0, vP[2] = {0, 0};
        if (!BIG && MODE != 1) {
#pragma unroll
            for (int k = 0; k < 5; k++) { const int e = t + k * LS; vS[k] = e < nS ? V[e] : 0.0; }
        }
        if (MODE != 1 && t < n6) { vT[0] = V[nS + t]; vT[1] = V[nS + n6 + t]; vT[2] = V[nS + 2 * n6 + t]; }
        if (t < d.prior_H_sz) vP[0] = PH[t];
        if (t + LS < d.prior_H_sz) vP[1] = PH[t + LS];
        // IMU factors (precomputed J^T J, local order pose_i sb_i pose_j sb_j)
        auto imu_frame_fetch = [&](int e) -> double {       // per frame: pose diag (21), sb diag (45), pose x sb (54)
            const int I = e / 120;
            int q = e - 120 * I;
            const bool hasA = I >= 1 && !skipL[I - 1], hasB = I <= N - 2 && !skipL[I];
            const double *HA = H + (size_t)(I - 1) * ISV_IMU_H, *HB = H + (size_t)I * ISV_IMU_H;
            int ia, ib;
            if (q < 21) {
                const int r = triAB[2 * q], c = triAB[2 * q + 1];
                ia = pairidx2(15 + r, 15 + c); ib = pairidx2(r, c);
            } else if (q < 66) {
                q -= 21;
                const int r = triAB[2 * q], c = triAB[2 * q + 1];
                ia = pairidx2(21 + r, 21 + c); ib = pairidx2(6 + r, 6 + c);
            } else {
                q -= 66;
                const int r = q / 9, c = q - 9 * r;         // pose row r, speed/bias column c of frame I
                ia = pairidx2(21 + c, 15 + r); ib = pairidx2(6 + c, r);
            }
            double v = 0;
            if (hasA) v += HA[ia];
            if (hasB) v += HB[ib];
            return v;
        };
        auto imu_frame_apply = [&](int e, double v) {
            const int I = e / 120;
            int q = e - 120 * I;
            if (q < 21) {
                const int r = triAB[2 * q], c = triAB[2 * q + 1];
                Spp[(MODE == 1 ? sbt(I, I, N) : sblk(I, I, N)) + r * 6 + c] += v;
                if (r == c) hdiag[15 * I + r] += v;
            } else if (q < 66) {
                q -= 21;
                const int r = triAB[2 * q], c = triAB[2 * q + 1];
                Dss[I * 81 + r * 9 + c] = v;
                if (r == c) hdiag[15 * I + 6 + r] = v;
            } else {
                Ysb[yo[I] + (I - nlo(I, M)) * 54 + (q - 66)] = v;
            }
        };
        auto imu_pair_fetch = [&](int e) -> double {        // per factor: the blocks between frames I and I + 1
            const int I = e / 225;
            int q = e - 225 * I;
            if (skipL[I]) return 0.0;
            const double *HB = H + (size_t)I * ISV_IMU_H;
            int ib;
            if (q < 36) { const int r = q / 6, c = q - 6 * r; ib = pairidx2(15 + r, c); }
            else if (q < 90) { q -= 36; const int r = q / 9, c = q - 9 * r; ib = pairidx2(15 + r, 6 + c); }      // pose_{I+1} x sb_I
            else if (q < 144) { q -= 90; const int r = q / 9, c = q - 9 * r; ib = pairidx2(21 + c, r); }          // pose_I x sb_{I+1}
            else { q -= 144; const int r = q / 9, c = q - 9 * r; ib = pairidx2(21 + r, 6 + c); }                  // sb_{I+1} (r) x sb_I (c)
            return HB[ib];
        };
        auto imu_pair_apply = [&](int e, double v) {
            const int I = e / 225;
            int q = e - 225 * I;
            if (q < 36) Spp[(MODE == 1 ? sbt(I + 1, I, N) : sblk(I + 1, I, N)) + q] += v;
            else if (q < 90) Ysb[yo[I] + (I + 1 - nlo(I, M)) * 54 + (q - 36)] = v;
            else if (q < 144) Ysb[yo[I + 1] + (I - nlo(I + 1, M)) * 54 + (q - 90)] = v;
            else {
                q -= 144;
                const int r = q / 9, c = q - 9 * r;         // rows = parent, cols = child
                if (I < M) Css[I * 81 + r * 9 + c] = v; else Css[(I + 1) * 81 + c * 9 + r] = v;
            }
        };
        if (t < n) {
            const int I = t / 15, r = t - 15 * I;
            if (I >= 1 && !skipL[I - 1]) vG += H[(size_t)(I - 1) * ISV_IMU_H + 465 + 15 + r];
            if (I <= N - 2 && !skipL[I]) vG += H[(size_t)I * ISV_IMU_H + 465 + r];
        }
        // ---- reprojection part from k_sweep / k_rank1_mfma (same packed layout) ----------------------
        for (int e = t; e < n; e += LS) { g[e] = 0.0; bs[e] = 0.0; hdiag[e] = 0.0; }
        if (MODE == 1) { for (int e = t; e < (2 * N - 1) * 36; e += LS) Spp[e] = 0.0; }        // (S_nv accumulates here, block-tridiagonal: sbt())
        else if (!BIG) {
#pragma unroll
            for (int k = 0; k < 5; k++) { const int e = t + k * LS; if (e < nS) Spp[e] = vS[k]; }
        }
        // second register stage (after the first was stored: both at once cost spills under the 128-VGPR cap of two
        // workgroups per CU): the IMU blocks, in flight across the barrier and the zeroing below (BIG: across the copy of
        // the pose blocks as well)
#pragma unroll
        for (int k = 0; k < KF; k++) { const int e = t + k * LS; vF[k] = e < N * 120 ? imu_frame_fetch(e) : 0.0; }
#pragma unroll
        for (int k = 0; k < KX; k++) { const int e = t + k * LS; vX[k] = e < (N - 1) * 225 ? imu_pair_fetch(e) : 0.0; }
        if (BIG && MODE != 1) {
            for (int e0 = t; e0 < nS; e0 += 4 * LS) {       // four loads in flight per trip
                double v4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int e = e0 + k * LS; v4[k] = e < nS ? V[e] : 0.0; }
#pragma unroll
                for (int k = 0; k < 4; k++) { const int e = e0 + k * LS; if (e < nS) Spp[e] = v4[k]; }
            }
        }
        __syncthreads();
        if (MODE != 1 && t < n6) {
            const int fa = t / 6, r = t - 6 * fa;
            hdiag[15 * fa + r] = vT[0]; g[15 * fa + r] = vT[1]; bs[15 * fa + r] = vT[2];
        }
        if (MODE == 1) {
        } else if (attempt == 0) {
            for (int l = l0 + t; l < l1; l += LS) gmax_l = fmax(gmax_l, fabs(d.lmG[l]));
        } else {
            // ---- mu retry: T -= sum_l (c_l(mu) - c_l(mu0)) w_l w_l^T, bs -= sum_l dc_l g_l w_l --------
            double *wS = Dss;                               // [RCH][n6 + 1] staging (Dss/Css/Ysb are rebuilt below)
            double *dC = red, *dG = red + RCH;
            const int wld = n6 + 1;
            for (int lb = l0; lb < l1; lb += RCH) {
                const int cnt = (l1 - lb) < RCH ? (l1 - lb) : RCH;
                __syncthreads();
                for (int e = t; e < cnt * n6; e += LS) {         // expand the packed w vectors to dense rows
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
                for (int e = t; e < nS; e += LS) {               // owner-computes: entry e of the packed pose blocks
                    int I, J, r, c;
                    spp_decode(e, I, J, r, c);
                    const int ra = 6 * I + r, cb = 6 * J + c;
                    double acc = 0;
                    for (int l = 0; l < cnt; l++) acc += dC[l] * wS[l * wld + ra] * wS[l * wld + cb];
                    Spp[e] -= acc;
                }
                if (t < n6) {
                    double accb = 0;
                    for (int l = 0; l < cnt; l++) accb += dG[l] * wS[l * wld + t];
                    bs[15 * (t / 6) + t % 6] -= accb;
                }
            }
            __syncthreads();
        }
        for (int e = t; e < 162 * N + ytot; e += LS) Dss[e] = 0.0;      // Dss, Css, Ysb are contiguous
        red[t] = vP[0];                                                  // prior blocks staged in LDS
        if (t + LS < nred) red[t + LS] = vP[1];
        __syncthreads();
        STAMP(0);
#pragma unroll
        for (int k = 0; k < KF; k++) { const int e = t + k * LS; if (e < N * 120) imu_frame_apply(e, vF[k]); }
#pragma unroll
        for (int k = 0; k < KX; k++) { const int e = t + k * LS; if (e < (N - 1) * 225 && !skipL[e / 225]) imu_pair_apply(e, vX[k]); }
        if (t < n) g[t] += vG;
        __syncthreads();
        STAMP(1);
        if (d.est_ex && t < 9) {                           // the pseudo-frame's speed/bias block is a dummy: unit diagonal, no coupling, zero gradient
            Dss[(N - 1) * 81 + t * 9 + t] = 1.0;           // (after the barrier: the IMU gather above stores zeros there)
            hdiag[15 * (N - 1) + 6 + t] = 1.0;
        }
        // ---- prior factors (precomputed J^T J, staged in red[]) ---------------------------------------
        // Three conflict-free phases instead of a barrier per factor: factors of one phase touch disjoint entries
        //   A: Linear9 + the relative-pose factors (k, k+1) with k even     B: SE3 prior (pose 0) + those with k odd
        //   C: the roll/pitch factors, entry e of EVERY factor by the same thread (two factors on one pose stay ordered)
        {
            auto prior_entry = [&](int q, int e) {
                int ncol, off, c0, c1 = 0;
                if (q == 0) { ncol = 6; off = PH_SE3; c0 = 0; }
                else if (q == 1) { ncol = 9; off = PH_LIN9; c0 = 15 * (d.Nvo - 1) + 6; }
                else if (q < 1 + d.Nvo) { const int k = q - 2; ncol = 12; off = PH_REL0 + PH_REL_SZ * k; c0 = 15 * k; c1 = 15 * (k + 1); }
                else { const int m = q - 1 - d.Nvo; ncol = 6; off = PH_REL0 + PH_REL_SZ * (d.Nvo - 1) + PH_RP_SZ * m; c0 = 15 * d.rollpitch[(size_t)w * d.max_rp + m].index; }
                const int np2 = ncol * (ncol + 1) / 2;
                if (e >= np2 + ncol) return;
                const double v = red[off + e];
                if (e < np2) {
                    const int aa = triAB[2 * e], bb = triAB[2 * e + 1];
                    const int ga = (aa < 6 || ncol != 12) ? c0 + aa : c1 + aa - 6;
                    const int gb = (bb < 6 || ncol != 12) ? c0 + bb : c1 + bb - 6;
                    const int Ia = ga / 15, ra = ga - 15 * Ia, Ib = gb / 15, rb = gb - 15 * Ib;
                    if (ra < 6) Spp[(MODE == 1 ? sbt(Ia, Ib, N) : sblk(Ia, Ib, N)) + ra * 6 + rb] += v;
                    else Dss[Ia * 81 + (ra - 6) * 9 + (rb - 6)] += v;
                    if (aa == bb) hdiag[ga] += v;
                } else {
                    const int aa = e - np2;
                    const int ga = (aa < 6 || ncol != 12) ? c0 + aa : c1 + aa - 6;
                    g[ga] += v;
                }
            };
            const int nrel = d.Nvo - 1, nrp = d.n_rp[w];
            // phase A: slot 0 = Linear9 (54 entries), slots 1.. = relative-pose factors 0, 2, 4, .. (90 entries each)
            for (int e = t; e < 90 * (1 + (nrel + 1) / 2); e += LS) { const int sl = e / 90; prior_entry(sl == 0 ? 1 : 2 + 2 * (sl - 1), e - 90 * sl); }
            __syncthreads();
            // phase B: slot 0 = SE3 prior (27 entries), slots 1.. = relative-pose factors 1, 3, ..
            for (int e = t; e < 90 * (1 + nrel / 2); e += LS) { const int sl = e / 90; prior_entry(sl == 0 ? 0 : 2 + 2 * (sl - 1) + 1, e - 90 * sl); }
            __syncthreads();
            if (t < 27) for (int m = 0; m < nrp; m++) prior_entry(1 + d.Nvo + m, t);
            __syncthreads();
        }
        STAMP(2);
        // ---- Jacobi scaling, LM diagonal, Cauchy data -------------------------------------------------
        { PHASE_IDS();
        if (MODE == 1) {
            // the non-visual pose blocks, gradient and diagonal as assembled (unscaled): the pose kernel adds the visual part
            for (int e = t; e < (2 * N - 1) * 36; e += LS) CS_snv[e] = Spp[e];
            for (int e = t; e < n; e += LS) { CS_g[e] = g[e]; CS_hd[e] = hdiag[e]; }
        }
        for (int e = t; e < n; e += LS) {
            if (MODE == 1 && (e % 15) < 6) { sc[e] = 1.0; u[e] = 0.0; y[e] = 0.0; continue; }      // pose rows: scaled by the pose kernel (their scale needs the visual diagonal)
            double s;
            if (iteration == 0) { s = 1.0 / (1.0 + sqrt(hdiag[e])); d.scale_p[(size_t)w * n + e] = s; }
            else s = d.scale_p[(size_t)w * n + e];
            const double D2 = fmin(fmax(s * s * hdiag[e], 1e-6), 1e32);
            sc[e] = s; D[e] = sqrt(D2);
            d.diag_p[(size_t)w * n + e] = D[e];
            d.grad_p[(size_t)w * n + e] = s * g[e] / D[e];
            u[e] = s * s * g[e] / D2;
            d.up[(size_t)w * n + e] = u[e];
            y[e] = s * (g[e] + bs[e]);
        }
        __syncthreads();
        }
        {   // qT = u^T T u on the unscaled blocks, then scale in place and add the LM diagonal
            PHASE_IDS();
            double accq = 0;
            if (MODE == 1) {
                // t_Y[pose row] = sum over the speed/bias columns of (unscaled Y) u: the pose kernel adds 2 u_pose . t_Y to u^T T u
                for (int rho = t; rho < n6; rho += LS) {
                    const int a = rho / 6, r = rho - 6 * a;
                    double sY = 0;
                    for (int i = (a > 0 ? a - 1 : 0); i <= (a + 1 < N ? a + 1 : N - 1); i++) {      // (before the elimination only the blocks of the IMU factors are filled)
                        const int lo = nlo(i, M);
                        if (a >= lo && a <= nhi(i, M, N)) {
                            const double *Yr = Ysb + yo[i] + (a - lo) * 54 + r * 9, *ui = u + 15 * i + 6;
#pragma unroll
                            for (int kk = 0; kk < 9; kk++) sY += Yr[kk] * ui[kk];
                        }
                    }
                    CS_tY[rho] = sY;
                }
                __syncthreads();
            }
            for (int e = t; e < (MODE == 1 ? 0 : nS); e += LS) {
                int I, J, r, c;
                spp_decode(e, I, J, r, c);
                if (I == J && r < c) continue;
                const int gi = 15 * I + r, gj = 15 * J + c;
                const double v = Spp[e];
                accq += (gi == gj ? 1.0 : 2.0) * v * u[gi] * u[gj];
                double sv = v * sc[gi] * sc[gj];
                if (gi == gj) sv += mu * D[gi] * D[gi];
                Spp[e] = sv;
            }
            for (int e = t; e < 81 * N; e += LS) {
                const int I = e / 81, rc = e - 81 * I, r = rc / 9, c = rc - 9 * r;
                if (r < c) continue;
                const int gi = 15 * I + 6 + r, gj = 15 * I + 6 + c;
                const double v = Dss[e];
                accq += (gi == gj ? 1.0 : 2.0) * v * u[gi] * u[gj];
                double sv = v * sc[gi] * sc[gj];
                if (gi == gj) sv += mu * D[gi] * D[gi];
                Dss[e] = sv;
            }
            for (int e = t; e < 81 * N; e += LS) {
                const int i = e / 81, rc = e - 81 * i, r = rc / 9, c = rc - 9 * r, pp = npar(i, M);
                if (pp < 0) continue;
                const int gi = 15 * pp + 6 + r, gj = 15 * i + 6 + c;
                const double v = Css[e];
                accq += 2.0 * v * u[gi] * u[gj];
                Css[e] = v * sc[gi] * sc[gj];
            }
            for (int e = t; e < ytot; e += LS) {
                int i = 0;
                i = yNode[e / 54];          // (long windows searched yo[] per ENTRY here: ~9 dependent LDS reads each)
                const int q = e - yo[i], ai = q / 54, rc = q - 54 * ai, r = rc / 9, c = rc - 9 * r;
                const int gi = 15 * (nlo(i, M) + ai) + r, gj = 15 * i + 6 + c;
                const double v = Ysb[e];
                accq += 2.0 * v * u[gi] * u[gj];
                Ysb[e] = v * sc[gi] * sc[gj];
            }
            __syncthreads();                               // every thread is done with u (it lives in red)
            // fixed-shape sum: butterfly inside every wavefront, then the eight wavefront sums in order
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) accq += __shfl_xor(accq, off);
            if (lane == 0) red[wv] = accq;
            __syncthreads();
            if (t == 0) {
                const double qsum = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
                if (MODE == 1) CS_sc[0] = qsum; else st.qT = qsum;
            }
        }
        __syncthreads();
        STAMP(3);
        // ---- speed/bias chains: wavefront 0 forward (0 .. M-1), wavefront 1 backward (N-1 .. M+1) ----
        // node i: (1) D_i -> inverse Cholesky factor; (2) rows of [C_i ; Y_i] times L_i^-T;
        //         (3) downdate the parent's diagonal block and pose coupling (not for children of M here:
        //             both chains end in M, those two downdates are applied after the join).
        { PHASE_IDS();
        auto node_rows = [&](int i, bool has_par) {            // step (2), executed by one wavefront
            // the right-hand side rides along as one more row: z_i^T = y_i^T L_i^-T (forward substitution)
            const int nr = nhi(i, M, N) - nlo(i, M) + 1, nrows = (has_par ? 9 : 0) + 6 * nr + 1;
            const double *Li = Dss + i * 81;
            for (int rho = lane; rho < nrows; rho += 64) {
                double *ptr = (has_par && rho < 9) ? Css + i * 81 + rho * 9
                            : (rho == nrows - 1 ? y + 15 * i + 6 : Ysb + yo[i] + (rho - (has_par ? 9 : 0)) * 9);
                double v[9], o[9];
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = ptr[k];
#pragma unroll
                for (int c = 0; c < 9; c++) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k <= c; k++) s += v[k] * Li[c * 9 + k];
                    o[c] = s;
                }
#pragma unroll
                for (int k = 0; k < 9; k++) ptr[k] = o[k];
            }
            WSYNC();
        };
        auto node_downdate = [&](int i, int lid, int nl) {     // step (3) with nl lanes, lane id lid
            const int pp = npar(i, M), lo = nlo(i, M), nr = nhi(i, M, N) - lo + 1;
            const double *C = Css + i * 81, *Yi = Ysb + yo[i];
            double *Yp = Ysb + yo[pp] + (lo - nlo(pp, M)) * 54;
            for (int e = lid; e < 45 + 54 * nr + 9; e += nl) {
                if (e >= 45 + 54 * nr) {                      // right-hand side of the parent: y_p -= C_i z_i
                    const int r = e - 45 - 54 * nr;
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += C[r * 9 + k] * y[15 * i + 6 + k];
                    y[15 * pp + 6 + r] -= s;
                } else if (e < 45) {
                    const int r = triAB[2 * e], c = triAB[2 * e + 1];
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += C[r * 9 + k] * C[c * 9 + k];
                    Dss[pp * 81 + r * 9 + c] -= s;
                } else {
                    const int q = e - 45, row = q / 9, c = q - 9 * row;     // row = (a - lo) * 6 + r
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += Yi[row * 9 + k] * C[c * 9 + k];
                    Yp[row * 9 + c] -= s;
                }
            }
        };
        // Software pipeline over the chain nodes, one block barrier per node.  What the NEXT node's factorisation waits for is
        // small -- C_i L_i^-T (9 rows), z_i, the 45 + 9 entries of the parent's diagonal block and right-hand side -- and stays
        // on wavefronts 0 (forward chain) / 1 (backward chain).  The bulk, Y_i L_i^-T and the parent's pose coupling
        // Y_p -= Y_i C_i^T, follows one node behind on wavefronts 2 / 3, a lane per row of Y_i (the transformed row stays in
        // registers for its own downdate).  Entry for entry the same sums as node_rows / node_downdate.
        auto crit_rows = [&](int i) {                          // C_i and z_i
            const double *Li = Dss + i * 81;
            if (lane < 10) {
                double *ptr = lane < 9 ? Css + i * 81 + lane * 9 : y + 15 * i + 6;
                double v[9], o[9];
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = ptr[k];
#pragma unroll
                for (int c = 0; c < 9; c++) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k <= c; k++) s += v[k] * Li[c * 9 + k];
                    o[c] = s;
                }
#pragma unroll
                for (int k = 0; k < 9; k++) ptr[k] = o[k];
            }
            WSYNC();
        };
        auto crit_downdate = [&](int i) {                      // D_p -= C_i C_i^T, y_p -= C_i z_i
            const int pp = npar(i, M);
            const double *C = Css + i * 81;
            if (lane < 45) {
                const int r = triAB[2 * lane], c = triAB[2 * lane + 1];
                double s = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) s += C[r * 9 + k] * C[c * 9 + k];
                Dss[pp * 81 + r * 9 + c] -= s;
            } else if (lane < 54) {
                const int r = lane - 45;
                double s = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) s += C[r * 9 + k] * y[15 * i + 6 + k];
                y[15 * pp + 6 + r] -= s;
            }
            WSYNC();
        };
        auto y_rows = [&](int i, bool downdate) {              // Y_i L_i^-T, then Y_p -= Y_i C_i^T (C_i already transformed)
            const int pp = npar(i, M), lo = nlo(i, M), nr = nhi(i, M, N) - lo + 1;
            const double *Li = Dss + i * 81, *C = Css + i * 81;
            for (int rho = lane; rho < 6 * nr; rho += 64) {
                double *ptr = Ysb + yo[i] + rho * 9;
                double v[9], o[9];
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = ptr[k];
#pragma unroll
                for (int c = 0; c < 9; c++) {
                    double s = 0;
#pragma unroll
                    for (int k = 0; k <= c; k++) s += v[k] * Li[c * 9 + k];
                    o[c] = s;
                }
#pragma unroll
                for (int k = 0; k < 9; k++) ptr[k] = o[k];
                if (downdate) {
                    double *Yp = Ysb + yo[pp] + (lo - nlo(pp, M)) * 54 + rho * 9;
#pragma unroll
                    for (int c = 0; c < 9; c++) {
                        double s = 0;
#pragma unroll
                        for (int k = 0; k < 9; k++) s += o[k] * C[c * 9 + k];
                        Yp[c] -= s;
                    }
                }
            }
        };
        {
            const int cntF = M, cntB = N - 1 - M, steps = (cntF > cntB ? cntF : cntB) + 1;
            for (int k = 0; k < steps; k++) {
                if (wv < 2) {
                    const int cnt = wv == 0 ? cntF : cntB;
                    if (k < cnt) {
                        const int i = wv == 0 ? k : N - 1 - k;
                        if (chol_inv_block<9>(Dss + i * 81, lane)) { if (lane == 0) flag[0] = 1; }
                        else {
                            crit_rows(i);
                            if (npar(i, M) != M) crit_downdate(i);
                        }
                    }
                } else if (wv < 4) {
                    const int cnt = wv == 2 ? cntF : cntB, kk = k - 1;
                    if (kk >= 0 && kk < cnt) {
                        const int i = wv == 2 ? kk : N - 1 - kk;
                        y_rows(i, npar(i, M) != M);
                    }
                }
                __syncthreads();
                if (flag[0]) break;
            }
        }
        if (!flag[0]) {
            if (M >= 1) node_downdate(M - 1, t, LS);
            __syncthreads();
            if (M + 1 <= N - 1) node_downdate(M + 1, t, LS);
            __syncthreads();
            if (wv == 0) {
                if (chol_inv_block<9>(Dss + M * 81, lane)) { if (lane == 0) flag[0] = 1; }
                else node_rows(M, false);
            }
            __syncthreads();
        }
        }
        STAMP(4);
        if (!flag[0]) {
            // ---- pose system: Spp -= sum_i Y_i Y_i^T ---------------------------------------------------
            { PHASE_IDS();
            if constexpr (BIG) {
            // (round 3) a thread owns a 3 x 6 half of a 6x6 pose block: per k it reads three values of Y_I and six of Y_J for
            // eighteen products (9 LDS reads per 18 FMAs; one entry per thread needed 2 per FMA, the 2 x 3 sub-blocks of the
            // first rewrite 5 per 6).  The phase is LDS-bandwidth bound (every entry re-reads both nine-vectors from every
            // covering node): 30.6 -> 15.4 us with 2 x 3 at N = 18 and 10.7 us with 3 x 6.  Long windows only: under the 128-VGPR
            // cap of the two-per-CU instantiation the eighteen accumulators spill (7.6 against 7.1 us at N = 11, 81 against 79 us
            // for the kernel).  Entry for entry the same sums in the same order (nodes ascending, k ascending).
            for (int item = t; item < N * (N + 1); item += LS) {
                const int q = item >> 1, r0 = 3 * (item & 1);
                const int ij = blkIJ[q], I = ij & 255, J = ij >> 8;
                double acc[3][6];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 6; c++) acc[r][c] = 0;
                for (int i = 0; i < N; i++) {
                    const int lo = nlo(i, M), hi = nhi(i, M, N);
                    if (J >= lo && I <= hi) {
                        const double *YI = Ysb + yo[i] + (I - lo) * 54 + r0 * 9, *YJ = Ysb + yo[i] + (J - lo) * 54;
#pragma unroll
                        for (int k = 0; k < 9; k++) {
                            const double a0 = YI[k], a1 = YI[9 + k], a2 = YI[18 + k];
                            double bb[6];
#pragma unroll
                            for (int c = 0; c < 6; c++) bb[c] = YJ[9 * c + k];
#pragma unroll
                            for (int c = 0; c < 6; c++) { acc[0][c] += a0 * bb[c]; acc[1][c] += a1 * bb[c]; acc[2][c] += a2 * bb[c]; }
                        }
                    }
                }
                double *B = (MODE == 1 ? CS_dlt : Spp) + q * 36 + r0 * 6;
                const bool dg = I == J;
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 6; c++)
                        if (!dg || r0 + r >= c) { if (MODE == 1) B[r * 6 + c] = -acc[r][c]; else B[r * 6 + c] -= acc[r][c]; }      // (a diagonal block keeps its lower triangle)
            }
            } else {
            // (round 3) a thread owns a 2 x 3 sub-block of a 6x6 pose block instead of one entry: per k it reads two values of
            // Y_I and three of Y_J for six products (5 LDS reads per 6 FMAs instead of 12).  The phase was LDS-bandwidth
            // bound (every entry re-read both nine-vectors from every covering node): 30.6 us at N = 18, 9.2 us at N = 11.
            // Entry for entry the same sums in the same order (nodes ascending, k ascending).
            for (int item = t; item < N * (N + 1) / 2 * 6; item += LS) {
                const int q = item / 6, sbk = item - 6 * q, r0 = 2 * (sbk >> 1), c0 = 3 * (sbk & 1);
                const int ij = blkIJ[q], I = ij & 255, J = ij >> 8;
                if (I == J && r0 + 1 < c0) continue;            // rows {0, 1} x columns {3, 4, 5} of a diagonal block: above the diagonal
                double s00 = 0, s01 = 0, s02 = 0, s10 = 0, s11 = 0, s12 = 0;
                for (int i = 0; i < N; i++) {
                    const int lo = nlo(i, M), hi = nhi(i, M, N);
                    if (J >= lo && I <= hi) {
                        const double *YI = Ysb + yo[i] + (I - lo) * 54 + r0 * 9, *YJ = Ysb + yo[i] + (J - lo) * 54 + c0 * 9;
#pragma unroll
                        for (int k = 0; k < 9; k++) {
                            const double a0 = YI[k], a1 = YI[9 + k], b0 = YJ[k], b1 = YJ[9 + k], b2 = YJ[18 + k];
                            s00 += a0 * b0; s01 += a0 * b1; s02 += a0 * b2;
                            s10 += a1 * b0; s11 += a1 * b1; s12 += a1 * b2;
                        }
                    }
                }
                double *B = (MODE == 1 ? CS_dlt : Spp) + q * 36 + r0 * 6 + c0;
                const bool dg = I == J;
                if (MODE == 1) {        // the chain kernel hands -sum_i Y_i' Y_i'^T over (the lower triangle of a diagonal block)
                    if (!dg || r0 >= c0) B[0] = -s00;
                    if (!dg || r0 >= c0 + 1) B[1] = -s01;
                    if (!dg || r0 >= c0 + 2) B[2] = -s02;
                    if (!dg || r0 + 1 >= c0) B[6] = -s10;
                    if (!dg || r0 + 1 >= c0 + 1) B[7] = -s11;
                    if (!dg || r0 + 1 >= c0 + 2) B[8] = -s12;
                } else {
                if (!dg || r0 >= c0) B[0] -= s00;
                if (!dg || r0 >= c0 + 1) B[1] -= s01;
                if (!dg || r0 >= c0 + 2) B[2] -= s02;
                if (!dg || r0 + 1 >= c0) B[6] -= s10;
                if (!dg || r0 + 1 >= c0 + 1) B[7] -= s11;
                if (!dg || r0 + 1 >= c0 + 2) B[8] -= s12;
                }
            }
            }
            if (t >= LS - n6) {                                 // pose right-hand side -= sum_i Y_i z_i
                const int rho = t - (LS - n6), a = rho / 6, r = rho - 6 * a;
                double s = 0;
                for (int i = 0; i < N; i++) {
                    const int lo = nlo(i, M);
                    if (a >= lo && a <= nhi(i, M, N)) {
                        const double *Yr = Ysb + yo[i] + (a - lo) * 54 + r * 9, *zi = y + 15 * i + 6;
#pragma unroll
                        for (int kk = 0; kk < 9; kk++) s += Yr[kk] * zi[kk];
                    }
                }
                y[15 * a + r] -= s;
            }
            }
            __syncthreads();
        }
        if (MODE == 1) {
            // hand-over to the pose kernel (same workgroup index, after the join of the two streams)
            PHASE_IDS();
            if (!flag[0]) {
                for (int e = t; e < tailsz; e += LS) CS_tail[e] = Dss[e];          // L_i^-1 | C_i' | Y_i'
                for (int e = t; e < n; e += LS) CS_y[e] = y[e];                    // z_i; pose rows: -sum_i Y_i' z_i
            }
            if (t == 0) CS_sc[1] = flag[0] ? 1.0 : 0.0;
            STAMP(5);
            return;
        }
        }   // (!pre)
        STAMP(5);
        if (!flag[0]) {
            // ---- blocked Cholesky of the pose system (6x6 blocks; diagonal blocks hold L_JJ^-1 afterwards)
            // Look-ahead: while wavefronts 1..7 apply the trailing update of step J, wavefront 0 updates the NEXT diagonal block
            // first and factors it, so the factorisation's dependent pivots and one barrier per step leave the critical path.
            // Entry for entry the same sums as a plain right-looking sweep.
            { PHASE_IDS();
            if (wv == 0 && chol_inv_block<6>(Spp + sblk(0, 0, N), lane)) { if (lane == 0) flag[0] = 1; }
            __syncthreads();
            for (int J = 0; J < N; J++) {
                if (flag[0]) break;
                const int m = N - J - 1;
                const double *Li = Spp + sblk(J, J, N);
                for (int rr = t; rr < m * 6 + 1; rr += LS) {    // panel rows: X = A L_JJ^-T; last row = rhs (z_J)
                    double *A = rr < m * 6 ? Spp + sblk(J + 1, J, N) + rr * 6 : y + 15 * J;   // blocks (J+1.., J) are contiguous
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
                // trailing update: S[I,K] -= X_I X_K^T for I >= K > J (packed columns J+1.. are contiguous)
                const int e0 = sblk(J + 1, J + 1, N), cntT = m * (m + 1) / 2 * 36;
                const double *X = Spp + sblk(J + 1, J, N);       // X_I at (I - J - 1) * 36
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
                    // (the trailing triangle is the tail of the packed storage: absolute block = first trailing block + q)
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
                    WSYNC();
                    if (chol_inv_block<6>(Spp + e0, lane)) { if (lane == 0) flag[0] = 1; }
                } else {
                    if constexpr (BIG) {
                    // long windows: 3 x 6 halves (9 LDS reads per 18 products), as in their Y Y^T phase: the first steps of an
                    // 18-frame window have 900 2 x 3 sub-blocks for 448 threads, i.e. three passes; 300 halves are one
                    const int nit = (m * (m + 1) / 2 - 1) * 2;
                    for (int it = t - 64; it < nit + m * 6; it += LS - 64) {
                        if (it >= nit) { trailing_entry(cntT + (it - nit)); continue; }      // rhs rows
                        const int q = 1 + (it >> 1), r0 = 3 * (it & 1);
                        const int ij = blkIJ[e0 / 36 + q];
                        const int ia = (ij & 255) - (J + 1), ca = (ij >> 8) - (J + 1);
                        const bool dg = ia == ca;
                        const double *XI = X + ia * 36 + r0 * 6, *XK = X + ca * 36;
                        double acc[3][6];
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int c = 0; c < 6; c++) acc[r][c] = 0;
#pragma unroll
                        for (int k = 0; k < 6; k++) {
                            const double a0 = XI[k], a1 = XI[6 + k], a2 = XI[12 + k];
#pragma unroll
                            for (int c = 0; c < 6; c++) { const double bk = XK[6 * c + k]; acc[0][c] += a0 * bk; acc[1][c] += a1 * bk; acc[2][c] += a2 * bk; }
                        }
                        double *Bq = Spp + e0 + q * 36 + r0 * 6;
#pragma unroll
                        for (int r = 0; r < 3; r++)
#pragma unroll
                            for (int c = 0; c < 6; c++)
                                if (!dg || r0 + r >= c) Bq[r * 6 + c] -= acc[r][c];
                    }
                    } else {
                    // (round 3) 2 x 3 sub-blocks per thread, as in the Y Y^T phase: 5 LDS reads per 6 products instead of 12
                    const int nit = (m * (m + 1) / 2 - 1) * 6;         // sub-blocks of the trailing blocks after the first one
                    for (int it = t - 64; it < nit + m * 6; it += LS - 64) {
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
                }
                __syncthreads();
            }
            }
        }
        __syncthreads();
        STAMP(6);
        if (flag[0] || attempt < d.force_retry) {
            mu *= 10.0; attempt++;
            __syncthreads();
            if (t == 0) flag[0] = 0;
            __syncthreads();
            continue;
        }
        // ---- backward substitution (the forward one rode along with the factorisation) ---------------
        // pose block: one wavefront, right-hand side in REGISTERS: yp0 holds frames 0..9 (lane = 6 frame
        // + r), yp1 frames 10.. ; pivot vectors travel by v_readlane, every lane forms its own row's dot
        // product: the N dependent steps need no LDS round trip and no barrier.
        if (wv == 0) {
            PHASE_IDS();
            const int q6 = lane / 6, c6 = lane - 6 * q6;
            const int fA = q6, fB = 10 + q6;                    // my frame in yp0 / yp1
            const bool hasPA = q6 < 10 && fA < N, hasPB = q6 < 10 && fB < N;
            double yp0 = hasPA ? y[15 * fA + c6] : 0.0, yp1 = hasPB ? y[15 * fB + c6] : 0.0;
            for (int J = N - 1; J >= 0; J--) {                  // right-looking
                const int b = 6 * (J % 10);
                double v[6];
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = J < 10 ? readlane_d2(yp0, b + k) : readlane_d2(yp1, b + k);
                const double *Lc = Spp + sblk(J, J, N) + c6;    // column c6 of L_JJ^-1
                double x = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) x += Lc[k * 6] * v[k];
                if (J < 10) { if (hasPA && fA == J) yp0 = x; } else { if (hasPB && fB == J) yp1 = x; }
#pragma unroll
                for (int k = 0; k < 6; k++) v[k] = J < 10 ? readlane_d2(yp0, b + k) : readlane_d2(yp1, b + k);
                if (hasPA && fA < J) {
                    const double *Lb = Spp + sblk(J, fA, N) + c6;
#pragma unroll
                    for (int k = 0; k < 6; k++) yp0 -= Lb[k * 6] * v[k];
                }
                if (N > 10 && hasPB && fB < J) {
                    const double *Lb = Spp + sblk(J, fB, N) + c6;
#pragma unroll
                    for (int k = 0; k < 6; k++) yp1 -= Lb[k * 6] * v[k];
                }
            }
            if (hasPA) y[15 * fA + c6] = yp0;
            if (hasPB) y[15 * fB + c6] = yp1;
        }
        __syncthreads();
        // chain rhs -= Y_i^T x_pose: thread = (speed/bias row, quarter of the pose blocks), folded in fixed order
        { PHASE_IDS();
        for (int tq = t; tq < 36 * N; tq += LS) {
            const int o = tq >> 2, part = tq & 3, i = o / 9, c = o - 9 * i, lo = nlo(i, M), nr = nhi(i, M, N) - lo + 1;
            const double *Yc = Ysb + yo[i] + c;
            double s = 0;
            for (int a = part; a < nr; a += 4) {
                const double *xa = y + 15 * (lo + a);
#pragma unroll
                for (int r = 0; r < 6; r++) s += Yc[(a * 6 + r) * 9] * xa[r];
            }
            red[tq] = s;
        }
        __syncthreads();
        if (t < 9 * N) {
            const int i = t / 9, c = t - 9 * i;
            y[15 * i + 6 + c] -= (red[4 * t] + red[4 * t + 1]) + (red[4 * t + 2] + red[4 * t + 3]);
        }
        }
        __syncthreads();
        // chains, reverse elimination order: x_i = L_i^-T (z_i - C_i^T x_parent); wavefront 0 takes M and the
        // forward chain (lane = 9 i + c), wavefront 1 the backward chain (lane = 9 (i - M - 1) + c)
        if (BIG) {
            // long windows: one node at a time on lanes 0..8, the parent's x through LDS (measured against two chain positions per
            // lane in registers with the parent by v_readlane, the short-window form: 21.4 us of solves at N = 18 against 17.8 -- the
            // run-time chain positions cost more integer work than the two wave syncs per node)
            auto node_bwd_lds = [&](int i, int pp) {
                const int cl = lane < 9 ? lane : 0;
                double sv = y[15 * i + 6 + cl];
                if (pp >= 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) sv -= Css[i * 81 + k * 9 + cl] * y[15 * pp + 6 + k];
                }
                double x = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) x += Dss[i * 81 + k * 9 + cl] * readlane_d2(sv, k);
                WSYNC();
                if (lane < 9) y[15 * i + 6 + lane] = x;
                WSYNC();
            };
            if (wv == 0) node_bwd_lds(M, -1);
            __syncthreads();
            if (wv == 0) { for (int i = M - 1; i >= 0; i--) node_bwd_lds(i, i + 1); }
            else if (wv == 1) { for (int i = M + 1; i <= N - 1; i++) node_bwd_lds(i, i - 1); }
        } else {
        if (wv < 2) {
            PHASE_IDS();
            const int q9 = lane / 9, c9 = lane - 9 * q9;
            const int mynode = wv == 0 ? q9 : M + 1 + q9;
            const bool has = q9 < 7 && (wv == 0 ? mynode <= M : mynode <= N - 1);
            double ys = has ? y[15 * mynode + 6 + c9] : 0.0;
            auto sb_base = [&](int i) { return i > M ? 9 * (i - M - 1) : 9 * i; };
            auto node_bwd = [&](int i, int pp, bool par_in_lds) {
                const int b = sb_base(i);
                const bool mine = has && mynode == i;
                double v[9];
                if (pp >= 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) v[k] = par_in_lds ? y[15 * pp + 6 + k] : readlane_d2(ys, sb_base(pp) + k);
                    const double *Cc = Css + i * 81 + c9;       // column c9 of C_i
                    double s = 0;
#pragma unroll
                    for (int k = 0; k < 9; k++) s += Cc[k * 9] * v[k];
                    if (mine) ys -= s;
                }
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = readlane_d2(ys, b + k);
                const double *Lc = Dss + i * 81 + c9;           // column c9 of L_i^-1
                double x = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) x += Lc[k * 9] * v[k];
                if (mine) ys = x;
            };
            if (wv == 0) {
                node_bwd(M, -1, false);
                if (has && mynode == M) y[15 * M + 6 + c9] = ys;
            }
            // x_M crosses to wavefront 1 through LDS
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }
        __syncthreads();
        if (wv < 2) {
            // (re-declared: the block barrier above must be reached by every wavefront)
            PHASE_IDS();
            const int q9 = lane / 9, c9 = lane - 9 * q9;
            const int mynode = wv == 0 ? q9 : M + 1 + q9;
            const bool has = q9 < 7 && (wv == 0 ? mynode <= M : mynode <= N - 1);
            double ys = has ? y[15 * mynode + 6 + c9] : 0.0;
            auto sb_base = [&](int i) { return i > M ? 9 * (i - M - 1) : 9 * i; };
            auto node_bwd = [&](int i, int pp, bool par_in_lds) {
                const int b = sb_base(i);
                const bool mine = has && mynode == i;
                double v[9];
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = par_in_lds ? y[15 * pp + 6 + k] : readlane_d2(ys, sb_base(pp) + k);
                const double *Cc = Css + i * 81 + c9;           // column c9 of C_i
                double s = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) s += Cc[k * 9] * v[k];
                if (mine) ys -= s;
#pragma unroll
                for (int k = 0; k < 9; k++) v[k] = readlane_d2(ys, b + k);
                const double *Lc = Dss + i * 81 + c9;           // column c9 of L_i^-1
                double x = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) x += Lc[k * 9] * v[k];
                if (mine) ys = x;
            };
            if (wv == 0) { for (int i = M - 1; i >= 0; i--) node_bwd(i, i + 1, false); }
            else { for (int i = M + 1; i <= N - 1; i++) node_bwd(i, i - 1, i - 1 == M); }
            if (has) y[15 * mynode + 6 + c9] = ys;
        }
        }
        __syncthreads();
        STAMP(7);
        break;
    }
    if (!ls_fail) {
        for (int e = t; e < n; e += LS) {
            d.zp[(size_t)w * n + e] = d.scale_p[(size_t)w * n + e] * y[e];
            d.gn_p[(size_t)w * n + e] = -D[e] * y[e];
        }
    }
    double cost_w = 0.0;
    {
        double m = gmax_l;
        for (int i = t; i < N; i += LS) {
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
        cost_w = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
        if (t == 0) {
            double mm = red[8];
            for (int k = 1; k < 8; k++) mm = fmax(mm, red[8 + k]);
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
    STAMP(8);
}

template <bool BIG, int NC, int MODE>
__global__ __launch_bounds__(LS, (BIG || MODE != 0) ? 2 : 4) void k_build_solve_sb(DevBatch d) {       // (the split solve runs one workgroup per CU: no 128-VGPR cap)
    extern __shared__ __align__(16) double lds[];
    build_solve_sb_body<BIG, NC, MODE>(d, lds);
}
// k_lin_gram_chain (round 5): the two things of an iteration that depend on the state alone and not on each other, in ONE launch on the
// solve stream -- blockIdx.y = 0: k_lin_gram's workgroup of the window (eight wavefronts); blockIdx.y = 1: the chain half of the split
// solve (MODE 1), which needs the IMU / prior blocks k_front left.  Small batches only (every workgroup finds a CU of its own).
// NT > 0 (handles whose elimination is not split over workgroups): the workgroup that linearised the window goes straight on to its
// rank-1 downdates (k_rank1_mfma's routine with its NT (NT + 1) / 2 output tiles dealt over <= 8 wavefronts: a tile is summed by one
// wavefront in landmark order either way, the bits are k_rank1_mfma's) -- the chain half beside it takes longer than k_lin_gram alone.
// Instantiated for the reference's 18 frames (64 windows: 2.76 -> 2.63 ms); at 11 frames it LOSES (128 windows 1.93 -> 2.03 ms, 256: 2.39 -> 2.70):
// eight wavefronts with two tiles each are slower than k_rank1_mfma's fifteen, and the role becomes the launch's long pole.
template <bool EX, bool BIG, int NC, int NT>
__global__ __launch_bounds__(LS, 2) void k_lin_gram_chain(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    static_assert(LS == 64 * LG_WAVES_SMALL, "k_lin_gram's eight-wavefront form");
    if (blockIdx.y == 0) {
        lin_gram_body<EX, LG_WAVES_SMALL>(d, lds);
        if constexpr (NT > 0) {
            constexpr int ntiles = NT * (NT + 1) / 2, TPW = (ntiles + 7) / 8, nwaves = (ntiles + TPW - 1) / TPW;
            // W, the landmark pieces and Tvis written by this workgroup's wavefronts are read back by others below
            __syncthreads();                                   // (same workgroup: the barrier's workgroup-scope release / acquire)
            if ((int)threadIdx.x >= 64 * nwaves) return;       // (a wavefront without tiles; it is past the last barrier it shares)
            rank1_body<NT, TPW, 64, 1, EX>(d);
        }
    }
    else build_solve_sb_body<BIG, NC, 1>(d, lds);
}
// k_pose_dogleg (round 5): the pose half of the split solve, the dogleg step and the step control of a window in ONE launch -- three stages
// of one workgroup per window that followed each other as separate launches (~4 us each on this GPU for a small batch, plus the reload
// of what the previous stage had in LDS).  The same routines in the same order: bit for bit the three-launch sequence.
template <bool BIG, int NC, bool EX>
__global__ __launch_bounds__(LS, 2) void k_pose_dogleg(DevBatch d) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double red[256];
    __shared__ int s_accept;
    build_solve_sb_body<BIG, NC, 2>(d, lds);
    // z_p, the Gauss-Newton step, u_p and the solver scalars this workgroup wrote are read back below by other threads of the SAME
    // workgroup: the barrier's workgroup-scope release / acquire is what that needs (an agent-scope fence writes the XCD's L2 back on
    // this GPU: measured, it cost the 18-frame window more than the launch it saved)
    __syncthreads();
    if (threadIdx.x >= 256) return;                   // (the dogleg and the step control are 256-thread routines)
    dogleg_body<EX>(d, blockIdx.x, threadIdx.x, red);
    __syncthreads();                                  // candidate states, per-factor costs and model pieces of this window are written
    step_control_body<true, EX>(d, blockIdx.x, threadIdx.x, lds, red, s_accept);
}
template __global__ void k_pose_dogleg<false, 11, false>(DevBatch);
template __global__ void k_pose_dogleg<false, 0, false>(DevBatch);
template __global__ void k_pose_dogleg<true, 0, false>(DevBatch);
template __global__ void k_pose_dogleg<false, 0, true>(DevBatch);
template __global__ void k_pose_dogleg<true, 0, true>(DevBatch);
template __global__ void k_lin_gram_chain<false, false, 0, 0>(DevBatch);
template __global__ void k_lin_gram_chain<false, false, 11, 0>(DevBatch);
template __global__ void k_lin_gram_chain<false, true, 0, 0>(DevBatch);
template __global__ void k_lin_gram_chain<true, false, 0, 0>(DevBatch);
template __global__ void k_lin_gram_chain<true, true, 0, 0>(DevBatch);
template __global__ void k_lin_gram_chain<false, true, 0, 7>(DevBatch);        // the reference's 18 frames
template __global__ void k_build_solve_sb<false, 0, 0>(DevBatch);
template __global__ void k_build_solve_sb<false, 11, 0>(DevBatch);
template __global__ void k_build_solve_sb<true, 0, 0>(DevBatch);
template __global__ void k_build_solve_sb<false, 0, 1>(DevBatch);
template __global__ void k_build_solve_sb<false, 11, 1>(DevBatch);
template __global__ void k_build_solve_sb<true, 0, 1>(DevBatch);
template __global__ void k_build_solve_sb<false, 0, 2>(DevBatch);
template __global__ void k_build_solve_sb<false, 11, 2>(DevBatch);
template __global__ void k_build_solve_sb<true, 0, 2>(DevBatch);
size_t build_solve_cs_doubles(int N) { return sb_cs_doubles(N); }

size_t build_solve_sb_bytes(int N, int prior_H_sz) {
    const int M = N / 2;
    size_t ytot = 0;
    for (int i = 0; i < N; i++) ytot += (size_t)((i < M ? i + 1 : N - 1) - (i > M ? i - 1 : 0) + 1) * 54;
    const size_t n = 15 * (size_t)N, nS = (size_t)N * (N + 1) / 2 * 36;
    size_t tail = 162 * (size_t)N + ytot;
    const size_t stage = (size_t)RCH * (6 * (size_t)N + 1);     // retry staging lives in the Dss/Css/Ysb region
    if (tail < stage) tail = stage;
    const size_t nred = (size_t)sb_nred(N, prior_H_sz);
    return (3 * n + nred + 16 + 16 + 2 + ((size_t)N * (N + 1) / 2 + 3) / 4 + 1 + 20 + 24 + nS + tail + 2) * sizeof(double);
}
