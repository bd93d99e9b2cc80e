// isv_dogleg.h -- the bodies of k_dogleg / k_step_control (see isv_solver.hip for the kernels and the description) as device
// routines shared with k_pose_dogleg (isv_build_solve_sb.hip: the pose half of the split solve, the dogleg and the step control of a
// window in ONE launch, small batches).
#pragma once
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"
#include "isv_proj_factor.h"
#include "isv_prior_factor.h"
#include "isv_imu_factor.h"

// ------------------------------------------------------------------------------------------
// block-wide deterministic sums of per-thread partials, K values at once (256 threads = four wavefronts): a shuffle tree
// inside every wavefront, then the four wavefront partials in fixed order through LDS -- TWO block barriers per call.
// (Rounds 1-2 summed one value at a time through an eight-level LDS tree with a block barrier per level: the nine sums
// of a dogleg + step-control pass cost ~90 barriers, a fifth of k_dogleg<true>'s critical path.)
template <int K>
DEV void block_sums(double (&v)[K], double *red /* >= 4 K doubles */, int t) {
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    }
    if ((t & 63) == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) red[(t >> 6) * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = ((red[k] + red[K + k]) + red[2 * K + k]) + red[3 * K + k];
    __syncthreads();
}

// DoglegStrategy::ComputeTraditionalDoglegStep + undo of the scalings + Evaluator::Plus.
// One workgroup per running window.
#ifdef ISV_STAMP
#define DSTAMP(k) do { if (t == (k >= 56 ? 64 : 0)) { unsigned long long now_ = wall_clock64(); d.dbg[(size_t)w * 64 + (k)] += (double)(now_ - (k >= 56 ? t1_last : t_last)); if (k >= 56) t1_last = now_; else t_last = now_; } } while (0)
#else
#define DSTAMP(k) do {} while (0)
#endif
// back-substitution of ONE eliminated landmark (schur_eliminator BackSubstitute) + its term of the Cauchy-point denominator,
// from the packed w vectors; zs / us: the window's z_p / u_p in LDS.  Shared by k_dogleg and the multi-workgroup k_backsub_split.
template <bool EX>
DEV void backsub_landmark(const DevBatch &d, const int l, const int fw0, const double *zs, const double *us, const double mu) {
            // (round 3: ONE metadata word instead of three dependent index loads; the landmark scalars are requested
            // before the w loop; two observations' w vectors are in flight per trip -- clamped address, masked add: same
            // order of additions, same bits)
            const unsigned m0 = d.lm_meta[l];
            const int h = (int)(m0 & 255), k = (int)((m0 >> 8) & 255);
            const double *wv = d.W + (size_t)(fw0 + (int)(m0 >> 16) + l) * 6;     // slots of frames h .. h + k - 1
            const double sl = d.scale_l[l], E = d.lmE[l], gl = d.lmG[l], Dl = d.diag_l[l];
            double wz = 0, wu = 0;        // w_l^T z_p, w_l^T u_p
            for (int o = 0; o < k; o += 2) {
                const int o1 = o + 1 < k ? o + 1 : o;
                const double2 a01 = *reinterpret_cast<const double2 *>(wv + 6 * o), a23 = *reinterpret_cast<const double2 *>(wv + 6 * o + 2),
                              a45 = *reinterpret_cast<const double2 *>(wv + 6 * o + 4);
                const double2 b01 = *reinterpret_cast<const double2 *>(wv + 6 * o1), b23 = *reinterpret_cast<const double2 *>(wv + 6 * o1 + 2),
                              b45 = *reinterpret_cast<const double2 *>(wv + 6 * o1 + 4);
                const double *z = zs + 15 * (h + o), *u = us + 15 * (h + o);
                wz += a01.x * z[0] + a01.y * z[1] + a23.x * z[2] + a23.y * z[3] + a45.x * z[4] + a45.y * z[5];
                wu += a01.x * u[0] + a01.y * u[1] + a23.x * u[2] + a23.y * u[3] + a45.x * u[4] + a45.y * u[5];
                if (o + 1 < k) {
                    const double *z1 = zs + 15 * (h + o + 1), *u1 = us + 15 * (h + o + 1);
                    wz += b01.x * z1[0] + b01.y * z1[1] + b23.x * z1[2] + b23.y * z1[3] + b45.x * z1[4] + b45.y * z1[5];
                    wu += b01.x * u1[0] + b01.y * u1[1] + b23.x * u1[2] + b23.y * u1[3] + b45.x * u1[4] + b45.y * u1[5];
                }
            }
            if (EX) {                                         // the extrinsic block (pseudo-frame Nr) couples to every landmark
                const double *we = d.Wex + (size_t)l * 6, *z = zs + 15 * d.Nr, *u = us + 15 * d.Nr;
                for (int c6 = 0; c6 < 6; c6++) { wz += we[c6] * z[c6]; wu += we[c6] * u[c6]; }
            }
            const double Es = sl * sl * E, Dl2 = Dl * Dl;
            // scaled-space y_l = (g'_l - w'_l^T y_p) / (E'_l + mu D_l^2),  w'^T y_p = s_l w^T (Sc_p y_p) = s_l wz
            const double yl = (sl * gl - sl * wz) / (Es + mu * Dl2);
            d.gn_l[l] = -Dl * yl;
            const double ul = sl * sl * gl / Dl2, cl = sl * sl / (Es + mu * Dl2);
            d.lm_aterm[l] = cl * wu * wu + 2.0 * ul * wu + E * ul * ul;
}

template <bool EX>
DEV void dogleg_body(DevBatch &d, const int w, const int t, double *red) {
    SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING) return;
    const int n = d.np, N = d.N, l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    const double *gp = d.grad_p + (size_t)w * n, *gnp = d.gn_p + (size_t)w * n;
    const double *Dp = d.diag_p + (size_t)w * n, *scp = d.scale_p + (size_t)w * n;
    double *dp = d.delta_p + (size_t)w * n;
    if (st.ls_fail) {
        if (t == 0) { st.step_valid = 0; st.iteration += 1; st.fresh = 0; }
        return;
    }
#ifdef ISV_STAMP
    unsigned long long t_last = wall_clock64(), t1_last = t_last;
#endif
    if (d.lds_T) {
        // the window's prior factor records go to LDS NOW, by every thread: their global loads are in flight behind the
        // back-substitution instead of opening the candidate evaluation's serial path (5 us of the slowest wavefront there);
        // the region lies behind everything the phases before the candidate evaluation touch (zs / us: 2 n doubles)
        extern __shared__ __align__(16) double dynp[];
        double *const sPr = dynp + (size_t)(N - 1) * 48 + (size_t)d.n_prior_slots * 16;
        prior_stage_records(d, w, sPr + (size_t)d.n_prior_slots * 20, t, 256);
        // ... and their J^T J record at x (the priors' model pieces walked it entry by entry from global memory)
        double *const sPHw = sPr + prior_lds_bytes(d.n_prior_slots, false) / sizeof(double) + 992 + n;
        const double *PHg = d.prior_H + (size_t)w * d.prior_H_sz;
        for (int i0 = t; i0 < (d.dg_stage_ph ? d.prior_H_sz : 0); i0 += 4 * 256) {          // (up to 1024 doubles: one trip, four loads in flight)
            double v4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + 256 * u; v4[u] = PHg[i < d.prior_H_sz ? i : d.prior_H_sz - 1]; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = i0 + 256 * u; if (i < d.prior_H_sz) sPHw[i] = v4[u]; }
        }
    }
    double a = 0, b = 0, c = 0, e = 0;
    for (int i = t; i < n; i += 256) { a += gp[i] * gp[i]; b += gnp[i] * gnp[i]; c += gp[i] * gnp[i]; }
    if (st.fresh && !d.bs_split) {          // (bs_split: k_backsub_split has done this on many CUs)
        // back-substitution of the eliminated landmarks (schur_eliminator BackSubstitute) + the landmark
        // terms of the Cauchy-point denominator, from the w vectors (one landmark per thread and pass)
        extern __shared__ __align__(16) double dyn0[];
        double *zs = dyn0, *us = dyn0 + n;       // (this space is reused by the candidate evaluation below)
        for (int i = t; i < n; i += 256) { zs[i] = d.zp[(size_t)w * n + i]; us[i] = d.up[(size_t)w * n + i]; }
        __syncthreads();
        const double mu = st.mu;
        const int fw0 = d.f_off[w];
        for (int l = l0 + t; l < l1; l += 256) backsub_landmark<EX>(d, l, fw0, zs, us, mu);
    }
    // (round 4: four trips' loads in flight in the three landmark loops -- clamped addresses, masked updates in the same order:
    // same bits; a 2000-landmark window waited a memory latency per trip, 8 trips x 3 loops)
    for (int lq = l0 + t; lq < l1; lq += 4 * 256) {
        double gl4[4], nl4[4], at4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int l = lq + 256 * u < l1 ? lq + 256 * u : l1 - 1; gl4[u] = d.grad_l[l]; nl4[u] = d.gn_l[l]; at4[u] = d.lm_aterm[l]; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (lq + 256 * u < l1) { a += gl4[u] * gl4[u]; b += nl4[u] * nl4[u]; c += gl4[u] * nl4[u]; e += at4[u]; }
    }
    DSTAMP(48);
    double sums4[4] = {a, b, c, e};
    block_sums<4>(sums4, red, t);
    const double g2 = sums4[0], gn2 = sums4[1], gdotgn = sums4[2], aterm = sums4[3];
    double alpha = st.alpha;
    if (st.fresh) alpha = g2 / (st.qT + aterm);
    const double radius = st.radius, gn_norm = sqrt(gn2), g_norm = sqrt(g2);
    // step = cg * gradient_ + cn * gauss_newton_step_   (scaled coordinates)
    double cg, cn, step_norm_scaled;
    bool need_norm = false;
    if (gn_norm <= radius) { cg = 0; cn = 1; step_norm_scaled = gn_norm; }
    else if (g_norm * alpha >= radius) { cg = -(radius / g_norm); cn = 0; step_norm_scaled = radius; }
    else {
        const double b_dot_a = -alpha * gdotgn;
        const double a_sq = pow(alpha * g_norm, 2.0);
        const double bma_sq = a_sq - 2 * b_dot_a + pow(gn_norm, 2.0);
        const double cc = b_dot_a - a_sq;
        const double dd = sqrt(cc * cc + bma_sq * (pow(radius, 2.0) - a_sq));
        const double beta = (cc <= 0) ? (dd - cc) / bma_sq : (radius * radius - a_sq) / (dd + cc);
        cg = -alpha * (1.0 - beta); cn = beta; step_norm_scaled = 0; need_norm = true;
    }
    double sn = 0;
    for (int i = t; i < n; i += 256) {
        const double s = cg * gp[i] + cn * gnp[i];
        sn += s * s;
        dp[i] = s / Dp[i] * scp[i];
    }
    for (int lq = l0 + t; lq < l1; lq += 4 * 256) {
        double gl4[4], nl4[4], dg4[4], sc4[4], lam4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int l = lq + 256 * u < l1 ? lq + 256 * u : l1 - 1; gl4[u] = d.grad_l[l]; nl4[u] = d.gn_l[l]; dg4[u] = d.diag_l[l]; sc4[u] = d.scale_l[l]; lam4[u] = d.lam[l]; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (lq + 256 * u < l1) {
            const int l = lq + 256 * u;
            const double s = cg * gl4[u] + cn * nl4[u];
            sn += s * s;
            const double dl = s / dg4[u] * sc4[u];
            d.delta_l[l] = dl;
            d.clam[l] = lam4[u] + dl;
        }
    }
    double sums1[1] = {sn};
    block_sums<1>(sums1, red, t);
    const double sn_tot = sums1[0];
    if (need_norm) step_norm_scaled = sqrt(sn_tot);
    __syncthreads();
    // candidate = Plus(x, delta); ambient step norm and |x|
    double dn = 0, xn = 0;
    for (int i = t; i < N; i += 256) {
        const double *x = d.pose + ((size_t)w * N + i) * 7, *sb = d.sb + ((size_t)w * N + i) * 9;
        double *xc = d.cpose + ((size_t)w * N + i) * 7, *sc = d.csb + ((size_t)w * N + i) * 9;
        double xp[7];
        pose_plus(x, dp + 15 * i, xp);
        for (int k = 0; k < 7; k++) { xc[k] = xp[k]; const double df = x[k] - xp[k]; dn += df * df; xn += x[k] * x[k]; }
        for (int k = 0; k < 9; k++) { const double v = sb[k] + dp[15 * i + 6 + k]; sc[k] = v; const double df = sb[k] - v; dn += df * df; xn += sb[k] * sb[k]; }
    }
    for (int lq = l0 + t; lq < l1; lq += 4 * 256) {
        double lam4[4], cl4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int l = lq + 256 * u < l1 ? lq + 256 * u : l1 - 1; lam4[u] = d.lam[l]; cl4[u] = d.clam[l]; }
#pragma unroll
        for (int u = 0; u < 4; u++) if (lq + 256 * u < l1) { const double df = lam4[u] - cl4[u]; dn += df * df; xn += lam4[u] * lam4[u]; }
    }
    DSTAMP(49);
    double sums2[2] = {dn, xn};
    block_sums<2>(sums2, red, t);
    const double dn_tot = sums2[0], xn_tot = sums2[1];
    if (t == 0) {
        st.alpha = alpha; st.dogleg_step_norm = step_norm_scaled;
        st.step_norm = sqrt(dn_tot); st.x_norm = sqrt(xn_tot);
        st.step_valid = 1; st.iteration += 1; st.fresh = 0;
    }
    if (!d.lds_T) return;                  // (generic path: separate candidate kernels on the side stream)
    // ---- IMU and prior factors at the candidate point + their model cost change, in this workgroup:
    //      wavefront 0: raw IMU residuals (lane per factor); wavefront 1: the prior factors;
    //      wavefronts 2, 3: model pieces  delta^T g + delta^T H delta / 2  from the J^T J blocks at x.
    extern __shared__ __align__(16) double dyn[];
    const int NIw = N - 1, slots = d.n_prior_slots;
    double *sImu = dyn;                            // [NIw][16] raw residual
    double *sMod = sImu + NIw * 16;                // [NIw][32] model pieces per tangent row
    double *sPm = sMod + NIw * 32;                 // [slots][16]
    double *sPrior = sPm + (size_t)slots * 16;     // prior_linearize_body scratch
    {   // the tangent step in LDS for the model pieces (its own region behind the prior scratch; every thread wrote its entries above)
        double *sDpw = sPrior + prior_lds_bytes(slots, false) / sizeof(double) + 992;
        for (int i = t; i < n; i += 256) sDpw[i] = dp[i];
    }
    __syncthreads();                               // candidate states and delta_p are visible to the workgroup
    DSTAMP(50);
#ifdef ISV_STAMP
    t1_last = wall_clock64();
#endif
    const int lane = t & 63, wv = t >> 6;
    // the prior factors at the candidate: raw residuals (lane per prior; the kinds of a wavefront's lanes run one after
    // another) split by kind over wavefronts 0 and 1, the sqrt_info rows and costs by wavefront 3 after the barrier
    double *const sRaw = sPrior, *const sW = sPrior + (size_t)slots * 10;
    const PriorRecs PR = prior_recs_at(d, sPrior + (size_t)slots * 20);       // staged at the top of this function
    const int n_rp = d.n_rp[w];
    double *const sHb = sPrior + prior_lds_bytes(slots, false) / sizeof(double);      // [2][496] J^T J records, then the weighted IMU residuals
    const double *sDp = sHb + 992;                         // delta_p of this window (staged above)
    if (wv == 0) {
        if (lane < NIw) {
            const size_t f = (size_t)w * NIw + lane;
            if (!d.imu_skip[f]) {
                const double *pi = d.cpose + ((size_t)w * N + lane) * 7, *si = d.csb + ((size_t)w * N + lane) * 9;
                double r15[15];
                imu_raw_residual(d.G, d.imu_in + f * ISV_IMU_IN, pi, pi + 7, si, si + 9, r15);
#pragma unroll
                for (int k = 0; k < 15; k++) sImu[lane * 16 + k] = r15[k];
            }
        }
        prior_phase1<false>(d, PR, d.cpose, d.csb, w, n_rp, sRaw, lane, 0x8u);     // roll / pitch
    } else if (wv == 1) {
        prior_phase1<false>(d, PR, d.cpose, d.csb, w, n_rp, sRaw, lane, 0x6u);     // Linear9, relative poses
        // model pieces of the prior factors from their J^T J blocks at x (staged at the top)
        const double *PH = d.dg_stage_ph ? sDp + n : d.prior_H + (size_t)w * d.prior_H_sz;
        for (int e = lane; e < slots * 12; e += 64) {
            const int q = e / 12, a = e - 12 * q;
            int ncol, off, c0, c1 = 0;
            bool valid = true;
            if (q == 0) { ncol = 6; off = PH_SE3; c0 = 0; }
            else if (q == 1) { ncol = 9; off = PH_LIN9; c0 = 15 * (d.Nvo - 1) + 6; }
            else if (q < 1 + d.Nvo) { const int k = q - 2; ncol = 12; off = PH_REL0 + PH_REL_SZ * k; c0 = 15 * k; c1 = 15 * (k + 1); }
            else {
                const int m = q - 1 - d.Nvo; ncol = 6; off = PH_REL0 + PH_REL_SZ * (d.Nvo - 1) + PH_RP_SZ * m;
                valid = m < n_rp; c0 = valid ? 15 * PR.rollpitch[m].index : 0;
            }
            double v = 0;
            if (valid && a < ncol) {
                const int np2 = ncol * (ncol + 1) / 2;
                double s = 0;
                for (int b = 0; b < ncol; b++) {
                    const int gb = (b < 6 || ncol != 12) ? c0 + b : c1 + b - 6;
                    s += PH[off + (a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a)] * sDp[gb];
                }
                const int ga = (a < 6 || ncol != 12) ? c0 + a : c1 + a - 6;
                v = sDp[ga] * (PH[off + np2 + a] + 0.5 * s);
            }
            sPm[q * 16 + a] = v;
        }
    } else {
        // IMU factor q, tangent row a (30 per factor): H row dot delta, packed pairs (max, min).
        // (round 3) Each of the two wavefronts STAGES its factor's 495-double J^T J record in LDS with coalesced loads and
        // forms the 30 row products from there; a lane used to walk "its" row of the packed triangle straight from global
        // memory -- 30 scattered loads per row, three rows per lane: 37 us, the slowest wavefront of the workgroup (the
        // other three waited 31 us for it).  Same order of additions per row.  The NEXT factor's record is requested as soon
        // as this one's is in LDS: its memory latency runs behind the row products.
        double *sH = sHb + (size_t)(wv - 2) * 496;
        double hv[8];
        auto fetch = [&](int q) {
            const double *H = d.imu_H + ((size_t)w * NIw + (q < NIw ? q : NIw - 1)) * ISV_IMU_H;
#pragma unroll
            for (int k = 0; k < 8; k++) { const int e = lane + 64 * k; hv[k] = H[e < ISV_IMU_H ? e : ISV_IMU_H - 1]; }
        };
        if (wv - 2 < NIw) fetch(wv - 2);
        if (wv == 3) prior_phase1<false>(d, PR, d.cpose, d.csb, w, n_rp, sRaw, lane, 0x1u);      // SE3 prior (behind the first record's latency)
        for (int q = wv - 2; q < NIw; q += 2) {
            ISV_WSYNC();                                   // the previous factor's rows have been read
#pragma unroll
            for (int k = 0; k < 8; k++) { const int e = lane + 64 * k; if (e < ISV_IMU_H) sH[e] = hv[k]; }
            ISV_WSYNC();
            if (q + 2 < NIw) fetch(q + 2);
            if (lane < 30) {
                const int a = lane;
                const double *dd = sDp + 15 * q;
                double s = 0;
#pragma unroll
                for (int b = 0; b < 30; b++) s += sH[a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a] * dd[b];
                sMod[q * 32 + a] = dd[a] * (sH[465 + a] + 0.5 * s);
            }
        }
    }
    DSTAMP(51);
    DSTAMP(56);
#ifdef ISV_STAMP
    if (t == 128) d.dbg[(size_t)w * 64 + 62] += (double)(wall_clock64() - t1_last);      // wavefront 2: the model pieces
#endif
    __syncthreads();
    DSTAMP(52);
    // sqrt_info-weighted IMU residuals -> cost; fixed-order sums of the model pieces; wavefront 3 (idle in the IMU loop for
    // every window length: 15 (N - 1) <= 192 only fails beyond N = 13, where it simply comes later): the priors' rows and costs
    double *sR2 = sHb;                             // (the J^T J staging is free again)
    if (wv == 3) prior_residual_costs(d, PR, w, n_rp, sRaw, sW, d.prior_cost_c, lane);
    for (int tq = t; tq < NIw * 15; tq += 256) {
        const int q = tq / 15, row = tq - 15 * q;
        const size_t f = (size_t)w * NIw + q;
        const double *S = d.imu_sqrt + f * 225 + row * 15;
        double r = 0;
#pragma unroll
        for (int k = 0; k < 15; k++) r += S[k] * sImu[q * 16 + k];
        sR2[tq] = r * r;
    }
    __syncthreads();
    if (t < NIw) {
        const size_t f = (size_t)w * NIw + t;
        double c2 = 0, m = 0;
        if (!d.imu_skip[f]) {
            for (int k = 0; k < 15; k++) c2 += sR2[t * 15 + k];
            for (int a = 0; a < 30; a++) m += sMod[t * 32 + a];
        }
        d.imu_cost_c[f] = 0.5 * c2;                 // no loss function on IMU factors
        d.imu_model[f] = m;
    } else if (t >= 64 && t < 64 + slots) {
        const int q = t - 64;
        double m = 0;
        for (int a = 0; a < 12; a++) m += sPm[q * 16 + a];
        d.prior_model[(size_t)w * slots + q] = m;
    }
    DSTAMP(53);
}
template <bool FUSED, bool EX> DEV void step_control_body(DevBatch &d, const int w, const int t, double *cl, double *red, int &s_accept);

// ------------------------------------------------------------------------------------------
// Candidate cost + model cost change (fixed-shape reductions), then TrustRegionMinimizer's step
// validity / tolerances / acceptance and DoglegStrategy's radius / mu update.  One workgroup per window.
// FUSED (LDS solver path): the reprojection factors' candidate cost and model cost change are evaluated HERE, one
// factor per thread in the same thread -> factor order the separate k_proj_linearize<1> pass summed them in, instead
// of being written to fcost_c / fmodel by a tile-grid kernel and read back: one launch and one round trip less per
// iteration.  Dynamic LDS: candidate poses [N][12] | extrinsic [12] | poses at x [N][12] | tangent step [N][6].
template <bool FUSED, bool EX>
DEV void step_control_body(DevBatch &d, const int w, const int t, double *cl, double *red, int &s_accept) {
    SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING) return;
    const int N = d.N;
    double S = 0, M = 0;
    if (st.step_valid) {
        double s = 0, m = 0;
        if (FUSED) {
            double *sC = cl, *sEx = cl + N * 12, *sX = sEx + 12, *sD = sX + N * 12, *sL = sD + N * 6;
            const int l0w = d.lm_off[w], Lww = d.lm_off[w + 1] - l0w;
            const bool stage = d.ctl_stage_lm != 0;
            if (stage) {          // what the factor loop gathers by landmark index: host point | candidate, current inverse depth | step
                for (int e = t; e < Lww; e += 256) {
                    const double *pp = d.lm_pts_i + (size_t)(l0w + e) * 3;
                    sL[6 * e] = pp[0]; sL[6 * e + 1] = pp[1]; sL[6 * e + 2] = pp[2];
                    sL[6 * e + 3] = d.clam[l0w + e]; sL[6 * e + 4] = d.lam[l0w + e]; sL[6 * e + 5] = d.delta_l[l0w + e];
                }
            }
            if (t < N) {
                const double *p = d.cpose + ((size_t)w * N + t) * 7;
                double R[9]; q_to_R(q_from_pose(p), R);
#pragma unroll
                for (int k = 0; k < 9; k++) sC[t * 12 + k] = R[k];
                sC[t * 12 + 9] = p[0]; sC[t * 12 + 10] = p[1]; sC[t * 12 + 11] = p[2];
            } else if (t >= 64 && t < 64 + N) {
                const int fr = t - 64;
                const double *p = d.pose + ((size_t)w * N + fr) * 7;
                double R[9]; q_to_R(q_from_pose(p), R);
#pragma unroll
                for (int k = 0; k < 9; k++) sX[fr * 12 + k] = R[k];
                sX[fr * 12 + 9] = p[0]; sX[fr * 12 + 10] = p[1]; sX[fr * 12 + 11] = p[2];
                const double *dp = d.delta_p + (size_t)w * d.np + 15 * fr;
#pragma unroll
                for (int k = 0; k < 6; k++) sD[fr * 6 + k] = dp[k];
            } else if (t == 128) {
                const double *e = d.ex + (size_t)w * 7;
                double R[9]; q_to_R(q_from_pose(e), R);
#pragma unroll
                for (int k = 0; k < 9; k++) sEx[k] = R[k];
                sEx[9] = e[0]; sEx[10] = e[1]; sEx[11] = e[2];
            }
            __syncthreads();
            // the extrinsic: constant (d.ex), or -- when it is estimated -- the pseudo-frame's pose block, candidate and x
            const double *exC = EX ? sC + d.Nr * 12 : sEx, *exX = EX ? sX + d.Nr * 12 : sEx;
            double ric[9], tic[3];
#pragma unroll
            for (int k = 0; k < 9; k++) ric[k] = exC[k];
#pragma unroll
            for (int k = 0; k < 3; k++) tic[k] = exC[9 + k];
            // the factor records and observations are read one trip ahead (nothing in them depends on the arithmetic)
            const int f_end = d.f_off[w + 1];
            FactorRec rec_n = {0, 0}; double2 pj_n = make_double2(0, 0);
            auto issue = [&](int f) { const int fc = f < f_end ? f : f_end - 1; rec_n = d.f_rec[fc]; pj_n = *reinterpret_cast<const double2 *>(d.f_pts_j + (size_t)fc * 2); };
            if (d.f_off[w] < f_end) issue(d.f_off[w] + t);
            for (int f = d.f_off[w] + t; f < f_end; f += 256) {
                const FactorRec rec = rec_n;
                const double2 pj = pj_n;
                if (f + 256 < f_end) issue(f + 256);
                const int fi = rec.ij & 255, fj = (rec.ij >> 8) & 255;
                double pi3[3], clam_l, lam_l, dl_l;
                if (stage) {
                    const double *q6 = sL + 6 * (rec.lm - l0w);
                    pi3[0] = q6[0]; pi3[1] = q6[1]; pi3[2] = q6[2]; clam_l = q6[3]; lam_l = q6[4]; dl_l = q6[5];
                } else {
                    const double *pp = d.lm_pts_i + (size_t)rec.lm * 3;
                    pi3[0] = pp[0]; pi3[1] = pp[1]; pi3[2] = pp[2]; clam_l = d.clam[rec.lm]; lam_l = d.lam[rec.lm]; dl_l = d.delta_l[rec.lm];
                }
                double Ri[9], Rj[9], Pi[3], Pj[3], r0, r1, Ji[12], Jj[12], Jl[2];
#pragma unroll
                for (int k = 0; k < 9; k++) { Ri[k] = sC[fi * 12 + k]; Rj[k] = sC[fj * 12 + k]; }
#pragma unroll
                for (int k = 0; k < 3; k++) { Pi[k] = sC[fi * 12 + 9 + k]; Pj[k] = sC[fj * 12 + 9 + k]; }
                proj_factor<false>(Ri, Pi, Rj, Pj, ric, tic, d.proj_sqrt_info, clam_l, pi3[0], pi3[1], pi3[2], pj.x, pj.y, r0, r1, Ji, Jj, Jl);
                s += 0.5 * log(1.0 + (r0 * r0 + r1 * r1));          // CauchyLoss(1.0): rho = log(1 + s)
                // model cost change piece (J delta)^T (r + J delta / 2) at x by the directional derivative
                double rx0, rx1, m0, m1;
#pragma unroll
                for (int k = 0; k < 9; k++) { Ri[k] = sX[fi * 12 + k]; Rj[k] = sX[fj * 12 + k]; }
#pragma unroll
                for (int k = 0; k < 3; k++) { Pi[k] = sX[fi * 12 + 9 + k]; Pj[k] = sX[fj * 12 + 9 + k]; }
                if (EX) {
                    double ricX[9], ticX[3];
#pragma unroll
                    for (int k = 0; k < 9; k++) ricX[k] = exX[k];
#pragma unroll
                    for (int k = 0; k < 3; k++) ticX[k] = exX[9 + k];
                    proj_residual_dir_ex(Ri, Pi, Rj, Pj, ricX, ticX, d.proj_sqrt_info, lam_l, pi3[0], pi3[1], pi3[2], pj.x, pj.y,
                                         sD + fi * 6, sD + fj * 6, sD + d.Nr * 6, dl_l, rx0, rx1, m0, m1);
                } else
                proj_residual_dir(Ri, Pi, Rj, Pj, ric, tic, d.proj_sqrt_info, lam_l, pi3[0], pi3[1], pi3[2], pj.x, pj.y,
                                  sD + fi * 6, sD + fj * 6, dl_l, rx0, rx1, m0, m1);
                const double rp = 1.0 / (1.0 + (rx0 * rx0 + rx1 * rx1));       // corrector: r, J scaled by sqrt(rho')
                m += rp * (m0 * (rx0 + m0 / 2.0) + m1 * (rx1 + m1 / 2.0));
            }
        } else {
            // (round 4: eight trips' loads in flight -- clamped addresses, masked adds in the same order: same bits; one load per
            // trip made the 30 000-factor window of BASELINE config 5 wait a memory latency 117 times: 36 us per launch)
            const int f_end = d.f_off[w + 1];
            for (int f0 = d.f_off[w] + t; f0 < f_end; f0 += 8 * 256) {
                double cs[8], ms[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int f = f0 + 256 * u < f_end ? f0 + 256 * u : f_end - 1; cs[u] = d.fcost_c[f]; ms[u] = d.fmodel[f]; }
#pragma unroll
                for (int u = 0; u < 8; u++) if (f0 + 256 * u < f_end) { s += cs[u]; m += ms[u]; }
            }
        }
        for (int i = t; i < N - 1; i += 256) { s += d.imu_cost_c[(size_t)w * (N - 1) + i]; m += d.imu_model[(size_t)w * (N - 1) + i]; }
        for (int i = t; i < d.n_prior_slots; i += 256) { s += d.prior_cost_c[(size_t)w * d.n_prior_slots + i]; m += d.prior_model[(size_t)w * d.n_prior_slots + i]; }
        double sm[2] = {s, m};
        block_sums<2>(sm, red, t);
        S = sm[0]; M = sm[1];
    }
    if (t == 0) {
        s_accept = 0;
        const int it = st.iteration;
        double *tc = d.trace_cost + (size_t)w * ISV_MAX_TRACE, *tr = d.trace_radius + (size_t)w * ISV_MAX_TRACE;
        double *ts = d.trace_step + (size_t)w * ISV_MAX_TRACE; int32_t *ta = d.trace_acc + (size_t)w * ISV_MAX_TRACE;
        const double model_cost_change = -M;
        const bool valid = st.step_valid && (model_cost_change > 0.0) && !(it <= d.force_invalid);
        d.cost_c[w] = S; d.model[w] = M;
        if (!valid) {                                       // HandleInvalidStep
            st.invalid += 1;
            if (st.invalid >= 5) st.termination = st.ls_fail ? ISV_TERM_LINEAR_SOLVER : ISV_TERM_INVALID_STEPS;
            else {
                st.mu *= 10.0; st.reuse = 0; st.need_linearize = 1;     // StepIsInvalid: redo the GN solve with a larger mu
                tc[it] = st.x_cost; tr[it] = st.radius; ts[it] = 0; ta[it] = 0;
                if (it >= d.max_iter) st.termination = ISV_TERM_MAX_ITERATIONS;
            }
        } else {
            st.invalid = 0;
            const double cand_cost = S, step_norm = st.step_norm;
            if (step_norm <= 1e-8 * (st.x_norm + 1e-8)) {
                st.termination = ISV_TERM_PARAMETER_TOL; tc[it] = st.x_cost; tr[it] = st.radius; ts[it] = step_norm; ta[it] = 0;
            } else if (fabs(st.x_cost - cand_cost) <= 1e-6 * st.x_cost) {
                st.termination = ISV_TERM_FUNCTION_TOL; tc[it] = st.x_cost; tr[it] = st.radius; ts[it] = step_norm; ta[it] = 0;
            } else {
                const double rel = (st.x_cost - cand_cost) / model_cost_change;
                if (rel > 1e-3) {                           // HandleSuccessfulStep
                    s_accept = 1;
                    st.x_cost = cand_cost;                   // refreshed by the next linearisation
                    if (rel < 0.25) st.radius *= 0.5;
                    if (rel > 0.75) st.radius = fmax(st.radius, 3.0 * st.dogleg_step_norm);
                    st.mu = fmax(1e-8, 2.0 * st.mu / 10.0);
                    st.reuse = 0; st.need_linearize = 1; st.num_successful += 1;
                    tc[it] = cand_cost; tr[it] = st.radius; ts[it] = step_norm; ta[it] = 1;
                } else {                                    // StepRejected
                    st.radius *= 0.5; st.reuse = 1;
                    tc[it] = cand_cost; tr[it] = st.radius; ts[it] = step_norm; ta[it] = 0;
                }
                // FinalizeIterationAndCheckIfMinimizerCanContinue (gradient tolerance: k_build_solve)
                if (it >= d.max_iter) st.termination = ISV_TERM_MAX_ITERATIONS;
                else if (st.radius <= d.min_radius) st.termination = ISV_TERM_MIN_RADIUS;
            }
        }
    }
    __syncthreads();
    if (s_accept) {
        const int l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
        for (int i = t; i < N * 7; i += 256) d.pose[(size_t)w * N * 7 + i] = d.cpose[(size_t)w * N * 7 + i];
        for (int i = t; i < N * 9; i += 256) d.sb[(size_t)w * N * 9 + i] = d.csb[(size_t)w * N * 9 + i];
        for (int l = l0 + t; l < l1; l += 256) d.lam[l] = d.clam[l];
    }
}
