// isv_lin_gram.h -- the body of k_lin_gram (see isv_visual.hip for the description), as a device routine shared by the
// kernels that run it: k_lin_gram (isv_visual.hip) and k_lin_gram_chain (isv_build_solve_sb.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "isv_kernels.h"
#include "isv_device_math.h"
#include "isv_proj_factor.h"

__host__ __device__ inline int tvis_col2(int a, int N) { return 36 * (a * N - a * (a - 1) / 2); }
typedef double double4g __attribute__((ext_vector_type(4)));
// LDS row of one factor: r(2) J_i(12) J_j(12) [+ J_ex(12) when the extrinsic is estimated] + 1 (odd stride)
__host__ __device__ constexpr int lg_xld(bool ex) { return ex ? 39 : 27; }
#define LGSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// LGW wavefronts per window: 4 for batches (158 VGPRs -> three workgroups per CU), 8 while every window has a CU of its own
// (twice the wavefronts on the factor stream; which wavefront sums a pair group does not change the sums)
// (a device routine since round 5: k_lin_gram runs it alone, k_lin_gram_chain -- isv_build_solve_sb.hip -- beside the chain half of the split solve)
template <bool EX, int LGW>
DEV void lin_gram_body(const DevBatch &d, double *lds) {
    constexpr int LG_XLD = lg_xld(EX);
    const int w = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || !st.need_linearize) return;
#ifdef ISV_STAMP
    // phase stamps of wavefront 0 (scripts/stamp_bs.py), accumulated in registers and written once at the end: a
    // read-modify-write of d.dbg per stamp would cost a memory round trip of its own
    unsigned long long t_last = wall_clock64(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int st_slot[12] = {24, 25, 26, 27, 28, 29, 30, 57, 58, 59, 60, 61};
#define LGSTAMP(k) do { unsigned long long now_ = wall_clock64(); _Pragma("unroll") for (int q_ = 0; q_ < 12; q_++) if (st_slot[q_] == (k)) st_acc[q_] += now_ - t_last; t_last = now_; } while (0)
#else
#define LGSTAMP(k) do {} while (0)
#endif
    // N: the REAL frames (pairs, pose blocks of the factors); Nd: pose blocks of the reduced system (+ the extrinsic's
    // pseudo-frame, index N, when it is estimated)
    const int N = d.Nr, Nd = d.N, NP = N * (N - 1) / 2;
    double *sPose = lds;                               // [N][12] R (row-major) | P
    double *sEx = sPose + N * 12;                      // [12]
    double *sX = sEx + 12 + wv * 16 * LG_XLD;          // this wavefront's 16-factor tile
    double *pbase = sEx + 12 + LGW * 16 * LG_XLD;
    double *Pjj = d.sw_global ? d.sw_part + (size_t)w * NP * 84 : pbase;      // pair partials (see k_sweep_mfma)
    double *Phh = Pjj + NP * 36, *Pgj = Phh + NP * 36, *Pgh = Pgj + NP * 6;
    int *offL = (int *)(d.sw_global ? pbase : pbase + NP * 84);                // [NP + 1] group starts
    double *after_off = (double *)offL + (NP + 2) / 2 + 1;
    int *sSched = (int *)after_off;                                            // [NP] the pair -> wavefront schedule (h | j << 8 | p << 16)
    double *sLam = after_off + (NP + 1) / 2 + 1;                               // [lg_lcap] inverse depths of the window's landmarks at x
    double *sPts = sLam + d.lg_lcap;                                           // [lg_lcap][3] host-frame points
    const int fw0 = d.f_off[w], l0 = d.lm_off[w], Lw = d.lm_off[w + 1] - l0;
    const int2 *prec = (const int2 *)d.pg_rec + fw0;   // the factor stream: wavefront-major, then pair group (upload order)
    const double2 *ppts = (const double2 *)d.pg_pts + fw0;
    const int *sched = sSched, *soff = d.pg_sched_off + (size_t)w * (ISV_SWEEP_WAVES + 1);
    double *out = d.Tvis + (size_t)w * d.tvis_sz;
    if (t < N) {
        const double *p = d.pose + ((size_t)w * Nd + t) * 7;
        double R[9]; q_to_R(q_from_pose(p), R);
#pragma unroll
        for (int k = 0; k < 9; k++) sPose[t * 12 + k] = R[k];
        sPose[t * 12 + 9] = p[0]; sPose[t * 12 + 10] = p[1]; sPose[t * 12 + 11] = p[2];
    } else if (t == 64) {
        const double *e = EX ? d.pose + ((size_t)w * Nd + N) * 7 : d.ex + (size_t)w * 7;      // the estimated extrinsic lives in the pseudo-frame's pose block
        double R[9]; q_to_R(q_from_pose(e), R);
#pragma unroll
        for (int k = 0; k < 9; k++) sEx[k] = R[k];
        sEx[9] = e[0]; sEx[10] = e[1]; sEx[11] = e[2];
    }
    for (int e = t; e <= NP; e += (64 * LGW)) offL[e] = d.pg_off[(size_t)w * (NP + 1) + e];
    // (round 3) everything a factor GATHERS by landmark index -- the inverse depth and the host point -- and the schedule
    // the group bookkeeping reads are staged once, with coalesced loads: inside the factor loop they were dependent
    // global loads (record -> landmark -> depth / point; one sched[] word per finished group), a memory latency each,
    // on a kernel whose chunks otherwise compute from registers and LDS
    for (int e = t; e < NP; e += (64 * LGW)) sSched[e] = d.pg_sched[(size_t)w * NP + e];
    for (int e = t; e < Lw; e += (64 * LGW)) sLam[e] = d.lam[l0 + e];
    for (int e = t; e < 3 * Lw; e += (64 * LGW)) sPts[e] = d.lm_pts_i[(size_t)l0 * 3 + e];
    __syncthreads();
    LGSTAMP(24);
    const int i = lane & 15, kq = lane >> 4, row2 = kq & 1, fsel = kq >> 1;
    // element of the LDS factor row that operand column i takes: J_i row row2 | J_j row row2 | r[row2]
    const int eoff = i < 6 ? 2 + row2 * 6 + i : (i < 12 ? 14 + row2 * 6 + (i - 6) : row2);
    const bool colok = i < 13;
    // sweep-schedule slices [sl0, sl1) of this wavefront (an even split when LGW divides ISV_SWEEP_WAVES, else 2-3-3 ...)
    const int sl0 = (wv * ISV_SWEEP_WAVES) / LGW, sl1 = ((wv + 1) * ISV_SWEEP_WAVES) / LGW;
    const int q0 = soff[sl0], q1 = soff[sl1];
    const int *wst = d.pg_wstart + (size_t)w * (ISV_SWEEP_WAVES + 1);
    const int s0 = wst[sl0], s1 = wst[sl1];            // this wavefront's slice of the factor stream
    auto gsize = [&](int qq) { const int pp = sched[qq] >> 16; return offL[pp + 1] - offL[pp]; };
    // one group's accumulator tile -> its five pieces.  C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    double *exp_w = EX ? d.ex_part + (size_t)w * NP * 114 : nullptr;
    auto flush_ex = [&](int qq, const double4g &a2, const double4g &a3) {       // J_ex^T [J_i | J_j | r] and J_ex^T J_ex of one group
        const int p = sched[qq] >> 16;
#pragma unroll
        for (int reg = 0; reg <= 1; reg++) {
            const int row = kq + 4 * reg;
            if (row < 6) {
                if (i < 6) { exp_w[p * 114 + row * 6 + i] = a2[reg]; exp_w[p * 114 + 72 + row * 6 + i] = a3[reg]; }
                else if (i < 12) exp_w[p * 114 + 36 + row * 6 + (i - 6)] = a2[reg];
                else if (i == 12) exp_w[p * 114 + 108 + row] = a2[reg];
            }
        }
    };
    auto flush = [&](int qq, const double4g &acc) {
        const int rec = sched[qq], h = rec & 255, j = (rec >> 8) & 255, p = rec >> 16;
#pragma unroll
        for (int reg = 0; reg <= 2; reg++) {
            const int row = kq + 4 * reg;
            if (row < 6) {
                if (i < 6) Phh[p * 36 + row * 6 + i] = acc[reg];
                else if (i == 12) Pgh[p * 6 + row] = acc[reg];
            } else if (row < 12) {
                const int rr = row - 6;
                if (i < 6) out[tvis_col2(h, Nd) + (j - h) * 36 + rr * 6 + i] = acc[reg];        // block (j, h)
                else if (i < 12) Pjj[p * 36 + rr * 6 + (i - 6)] = acc[reg];
                else if (i == 12) Pgj[p * 6 + rr] = acc[reg];
            }
        }
    };
    // pairs nobody observes still own a slot of every sum below: zero them
    for (int q = q0; q < q1; q++) if (gsize(q) == 0) { flush(q, double4g{0, 0, 0, 0}); if (EX) flush_ex(q, double4g{0, 0, 0, 0}, double4g{0, 0, 0, 0}); }
    // The factor stream of this wavefront is its pair groups back to back (upload order), visited in FULL 64-lane
    // chunks: a chunk may span several groups, every lane reads its own pair's pose blocks.  The Gram accumulator
    // follows the group boundaries: rows of the 16-factor LDS tile outside the current group are masked to zero.
    int q = q0;
    while (q < q1 && gsize(q) == 0) q++;
    int gend = s0 + (q < q1 ? gsize(q) : 0);
    double4g acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    const int eoffx = 26 + row2 * 6 + (i < 6 ? i : 0);  // J_ex row row2, column i of the LDS factor row
    // the factor stream (record + observing view's point) is loaded ONE CHUNK AHEAD: nothing in it depends on the arithmetic
    int2 rc_n = make_int2(0, 0); double2 pj_n = make_double2(0, 0);
    auto issue = [&](int pos) { const int p = pos + lane < s1 ? pos + lane : s1 - 1; rc_n = prec[p]; pj_n = ppts[p]; };
    if (s0 < s1) issue(s0);
    LGSTAMP(25);
    for (int pos = s0; pos < s1; pos += 64) {
        const int cnt = (s1 - pos) < 64 ? (s1 - pos) : 64;
        double r0 = 0, r1 = 0, Ji[12], Jj[12], Jl[2], Jex[12];
        const int2 rc = rc_n;                              // {global landmark, f_rel | h << 16 | j << 24}
        const double2 pj = pj_n;
        if (pos + 64 < s1) issue(pos + 64);
        if (lane < cnt) {
            const int h = (rc.y >> 16) & 255, j = (rc.y >> 24) & 255;
            double ric[9], tic[3], Ri[9], Rj[9], Pi[3], Pj[3];
#pragma unroll
            for (int k = 0; k < 9; k++) { ric[k] = sEx[k]; Ri[k] = sPose[h * 12 + k]; Rj[k] = sPose[j * 12 + k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) { tic[k] = sEx[9 + k]; Pi[k] = sPose[h * 12 + 9 + k]; Pj[k] = sPose[j * 12 + 9 + k]; }
            const double *pi3 = sPts + (rc.x - l0) * 3;
            const double lam_l = sLam[rc.x - l0];
            proj_factor<true>(Ri, Pi, Rj, Pj, ric, tic, d.proj_sqrt_info, lam_l, pi3[0], pi3[1], pi3[2], pj.x, pj.y, r0, r1, Ji, Jj, Jl);
            // CauchyLoss(1.0): rho = log(1 + s); the Corrector scales r and J by sqrt(rho') = 1 / sqrt(1 + s)
            const double sum = 1.0 + (r0 * r0 + r1 * r1);
            const double sc = sqrt(fmax(1.0 / sum, 2.2250738585072014e-308));
            const size_t f = (size_t)fw0 + (rc.y & 0xffff);
            d.fcost[f] = 0.5 * log(sum);
            r0 *= sc; r1 *= sc;
#pragma unroll
            for (int k = 0; k < 12; k++) { Ji[k] *= sc; Jj[k] *= sc; }
            Jl[0] *= sc; Jl[1] *= sc;
            // observing frame's w = J_j^T J_l into the landmark's packed W slot (observation index f + landmark + 1)
            double2 *wd = (double2 *)(d.W + (f + rc.x + 1) * 6);
            wd[0] = make_double2(Jj[0] * Jl[0] + Jj[6] * Jl[1], Jj[1] * Jl[0] + Jj[7] * Jl[1]);
            wd[1] = make_double2(Jj[2] * Jl[0] + Jj[8] * Jl[1], Jj[3] * Jl[0] + Jj[9] * Jl[1]);
            wd[2] = make_double2(Jj[4] * Jl[0] + Jj[10] * Jl[1], Jj[5] * Jl[0] + Jj[11] * Jl[1]);
            // the factor's pieces of the landmark scalars: E, g, host-frame w
            double2 *fl = (double2 *)(d.flm + f * 8);
            fl[0] = make_double2(Jl[0] * Jl[0] + Jl[1] * Jl[1], Jl[0] * r0 + Jl[1] * r1);
            fl[1] = make_double2(Ji[0] * Jl[0] + Ji[6] * Jl[1], Ji[1] * Jl[0] + Ji[7] * Jl[1]);
            fl[2] = make_double2(Ji[2] * Jl[0] + Ji[8] * Jl[1], Ji[3] * Jl[0] + Ji[9] * Jl[1]);
            fl[3] = make_double2(Ji[4] * Jl[0] + Ji[10] * Jl[1], Ji[5] * Jl[0] + Ji[11] * Jl[1]);
            if (EX) {                                   // J_ex (projection_factor.cpp:100-111), same corrector scale; its piece of the landmark's w
                proj_jac_ex(Ri, Pi, Rj, Pj, ric, tic, d.proj_sqrt_info, lam_l, pi3[0], pi3[1], pi3[2], Jex);
#pragma unroll
                for (int k = 0; k < 12; k++) Jex[k] *= sc;
                double2 *fx = (double2 *)(d.flmx + f * 6);
                fx[0] = make_double2(Jex[0] * Jl[0] + Jex[6] * Jl[1], Jex[1] * Jl[0] + Jex[7] * Jl[1]);
                fx[1] = make_double2(Jex[2] * Jl[0] + Jex[8] * Jl[1], Jex[3] * Jl[0] + Jex[9] * Jl[1]);
                fx[2] = make_double2(Jex[4] * Jl[0] + Jex[10] * Jl[1], Jex[5] * Jl[0] + Jex[11] * Jl[1]);
            }
        }
        LGSTAMP(26);
        // Gram: 16 factors per round through the wave-private LDS tile
        for (int rq = 0; rq * 16 < cnt; rq++) {
            if ((lane >> 4) == rq) {
                double *row = sX + (lane & 15) * LG_XLD;
                row[0] = r0; row[1] = r1;
#pragma unroll
                for (int k = 0; k < 12; k++) { row[2 + k] = Ji[k]; row[14 + k] = Jj[k]; }
                if (EX) {
#pragma unroll
                    for (int k = 0; k < 12; k++) row[26 + k] = Jex[k];
                }
            }
            LGSYNC();
            LGSTAMP(57);
            double v[8], vx[8];
#pragma unroll
            for (int u2 = 0; u2 < 8; u2++) v[u2] = colok ? sX[(2 * u2 + fsel) * LG_XLD + eoff] : 0.0;
            if (EX) {
#pragma unroll
                for (int u2 = 0; u2 < 8; u2++) vx[u2] = i < 6 ? sX[(2 * u2 + fsel) * LG_XLD + eoffx] : 0.0;
            }
            LGSTAMP(58);
            const int rs = pos + 16 * rq, re = (rs + 16) < (pos + cnt) ? (rs + 16) : (pos + cnt);
            int cur = rs;
            while (cur < re) {                              // the segments of this round, one per group it touches
                const int se = gend < re ? gend : re;
                const int a = cur - rs, b = se - rs;        // tile rows [a, b) belong to the current group
#pragma unroll
                for (int u2 = 0; u2 < 8; u2++) {
                    if (2 * u2 + 1 >= a && 2 * u2 < b) {
                        const int src = 2 * u2 + fsel;
                        const double x = (src >= a && src < b) ? v[u2] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc, 0, 0, 0);
                        if (EX) {                       // D[r][c] = sum_k A[k][r] B[k][c]: rows = J_ex columns
                            const double xe = (src >= a && src < b) ? vx[u2] : 0.0;
                            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(xe, x, acc2, 0, 0, 0);
                            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(xe, xe, acc3, 0, 0, 0);
                        }
                    }
                }
                cur = se;
                if (se == gend) {                           // the group is complete
                    flush(q, acc);
                    acc = double4g{0, 0, 0, 0};
                    if (EX) { flush_ex(q, acc2, acc3); acc2 = double4g{0, 0, 0, 0}; acc3 = double4g{0, 0, 0, 0}; }
                    q++;
                    while (q < q1 && gsize(q) == 0) q++;
                    gend += q < q1 ? gsize(q) : 0;
                }
            }
            LGSTAMP(59);
            LGSYNC();
        }
        LGSTAMP(27);
    }
    LGSTAMP(28);
    __syncthreads();
    LGSTAMP(29);
    // fold the pair partials into the diagonal blocks, the Jacobi-scaling diagonal and the gradient (fixed order)
    const int tail = 36 * (Nd * (Nd + 1) / 2);
    auto pidx = [N](int hh, int jj) { return hh * N - hh * (hh + 1) / 2 + (jj - hh - 1); };
    if (EX) {
        // the extrinsic's block row of the pose system: (ex, a) for every real frame a, (ex, ex), its diagonal and gradient
        for (int tq = t; tq < (N + 1) * 36 + 6; tq += (64 * LGW)) {
            double sum = 0.0;
            if (tq < N * 36) {
                const int a = tq / 36, rc = tq - 36 * a;
                for (int j2 = a + 1; j2 < N; j2++) sum += exp_w[pidx(a, j2) * 114 + rc];
                for (int h2 = 0; h2 < a; h2++) sum += exp_w[pidx(h2, a) * 114 + 36 + rc];
                out[tvis_col2(a, Nd) + (N - a) * 36 + rc] = sum;
            } else if (tq < (N + 1) * 36) {
                const int rc = tq - N * 36;
                for (int p2 = 0; p2 < NP; p2++) sum += exp_w[p2 * 114 + 72 + rc];
                out[tvis_col2(N, Nd) + rc] = sum;
                if (rc / 6 == rc % 6) out[tail + 6 * N + rc / 6] = sum;
            } else {
                const int r = tq - (N + 1) * 36;
                for (int p2 = 0; p2 < NP; p2++) sum += exp_w[p2 * 114 + 108 + r];
                out[tail + 6 * Nd + 6 * N + r] = sum;
            }
        }
    }
    for (int tq = t; tq < N * 42; tq += (64 * LGW)) {
        if (tq < N * 36) {
            const int a = tq / 36, rc = tq - 36 * a, r = rc / 6, c = rc - 6 * r;
            double s = 0.0;
            for (int j2 = a + 1; j2 < N; j2++) s += Phh[pidx(a, j2) * 36 + rc];
            for (int h2 = 0; h2 < a; h2++) s += Pjj[pidx(h2, a) * 36 + rc];
            out[tvis_col2(a, Nd) + rc] = s;
            if (r == c) out[tail + 6 * a + r] = s;
        } else {
            const int q = tq - N * 36, a = q / 6, r = q - 6 * a;
            double s = 0.0;
            for (int j2 = a + 1; j2 < N; j2++) s += Pgh[pidx(a, j2) * 6 + r];
            for (int h2 = 0; h2 < a; h2++) s += Pgj[pidx(h2, a) * 6 + r];
            out[tail + 6 * Nd + 6 * a + r] = s;
        }
    }
    LGSTAMP(30);
#ifdef ISV_STAMP
    if (t == 0) for (int q_ = 0; q_ < 12; q_++) d.dbg[(size_t)w * 64 + st_slot[q_]] += (double)st_acc[q_];
#endif
}
