// isv_prior_factor.h -- the prior factors of a window (SE3Prior / Linear9 / RelativePose / RollPitch) as one
// wavefront-sized routine, shared by k_prior_linearize (isv_linearize.hip) and k_dogleg (isv_solver.hip).
#pragma once
#include "isv_device_types.h"
#include "isv_device_math.h"

// ------------------------------------------------------------------------------------------
// Prior factors: one wavefront per window.  Slot 0 = SE3 prior, 1 = Linear9, 2..Nvo = relative pose, then
// roll-pitch.  Phase 1: one lane per prior forms the raw residual and raw Jacobian blocks in LDS (the
// SO(3) log / right-Jacobian calls are shared by the SE3 and relative-pose lanes).  Phase 2: all lanes form
// sqrt_info * [raw r | raw J] entry by entry.  Phase 3: CauchyLoss corrector and strips.  Phase 4: J^T J
// pairs and J^T r for k_build_solve.  Same operation order per entry as a scalar evaluation, so the
// numbers do not depend on the lane split.
#define PRL_RAW 82      // per-slot LDS: raw r (9) | raw J (72)
#define PRL_W 90        // per-slot LDS: r (9) | J (81)
#define PRL_S 82        // per-slot LDS: sqrt_info (<= 81)
// unweighted prior Jacobians at a pose (EvaluateOnlyJacobians of the reference factors), 6-column form
DEV void relpose_jac(const double *dt, const double *dR, const double *pi, const double *pj, double *res, double *Ji, double *Jj) {
    Quat Qi = q_from_pose(pi), Qj = q_from_pose(pj);
    double Ri[9], Rj[9], dd[3], qd[3], M1[9], M2[9], lg[3], Jr[9], S[9], nJ[9], T1[9], T2[9];
    q_to_R(Qi, Ri); q_to_R(Qj, Rj);
    for (int k = 0; k < 3; k++) dd[k] = pj[k] - pi[k];
    q_rot(q_inv(Qi), dd, qd);
    m3_mul_nt(dR, Rj, M1); m3_mul(M1, Ri, M2);
    so3_log(q_from_R(M2), lg);
    for (int k = 0; k < 3; k++) { res[k] = dt[k] - qd[k]; res[3 + k] = lg[k]; }
    so3_rjac_inv(lg, Jr); skew3(qd, S);
    for (int k = 0; k < 36; k++) { Ji[k] = 0; Jj[k] = 0; }
    for (int k = 0; k < 9; k++) nJ[k] = -Jr[k];
    m3_mul_nt(nJ, Ri, T1); m3_mul(T1, Rj, T2);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        Ji[a * 6 + b] = Ri[b * 3 + a]; Ji[a * 6 + 3 + b] = -S[a * 3 + b]; Ji[(3 + a) * 6 + 3 + b] = Jr[a * 3 + b];
        Jj[a * 6 + b] = -Ri[b * 3 + a]; Jj[(3 + a) * 6 + 3 + b] = T2[a * 3 + b];
    }
}


// The window's prior factor records are STAGED IN LDS once per call (round 3): one coalesced pass instead of field-by-field
// global loads on the serial path of a lane per prior -- every `f.R`, `f.delta_t`, `sqrt_info[row * dim + k]`, `index` used to
// be a memory round trip of its own (the candidate evaluation of 13 priors took 20 us inside k_dogleg, on ONE wavefront the
// other three waited for).  Layout (8-byte words): SE3 | Linear9 | relative pose x (Nvo - 1) | roll/pitch x max_rp.
struct PriorRecs {
    const isv_se3_prior_t *se3; const isv_linear9_t *lin9; const isv_relpose_t *relpose; const isv_rollpitch_t *rollpitch;
};
DEV PriorRecs prior_recs_at(const DevBatch &d, const double *dst) {          // the staged records' addresses (no copy)
    constexpr int SW = sizeof(isv_se3_prior_t) / 8, LW = sizeof(isv_linear9_t) / 8, RW = sizeof(isv_relpose_t) / 8;
    const uint64_t *o = (const uint64_t *)dst;
    const int n0 = SW, n1 = n0 + LW, n2 = n1 + RW * (d.Nvo - 1);
    PriorRecs r;
    r.se3 = (const isv_se3_prior_t *)o; r.lin9 = (const isv_linear9_t *)(o + n0);
    r.relpose = (const isv_relpose_t *)(o + n1); r.rollpitch = (const isv_rollpitch_t *)(o + n2);
    return r;
}
DEV PriorRecs prior_stage_records(const DevBatch &d, int w, double *dst, int t, int nthr) {
    constexpr int SW = sizeof(isv_se3_prior_t) / 8, LW = sizeof(isv_linear9_t) / 8, RW = sizeof(isv_relpose_t) / 8, PW = sizeof(isv_rollpitch_t) / 8;
    const int nrel = d.Nvo - 1;
    uint64_t *o = (uint64_t *)dst;
    const uint64_t *a = (const uint64_t *)(d.se3 + w), *b = (const uint64_t *)(d.lin9 + w);
    const uint64_t *c = (const uint64_t *)(d.relpose + (size_t)w * nrel), *e = (const uint64_t *)(d.rollpitch + (size_t)w * d.max_rp);
    const int n0 = SW, n1 = n0 + LW, n2 = n1 + RW * nrel, n3 = n2 + PW * d.max_rp;
    // two words per trip, the source chosen by address (one load instruction each, no branch), both loads before the stores
    for (int k0 = t; k0 < n3; k0 += 2 * nthr) {
        uint64_t v[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int k = (k0 + u * nthr) < n3 ? k0 + u * nthr : n3 - 1;
            const uint64_t *src = k < n0 ? a + k : (k < n1 ? b + (k - n0) : (k < n2 ? c + (k - n1) : e + (k - n2)));
            v[u] = *src;
        }
#pragma unroll
        for (int u = 0; u < 2; u++) { const int k = k0 + u * nthr; if (k < n3) o[k] = v[u]; }
    }
    return prior_recs_at(d, dst);
}
struct PriorDesc { int kind, strip_off, H_off, valid; const double *S; };   // kind 0 SE3, 1 Linear9, 2 relpose, 3 rollpitch
DEV PriorDesc prior_desc(const DevBatch &d, const PriorRecs &R, int s, int n_rp) {
    PriorDesc p;
    if (s == 0) { p.kind = 0; p.strip_off = PR_SE3; p.H_off = PH_SE3; p.valid = 1; p.S = R.se3->sqrt_info; }
    else if (s == 1) { p.kind = 1; p.strip_off = PR_LIN9; p.H_off = PH_LIN9; p.valid = 1; p.S = R.lin9->sqrt_info; }
    else if (s < 1 + d.Nvo) {
        const int i = s - 2;
        p.kind = 2; p.strip_off = PR_REL0 + PR_REL_SZ * i; p.H_off = PH_REL0 + PH_REL_SZ * i; p.valid = 1;
        p.S = R.relpose[i].sqrt_info;
    } else {
        const int m = s - 1 - d.Nvo;
        p.kind = 3; p.strip_off = PR_REL0 + PR_REL_SZ * (d.Nvo - 1) + PR_RP_SZ * m;
        p.H_off = PH_REL0 + PH_REL_SZ * (d.Nvo - 1) + PH_RP_SZ * m; p.valid = m < n_rp;
        p.S = R.rollpitch[p.valid ? m : 0].sqrt_info;
    }
    return p;
}
// phase 2 for one prior of shape DIM x (NBLK blocks of BW): wr = [S raw | S rawJ]
template <int DIM, int NBLK, int BW, bool COPY_S, bool JAC>
DEV void prior_weight(const double *S, const double *raw, const double *rawJ, double *wr, int t) {
    constexpr int per = 1 + (JAC ? NBLK * BW : 0);
    for (int e = t; e < DIM * per; e += 64) {
        const int row = e / per, c = e - row * per;
        double v = 0;
        if (c == 0) {
#pragma unroll
            for (int k = 0; k < DIM; k++) v += S[row * DIM + k] * raw[k];
            wr[row] = v;
        } else {
            const int col = c - 1, blk = col / BW, cc = col - blk * BW;
            if (COPY_S) v = S[row * DIM + cc];
            else {
#pragma unroll
                for (int k = 0; k < DIM; k++) v += S[row * DIM + k] * rawJ[blk * DIM * BW + k * BW + cc];
            }
            wr[9 + blk * DIM * BW + row * BW + cc] = v;
        }
    }
}
// phase 3: corrector scale of r and J in place + strip; returns 0.5 rho
template <int DIM, int NBLK, int BW, bool JAC, bool WAVE>
DEV double prior_correct(double *wr, double *strip_o, int t) {
    double ssum = 0;
#pragma unroll
    for (int k = 0; k < DIM; k++) ssum += wr[k] * wr[k];
    const double sum = 1.0 + ssum;
    if (JAC) {
        const double sc = sqrt(fmax(1.0 / sum, 2.2250738585072014e-308));
        if (WAVE) ISV_WSYNC(); else __syncthreads();          // every lane has read r before it is rescaled
        constexpr int nj = DIM * NBLK * BW;
        for (int e = t; e < DIM + nj; e += 64) {
            double *q = e < DIM ? wr + e : wr + 9 + (e - DIM);
            const double v = *q * sc;
            *q = v; strip_o[e] = v;
        }
    }
    return 0.5 * log(sum);
}
// phase 4: J^T J pairs (a >= b at a(a+1)/2 + b) then J^T r
template <int DIM, int NBLK, int BW>
DEV void prior_H(const double *wr, double *H, int t) {
    constexpr int ncol = NBLK * BW, npair = ncol * (ncol + 1) / 2;
    const double *J = wr + 9;
    for (int e = t; e < npair + ncol; e += 64) {
        double v = 0;
        if (e < npair) {
            int a = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            if ((a + 1) * (a + 2) / 2 <= e) a++;
            if (a * (a + 1) / 2 > e) a--;
            const int b = e - a * (a + 1) / 2;
            const double *Ja = J + (a / BW) * DIM * BW + (a % BW), *Jb = J + (b / BW) * DIM * BW + (b % BW);
#pragma unroll
            for (int k = 0; k < DIM; k++) v += Ja[k * BW] * Jb[k * BW];
        } else {
            const int a = e - npair;
            const double *Ja = J + (a / BW) * DIM * BW + (a % BW);
#pragma unroll
            for (int k = 0; k < DIM; k++) v += Ja[k * BW] * wr[k];
        }
        H[e] = v;
    }
}

// ---- phase 1: raw residual / raw Jacobian blocks, one lane per prior; `kinds`: bit k set = evaluate the priors of kind k
//      (k_dogleg splits the kinds over two wavefronts: the kinds of a wavefront's lanes diverge, i.e. run one after another)
//      (s0, sstride): lane t takes slots s0 + sstride (t + 64 k): several wavefronts can stride the slots (k_front)
template <bool JAC>
DEV void prior_phase1(const DevBatch &d, const PriorRecs &R, const double *pose_src, const double *sb_src, int w, int n_rp,
                      double *sRaw, int t, unsigned kinds, int s0 = 0, int sstride = 1) {
    constexpr int RAWS = JAC ? PRL_RAW : 10;
    const int slots = d.n_prior_slots, N = d.N;
    const double *poseW = pose_src + (size_t)w * N * 7;
    for (int s = s0 + sstride * t; s < slots; s += 64 * sstride) {
        double *raw = sRaw + s * RAWS, *rawJ = raw + 9;
        const int kind = s == 0 ? 0 : (s == 1 ? 1 : (s < 1 + d.Nvo ? 2 : 3));
        if (!((kinds >> kind) & 1u)) continue;
        Quat rr = Quat{1, 0, 0, 0};
        double Ri[9], Rj[9], qd[3], lg[3], Jr[9];
        const double *pi = poseW;
        if (kind == 0) {
            // SE3PriorFactor::Evaluate  se3_prior_factor.h:21-53
            Quat ri = q_normalized(q_from_pose(poseW)), rp = q_from_R(R.se3->R);
            rr = so3_mul(q_conj(rp), ri);
        } else if (kind == 2) {
            // RelativePoseFactor::Evaluate  relative_pose_factor.h:27-70
            const isv_relpose_t &f = R.relpose[s - 2];
            pi = poseW + (s - 2) * 7;
            const double *pj = pi + 7;
            Quat Qi = q_from_pose(pi), Qj = q_from_pose(pj);
            double dd[3], M1[9], M2[9];
            q_to_R(Qi, Ri); q_to_R(Qj, Rj);
#pragma unroll
            for (int k = 0; k < 3; k++) dd[k] = pj[k] - pi[k];
            q_rot(q_inv(Qi), dd, qd);
            m3_mul_nt(f.delta_R, Rj, M1); m3_mul(M1, Ri, M2);
            rr = q_from_R(M2);
        }
        so3_log(rr, lg);
        if (JAC) so3_rjac_inv(lg, Jr);
        if (kind == 0) {
            const isv_se3_prior_t &f = *R.se3;
#pragma unroll
            for (int k = 0; k < 3; k++) { raw[k] = poseW[k] - f.t[k]; raw[3 + k] = lg[k]; }
            if (JAC) {
#pragma unroll
                for (int k = 0; k < 36; k++) rawJ[k] = 0;
                rawJ[0] = rawJ[7] = rawJ[14] = 1.0;
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) rawJ[(3 + a) * 6 + 3 + b] = Jr[a * 3 + b];
            }
        } else if (kind == 1) {
            // Linear9Factor::Evaluate  linear9_factor.h:20-44 (Jacobian = sqrt_info)
            const isv_linear9_t &f = *R.lin9;
            const double *sb = sb_src + ((size_t)w * N + (d.Nvo - 1)) * 9;
#pragma unroll
            for (int k = 0; k < 9; k++) raw[k] = sb[k] - f.VB[k];
        } else if (kind == 2) {
            const isv_relpose_t &f = R.relpose[s - 2];
#pragma unroll
            for (int k = 0; k < 3; k++) { raw[k] = f.delta_t[k] - qd[k]; raw[3 + k] = lg[k]; }
            if (JAC) {
                double S[9], T1[9], T2[9], nJ[9];
                skew3(qd, S);
#pragma unroll
                for (int k = 0; k < 72; k++) rawJ[k] = 0;
#pragma unroll
                for (int k = 0; k < 9; k++) nJ[k] = -Jr[k];
                m3_mul_nt(nJ, Ri, T1);                               // -J Ri^T
                m3_mul(T1, Rj, T2);
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) {
                        rawJ[a * 6 + b] = Ri[b * 3 + a];                 // Ri^T
                        rawJ[a * 6 + 3 + b] = -S[a * 3 + b];
                        rawJ[(3 + a) * 6 + 3 + b] = Jr[a * 3 + b];
                        rawJ[36 + a * 6 + b] = -Ri[b * 3 + a];
                        rawJ[36 + (3 + a) * 6 + 3 + b] = T2[a * 3 + b];
                    }
            }
        } else if (s - 1 - d.Nvo < n_rp) {
            // RollPitchFactor::Evaluate  rollpitch_factor.h:26-57
            const isv_rollpitch_t &f = R.rollpitch[s - 1 - d.Nvo];
            const double *p = poseW + f.index * 7;
            Quat Rq = q_normalized(q_from_pose(p)), Rm = q_from_R(f.R);
            double nZ[3] = {0, 0, -1.0}, v[3];
            q_rot(so3_mul(Rm, q_conj(Rq)), nZ, v);
            raw[0] = v[0]; raw[1] = v[1];
            if (JAC) {
                double S[9], Rmm[9], Bm[9];
                skew3(v, S); q_to_R(Rm, Rmm); m3_mul(S, Rmm, Bm);
#pragma unroll
                for (int k = 0; k < 12; k++) rawJ[k] = 0;
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) rawJ[a * 6 + 3 + b] = Bm[a * 3 + b];
            }
        }
    }
}
// residual-only evaluation (the candidate point), phases 2 and 3 for ALL priors of the window at once instead of slot after
// slot -- lane per (prior, row) forms r = sqrt_info raw straight from the factor records, lane per prior the cost.  Same
// operation order per entry as prior_weight / prior_correct, so the candidate cost is the same function of the state, bit for
// bit, as the cost at x the step control compares it with.  One wavefront (t = lane).
DEV void prior_residual_costs(const DevBatch &d, const PriorRecs &R, int w, int n_rp, const double *sRaw, double *sW, double *cost_out, int t) {
    constexpr int RAWS = 10, WS = 10;
    const int slots = d.n_prior_slots;
    for (int e = t; e < slots * 9; e += 64) {
        const int s = e / 9, row = e - 9 * s;
        const PriorDesc p = prior_desc(d, R, s, n_rp);
        const int dim = p.kind == 1 ? 9 : (p.kind == 3 ? 2 : 6);
        if (!p.valid || row >= dim) continue;
        const double *raw = sRaw + s * RAWS;
        double v = 0;
        for (int k = 0; k < dim; k++) v += p.S[row * dim + k] * raw[k];
        sW[s * WS + row] = v;
    }
    ISV_WSYNC();
    for (int s = t; s < slots; s += 64) {
        const PriorDesc p = prior_desc(d, R, s, n_rp);
        const int dim = p.kind == 1 ? 9 : (p.kind == 3 ? 2 : 6);
        double cost = 0.0;
        if (p.valid) {
            const double *wr = sW + s * WS;
            double ssum = 0;
            for (int k = 0; k < dim; k++) ssum += wr[k] * wr[k];
            cost = 0.5 * log(1.0 + ssum);
        }
        cost_out[(size_t)w * slots + s] = cost;
    }
}

// WAVE: executed by ONE wavefront of a larger workgroup (lane ids, wave-level LDS ordering instead of barriers)
// (s0, sstride) with sstride > 1 (WAVE only): `sstride` wavefronts of the workgroup share the window's priors, wavefront s0 takes the
// slots s0, s0 + sstride, ... through all four phases (a slot's arithmetic does not depend on the lane split); the records are staged
// by all of them behind ONE block barrier, which every wavefront that calls this must reach
template <bool JAC, bool WAVE>
DEV void prior_linearize_body(const DevBatch &d, const double *pose_src, const double *sb_src, double *cost_out, int gate, int w, double *lds, int s0 = 0, int sstride = 1) {
    const int t = WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const int slots = d.n_prior_slots;
    if (gate) {
        const SolveState &ss = d.st[w];
        if (!(ss.termination == ISV_TERM_RUNNING && (gate == 1 ? ss.need_linearize != 0 : ss.step_valid != 0))) return;
    }
    const int n_rp = d.n_rp[w];
    // residual-only evaluation (JAC = false, the candidate point in k_dogleg): no Jacobian blocks, 10-double slots
    constexpr int RAWS = JAC ? PRL_RAW : 10, WS = JAC ? PRL_W : 10;
    double *sRaw = lds, *sW = lds + (size_t)slots * RAWS, *sRec = sW + (size_t)slots * WS;
    double *strip = d.prior_strip + (size_t)w * d.prior_strip_sz;
    double *PH = d.prior_H + (size_t)w * d.prior_H_sz;
    const PriorRecs R = sstride > 1 ? prior_stage_records(d, w, sRec, (int)threadIdx.x, 64 * sstride) : prior_stage_records(d, w, sRec, t, 64);
    if (WAVE && sstride == 1) ISV_WSYNC(); else __syncthreads();
    prior_phase1<JAC>(d, R, pose_src, sb_src, w, n_rp, sRaw, t, 0xFu, s0, sstride);
    if (WAVE) ISV_WSYNC(); else __syncthreads();
    if (!JAC) {
        prior_residual_costs(d, R, w, n_rp, sRaw, sW, cost_out, t);     // (64 threads either way: the wavefront-level ordering is the block's)
        return;
    }
    // ---- phase 2: sqrt_info * [raw r | raw J] ----
    for (int s = s0; s < slots; s += sstride) {
        const PriorDesc p = prior_desc(d, R, s, n_rp);
        if (!p.valid) continue;
        const double *raw = sRaw + s * RAWS, *rawJ = raw + 9, *S = p.S;
        double *wr = sW + s * WS;
        if (p.kind == 0) prior_weight<6, 1, 6, false, JAC>(S, raw, rawJ, wr, t);
        else if (p.kind == 1) prior_weight<9, 1, 9, true, JAC>(S, raw, rawJ, wr, t);
        else if (p.kind == 2) prior_weight<6, 2, 6, false, JAC>(S, raw, rawJ, wr, t);
        else prior_weight<2, 1, 6, false, JAC>(S, raw, rawJ, wr, t);
    }
    if (WAVE) ISV_WSYNC(); else __syncthreads();
    // ---- phase 3: CauchyLoss corrector (scale r and J by sqrt(rho')), cost, strips ----
    for (int s = s0; s < slots; s += sstride) {
        const PriorDesc p = prior_desc(d, R, s, n_rp);
        double *wr = sW + s * WS, *so = strip + p.strip_off;
        double cost = 0.0;
        if (!p.valid) { if (JAC) for (int e = t; e < PR_RP_SZ; e += 64) so[e] = 0.0; }
        else if (p.kind == 0) cost = prior_correct<6, 1, 6, JAC, WAVE>(wr, so, t);
        else if (p.kind == 1) cost = prior_correct<9, 1, 9, JAC, WAVE>(wr, so, t);
        else if (p.kind == 2) cost = prior_correct<6, 2, 6, JAC, WAVE>(wr, so, t);
        else cost = prior_correct<2, 1, 6, JAC, WAVE>(wr, so, t);
        if (t == 0) cost_out[(size_t)w * slots + s] = cost;
    }
    if (!JAC) return;
    if (WAVE) ISV_WSYNC(); else __syncthreads();
    // ---- phase 4: J^T J and J^T r ----
    for (int s = s0; s < slots; s += sstride) {
        const PriorDesc p = prior_desc(d, R, s, n_rp);
        if (!p.valid) continue;
        const double *wr = sW + s * WS;
        if (p.kind == 0) prior_H<6, 1, 6>(wr, PH + p.H_off, t);
        else if (p.kind == 1) prior_H<9, 1, 9>(wr, PH + p.H_off, t);
        else if (p.kind == 2) prior_H<6, 2, 6>(wr, PH + p.H_off, t);
        else prior_H<2, 1, 6>(wr, PH + p.H_off, t);
    }
}
