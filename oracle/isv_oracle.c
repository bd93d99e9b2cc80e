/*
 * isv_oracle.c -- CPU ORACLE: plain-C restatement of the IS-VINS per-frame sliding-window solve
 *   Estimator::backendOptimization()  /root/reference/src/estimator.cpp:1541-1562
 *     vector2double :474-516, problemSolve :1004-1146, double2vector :518-594,
 *     MargForward :1149-1352, MargBackward :1354-1539
 * plus the part of Ceres-Solver 2.0.0 (external dependency, NOT under /root/reference; version
 * named in README.md:22) that problemSolve() drives: ceres::Solve with DENSE_SCHUR + DOGLEG
 * (TRADITIONAL_DOGLEG) + CauchyLoss(1.0) + PoseLocalParameterization, restated from Ceres'
 * published algorithm (trust_region_minimizer.cc, dogleg_strategy.cc, corrector.cc,
 * schur_eliminator_impl.h, trust_region_step_evaluator.cc).
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / reported baseline.  The product path
 * (is-vins_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this image (Eigen3, Ceres, Sophus, OpenCV
 * absent; no network) and ships no golden vectors or unit tests for this path.
 * What pins this restatement instead: finite-difference Jacobian checks that mirror the
 * reference's own check() routines, the sparsification "zero test"/KLD the reference computes,
 * Schur == dense-normal-equations identities, and first-order optimality of converged solves
 * (tests/test_oracle_*.py).
 *
 * Deviations that are mathematically neutral (documented in DESIGN.md): Ceres' automatic Schur
 * ordering may also eliminate some speed-bias blocks as e-blocks; here exactly the landmarks are
 * eliminated.  Eigen's PartialPivLU/FullPivLU/LLT/SelfAdjointEigenSolver/BDCSVD are restated by
 * textbook algorithms of the same mathematical definition.
 */
#include <stdio.h>
#include <float.h>
#include "isvo_factors.h"

/* ------------------------------------------------------------------------------------------ */
/* Problem = what problemSolve() adds to ceres::Problem (src/estimator.cpp:1022-1117)           */
typedef struct {
    int dim, nb, robust;
    int col[4], wid[4];      /* tangent column offset / width of each non-constant block    */
    int joff[4];             /* offset of the dim x wid row-major Jacobian block in jac[]    */
    int roff;                /* offset in the residual vector                                */
    int kind, a, b, c;       /* 0 imu(a=i) 1 proj(a=i,b=j,c=l) 2 se3 3 lin9 4 relpose(a) 5 rollpitch(a) */
    int obs0, obsj;          /* proj: observation indices of pts_i and pts_j                 */
} rblock_t;

typedef struct {
    const isv_config_t *cfg;
    const isv_window_t *w;
    int N, Nvo, L, np, ncols, nrb, nres, njac, namb;
    rblock_t *rb;
    double *imu_sqrt_info;   /* [N-1][225] */
    /* current point (ambient) */
    double *pose, *sb, *lam, ex[7];
    /* ESTIMATE_EXTRINSIC (src/estimator.cpp:1028-1036): the extrinsic block is free, tangent columns 15N .. 15N+5 */
    int est_ex;
    double cex[7];           /* candidate extrinsic */
    const double *ex_use;    /* the extrinsic evaluate() reads (x or the candidate) */
    /* evaluation outputs */
    double *res, *res_c, *jac, *grad;   /* res: residuals at x (with jac); res_c: candidate-point scratch */
} problem_t;

static void build_problem(problem_t *P, const isv_config_t *cfg, const isv_window_t *w) {
    memset(P, 0, sizeof(*P));
    P->cfg = cfg; P->w = w;
    int N = cfg->n_frames, L = w->n_landmarks;
    P->est_ex = cfg->estimate_extrinsic != 0;
    P->ex_use = P->ex;
    P->N = N; P->Nvo = cfg->n_vo; P->L = L; P->np = 15 * N + (P->est_ex ? 6 : 0); P->ncols = P->np + L;
    P->namb = 16 * N + L + (P->est_ex ? 7 : 0);
    int F = w->n_obs - L;
    int maxrb = (N - 1) + F + 2 + (cfg->n_vo - 1) + w->n_rollpitch;
    P->rb = (rblock_t *)calloc(maxrb, sizeof(rblock_t));
    int n = 0, roff = 0, joff = 0;
    /* IMU factors i -> i+1 for ALL i, skipped when sum_dt > 10 (:1040-1051), no loss */
    for (int i = 0; i < N - 1; i++) {
        if (w->imu[i].sum_dt > 10.0) continue;
        rblock_t *r = &P->rb[n++];
        r->kind = 0; r->a = i; r->dim = 15; r->nb = 4; r->robust = 0;
        r->col[0] = 15 * i; r->wid[0] = 6; r->col[1] = 15 * i + 6; r->wid[1] = 9;
        r->col[2] = 15 * (i + 1); r->wid[2] = 6; r->col[3] = 15 * (i + 1) + 6; r->wid[3] = 9;
        r->roff = roff; roff += 15;
        for (int k = 0; k < 4; k++) { r->joff[k] = joff; joff += 15 * r->wid[k]; }
    }
    /* projection factors host -> every later view (:1057-1092), CauchyLoss(1.0) */
    for (int l = 0; l < L; l++) {
        int h = w->lm_start_frame[l], o0 = w->lm_obs_ptr[l], o1 = w->lm_obs_ptr[l + 1];
        for (int o = o0 + 1; o < o1; o++) {
            rblock_t *r = &P->rb[n++];
            r->kind = 1; r->a = h; r->b = h + (o - o0); r->c = l; r->obs0 = o0; r->obsj = o;
            r->dim = 2; r->nb = P->est_ex ? 4 : 3; r->robust = 1;
            r->col[0] = 15 * r->a; r->wid[0] = 6; r->col[1] = 15 * r->b; r->wid[1] = 6;
            if (P->est_ex) { r->col[2] = 15 * N; r->wid[2] = 6; r->col[3] = P->np + l; r->wid[3] = 1; }     /* pose_i, pose_j, ex, lambda (:1077) */
            else { r->col[2] = P->np + l; r->wid[2] = 1; }
            r->roff = roff; roff += 2;
            for (int k = 0; k < r->nb; k++) { r->joff[k] = joff; joff += 2 * r->wid[k]; }
        }
    }
    /* priors (:1102-1117), all with CauchyLoss(1.0) */
    { rblock_t *r = &P->rb[n++]; r->kind = 2; r->dim = 6; r->nb = 1; r->robust = 1;
      r->col[0] = 0; r->wid[0] = 6; r->roff = roff; roff += 6; r->joff[0] = joff; joff += 36; }
    { rblock_t *r = &P->rb[n++]; r->kind = 3; r->dim = 9; r->nb = 1; r->robust = 1;
      r->col[0] = 15 * (cfg->n_vo - 1) + 6; r->wid[0] = 9; r->roff = roff; roff += 9; r->joff[0] = joff; joff += 81; }
    for (int i = 0; i < cfg->n_vo - 1; i++) {
        rblock_t *r = &P->rb[n++]; r->kind = 4; r->a = i; r->dim = 6; r->nb = 2; r->robust = 1;
        r->col[0] = 15 * i; r->wid[0] = 6; r->col[1] = 15 * (i + 1); r->wid[1] = 6;
        r->roff = roff; roff += 6;
        for (int k = 0; k < 2; k++) { r->joff[k] = joff; joff += 36; }
    }
    for (int i = 0; i < w->n_rollpitch; i++) {
        rblock_t *r = &P->rb[n++]; r->kind = 5; r->a = i; r->dim = 2; r->nb = 1; r->robust = 1;
        r->col[0] = 15 * w->rollpitch[i].index; r->wid[0] = 6;
        r->roff = roff; roff += 2; r->joff[0] = joff; joff += 12;
    }
    P->nrb = n; P->nres = roff; P->njac = joff;
    P->pose = (double *)calloc(7 * N, 8); P->sb = (double *)calloc(9 * N, 8);
    P->lam = (double *)calloc(L > 0 ? L : 1, 8);
    P->res = (double *)calloc(roff, 8); P->res_c = (double *)calloc(roff, 8); P->jac = (double *)calloc(joff, 8);
    P->grad = (double *)calloc(P->ncols, 8);
    P->imu_sqrt_info = (double *)calloc(225 * (N - 1), 8);
    for (int i = 0; i < N - 1; i++) isvo_imu_sqrt_info(w->imu[i].covariance, P->imu_sqrt_info + 225 * i);
}
static void free_problem(problem_t *P) {
    free(P->rb); free(P->pose); free(P->sb); free(P->lam); free(P->res); free(P->res_c); free(P->jac); free(P->grad);
    free(P->imu_sqrt_info);
}

/* drop the 7th pose column: PoseLocalParameterization::ComputeJacobian = [I6; 0]
 * (pose_local_parameterization.cpp:20-27); J_local = J * [I6;0] */
static void take6(const double *J7, int rows, double *J6) {
    for (int r = 0; r < rows; r++) for (int c = 0; c < 6; c++) J6[r * 6 + c] = J7[r * 7 + c];
}

/* ceres ResidualBlock::Evaluate for every block: cost, corrected residuals and Jacobians.
 * CauchyLoss(a=1): rho = [log(1+s), max(min, 1/(1+s)), -1/(1+s)^2]; rho'' < 0 so the Corrector
 * scales residual and Jacobian by sqrt(rho') (corrector.cc). */
static double evaluate(problem_t *P, const double *pose, const double *sb, const double *lam,
                       int want_jac) {
    const isv_window_t *w = P->w; const isv_config_t *cfg = P->cfg;
    double cost = 0;
    for (int n = 0; n < P->nrb; n++) {
        rblock_t *r = &P->rb[n];
        double *res = (want_jac ? P->res : P->res_c) + r->roff;
        double J0[15 * 9], J1[15 * 9], J2[15 * 9], J3[15 * 9];
        double *jb[4] = {NULL, NULL, NULL, NULL};
        if (want_jac) for (int k = 0; k < r->nb; k++) jb[k] = P->jac + r->joff[k];
        switch (r->kind) {
        case 0: {
            int i = r->a;
            isvo_imu_eval(&w->imu[i], cfg->gravity, pose + 7 * i, sb + 9 * i, pose + 7 * (i + 1), sb + 9 * (i + 1),
                          P->imu_sqrt_info + 225 * i, res, want_jac ? J0 : NULL, want_jac ? jb[1] : NULL,
                          want_jac ? J2 : NULL, want_jac ? jb[3] : NULL);
            if (want_jac) { take6(J0, 15, jb[0]); take6(J2, 15, jb[2]); }
        } break;
        case 1: {
            isvo_proj_eval(pose + 7 * r->a, pose + 7 * r->b, P->ex_use, lam[r->c], w->obs_point + 3 * r->obs0,
                           w->obs_point + 3 * r->obsj, cfg->proj_sqrt_info, 1, res,
                           want_jac ? J0 : NULL, want_jac ? J1 : NULL, (want_jac && P->est_ex) ? J2 : NULL, want_jac ? jb[r->nb - 1] : NULL);
            if (want_jac) { take6(J0, 2, jb[0]); take6(J1, 2, jb[1]); if (P->est_ex) take6(J2, 2, jb[2]); }
        } break;
        case 2:
            isvo_se3prior_eval(w->pose_prior, w->pose_prior->sqrt_info, pose + 7 * 0, res, want_jac ? J0 : NULL);
            if (want_jac) take6(J0, 6, jb[0]);
            break;
        case 3:
            isvo_linear9_eval(w->vb_prior, w->vb_prior->sqrt_info, sb + 9 * (cfg->n_vo - 1), res, jb[0]);
            break;
        case 4:
            isvo_relpose_eval(&w->relpose[r->a], w->relpose[r->a].sqrt_info, pose + 7 * r->a, pose + 7 * (r->a + 1),
                              res, want_jac ? J0 : NULL, want_jac ? J1 : NULL);
            if (want_jac) { take6(J0, 6, jb[0]); take6(J1, 6, jb[1]); }
            break;
        case 5: {
            const isv_rollpitch_t *f = &w->rollpitch[r->a];
            isvo_rollpitch_eval(f, f->sqrt_info, pose + 7 * f->index, res, want_jac ? J0 : NULL);
            if (want_jac) take6(J0, 2, jb[0]);
        } break;
        }
        (void)J3;
        double s = dotn(res, res, r->dim);
        if (r->robust) {
            double sum = 1.0 + s, inv = 1.0 / sum;
            double rho0 = log(sum), rho1 = inv > DBL_MIN ? inv : DBL_MIN;
            cost += 0.5 * rho0;
            double sc = sqrt(rho1);
            if (want_jac) for (int k = 0; k < r->nb; k++) for (int e = 0; e < r->dim * r->wid[k]; e++) jb[k][e] *= sc;
            for (int e = 0; e < r->dim; e++) res[e] *= sc;
        } else
            cost += 0.5 * s;
    }
    if (want_jac) {                /* gradient = J^T r with the UNSCALED Jacobian */
        memset(P->grad, 0, sizeof(double) * P->ncols);
        for (int n = 0; n < P->nrb; n++) {
            rblock_t *r = &P->rb[n];
            for (int k = 0; k < r->nb; k++) {
                const double *J = P->jac + r->joff[k];
                for (int e = 0; e < r->dim; e++) for (int c = 0; c < r->wid[k]; c++)
                    P->grad[r->col[k] + c] += J[e * r->wid[k] + c] * P->res[r->roff + e];
            }
        }
    }
    return cost;
}

static void jac_colnorm2(const problem_t *P, double *out) {
    memset(out, 0, sizeof(double) * P->ncols);
    for (int n = 0; n < P->nrb; n++) {
        const rblock_t *r = &P->rb[n];
        for (int k = 0; k < r->nb; k++) {
            const double *J = P->jac + r->joff[k];
            for (int e = 0; e < r->dim; e++) for (int c = 0; c < r->wid[k]; c++)
                out[r->col[k] + c] += J[e * r->wid[k] + c] * J[e * r->wid[k] + c];
        }
    }
}
static void jac_scale_cols(problem_t *P, const double *s) {
    for (int n = 0; n < P->nrb; n++) {
        rblock_t *r = &P->rb[n];
        for (int k = 0; k < r->nb; k++) {
            double *J = P->jac + r->joff[k];
            for (int e = 0; e < r->dim; e++) for (int c = 0; c < r->wid[k]; c++) J[e * r->wid[k] + c] *= s[r->col[k] + c];
        }
    }
}
static void jac_right_mul(const problem_t *P, const double *x, double *y /* nres, += */) {
    for (int n = 0; n < P->nrb; n++) {
        const rblock_t *r = &P->rb[n];
        for (int k = 0; k < r->nb; k++) {
            const double *J = P->jac + r->joff[k];
            for (int e = 0; e < r->dim; e++) {
                double s = 0;
                for (int c = 0; c < r->wid[k]; c++) s += J[e * r->wid[k] + c] * x[r->col[k] + c];
                y[r->roff + e] += s;
            }
        }
    }
}
static void jac_left_mul(const problem_t *P, const double *r_in, double *g /* ncols, += */) {
    for (int n = 0; n < P->nrb; n++) {
        const rblock_t *r = &P->rb[n];
        for (int k = 0; k < r->nb; k++) {
            const double *J = P->jac + r->joff[k];
            for (int e = 0; e < r->dim; e++) for (int c = 0; c < r->wid[k]; c++)
                g[r->col[k] + c] += J[e * r->wid[k] + c] * r_in[r->roff + e];
        }
    }
}

/* DENSE_SCHUR: min |J y - r|^2 + |D y|^2 with the landmarks eliminated (schur_eliminator_impl.h,
 * dense Cholesky on the reduced system, back substitution).  Returns 0 ok, 1 = LINEAR_SOLVER_FAILURE */
/* test hook: treat the first n linear solves of every iteration as failed (exercises the mu x 10 retry of
 * DoglegStrategy::ComputeGaussNewtonStep, which well-posed windows never reach) */
static int g_force_retry = 0;
void isvo_debug_force_retry(int n) { g_force_retry = n; }
/* test hooks for the terminations well-posed windows never reach (the device has the same hooks:
 * ISV_DEBUG_FORCE_INVALID / ISV_DEBUG_MIN_RADIUS): treat the first n trust-region steps as invalid
 * (model_cost_change <= 0, TrustRegionMinimizer::HandleInvalidStep), and override
 * Solver::Options::min_trust_region_radius (1e-32 by default) */
static int g_force_invalid = 0;
static double g_min_radius = 1e-32;
void isvo_debug_force_invalid(int n) { g_force_invalid = n; }
void isvo_debug_min_radius(double r) { g_min_radius = r > 0 ? r : 1e-32; }

static int dense_schur_solve(const problem_t *P, const double *D, double *y) {
    int np = P->np, L = P->L, N = P->N;
    double *S = (double *)calloc((size_t)np * np, 8), *g = (double *)calloc(np, 8);
    double *E = (double *)calloc(L > 0 ? L : 1, 8), *gl = (double *)calloc(L > 0 ? L : 1, 8);
    const int NF = N + 1;               /* pose-like blocks a landmark couples to: frames 0..N-1 and, when estimated, the extrinsic as "frame" N (columns 15N..) */
    double *W = (double *)calloc((size_t)(L > 0 ? L : 1) * NF * 6, 8);   /* W[l][frame][6] */
    for (int n = 0; n < P->nrb; n++) {
        const rblock_t *r = &P->rb[n];
        const double *res = P->res + r->roff;
        for (int k = 0; k < r->nb; k++) {
            const double *Jk = P->jac + r->joff[k];
            int ck = r->col[k], wk = r->wid[k];
            if (ck >= np) {                          /* landmark column */
                int l = ck - np;
                for (int e = 0; e < r->dim; e++) { E[l] += Jk[e] * Jk[e]; gl[l] += Jk[e] * res[e]; }
                continue;
            }
            for (int c = 0; c < wk; c++) { double s = 0; for (int e = 0; e < r->dim; e++) s += Jk[e * wk + c] * res[e]; g[ck + c] += s; }
            for (int m = 0; m < r->nb; m++) {
                const double *Jm = P->jac + r->joff[m];
                int cm = r->col[m], wm = r->wid[m];
                if (cm >= np) {                      /* pose x landmark -> W */
                    int l = cm - np, f = ck / 15;
                    for (int c = 0; c < wk; c++) { double s = 0; for (int e = 0; e < r->dim; e++) s += Jk[e * wk + c] * Jm[e]; W[((size_t)l * NF + f) * 6 + c] += s; }
                    continue;
                }
                for (int c = 0; c < wk; c++) for (int d = 0; d < wm; d++) {
                    double s = 0; for (int e = 0; e < r->dim; e++) s += Jk[e * wk + c] * Jm[e * wm + d];
                    S[(size_t)(ck + c) * np + cm + d] += s;
                }
            }
        }
    }
    for (int i = 0; i < np; i++) S[(size_t)i * np + i] += D[i] * D[i];
    for (int l = 0; l < L; l++) {
        int h = P->w->lm_start_frame[l], k = P->w->lm_obs_ptr[l + 1] - P->w->lm_obs_ptr[l];
        double ete = E[l] + D[np + l] * D[np + l];
        double inv = 1.0 / ete;
        const int kk = k + (P->est_ex ? 1 : 0);             /* the landmark's frames, then the extrinsic */
        for (int ia = 0; ia < kk; ia++) for (int a = 0; a < 6; a++) {
            const int fa = ia < k ? h + ia : N;
            double wa = W[((size_t)l * NF + fa) * 6 + a];
            g[15 * fa + a] -= wa * inv * gl[l];
            for (int ib = 0; ib < kk; ib++) for (int b = 0; b < 6; b++) {
                const int fb = ib < k ? h + ib : N;
                S[(size_t)(15 * fa + a) * np + 15 * fb + b] -= wa * inv * W[((size_t)l * NF + fb) * 6 + b];
            }
        }
    }
    int info = chol_lower(S, np);
    if (info == 0) {
        chol_solve(S, np, g);
        for (int i = 0; i < np; i++) y[i] = g[i];
        for (int l = 0; l < L; l++) {
            int h = P->w->lm_start_frame[l], k = P->w->lm_obs_ptr[l + 1] - P->w->lm_obs_ptr[l];
            double ete = E[l] + D[np + l] * D[np + l], s = gl[l];
            for (int ia = 0; ia < k + (P->est_ex ? 1 : 0); ia++) for (int a = 0; a < 6; a++) { const int fa = ia < k ? h + ia : N; s -= W[((size_t)l * NF + fa) * 6 + a] * y[15 * fa + a]; }
            y[np + l] = s / ete;
        }
        for (int i = 0; i < P->ncols; i++) if (!isfinite(y[i])) info = 1;
    }
    free(S); free(g); free(E); free(gl); free(W);
    return info != 0;
}

/* Evaluator::Plus over all blocks */
static void state_plus(const problem_t *P, const double *pose, const double *sb, const double *lam,
                       const double *delta, double *pose_o, double *sb_o, double *lam_o, double *ex_o) {
    if (P->est_ex && ex_o) isvo_pose_plus(P->ex, delta + 15 * P->N, ex_o);
    for (int i = 0; i < P->N; i++) {
        isvo_pose_plus(pose + 7 * i, delta + 15 * i, pose_o + 7 * i);
        for (int k = 0; k < 9; k++) sb_o[9 * i + k] = sb[9 * i + k] + delta[15 * i + 6 + k];
    }
    for (int l = 0; l < P->L; l++) lam_o[l] = lam[l] + delta[P->np + l];
}
static double amb_norm2_diff(const problem_t *P, const double *p0, const double *s0, const double *l0,
                             const double *p1, const double *s1, const double *l1, const double *e1 /* candidate extrinsic or NULL */) {
    double s = 0;
    if (P->est_ex) for (int i = 0; i < 7; i++) { double d = P->ex[i] - ((p1 && e1) ? e1[i] : 0); s += d * d; }
    for (int i = 0; i < 7 * P->N; i++) { double d = p0[i] - (p1 ? p1[i] : 0); s += d * d; }
    for (int i = 0; i < 9 * P->N; i++) { double d = s0[i] - (s1 ? s1[i] : 0); s += d * d; }
    for (int i = 0; i < P->L; i++) { double d = l0[i] - (l1 ? l1[i] : 0); s += d * d; }
    return s;
}
/* gradient_max_norm = |x - Plus(x, -gradient)|_inf (trust_region_minimizer.cc, EvaluateGradientAndJacobian) */
static double projected_gradient_max(const problem_t *P, double *scratch) {
    double *ng = scratch, *pp = ng + P->ncols, *ss = pp + 7 * P->N, *ll = ss + 9 * P->N;
    for (int i = 0; i < P->ncols; i++) ng[i] = -P->grad[i];
    double ee[7];
    state_plus(P, P->pose, P->sb, P->lam, ng, pp, ss, ll, ee);
    double m = 0;
    if (P->est_ex) for (int i = 0; i < 7; i++) m = fmax(m, fabs(P->ex[i] - ee[i]));
    for (int i = 0; i < 7 * P->N; i++) m = fmax(m, fabs(P->pose[i] - pp[i]));
    for (int i = 0; i < 9 * P->N; i++) m = fmax(m, fabs(P->sb[i] - ss[i]));
    for (int i = 0; i < P->L; i++) m = fmax(m, fabs(P->lam[i] - ll[i]));
    return m;
}

/* TrustRegionMinimizer::Minimize with DoglegStrategy(TRADITIONAL_DOGLEG), Ceres 2.0.0 defaults:
 * initial radius 1e4, max radius 1e16, min radius 1e-32, min_relative_decrease 1e-3,
 * function/gradient/parameter tolerance 1e-6/1e-10/1e-8, jacobi_scaling, monotonic steps,
 * min/max LM diagonal 1e-6/1e32, dogleg mu in [1e-8, 1] x10, max 5 consecutive invalid steps. */
static void minimize(problem_t *P, int max_iter, isv_summary_t *sum) {
    int n = P->ncols, N = P->N, L = P->L;
    double *scale = (double *)calloc(n, 8), *diag = (double *)calloc(n, 8), *gradient = (double *)calloc(n, 8);
    double *gn = (double *)calloc(n, 8), *step = (double *)calloc(n, 8), *delta = (double *)calloc(n, 8);
    double *lmD = (double *)calloc(n, 8), *tmp = (double *)calloc(n, 8);
    double *Jg = (double *)calloc(P->nres, 8), *mres = (double *)calloc(P->nres, 8);
    double *cp = (double *)calloc(7 * N, 8), *cs = (double *)calloc(9 * N, 8), *cl = (double *)calloc(L > 0 ? L : 1, 8);
    double *scratch = (double *)calloc(n + 16 * N + L + 8, 8);
    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_step_norm = 0;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    int reuse = 0, invalid = 0;
    memset(sum, 0, sizeof(*sum));

    double x_cost = evaluate(P, P->pose, P->sb, P->lam, 1);
    jac_colnorm2(P, scale);
    for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + sqrt(scale[i]));
    jac_scale_cols(P, scale);
    double gmax = projected_gradient_max(P, scratch);
    double x_norm = sqrt(amb_norm2_diff(P, P->pose, P->sb, P->lam, NULL, NULL, NULL, NULL));
    sum->initial_cost = x_cost; sum->trace_cost[0] = x_cost; sum->trace_radius[0] = radius;
    int it = 0, term = ISV_TERM_RUNNING;
    for (;;) {
        /* FinalizeIterationAndCheckIfMinimizerCanContinue */
        if (it >= max_iter) { term = ISV_TERM_MAX_ITERATIONS; break; }
        if (gmax <= 1e-10) { term = ISV_TERM_GRADIENT_TOL; break; }
        if (radius <= g_min_radius) { term = ISV_TERM_MIN_RADIUS; break; }
        it++;
        /* ---- DoglegStrategy::ComputeStep ---- */
        int ls_fail = 0;
        if (!reuse) {
            reuse = 1;
            jac_colnorm2(P, diag);
            for (int i = 0; i < n; i++) diag[i] = sqrt(fmin(fmax(diag[i], 1e-6), 1e32));
            memset(gradient, 0, 8 * n);
            jac_left_mul(P, P->res, gradient);
            for (int i = 0; i < n; i++) gradient[i] /= diag[i];
            for (int i = 0; i < n; i++) tmp[i] = gradient[i] / diag[i];
            memset(Jg, 0, 8 * P->nres);
            jac_right_mul(P, tmp, Jg);
            alpha = dotn(gradient, gradient, n) / dotn(Jg, Jg, P->nres);
            ls_fail = 1;
            int forced = g_force_retry;
            while (mu < max_mu) {
                if (forced > 0) { forced--; mu *= mu_inc; continue; }
                for (int i = 0; i < n; i++) lmD[i] = diag[i] * sqrt(mu);
                if (dense_schur_solve(P, lmD, gn)) { mu *= mu_inc; continue; }
                ls_fail = 0; break;
            }
            if (!ls_fail) for (int i = 0; i < n; i++) gn[i] *= -diag[i];
        }
        int valid = 0; double model_cost_change = 0;
        if (!ls_fail) {
            /* ComputeTraditionalDoglegStep */
            double gn_norm = 0;
            for (int i = 0; i < n; i++) gn_norm += gn[i] * gn[i];
            gn_norm = sqrt(gn_norm);
            double g_norm = sqrt(dotn(gradient, gradient, n));
            if (gn_norm <= radius) {
                for (int i = 0; i < n; i++) step[i] = gn[i];
                dogleg_step_norm = gn_norm;
            } else if (g_norm * alpha >= radius) {
                for (int i = 0; i < n; i++) step[i] = -(radius / g_norm) * gradient[i];
                dogleg_step_norm = radius;
            } else {
                double b_dot_a = -alpha * dotn(gradient, gn, n);
                double a_sq = pow(alpha * g_norm, 2.0);
                double bma_sq = a_sq - 2 * b_dot_a + pow(gn_norm, 2.0);
                double c = b_dot_a - a_sq;
                double d = sqrt(c * c + bma_sq * (pow(radius, 2.0) - a_sq));
                double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
                for (int i = 0; i < n; i++) step[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gn[i];
                dogleg_step_norm = sqrt(dotn(step, step, n));
            }
            for (int i = 0; i < n; i++) step[i] /= diag[i];
            /* model_cost_change = -(J step)^T (r + J step / 2) */
            memset(mres, 0, 8 * P->nres);
            jac_right_mul(P, step, mres);
            double mc = 0;
            for (int i = 0; i < P->nres; i++) mc += mres[i] * (P->res[i] + mres[i] / 2.0);
            model_cost_change = -mc;
            valid = model_cost_change > 0.0;
            if (it <= g_force_invalid) valid = 0;
        }
        if (!valid) {                                /* HandleInvalidStep */
            if (++invalid >= 5) { term = ls_fail ? ISV_TERM_LINEAR_SOLVER : ISV_TERM_INVALID_STEPS; break; }
            mu *= mu_inc; reuse = 0;                 /* DoglegStrategy::StepIsInvalid */
            if (it < ISV_MAX_TRACE) { sum->trace_cost[it] = x_cost; sum->trace_radius[it] = radius; sum->trace_step_norm[it] = 0; sum->trace_accepted[it] = 0; }
            continue;
        }
        invalid = 0;
        for (int i = 0; i < n; i++) delta[i] = step[i] * scale[i];
        state_plus(P, P->pose, P->sb, P->lam, delta, cp, cs, cl, P->cex);
        if (P->est_ex) P->ex_use = P->cex;
        double cand_cost = evaluate(P, cp, cs, cl, 0);
        P->ex_use = P->ex;
        double step_norm = sqrt(amb_norm2_diff(P, P->pose, P->sb, P->lam, cp, cs, cl, P->cex));
        int accepted = 0, stop = 0;
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { term = ISV_TERM_PARAMETER_TOL; stop = 1; }
        else if (fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { term = ISV_TERM_FUNCTION_TOL; stop = 1; }
        if (stop) {
            if (it < ISV_MAX_TRACE) { sum->trace_cost[it] = x_cost; sum->trace_radius[it] = radius; sum->trace_step_norm[it] = step_norm; sum->trace_accepted[it] = 0; }
            break;                                   /* the last step is NOT taken (Ceres) */
        }
        double rel = (x_cost - cand_cost) / model_cost_change;   /* monotonic StepQuality */
        if (rel > 1e-3) {                            /* HandleSuccessfulStep */
            accepted = 1;
            memcpy(P->pose, cp, 56 * N); memcpy(P->sb, cs, 72 * N); memcpy(P->lam, cl, 8 * L);
            if (P->est_ex) memcpy(P->ex, P->cex, 56);
            x_norm = sqrt(amb_norm2_diff(P, P->pose, P->sb, P->lam, NULL, NULL, NULL, NULL));
            x_cost = evaluate(P, P->pose, P->sb, P->lam, 1);
            jac_scale_cols(P, scale);
            gmax = projected_gradient_max(P, scratch);
            /* DoglegStrategy::StepAccepted */
            if (rel < 0.25) radius *= 0.5;
            if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
            mu = fmax(min_mu, 2.0 * mu / mu_inc);
            reuse = 0;
            sum->num_successful++;
        } else {                                     /* StepRejected */
            radius *= 0.5; reuse = 1;
        }
        if (it < ISV_MAX_TRACE) {
            sum->trace_cost[it] = accepted ? x_cost : cand_cost; sum->trace_radius[it] = radius;
            sum->trace_step_norm[it] = step_norm; sum->trace_accepted[it] = accepted;
        }
    }
    sum->iterations = it; sum->termination = term; sum->final_cost = x_cost; sum->status = ISV_OK;
    free(scale); free(diag); free(gradient); free(gn); free(step); free(delta); free(lmD); free(tmp);
    free(Jg); free(mres); free(cp); free(cs); free(cl); free(scratch);
}

/* ------------------------------------------------------------------------------------------ */
/* Estimator::vector2double  src/estimator.cpp:474-516 */
static void vector2double(const isv_window_t *w, int N, double *pose, double *sb, double *ex, double *lam) {
    for (int i = 0; i < N; i++) {
        quat_t q = q_from_R(w->Rs + 9 * i);
        double *p = pose + 7 * i;
        p[0] = w->Ps[3 * i]; p[1] = w->Ps[3 * i + 1]; p[2] = w->Ps[3 * i + 2];
        p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
        for (int k = 0; k < 3; k++) { sb[9 * i + k] = w->Vs[3 * i + k]; sb[9 * i + 3 + k] = w->Bas[3 * i + k]; sb[9 * i + 6 + k] = w->Bgs[3 * i + k]; }
    }
    quat_t q = q_from_R(w->ric);
    ex[0] = w->tic[0]; ex[1] = w->tic[1]; ex[2] = w->tic[2]; ex[3] = q.x; ex[4] = q.y; ex[5] = q.z; ex[6] = q.w;
    for (int l = 0; l < w->n_landmarks; l++) lam[l] = 1. / w->lm_depth[l];   /* getDepthVector :188-204 */
}

/* pseudo-measurement update after the solve  src/estimator.cpp:1133-1144 */
/* diagnostic hook (tests/test_sequence_long.py, the sensitivity study): skip the update() calls, i.e. keep every prior's
 * pseudo-measurement where marginalisation put it.  NOT the reference's behaviour. */
static int g_no_update = 0;
void isvo_debug_no_update(int on) { g_no_update = on; }
static void update_priors(const isv_config_t *cfg, isv_window_t *w, const double *pose, const double *sb) {
    int v = cfg->n_vo - 1;
    if (g_no_update) return;
    isvo_linear9_update(w->vb_prior, w->Vs + 3 * v, w->Bas + 3 * v, w->Bgs + 3 * v, sb + 9 * v);
    isvo_se3prior_update(w->pose_prior, w->Ps, w->Rs, pose);
    for (int i = 0; i < cfg->n_vo - 1; i++)
        isvo_relpose_update(&w->relpose[i], w->Ps + 3 * i, w->Rs + 9 * i, w->Ps + 3 * (i + 1), w->Rs + 9 * (i + 1),
                            pose + 7 * i, pose + 7 * (i + 1));
    for (int i = 0; i < w->n_rollpitch; i++) {
        int idx = w->rollpitch[i].index;
        isvo_rollpitch_update(&w->rollpitch[i], w->Rs + 9 * idx, pose + 7 * idx);
    }
}

/* Estimator::double2vector  src/estimator.cpp:518-594 (failure_occur is never set, :664) */
static void double2vector(const isv_config_t *cfg, isv_window_t *w, const double *pose, const double *sb,
                          const double *ex, const double *lam) {
    int N = cfg->n_frames;
    double origin_R0[3], origin_P0[3], origin_R00[3], R00[9], rot_diff[9];
    R2ypr(w->Rs, origin_R0);
    memcpy(origin_P0, w->Ps, 24);
    q_to_R(q_from_pose(pose), R00);
    R2ypr(R00, origin_R00);
    double y_diff = origin_R0[0] - origin_R00[0];
    double ypr[3] = {y_diff, 0, 0};
    ypr2R(ypr, rot_diff);
    if (fabs(fabs(origin_R0[1]) - 90) < 1.0 || fabs(fabs(origin_R00[1]) - 90) < 1.0) {
        double T[9]; m3_t(R00, T); mm(w->Rs, T, rot_diff, 3, 3, 3);
    }
    double t[3], Rn[9];
    m3v(rot_diff, w->vb_prior->VB + 6, t); memcpy(w->vb_prior->VB + 6, t, 24);          /* :549 (gyro-bias slot) */
    mm(rot_diff, w->pose_prior->R, Rn, 3, 3, 3); memcpy(w->pose_prior->R, Rn, 72);       /* :550 */
    for (int i = 0; i < N; i++) {
        double Ri[9], d[3];
        q_to_R(q_normalized(q_from_pose(pose + 7 * i)), Ri);
        mm(rot_diff, Ri, w->Rs + 9 * i, 3, 3, 3);
        for (int k = 0; k < 3; k++) d[k] = pose[7 * i + k] - pose[k];
        m3v(rot_diff, d, t);
        for (int k = 0; k < 3; k++) w->Ps[3 * i + k] = t[k] + origin_P0[k];
        m3v(rot_diff, sb + 9 * i, t);
        for (int k = 0; k < 3; k++) { w->Vs[3 * i + k] = t[k]; w->Bas[3 * i + k] = sb[9 * i + 3 + k]; w->Bgs[3 * i + k] = sb[9 * i + 6 + k]; }
    }
    memcpy(w->tic, ex, 24);
    q_to_R(q_from_pose(ex), w->ric);
    for (int l = 0; l < w->n_landmarks; l++) {        /* FeatureManager::setDepth :145-163 */
        w->lm_depth[l] = 1.0 / lam[l];
        if (w->lm_solve_flag) w->lm_solve_flag[l] = (w->lm_depth[l] < 0 || w->lm_depth[l] > 10) ? 2 : 1;
    }
}

/* add J_a^T W J_b into Lam (symmetric fill as the reference does, :1183-1201) */
static void add_hessian(double *Lam, int n, const double *Ja, int da, int ia, const double *Jb, int db, int ib,
                        const double *Wm, int dim) {
    /* JtW = Ja^T * W ; H = JtW * Jb */
    double JtW[15 * 15], H[15 * 15];
    mm_tn(Ja, Wm, JtW, da, dim, dim);
    mm(JtW, Jb, H, da, dim, db);
    for (int a = 0; a < da; a++) for (int b = 0; b < db; b++) Lam[(size_t)(ia + a) * n + ib + b] += H[a * db + b];
    if (ia != ib) for (int a = 0; a < da; a++) for (int b = 0; b < db; b++) Lam[(size_t)(ib + b) * n + ia + a] += H[a * db + b];
}
/* sqrt_info = LLT(M).matrixL().transpose() for an n x n SPD matrix M */
static int llt_upper(const double *M, int n, double *U) {
    double *Lw = (double *)malloc(8 * n * n);
    memcpy(Lw, M, 8 * n * n);
    int info = chol_lower(Lw, n);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) U[i * n + j] = Lw[j * n + i];
    free(Lw);
    return info;
}
static void compact6(const double *J7, int rows, double *J6) { take6(J7, rows, J6); }

/* J pseudo-inverse (Utility::pseudoInverse, utility.h:143-156, threshold eps*max(rows,cols)):
 * restated via the eigen-decomposition of J J^T: J^+ = J^T U diag(1/s^2) U^T over singular
 * values s > thr * s_max. J is m x n with m <= n. */
static void pinv_wide(const double *J, int m, int n, double eps, double *Jp /* n x m */) {
    double *G = (double *)malloc(8 * m * m), *w = (double *)malloc(8 * m), *V = (double *)malloc(8 * m * m);
    mm_nt(J, J, G, m, n, m);
    sym_eig(G, m, w, V);
    double smax = sqrt(fmax(w[m - 1], 0.0));
    double thr = eps * (m > n ? m : n);
    double *Ginv = (double *)calloc(m * m, 8);
    for (int k = 0; k < m; k++) {
        double s = sqrt(fmax(w[k], 0.0));
        if (!(s > thr * smax)) continue;
        for (int a = 0; a < m; a++) for (int b = 0; b < m; b++) Ginv[a * m + b] += V[a * m + k] * V[b * m + k] / w[k];
    }
    mm_tn(J, Ginv, Jp, n, m, m);
    free(G); free(w); free(V); free(Ginv);
}

/* shared sparsification step (estimator.cpp:1311-1331 / :1479-1497): eigen-truncate Lam at alpha,
 * return rank; U is n x rank (row-major with leading dim n), Dv the kept eigenvalues */
static int eig_truncate(const double *Lam, int n, double alpha, double *U, double *Dv) {
    double *w = (double *)malloc(8 * n), *V = (double *)malloc(8 * n * n);
    sym_eig(Lam, n, w, V);
    int rank = 0;
    for (int i = 0; i < n; i++) if (w[i] > alpha) {
        for (int k = 0; k < n; k++) U[k * n + rank] = V[k * n + i];
        Dv[rank++] = w[i];
    }
    free(w); free(V);
    return rank;
}
/* Sigma = (Jk U) D^-1 (Jk U)^T for a row block Jk (rows x n) */
static void project_cov(const double *Jk, int rows, int n, const double *U, const double *Dv, int rank, double *Sigma) {
    double *JU = (double *)malloc(8 * rows * rank);
    for (int a = 0; a < rows; a++) for (int k = 0; k < rank; k++) {
        double s = 0; for (int c = 0; c < n; c++) s += Jk[a * n + c] * U[c * n + k];
        JU[a * rank + k] = s;
    }
    for (int a = 0; a < rows; a++) for (int b = 0; b < rows; b++) {
        double s = 0; for (int k = 0; k < rank; k++) s += JU[a * rank + k] * (1.0 / Dv[k]) * JU[b * rank + k];
        Sigma[a * rows + b] = s;
    }
    free(JU);
}

/* Estimator::MargForward  src/estimator.cpp:1149-1352.  Linearises at para_* (the un-rotated
 * solve output) while Ri/ti come from the rotated Rs[0]/Ps[0] (quirk). */
static void marg_forward(const isv_config_t *cfg, const isv_window_t *w, const double *pose, const double *ex,
                         const double *lam, isv_marg_result_t *out) {
    /* MargPointIdx: factors with imu_i == 0 && gap == 1 (estimator.cpp:1082-1087) */
    int L = w->n_landmarks, n0 = 0;
    int *idx = (int *)malloc(4 * (L + 1));
    for (int l = 0; l < L; l++) if (w->lm_start_frame[l] == 0 && w->lm_obs_ptr[l + 1] - w->lm_obs_ptr[l] >= 2) idx[n0++] = l;
    int n = n0 + 12;
    double *Lam = (double *)calloc((size_t)n * n, 8);
    const double *sq = cfg->proj_sqrt_info;
    double info2[4]; mm_tn(sq, sq, info2, 2, 2, 2);
    /* order: T1 @0, T0 @6, landmarks @12.. */
    for (int m = 0; m < n0; m++) {
        int l = idx[m], o0 = w->lm_obs_ptr[l];
        double r[2], Ji7[14], Jj7[14], Jl[2], Ji[12], Jj[12];
        isvo_proj_eval(pose, pose + 7, ex, lam[l], w->obs_point + 3 * o0, w->obs_point + 3 * (o0 + 1), sq, 0, r, Ji7, Jj7, NULL, Jl);
        compact6(Ji7, 2, Ji); compact6(Jj7, 2, Jj);
        /* ParamMap order: pose[imu_i=0] (@6), pose[imu_j=1] (@0), ex (skipped), feature (@12+m) */
        const double *Jb[3] = {Ji, Jj, Jl}; int db[3] = {6, 6, 1}; int ib[3] = {6, 0, 12 + m};
        for (int j = 0; j < 3; j++) for (int k = j; k < 3; k++) add_hessian(Lam, n, Jb[j], db[j], ib[j], Jb[k], db[k], ib[k], info2, 2);
    }
    {   /* pose prior on T0 (:1203-1211) */
        double r[6], J7[42], J[36], W6[36];
        isvo_se3prior_eval(w->pose_prior, NULL, pose, r, J7); compact6(J7, 6, J);
        mm_tn(w->pose_prior->sqrt_info, w->pose_prior->sqrt_info, W6, 6, 6, 6);
        double JtW[36], H[36]; mm_tn(J, W6, JtW, 6, 6, 6); mm(JtW, J, H, 6, 6, 6);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Lam[(size_t)(6 + a) * n + 6 + b] += H[a * 6 + b];
    }
    {   /* relpose edge (0,1) (:1212-1238) */
        double r[6], Ji7[42], Jj7[42], Ji[36], Jj[36], W6[36];
        isvo_relpose_eval(&w->relpose[0], NULL, pose, pose + 7, r, Ji7, Jj7);
        compact6(Ji7, 6, Ji); compact6(Jj7, 6, Jj);
        mm_tn(w->relpose[0].sqrt_info, w->relpose[0].sqrt_info, W6, 6, 6, 6);
        const double *Jb[2] = {Ji, Jj}; int ib[2] = {6, 0};
        for (int j = 0; j < 2; j++) for (int k = j; k < 2; k++) add_hessian(Lam, n, Jb[j], 6, ib[j], Jb[k], 6, ib[k], W6, 6);
    }
    /* (i) pose-graph edge (:1240-1283) */
    double Lam_rp[144];
    for (int a = 0; a < 12; a++) for (int b = 0; b < 12; b++) Lam_rp[a * 12 + b] = Lam[(size_t)a * n + b];
    quat_t Qi = q_from_pose(pose), Qj = q_from_pose(pose + 7);
    double d[3] = {pose[7] - pose[0], pose[8] - pose[1], pose[9] - pose[2]};
    isv_relpose_t *pg = &out->combined.relative_pose;
    memset(pg, 0, sizeof(*pg));
    q_rot(q_inv(Qi), d, pg->delta_t);
    q_to_R(q_mul(q_inv(Qi), Qj), pg->delta_R);
    pg->imu_i = 0; pg->imu_j = 1;
    {
        double r[6], Ji7[42], Jj7[42], J[72], Jp[72], T[72], Om[36], cov[36];
        memset(J, 0, sizeof(J));
        isvo_relpose_eval(pg, NULL, pose, pose + 7, r, Ji7, Jj7);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { J[a * 12 + b] = Ji7[a * 7 + b]; J[a * 12 + 6 + b] = Jj7[a * 7 + b]; }
        pinv_wide(J, 6, 12, 1e-8, Jp);                    /* 12 x 6 */
        mm_tn(Jp, Lam_rp, T, 6, 12, 12);                  /* Jp^T Lam_rp : 6 x 12 */
        mm(T, Jp, Om, 6, 12, 6);
        inv_partial_lu(Om, cov, 6);
        llt_upper(Om, 6, pg->sqrt_info);
        memcpy(out->combined.covRel, cov, sizeof(cov));
    }
    out->combined.has_rollpitch = 0;
    if (w->n_rollpitch > 0 && w->rollpitch[0].index == 0) {
        out->combined.has_rollpitch = 1;
        out->combined.rollpitch = w->rollpitch[0];
        double W2[4]; mm_tn(w->rollpitch[0].sqrt_info, w->rollpitch[0].sqrt_info, W2, 2, 2, 2);
        inv_partial_lu(W2, out->combined.covAbs, 2);
    }
    out->combined.distance = sqrt(dotn(pg->delta_t, pg->delta_t, 3));
    out->combined.ts = w->header0;
    memcpy(out->combined.Ri, w->Rs, 72); memcpy(out->combined.ti, w->Ps, 24);
    /* (ii) new pose prior on T1 (:1286-1351) */
    int nm = n0 + 6;
    double *Lmm = (double *)malloc(8 * (size_t)nm * nm), *Lmm_inv = (double *)malloc(8 * (size_t)nm * nm);
    for (int a = 0; a < nm; a++) for (int b = 0; b < nm; b++) Lmm[(size_t)a * nm + b] = Lam[(size_t)(6 + a) * n + 6 + b];
    inv_full_lu(Lmm, Lmm_inv, nm);
    double Lprior[36];
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
        double s = Lam[(size_t)a * n + b];
        for (int p = 0; p < nm; p++) {
            double t = 0;
            for (int q = 0; q < nm; q++) t += Lmm_inv[(size_t)p * nm + q] * Lam[(size_t)b * n + 6 + q];
            s -= Lam[(size_t)a * n + 6 + p] * t;
        }
        Lprior[a * 6 + b] = s;
    }
    isv_se3_prior_t *fp = &out->forward_pose_prior;
    memset(fp, 0, sizeof(*fp));
    memcpy(fp->t, pose + 7, 24);
    q_to_R(q_from_pose(pose + 7), fp->R);            /* SE3PriorFactor(P1,Q1): R(R_new) from a quaternion */
    fp->index = 0;
    double r6[6], J7[42], Jr[36], covi[36], X[36];
    isvo_se3prior_eval(fp, NULL, pose + 7, r6, J7); compact6(J7, 6, Jr);
    /* FullPivHouseholderQR rank with threshold eps=1e-16 (estimator.cpp:8,1304-1305): restated by
     * full-pivot LU pivots against the same relative threshold */
    double Linv[36];
    int rank;
    {
        double A[36]; memcpy(A, Lprior, sizeof(A));
        /* rank by full pivoting */
        int rp[6], cpv[6]; double maxp = 0; rank = 0; double piv[6];
        for (int k = 0; k < 6; k++) {
            int pi = k, pj = k; double best = 0;
            for (int i = k; i < 6; i++) for (int j = k; j < 6; j++) if (fabs(A[i * 6 + j]) > best) { best = fabs(A[i * 6 + j]); pi = i; pj = j; }
            piv[k] = best; if (best > maxp) maxp = best;
            if (best == 0) { for (int q = k; q < 6; q++) piv[q] = 0; break; }
            for (int j = 0; j < 6; j++) { double t = A[k * 6 + j]; A[k * 6 + j] = A[pi * 6 + j]; A[pi * 6 + j] = t; }
            for (int i = 0; i < 6; i++) { double t = A[i * 6 + k]; A[i * 6 + k] = A[i * 6 + pj]; A[i * 6 + pj] = t; }
            for (int i = k + 1; i < 6; i++) { double f = A[i * 6 + k] / A[k * 6 + k]; for (int j = k; j < 6; j++) A[i * 6 + j] -= f * A[k * 6 + j]; }
            (void)rp; (void)cpv;
        }
        for (int k = 0; k < 6; k++) if (piv[k] > 1e-16 * maxp) rank++;
    }
    double kld = 0;
    if (rank == 6) {
        inv_full_lu(Lprior, Linv, 6);
        double T[36]; mm(Jr, Linv, T, 6, 6, 6); mm_nt(T, Jr, covi, 6, 6, 6);
    } else {
        double U[36], Dv[6];
        int rk = eig_truncate(Lprior, 6, cfg->alpha, U, Dv);
        project_cov(Jr, 6, 6, U, Dv, rk, covi);
    }
    inv_partial_lu(covi, X, 6);
    if (rank == 6) {
        double T[36], phi[36], pc[36];
        mm_tn(Jr, X, T, 6, 6, 6); mm(T, Jr, phi, 6, 6, 6);
        mm(phi, Linv, pc, 6, 6, 6);
        double a = 0; for (int k = 0; k < 6; k++) a += pc[k * 6 + k];
        kld = 0.5 * (a - log(det_lu(phi, 6)) - log(det_lu(Linv, 6)) - 6);
    }
    llt_upper(X, 6, fp->sqrt_info);
    out->forward_kld = kld; if (getenv("ISVO_DEBUG")) fprintf(stderr, "fwd rank=%d kld=%g\n", rank, kld);
    out->n_marg_landmarks = n0;
    free(idx); free(Lam); free(Lmm); free(Lmm_inv);
}

/* Estimator::MargBackward  src/estimator.cpp:1354-1539 */
static void marg_backward(const isv_config_t *cfg, const isv_window_t *w, const double *pose, const double *sb,
                          const double *imu_sqrt_info_v /* of factor n_vo-1 -> n_vo */, isv_marg_result_t *out) {
    int v = cfg->n_vo;           /* T1 = frame v (@0), VB1 (@6), T0 = frame v-1 (@15), VB0 (@21) */
    double Lam[900]; memset(Lam, 0, sizeof(Lam));
    {   /* VB prior on sb[v-1] (:1372-1380) */
        double W9[81]; mm_tn(w->vb_prior->sqrt_info, w->vb_prior->sqrt_info, W9, 9, 9, 9);
        for (int a = 0; a < 9; a++) for (int b = 0; b < 9; b++) Lam[(21 + a) * 30 + 21 + b] += W9[a * 9 + b];
    }
    {   /* IMU factor v-1 -> v, unweighted Jacobians, omega = sqrt_info^T sqrt_info (:1382-1412) */
        double r[15], Jpi7[105], Jsi[135], Jpj7[105], Jsj[135], Jpi[90], Jpj[90], Om[225];
        isvo_imu_eval(&w->imu[v - 1], cfg->gravity, pose + 7 * (v - 1), sb + 9 * (v - 1), pose + 7 * v, sb + 9 * v, NULL, r, Jpi7, Jsi, Jpj7, Jsj);
        compact6(Jpi7, 15, Jpi); compact6(Jpj7, 15, Jpj);
        mm_tn(imu_sqrt_info_v, imu_sqrt_info_v, Om, 15, 15, 15);
        const double *Jb[4] = {Jpi, Jsi, Jpj, Jsj}; int db[4] = {6, 9, 6, 9}; int ib[4] = {15, 21, 0, 6};
        for (int j = 0; j < 4; j++) for (int k = j; k < 4; k++) add_hessian(Lam, 30, Jb[j], db[j], ib[j], Jb[k], db[k], ib[k], Om, 15);
    }
    double Lmm[81], Lmm_inv[81], Lprior[441];
    for (int a = 0; a < 9; a++) for (int b = 0; b < 9; b++) Lmm[a * 9 + b] = Lam[(21 + a) * 30 + 21 + b];
    inv_full_lu(Lmm, Lmm_inv, 9);
    for (int a = 0; a < 21; a++) for (int b = 0; b < 21; b++) {
        double s = Lam[a * 30 + b];
        for (int p = 0; p < 9; p++) { double t = 0; for (int q = 0; q < 9; q++) t += Lmm_inv[p * 9 + q] * Lam[b * 30 + 21 + q]; s -= Lam[a * 30 + 21 + p] * t; }
        Lprior[a * 21 + b] = s;
    }
    const double *PSi = pose + 7 * (v - 1), *PSj = pose + 7 * v;
    quat_t Qi = q_from_pose(PSi), Qj = q_from_pose(PSj);
    double d[3] = {PSj[0] - PSi[0], PSj[1] - PSi[1], PSj[2] - PSi[2]};
    isv_relpose_t *rp = &out->backward_relpose; memset(rp, 0, sizeof(*rp));
    q_rot(q_inv(Qi), d, rp->delta_t); q_to_R(q_mul(q_inv(Qi), Qj), rp->delta_R);
    rp->imu_i = v - 1; rp->imu_j = v;
    double r6[6], Ji7[42], Jj7[42], Jrp_i[36], Jrp_j[36];
    isvo_relpose_eval(rp, NULL, PSi, PSj, r6, Ji7, Jj7); compact6(Ji7, 6, Jrp_i); compact6(Jj7, 6, Jrp_j);
    isv_linear9_t *vb = &out->backward_vb; memset(vb, 0, sizeof(*vb));
    memcpy(vb->VB, sb + 9 * v, 72); vb->index = v;
    isv_rollpitch_t *gp = &out->backward_rollpitch; memset(gp, 0, sizeof(*gp));
    q_to_R(Qi, gp->R); gp->index = v - 1;           /* RollPitchFactor(Qw): R(Rz) */
    double r2[2], Jg7[14], Jg[12], Jyaw[6];
    isvo_rollpitch_eval(gp, NULL, PSi, r2, Jg7); compact6(Jg7, 2, Jg);
    isvo_yaw_jac(PSi, NULL, Jyaw);
    /* Jr rows: relpose 0..5, vb 6..14, rollpitch 15..16, abs position 17..19, yaw 20; cols t1 R1 VB1 t0 R0 */
    double Jr[441]; memset(Jr, 0, sizeof(Jr));
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { Jr[a * 21 + 15 + b] += Jrp_i[a * 6 + b]; Jr[a * 21 + b] += Jrp_j[a * 6 + b]; }
    for (int a = 0; a < 9; a++) Jr[(6 + a) * 21 + 6 + a] += 1.0;
    for (int a = 0; a < 2; a++) for (int b = 0; b < 6; b++) Jr[(15 + a) * 21 + 15 + b] += Jg[a * 6 + b];
    for (int a = 0; a < 3; a++) Jr[(17 + a) * 21 + 15 + a] += 1.0;
    for (int b = 0; b < 6; b++) Jr[20 * 21 + 15 + b] += Jyaw[b];
    double U[441], Dv[21];
    int rank = eig_truncate(Lprior, 21, cfg->alpha, U, Dv);
    double S6[36], S9[81], S2[4], S3[9], S1[1], X6[36], X9[81], X2[4], X3[9], X1[1];
    project_cov(Jr + 0 * 21, 6, 21, U, Dv, rank, S6);   inv_partial_lu(S6, X6, 6);   llt_upper(X6, 6, rp->sqrt_info);
    project_cov(Jr + 6 * 21, 9, 21, U, Dv, rank, S9);   inv_partial_lu(S9, X9, 9);   llt_upper(X9, 9, vb->sqrt_info);
    project_cov(Jr + 15 * 21, 2, 21, U, Dv, rank, S2);  inv_partial_lu(S2, X2, 2);   llt_upper(X2, 2, gp->sqrt_info);
    project_cov(Jr + 17 * 21, 3, 21, U, Dv, rank, S3);  inv_partial_lu(S3, X3, 3);
    project_cov(Jr + 20 * 21, 1, 21, U, Dv, rank, S1);  X1[0] = 1.0 / S1[0];
    /* zero test / KLD (:1519-1534): A = (Jr U)^T X (Jr U) vs D */
    double X[441]; memset(X, 0, sizeof(X));
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) X[a * 21 + b] = X6[a * 6 + b];
    for (int a = 0; a < 9; a++) for (int b = 0; b < 9; b++) X[(6 + a) * 21 + 6 + b] = X9[a * 9 + b];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) X[(15 + a) * 21 + 15 + b] = X2[a * 2 + b];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) X[(17 + a) * 21 + 17 + b] = X3[a * 3 + b];
    X[20 * 21 + 20] = X1[0];
    double *JU = (double *)malloc(8 * 21 * rank), *XJU = (double *)malloc(8 * 21 * rank), *A = (double *)malloc(8 * rank * rank + 8);
    for (int a = 0; a < 21; a++) for (int k = 0; k < rank; k++) { double s = 0; for (int c = 0; c < 21; c++) s += Jr[a * 21 + c] * U[c * 21 + k]; JU[a * rank + k] = s; }
    mm(X, JU, XJU, 21, 21, rank);
    mm_tn(JU, XJU, A, rank, 21, rank);
    double tr = 0, ldinv = 0;
    for (int k = 0; k < rank; k++) { tr += A[k * rank + k] / Dv[k]; ldinv += log(1.0 / Dv[k]); }
    out->backward_kld = 0.5 * (tr - log(det_lu(A, rank)) - ldinv - 21);
    free(JU); free(XJU); free(A);
}

/* ------------------------------------------------------------------------------------------ */
/* public oracle API (ctypes-loaded by tests / bench cpu_baseline)                              */
int isvo_optimize(const isv_config_t *cfg, isv_window_t *w, isv_summary_t *sum, isv_marg_result_t *marg) {
    problem_t P; build_problem(&P, cfg, w);
    vector2double(w, cfg->n_frames, P.pose, P.sb, P.ex, P.lam);
    minimize(&P, cfg->num_iterations, sum);
    update_priors(cfg, w, P.pose, P.sb);
    double2vector(cfg, w, P.pose, P.sb, P.ex, P.lam);
    if (w->para_Pose) memcpy(w->para_Pose, P.pose, 56 * P.N);
    if (w->para_SpeedBias) memcpy(w->para_SpeedBias, P.sb, 72 * P.N);
    if (w->para_Ex_Pose) memcpy(w->para_Ex_Pose, P.ex, 56);
    if (w->para_Feature) memcpy(w->para_Feature, P.lam, 8 * P.L);
    if (marg) {
        memset(marg, 0, sizeof(*marg));
        if (w->margin_old) {
            marg->valid = 1;
            marg_forward(cfg, w, P.pose, P.ex, P.lam, marg);
            marg_backward(cfg, w, P.pose, P.sb, P.imu_sqrt_info + 225 * (cfg->n_vo - 1), marg);
        }
    }
    free_problem(&P);
    return ISV_OK;
}

/* isvo_optimize over n independent windows on `nthreads` host threads (static block partition over windows): the
 * all-cores leg of bench.py's cpu_baseline.  The reference's problemSolve is single-threaded (src/estimator.cpp:1122),
 * so more cores only help across windows. */
#include <pthread.h>
typedef struct { const isv_config_t *cfg; isv_window_t *const *w; isv_summary_t *sum; isv_marg_result_t *marg; int b0, b1; } mt_job_t;
static void *mt_worker(void *arg) {
    mt_job_t *j = (mt_job_t *)arg;
    for (int b = j->b0; b < j->b1; b++) isvo_optimize(j->cfg, j->w[b], &j->sum[b], j->marg ? &j->marg[b] : NULL);
    return NULL;
}
int isvo_optimize_batch_mt(const isv_config_t *cfg, isv_window_t *const *w, int n, int nthreads, isv_summary_t *sum, isv_marg_result_t *marg) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    mt_job_t *jobs = (mt_job_t *)malloc(sizeof(mt_job_t) * nthreads);
    for (int k = 0; k < nthreads; k++) {
        jobs[k] = (mt_job_t){cfg, w, sum, marg, (int)((long long)n * k / nthreads), (int)((long long)n * (k + 1) / nthreads)};
        pthread_create(&th[k], NULL, mt_worker, &jobs[k]);
    }
    for (int k = 0; k < nthreads; k++) pthread_join(th[k], NULL);
    free(th); free(jobs);
    return ISV_OK;
}

/* Estimator::initFactorGraph  src/estimator.cpp:667-1001 (one-time, INITIAL_STRUCTURE -> NON_LINEAR).
 *  1. solve the window WITHOUT prior factors: IMU + reprojection factors only, max_num_iterations = 3 NUM_ITERATIONS
 *     (:677-742; the 1 s wall-clock cap and num_threads = 2 of the reference are not restated);
 *  2. Lambda (15 Vo)^2 from the first Vo-1 IMU factors, unweighted Jacobians and omega = sqrt_info^T sqrt_info, order
 *     [T0..T_{Vo-1}, VB_{Vo-1}, VB_0..VB_{Vo-2}] (:744-806); Schur out VB_0..VB_{Vo-2} with a fullPivLu inverse (:810-817);
 *  3. new factors at the solved estimate: RelativePose (i, i+1) for i < Vo-1, SE3 prior on pose 0, Linear9 on
 *     speed/bias Vo-1, unweighted Jacobians stacked into Jr (:821-920);
 *  4. eigen-truncate Lambda_prior at ALPHA, per factor Sigma_i = (J_i U) D^-1 (J_i U)^T, sqrt_info = chol(Sigma_i^-1)^T
 *     (:927-974); KLD of the recovered factors against the truncated marginal (:976-989);
 *  5. double2vector() (:999).
 * The window's prior structs are OUTPUTS (vioRelativePoseEdges[1..], vioPosePriorEdge, vioVBPrior); no roll-pitch edge
 * exists after initialisation. */
int isvo_init_factor_graph(const isv_config_t *cfg, isv_window_t *w, isv_summary_t *sum, double *kld_out) {
    const int V = cfg->n_vo;
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memset(w->pose_prior, 0, sizeof(*w->pose_prior)); memcpy(w->pose_prior->R, I3, 72);
    memset(w->vb_prior, 0, sizeof(*w->vb_prior)); w->vb_prior->index = V - 1;
    for (int i = 0; i < V - 1; i++) { memset(&w->relpose[i], 0, sizeof(isv_relpose_t)); memcpy(w->relpose[i].delta_R, I3, 72); w->relpose[i].imu_i = i; w->relpose[i].imu_j = i + 1; }
    w->n_rollpitch = 0;
    problem_t P; build_problem(&P, cfg, w);
    vector2double(w, cfg->n_frames, P.pose, P.sb, P.ex, P.lam);
    isv_summary_t local; if (!sum) sum = &local;
    minimize(&P, 3 * cfg->num_iterations, sum);
    const double *pose = P.pose, *sb = P.sb;
    const int n = 15 * V, rr = 6 * V + 9, mmd = 9 * (V - 1);
    double *Lam = (double *)calloc((size_t)n * n, 8);
    for (int i = 0; i < V - 1; i++) {
        if (w->imu[i].sum_dt > 10.0) continue;
        double r[15], Jpi7[105], Jsi[135], Jpj7[105], Jsj[135], Jpi[90], Jpj[90], Om[225];
        isvo_imu_eval(&w->imu[i], cfg->gravity, pose + 7 * i, sb + 9 * i, pose + 7 * (i + 1), sb + 9 * (i + 1), NULL, r, Jpi7, Jsi, Jpj7, Jsj);
        compact6(Jpi7, 15, Jpi); compact6(Jpj7, 15, Jpj);
        const double *S = P.imu_sqrt_info + 225 * i;
        mm_tn(S, S, Om, 15, 15, 15);
        const int j = i + 1;
        const int si_off = (i == V - 1) ? 6 * V : 6 * V + 9 + 9 * i, sj_off = (j == V - 1) ? 6 * V : 6 * V + 9 + 9 * j;
        const double *Jb[4] = {Jpi, Jsi, Jpj, Jsj}; int db[4] = {6, 9, 6, 9}; int ib[4] = {6 * i, si_off, 6 * j, sj_off};
        for (int a = 0; a < 4; a++) for (int b = a; b < 4; b++) add_hessian(Lam, n, Jb[a], db[a], ib[a], Jb[b], db[b], ib[b], Om, 15);
    }
    double *Lmm = (double *)malloc(8 * mmd * mmd), *Lmm_inv = (double *)malloc(8 * mmd * mmd), *T = (double *)malloc(8 * mmd * rr);
    double *Lp = (double *)malloc(8 * rr * rr);
    for (int a = 0; a < mmd; a++) for (int b = 0; b < mmd; b++) Lmm[a * mmd + b] = Lam[(size_t)(rr + a) * n + rr + b];
    inv_full_lu(Lmm, Lmm_inv, mmd);
    for (int p = 0; p < mmd; p++) for (int b = 0; b < rr; b++) { double s = 0; for (int q = 0; q < mmd; q++) s += Lmm_inv[p * mmd + q] * Lam[(size_t)b * n + rr + q]; T[p * rr + b] = s; }
    for (int a = 0; a < rr; a++) for (int b = 0; b < rr; b++) { double s = Lam[(size_t)a * n + b]; for (int p = 0; p < mmd; p++) s -= Lam[(size_t)a * n + rr + p] * T[p * rr + b]; Lp[a * rr + b] = s; }
    /* recovered factors and their stacked unweighted Jacobians */
    double *Jr = (double *)calloc((size_t)rr * rr, 8);
    for (int i = 0; i < V - 1; i++) {
        const double *PSi = pose + 7 * i, *PSj = pose + 7 * (i + 1);
        quat_t Qi = q_from_pose(PSi), Qj = q_from_pose(PSj);
        double dd[3] = {PSj[0] - PSi[0], PSj[1] - PSi[1], PSj[2] - PSi[2]};
        isv_relpose_t *rp = &w->relpose[i]; memset(rp, 0, sizeof(*rp));
        q_rot(q_inv(Qi), dd, rp->delta_t); q_to_R(q_mul(q_inv(Qi), Qj), rp->delta_R);
        rp->imu_i = i; rp->imu_j = i + 1;
        double r6[6], Ji7[42], Jj7[42], Ji[36], Jj[36];
        isvo_relpose_eval(rp, NULL, PSi, PSj, r6, Ji7, Jj7); compact6(Ji7, 6, Ji); compact6(Jj7, 6, Jj);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { Jr[(size_t)(6 * i + a) * rr + 6 * i + b] += Ji[a * 6 + b]; Jr[(size_t)(6 * i + a) * rr + 6 * (i + 1) + b] += Jj[a * 6 + b]; }
    }
    {
        isv_se3_prior_t *pp = w->pose_prior; memset(pp, 0, sizeof(*pp));
        memcpy(pp->t, pose, 24); q_to_R(q_from_pose(pose), pp->R); pp->index = 0;
        double r6[6], J7[42], J6[36];
        isvo_se3prior_eval(pp, NULL, pose, r6, J7); compact6(J7, 6, J6);
        const int r0 = 6 * (V - 1);
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) Jr[(size_t)(r0 + a) * rr + b] += J6[a * 6 + b];
        isv_linear9_t *vb = w->vb_prior; memset(vb, 0, sizeof(*vb));
        memcpy(vb->VB, sb + 9 * (V - 1), 72); vb->index = V - 1;
        for (int a = 0; a < 9; a++) Jr[(size_t)(r0 + 6 + a) * rr + 6 * V + a] += 1.0;     /* Linear9Factor::EvaluateOnlyJacobians: identity */
    }
    double *U = (double *)malloc(8 * rr * rr), *Dv = (double *)malloc(8 * rr);
    const int rank = eig_truncate(Lp, rr, cfg->alpha, U, Dv);
    double *X = (double *)calloc((size_t)rr * rr, 8);
    int hdim = 0;
    for (int i = 0; i < V + 1; i++) {               /* V-1 relative poses, the pose prior, the speed/bias prior */
        const int rows = i < V ? 6 : 9;
        double Sg[81], Xi[81];
        project_cov(Jr + (size_t)hdim * rr, rows, rr, U, Dv, rank, Sg);
        inv_partial_lu(Sg, Xi, rows);
        double *dst = i < V - 1 ? w->relpose[i].sqrt_info : (i == V - 1 ? w->pose_prior->sqrt_info : w->vb_prior->sqrt_info);
        llt_upper(Xi, rows, dst);
        for (int a = 0; a < rows; a++) for (int b = 0; b < rows; b++) X[(size_t)(hdim + a) * rr + hdim + b] = Xi[a * rows + b];
        hdim += rows;
    }
    if (kld_out) {                                  /* A = (Jr U)^T X (Jr U) against D (:976-989) */
        double *JU = (double *)malloc(8 * rr * rank), *XJU = (double *)malloc(8 * rr * rank), *A = (double *)malloc(8 * rank * rank + 8);
        for (int a = 0; a < rr; a++) for (int k = 0; k < rank; k++) { double s = 0; for (int c = 0; c < rr; c++) s += Jr[(size_t)a * rr + c] * U[c * rr + k]; JU[a * rank + k] = s; }
        mm(X, JU, XJU, rr, rr, rank);
        mm_tn(JU, XJU, A, rank, rr, rank);
        double tr = 0, ldinv = 0;
        for (int k = 0; k < rank; k++) { tr += A[k * rank + k] / Dv[k]; ldinv += log(1.0 / Dv[k]); }
        /* the reference takes log(A.determinant()) and log(Dinv.determinant()) (:985-986), which overflow / underflow for
         * 6 Vo + 9 = 57 eigenvalues around 1e8 (its KLD is only a commented-out print); the log domain gives the intended value */
        *kld_out = 0.5 * (tr - logabsdet_lu(A, rank) - ldinv - rr);
        free(JU); free(XJU); free(A);
    }
    double2vector(cfg, w, P.pose, P.sb, P.ex, P.lam);
    if (w->para_Pose) memcpy(w->para_Pose, P.pose, 56 * P.N);
    if (w->para_SpeedBias) memcpy(w->para_SpeedBias, P.sb, 72 * P.N);
    if (w->para_Ex_Pose) memcpy(w->para_Ex_Pose, P.ex, 56);
    if (w->para_Feature) memcpy(w->para_Feature, P.lam, 8 * P.L);
    free(Lam); free(Lmm); free(Lmm_inv); free(T); free(Lp); free(Jr); free(U); free(Dv); free(X);
    free_problem(&P);
    return ISV_OK;
}

/* FeatureManager::triangulate  src/feature_tracker/feature_manager.cpp:206-258.  For every landmark without a positive
 * depth: DLT matrix A (2k x 4) over all k views in the host camera frame (:219-242), V = right singular vector of the
 * smallest singular value (Eigen::JacobiSVD, :244), depth = V[2] / V[3] (:245), INIT_DEPTH outside [0.1, 8] (:252-255).
 * The SVD here is a one-sided (Hestenes) Jacobi on the columns of A: A V = U S, V accumulated from the rotations. */
int isvo_triangulate(const isv_config_t *cfg, isv_window_t *w) {
    for (int l = 0; l < w->n_landmarks; l++) {
        if (w->lm_depth[l] > 0) continue;
        const int h = w->lm_start_frame[l], o0 = w->lm_obs_ptr[l], k = w->lm_obs_ptr[l + 1] - o0;
        double A[64][4], V[4][4];                      /* k <= 32 views */
        double R0[9], t0[3], tmp[3];
        mm(w->Rs + 9 * h, w->ric, R0, 3, 3, 3);
        m3v(w->Rs + 9 * h, w->tic, tmp);
        for (int c = 0; c < 3; c++) t0[c] = w->Ps[3 * h + c] + tmp[c];
        for (int o = 0; o < k; o++) {
            const int j = h + o;
            double R1[9], t1[3], dt[3], t[3], R[9], P[12], mt[3];
            mm(w->Rs + 9 * j, w->ric, R1, 3, 3, 3);
            m3v(w->Rs + 9 * j, w->tic, tmp);
            for (int c = 0; c < 3; c++) { t1[c] = w->Ps[3 * j + c] + tmp[c]; dt[c] = t1[c] - t0[c]; }
            m3tv(R0, dt, t);
            for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) { double s = 0; for (int q = 0; q < 3; q++) s += R0[q * 3 + a] * R1[q * 3 + c]; R[a * 3 + c] = s; }
            m3tv(R, t, mt);
            for (int a = 0; a < 3; a++) { for (int c = 0; c < 3; c++) P[a * 4 + c] = R[c * 3 + a]; P[a * 4 + 3] = -mt[a]; }
            const double *pt = w->obs_point + (size_t)(o0 + o) * 3;
            const double nrm = sqrt(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]);
            const double f[3] = {pt[0] / nrm, pt[1] / nrm, pt[2] / nrm};
            for (int c = 0; c < 4; c++) { A[2 * o][c] = f[0] * P[8 + c] - f[2] * P[c]; A[2 * o + 1][c] = f[1] * P[8 + c] - f[2] * P[4 + c]; }
        }
        const int m = 2 * k;
        for (int a = 0; a < 4; a++) for (int c = 0; c < 4; c++) V[a][c] = a == c;
        for (int sweep = 0; sweep < 60; sweep++) {
            int rotated = 0;
            for (int p = 0; p < 3; p++) for (int q = p + 1; q < 4; q++) {
                double app = 0, aqq = 0, apq = 0;
                for (int r = 0; r < m; r++) { app += A[r][p] * A[r][p]; aqq += A[r][q] * A[r][q]; apq += A[r][p] * A[r][q]; }
                if (fabs(apq) <= 1e-17 * sqrt(app * aqq) || apq == 0.0) continue;
                rotated = 1;
                const double theta = (aqq - app) / (2.0 * apq);
                const double tq = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(tq * tq + 1.0), s = tq * c;
                for (int r = 0; r < m; r++) { const double x = A[r][p], y = A[r][q]; A[r][p] = c * x - s * y; A[r][q] = s * x + c * y; }
                for (int r = 0; r < 4; r++) { const double x = V[r][p], y = V[r][q]; V[r][p] = c * x - s * y; V[r][q] = s * x + c * y; }
            }
            if (!rotated) break;
        }
        int best = 0; double bn = 1e300;
        for (int c = 0; c < 4; c++) { double nn = 0; for (int r = 0; r < m; r++) nn += A[r][c] * A[r][c]; if (nn < bn) { bn = nn; best = c; } }
        double dep = V[2][best] / V[3][best];
        if (dep < 0.1 || dep > 8.0) dep = cfg->init_depth;
        w->lm_depth[l] = dep;
    }
    return 0;
}

/* one ceres::Problem::Evaluate at the window's current state -> strips in the product's layout */
int isvo_linearize(const isv_config_t *cfg, const isv_window_t *w, double *proj_strips, double *imu_strips,
                   double *prior_res, double *cost) {
    problem_t P; build_problem(&P, cfg, w);
    vector2double(w, cfg->n_frames, P.pose, P.sb, P.ex, P.lam);
    double c = evaluate(&P, P.pose, P.sb, P.lam, 1);
    if (cost) *cost = c;
    int f = 0, pr = 0;
    for (int n = 0; n < P.nrb; n++) {
        rblock_t *r = &P.rb[n];
        if (r->kind == 1 && proj_strips) {
            double *s = proj_strips + (size_t)ISV_PROJ_STRIP * f++;
            s[0] = P.res[r->roff]; s[1] = P.res[r->roff + 1];
            memcpy(s + 2, P.jac + r->joff[0], 96); memcpy(s + 14, P.jac + r->joff[1], 96); memcpy(s + 26, P.jac + r->joff[2], 16);
        } else if (r->kind == 0 && imu_strips) {
            double *s = imu_strips + (size_t)ISV_IMU_STRIP * r->a;
            memcpy(s, P.res + r->roff, 120);
            memcpy(s + 15, P.jac + r->joff[0], 8 * 90); memcpy(s + 105, P.jac + r->joff[1], 8 * 135);
            memcpy(s + 240, P.jac + r->joff[2], 8 * 90); memcpy(s + 330, P.jac + r->joff[3], 8 * 135);
        } else if (r->kind >= 2 && prior_res) {
            memcpy(prior_res + pr, P.res + r->roff, 8 * r->dim); pr += r->dim;
        }
    }
    free_problem(&P);
    return ISV_OK;
}

/* normal equations at the current state, dense, for identity tests: H (ncols x ncols), g */
int isvo_normal_equations(const isv_config_t *cfg, const isv_window_t *w, double *H, double *g, int *ncols_out) {
    problem_t P; build_problem(&P, cfg, w);
    vector2double(w, cfg->n_frames, P.pose, P.sb, P.ex, P.lam);
    evaluate(&P, P.pose, P.sb, P.lam, 1);
    int n = P.ncols;
    if (ncols_out) *ncols_out = n;
    if (H) {
        memset(H, 0, 8 * (size_t)n * n);
        for (int k = 0; k < P.nrb; k++) {
            rblock_t *r = &P.rb[k];
            for (int a = 0; a < r->nb; a++) for (int b = 0; b < r->nb; b++) {
                const double *Ja = P.jac + r->joff[a], *Jb = P.jac + r->joff[b];
                for (int c = 0; c < r->wid[a]; c++) for (int d = 0; d < r->wid[b]; d++) {
                    double s = 0; for (int e = 0; e < r->dim; e++) s += Ja[e * r->wid[a] + c] * Jb[e * r->wid[b] + d];
                    H[(size_t)(r->col[a] + c) * n + r->col[b] + d] += s;
                }
            }
        }
    }
    if (g) memcpy(g, P.grad, 8 * n);
    free_problem(&P);
    return ISV_OK;
}
/* the DENSE_SCHUR solve of (J^T J + diag(D^2)) y = J^T r at the current state (unscaled J) */
int isvo_schur_solve(const isv_config_t *cfg, const isv_window_t *w, const double *D, double *y) {
    problem_t P; build_problem(&P, cfg, w);
    vector2double(w, cfg->n_frames, P.pose, P.sb, P.ex, P.lam);
    evaluate(&P, P.pose, P.sb, P.lam, 1);
    int rc = dense_schur_solve(&P, D, y);
    free_problem(&P);
    return rc;
}
/* cost only */
double isvo_cost(const isv_config_t *cfg, const isv_window_t *w) {
    problem_t P; build_problem(&P, cfg, w);
    vector2double(w, cfg->n_frames, P.pose, P.sb, P.ex, P.lam);
    double c = evaluate(&P, P.pose, P.sb, P.lam, 0);
    free_problem(&P);
    return c;
}

/* thin exports of the factor functions for finite-difference tests */
void isvo_x_proj(const double *pi, const double *pj, const double *ex, double lam, const double *pts_i, const double *pts_j,
                 const double *sqrt_info, int weighted, double *res, double *Ji, double *Jj, double *Jex, double *Jl) {
    isvo_proj_eval(pi, pj, ex, lam, pts_i, pts_j, sqrt_info, weighted, res, Ji, Jj, Jex, Jl);
}
void isvo_x_imu(const isv_imu_t *pre, const double *G, const double *pi, const double *si, const double *pj, const double *sj,
                int weighted, double *res, double *Jpi, double *Jsi, double *Jpj, double *Jsj, double *sqrt_info_out) {
    double sq[225];
    isvo_imu_sqrt_info(pre->covariance, sq);
    if (sqrt_info_out) memcpy(sqrt_info_out, sq, sizeof(sq));
    isvo_imu_eval(pre, G, pi, si, pj, sj, weighted ? sq : NULL, res, Jpi, Jsi, Jpj, Jsj);
}
void isvo_x_se3prior(const isv_se3_prior_t *f, int weighted, const double *pose, double *res, double *J) {
    isvo_se3prior_eval(f, weighted ? f->sqrt_info : NULL, pose, res, J);
}
void isvo_x_linear9(const isv_linear9_t *f, int weighted, const double *sb, double *res, double *J) {
    isvo_linear9_eval(f, weighted ? f->sqrt_info : NULL, sb, res, J);
}
void isvo_x_relpose(const isv_relpose_t *f, int weighted, const double *pi, const double *pj, double *res, double *Ji, double *Jj) {
    isvo_relpose_eval(f, weighted ? f->sqrt_info : NULL, pi, pj, res, Ji, Jj);
}
void isvo_x_rollpitch(const isv_rollpitch_t *f, int weighted, const double *pose, double *res, double *J) {
    isvo_rollpitch_eval(f, weighted ? f->sqrt_info : NULL, pose, res, J);
}
void isvo_x_yaw(const double *pose, double *res, double *J6) { isvo_yaw_jac(pose, res, J6); }
void isvo_x_pose_plus(const double *x, const double *d, double *xp) { isvo_pose_plus(x, d, xp); }
void isvo_x_preint_init(isv_imu_t *p, const double *ba, const double *bg) { isvo_preint_init(p, ba, bg); }
void isvo_x_preint_step(isv_imu_t *p, double dt, const double *a0, const double *g0, const double *a1, const double *g1, const double *noise4) {
    isvo_preint_step(p, dt, a0, g0, a1, g1, noise4);
}
void isvo_x_so3_log(const double *R, double *w) { so3_log(so3_from_R(R), w); }
void isvo_x_so3_exp(const double *w, double *R) { q_to_R(so3_exp(w), R); }
void isvo_x_rjacinv(const double *phi, double *J) { so3_rjac_inv(phi, J); }
void isvo_x_R2q(const double *R, double *q_xyzw) { quat_t q = q_from_R(R); q_xyzw[0] = q.x; q_xyzw[1] = q.y; q_xyzw[2] = q.z; q_xyzw[3] = q.w; }
void isvo_x_update_se3(isv_se3_prior_t *f, const double *P_old, const double *R_old, const double *pose_new) { isvo_se3prior_update(f, P_old, R_old, pose_new); }
void isvo_x_update_relpose(isv_relpose_t *f, const double *ti, const double *Ri, const double *tj, const double *Rj, const double *PSi, const double *PSj) { isvo_relpose_update(f, ti, Ri, tj, Rj, PSi, PSj); }
void isvo_x_update_rollpitch(isv_rollpitch_t *f, const double *R_old, const double *pose_new) { isvo_rollpitch_update(f, R_old, pose_new); }
