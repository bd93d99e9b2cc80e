/*
 * isv_pgo_oracle.c -- CPU ORACLE of the pose-graph optimisation that consumes the backend's CombinedFactors:
 *   CombinedFactors::operator+      /root/reference/include/factor/pose_graph_factors.h:27-51
 *   PoseGraph::optimizeCS           /root/reference/src/pose_graph/pose_graph.cpp:234-428   (one pass of the loop body)
 * plus the part of Ceres-Solver 2.0.0 (external, README.md:22) that optimizeCS drives with default options:
 * TrustRegionMinimizer + LevenbergMarquardtStrategy (levenberg_marquardt_strategy.cc), HuberLoss + Corrector
 * (loss_function.cc, corrector.cc), Jacobi scaling, and ceres::Covariance of the pose blocks (covariance_impl.cc:
 * (J^T J)^-1 in the tangent space, lifted with the local parameterisation's Jacobian), restated from the published
 * algorithm.  SPARSE_NORMAL_CHOLESKY is a linear-solver choice: the step solves the same normal equations, formed
 * densely here.
 *
 * TEST INFRASTRUCTURE ONLY (tests/ and nothing else loads it).  PARITY UNPINNED: the reference ships no pose-graph
 * fixtures and cannot be built here; what pins this file is in tests/test_oracle_pgo.py (finite differences, the
 * closed-loop known answer, covariance == inverse Hessian, operator+ against the SE(3) composition identities).
 *
 * Reference quirks kept (they change the stored results): the edge loop stops BEFORE cur_index, so the newest loop
 * edge is not part of the solve it triggered (:312-313); covariance blocks are fetched in the 7-dim AMBIENT space into a
 * 36-double buffer and mapped as a column-major 6x6 (:346-350), i.e. the stored `cov` is the first 36 doubles of the
 * row-major 7x7 [Sigma 0; 0 0] re-read column-major; the update() of every relative-pose factor runs AFTER updatePose
 * and therefore sees old == new (:375-378).  Not reproducible (undefined behaviour in the reference): updateCov of
 * cur_index reads poses_cov[param_index] one past the end (:374) -- cur's cov is left untouched here.
 */
#include <stdio.h>
#include <float.h>
#include "isvo_factors.h"
#include "../include/isvins_posegraph.h"

/* ------------------------------------------------------------------------------------------ */
/* Sophus::SE3d::Adj() with tangent order [upsilon; omega]: [[R, [t]x R], [0, R]] */
static void se3_adj(const double *R, const double *t, double *A /* 6x6 */) {
    double S[9], SR[9];
    skew(t, S); mm(S, R, SR, 3, 3, 3);
    memset(A, 0, 36 * sizeof(double));
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        A[a * 6 + b] = R[a * 3 + b]; A[a * 6 + 3 + b] = SR[a * 3 + b]; A[(3 + a) * 6 + 3 + b] = R[a * 3 + b];
    }
}
static int llt_upper6(const double *M, double *U) {
    double L[36]; memcpy(L, M, sizeof(L));
    int info = chol_lower(L, 6);
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) U[i * 6 + j] = j >= i ? L[j * 6 + i] : 0.0;
    return info;
}

/* CombinedFactors::operator+  pose_graph_factors.h:27-51 */
int isvo_combined_factors_add(isv_combined_factors_t *acc, int32_t *acc_length, int64_t *acc_vio_index,
                              const isv_combined_factors_t *other, int64_t other_vio_index) {
    const double *R0 = acc->relative_pose.delta_R, *t0 = acc->relative_pose.delta_t;
    const double *R1 = other->relative_pose.delta_R, *t1 = other->relative_pose.delta_t;
    double W[36], cov1[36], Adj[36], T[36], T2[36];
    mm_tn(other->relative_pose.sqrt_info, other->relative_pose.sqrt_info, W, 6, 6, 6);
    inv_partial_lu(W, cov1, 6);                                  /* covRel1 = (S^T S)^-1 */
    se3_adj(R0, t0, Adj);
    mm(Adj, cov1, T, 6, 6, 6); mm_nt(T, Adj, T2, 6, 6, 6);
    for (int k = 0; k < 36; k++) acc->covRel[k] += T2[k];        /* covRel += Adj covRel1 Adj^T */
    acc->has_rollpitch = other->has_rollpitch; acc->rollpitch = other->rollpitch;      /* rollPitchFactor = other's (covAbs is not touched) */
    double Rn[9], tn[3], v[3], info[36];
    mm(R0, R1, Rn, 3, 3, 3); m3v(R0, t1, v);
    for (int k = 0; k < 3; k++) tn[k] = v[k] + t0[k];            /* T0 * T1 */
    memcpy(acc->relative_pose.delta_R, Rn, sizeof(Rn)); memcpy(acc->relative_pose.delta_t, tn, sizeof(tn));
    inv_partial_lu(acc->covRel, info, 6);
    llt_upper6(info, acc->relative_pose.sqrt_info);              /* LLT(covRel^-1).matrixL().transpose() */
    acc->distance = sqrt(dotn(tn, tn, 3));
    (*acc_length)++;
    if (*acc_vio_index == -1) {
        memcpy(acc->ti, other->ti, 24); memcpy(acc->Ri, other->Ri, 72);
        *acc_vio_index = other_vio_index; acc->ts = other->ts;
    }
    return ISV_OK;
}

/* ------------------------------------------------------------------------------------------ */
typedef struct { int kind, a, b, kf, robust, dim, roff; } pg_edge_t;   /* kind 0 rollpitch(a), 1 relpose(a,b), 2 loop(a=connected,b=li) */
typedef struct {
    int P1;                        /* parameter blocks (local indices 0..P1-1) */
    int nfree, ncols, nres, nedges;
    int *col;                      /* tangent column offset of block k, -1 if constant */
    double *pose, *cand;           /* [P1][7] */
    pg_edge_t *edges;
    isv_relpose_t *loopf;          /* loop factors (one per loop edge) */
    const isv_pg_keyframe_t **kf;  /* keyframe of local index k */
    double huber;
    double *res, *jac;             /* residuals [nres]; Jacobians per edge: [dim x 6] for block a then block b, local param applied */
} pg_problem_t;

static void take6p(const double *J7, int rows, double *J6) { for (int r = 0; r < rows; r++) for (int c = 0; c < 6; c++) J6[r * 6 + c] = J7[r * 7 + c]; }

/* residual blocks at `pose` (cost, corrected residuals and Jacobians: ResidualBlock::Evaluate + Corrector) */
static double pg_evaluate(pg_problem_t *P, const double *pose, int want_jac) {
    double cost = 0;
    for (int e = 0; e < P->nedges; e++) {
        pg_edge_t *E = &P->edges[e];
        double r[6], Ja7[42], Jb7[42];
        double *Ja = P->jac + (size_t)e * 72, *Jb = Ja + 36;
        if (E->kind == 0) {
            const isv_rollpitch_t *f = &P->kf[E->a]->rollpitch;
            isvo_rollpitch_eval(f, f->sqrt_info, pose + 7 * E->a, r, want_jac ? Ja7 : NULL);
            if (want_jac) take6p(Ja7, 2, Ja);
        } else {
            const isv_relpose_t *f = E->kind == 1 ? &P->kf[E->a]->relative_pose : &P->loopf[E->kf];
            isvo_relpose_eval(f, f->sqrt_info, pose + 7 * E->a, pose + 7 * E->b, r, want_jac ? Ja7 : NULL, want_jac ? Jb7 : NULL);
            if (want_jac) { take6p(Ja7, 6, Ja); take6p(Jb7, 6, Jb); }
        }
        const double s = dotn(r, r, E->dim);
        double sc = 1.0;
        if (E->robust) {                       /* HuberLoss(a): rho = [s, 1, 0] (s <= a^2) or [2 a sqrt(s) - a^2, a / sqrt(s), .] */
            const double a = P->huber, b = a * a;
            if (s > b) { const double rr = sqrt(s); cost += 0.5 * (2.0 * a * rr - b); sc = sqrt(fmax(DBL_MIN, a / rr)); }
            else cost += 0.5 * s;
        } else cost += 0.5 * s;
        if (want_jac) {                        /* Corrector: rho'' <= 0 for Huber, so residual and Jacobians scale by sqrt(rho') */
            for (int k = 0; k < E->dim; k++) P->res[E->roff + k] = r[k] * sc;
            for (int k = 0; k < E->dim * 6; k++) Ja[k] *= sc;
            if (E->kind != 0) for (int k = 0; k < E->dim * 6; k++) Jb[k] *= sc;
        }
    }
    return cost;
}
/* iterate the (block, Jacobian) pairs of an edge */
#define PG_BLOCKS(P, e, E, nb, blk, Jp)                                                                         \
    const pg_edge_t *E = &(P)->edges[e]; const int nb = E->kind == 0 ? 1 : 2;                                   \
    const int blk[2] = {E->a, E->b}; const double *Jp[2] = {(P)->jac + (size_t)(e) * 72, (P)->jac + (size_t)(e) * 72 + 36};

static void pg_scale_cols(pg_problem_t *P, const double *s) {
    for (int e = 0; e < P->nedges; e++) {
        PG_BLOCKS(P, e, E, nb, blk, Jp)
        for (int k = 0; k < nb; k++) { const int c0 = P->col[blk[k]]; if (c0 < 0) continue;
            double *J = (double *)Jp[k];
            for (int r = 0; r < E->dim; r++) for (int c = 0; c < 6; c++) J[r * 6 + c] *= s[c0 + c]; }
    }
}
static void pg_colnorm2(const pg_problem_t *P, double *out) {
    memset(out, 0, sizeof(double) * P->ncols);
    for (int e = 0; e < P->nedges; e++) {
        PG_BLOCKS(P, e, E, nb, blk, Jp)
        for (int k = 0; k < nb; k++) { const int c0 = P->col[blk[k]]; if (c0 < 0) continue;
            for (int r = 0; r < E->dim; r++) for (int c = 0; c < 6; c++) out[c0 + c] += Jp[k][r * 6 + c] * Jp[k][r * 6 + c]; }
    }
}
static void pg_left_mul(const pg_problem_t *P, const double *rin, double *g) {      /* g += J^T r */
    for (int e = 0; e < P->nedges; e++) {
        PG_BLOCKS(P, e, E, nb, blk, Jp)
        for (int k = 0; k < nb; k++) { const int c0 = P->col[blk[k]]; if (c0 < 0) continue;
            for (int r = 0; r < E->dim; r++) for (int c = 0; c < 6; c++) g[c0 + c] += Jp[k][r * 6 + c] * rin[E->roff + r]; }
    }
}
static void pg_right_mul(const pg_problem_t *P, const double *x, double *y) {        /* y += J x */
    for (int e = 0; e < P->nedges; e++) {
        PG_BLOCKS(P, e, E, nb, blk, Jp)
        for (int k = 0; k < nb; k++) { const int c0 = P->col[blk[k]]; if (c0 < 0) continue;
            for (int r = 0; r < E->dim; r++) { double s = 0; for (int c = 0; c < 6; c++) s += Jp[k][r * 6 + c] * x[c0 + c]; y[E->roff + r] += s; } }
    }
}
/* H = J^T J (dense, ncols x ncols) */
static void pg_hessian(const pg_problem_t *P, double *H) {
    const int n = P->ncols;
    memset(H, 0, sizeof(double) * (size_t)n * n);
    for (int e = 0; e < P->nedges; e++) {
        PG_BLOCKS(P, e, E, nb, blk, Jp)
        for (int k = 0; k < nb; k++) for (int m = 0; m < nb; m++) {
            const int ck = P->col[blk[k]], cm = P->col[blk[m]]; if (ck < 0 || cm < 0) continue;
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
                double s = 0; for (int r = 0; r < E->dim; r++) s += Jp[k][r * 6 + a] * Jp[m][r * 6 + b];
                H[(size_t)(ck + a) * n + cm + b] += s;
            }
        }
    }
}
/* ---- the CPU BASELINE's linear algebra (bench.py, pose_graph_optimisation.cpp_baseline): the same LM loop on a SKYLINE
 * (envelope) Cholesky.  In the keyframes' order the normal equations are block tridiagonal plus one long row per loop
 * closure, and a Cholesky factor fills only inside the row envelope -- what a sparse direct solver (the reference's
 * SPARSE_NORMAL_CHOLESKY, pose_graph.cpp:266) exploits, and what k_pgo stores.  isvo_pgo_set_sparse(1) switches
 * pg_minimize and the covariance step to it; the dense path stays the CHECKER of the GPU tests (tests/test_oracle_pgo.py
 * compares the two).  Storage stays the dense n x n array, only envelope entries are touched. */
static int g_pgo_sparse = 0;
void isvo_pgo_set_sparse(int on) { g_pgo_sparse = on; }
/* first[i]: column of the first non-zero of row i of H (lower triangle) */
static void pg_envelope(const pg_problem_t *P, int *first) {
    const int n = P->ncols;
    for (int i = 0; i < n; i++) first[i] = i - i % 6;
    for (int e = 0; e < P->nedges; e++) {
        PG_BLOCKS(P, e, E, nb, blk, Jp)
        (void)Jp;
        for (int k = 0; k < nb; k++) for (int m = 0; m < nb; m++) {
            const int ck = P->col[blk[k]], cm = P->col[blk[m]];
            if (ck < 0 || cm < 0 || cm >= ck) continue;
            for (int a = 0; a < 6; a++) if (cm < first[ck + a]) first[ck + a] = cm;
        }
    }
}
static void pg_hessian_env(const pg_problem_t *P, double *H, const int *first) {
    const int n = P->ncols;
    for (int i = 0; i < n; i++) memset(H + (size_t)i * n + first[i], 0, sizeof(double) * (size_t)(i - first[i] + 1));
    for (int e = 0; e < P->nedges; e++) {
        PG_BLOCKS(P, e, E, nb, blk, Jp)
        for (int k = 0; k < nb; k++) for (int m = 0; m < nb; m++) {
            const int ck = P->col[blk[k]], cm = P->col[blk[m]]; if (ck < 0 || cm < 0 || cm > ck) continue;
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
                if (cm + b > ck + a) continue;
                double s = 0; for (int r = 0; r < E->dim; r++) s += Jp[k][r * 6 + a] * Jp[m][r * 6 + b];
                H[(size_t)(ck + a) * n + cm + b] += s;
            }
        }
    }
}
static int chol_skyline(double *A, int n, const int *first) {
    for (int i = 0; i < n; i++) {
        double *Li = A + (size_t)i * n;
        for (int j = first[i]; j < i; j++) {
            const double *Lj = A + (size_t)j * n;
            double s = Li[j];
            for (int k = first[i] > first[j] ? first[i] : first[j]; k < j; k++) s -= Li[k] * Lj[k];
            Li[j] = s / Lj[j];
        }
        double s = Li[i];
        for (int k = first[i]; k < i; k++) s -= Li[k] * Li[k];
        if (!(s > 0.0)) return 1;
        Li[i] = sqrt(s);
    }
    return 0;
}
/* L L^T x = b in place; from: first row with a non-zero right-hand side (unit vectors of the covariance step) */
static void chol_skyline_solve(const double *L, int n, const int *first, double *b, int from) {
    for (int i = from; i < n; i++) {
        const double *Li = L + (size_t)i * n;
        double s = b[i];
        for (int k = first[i] > from ? first[i] : from; k < i; k++) s -= Li[k] * b[k];
        b[i] = s / Li[i];
    }
    for (int i = n - 1; i >= 0; i--) {
        const double *Li = L + (size_t)i * n;
        const double x = b[i] / Li[i];
        b[i] = x;
        for (int k = first[i]; k < i; k++) b[k] -= Li[k] * x;
    }
}

static void pg_plus(const pg_problem_t *P, const double *pose, const double *delta, double *out) {
    for (int k = 0; k < P->P1; k++) {
        if (P->col[k] < 0) memcpy(out + 7 * k, pose + 7 * k, 56);
        else isvo_pose_plus(pose + 7 * k, delta + P->col[k], out + 7 * k);
    }
}
static double pg_norm2_diff(const pg_problem_t *P, const double *a, const double *b) {      /* over the NON-constant blocks (reduced program) */
    double s = 0;
    for (int k = 0; k < P->P1; k++) if (P->col[k] >= 0) for (int c = 0; c < 7; c++) { const double d = a[7 * k + c] - (b ? b[7 * k + c] : 0); s += d * d; }
    return s;
}

/* TrustRegionMinimizer::Minimize with LevenbergMarquardtStrategy, Ceres 2.0.0 defaults: initial radius 1e4, max 1e16,
 * min 1e-32, min/max LM diagonal 1e-6 / 1e32, min_relative_decrease 1e-3, tolerances 1e-6 / 1e-10 / 1e-8, jacobi scaling,
 * monotonic steps, at most 5 consecutive invalid steps. */
static void pg_minimize(pg_problem_t *P, int max_iter, isv_pgo_result_t *out) {
    const int n = P->ncols;
    double *scale = calloc(n + 1, 8), *diag = calloc(n + 1, 8), *g = calloc(n + 1, 8), *step = calloc(n + 1, 8), *delta = calloc(n + 1, 8);
    double *H = calloc((size_t)n * n + 1, 8), *mres = calloc(P->nres + 1, 8), *neg = calloc(n + 1, 8), *pp = calloc(7 * P->P1, 8);
    int *first = calloc(n + 1, sizeof(int));
    if (g_pgo_sparse) pg_envelope(P, first);
    double radius = 1e4, decrease_factor = 2.0;
    int reuse_diag = 0, invalid = 0, it = 0, term = ISV_TERM_RUNNING;
    double x_cost = pg_evaluate(P, P->pose, 1);
    pg_colnorm2(P, scale);
    for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + sqrt(scale[i]));
    pg_scale_cols(P, scale);
    /* gradient_max_norm = |x - Plus(x, -g)|_inf with the UNSCALED gradient */
    #define GMAX(dst) do { memset(g, 0, 8 * n); pg_left_mul(P, P->res, g); for (int i_ = 0; i_ < n; i_++) neg[i_] = -g[i_] / scale[i_]; \
        pg_plus(P, P->pose, neg, pp); double m_ = 0; for (int k_ = 0; k_ < P->P1; k_++) if (P->col[k_] >= 0) for (int c_ = 0; c_ < 7; c_++) m_ = fmax(m_, fabs(P->pose[7 * k_ + c_] - pp[7 * k_ + c_])); dst = m_; } while (0)
    double gmax; GMAX(gmax);
    double x_norm = sqrt(pg_norm2_diff(P, P->pose, NULL));
    out->initial_cost = x_cost; out->trace_cost[0] = x_cost;
    for (;;) {
        if (it >= max_iter) { term = ISV_TERM_MAX_ITERATIONS; break; }
        if (gmax <= 1e-10) { term = ISV_TERM_GRADIENT_TOL; break; }
        if (radius <= 1e-32) { term = ISV_TERM_MIN_RADIUS; break; }
        it++;
        /* ---- LevenbergMarquardtStrategy::ComputeStep ---- */
        if (!reuse_diag) {
            pg_colnorm2(P, diag);
            for (int i = 0; i < n; i++) diag[i] = fmin(fmax(diag[i], 1e-6), 1e32);
        }
        reuse_diag = 1;
        if (g_pgo_sparse) pg_hessian_env(P, H, first); else pg_hessian(P, H);
        for (int i = 0; i < n; i++) H[(size_t)i * n + i] += diag[i] / radius;          /* D = sqrt(diagonal / radius) */
        memset(g, 0, 8 * n); pg_left_mul(P, P->res, g);
        int ls_fail = (g_pgo_sparse ? chol_skyline(H, n, first) : chol_lower(H, n)) != 0;
        if (!ls_fail) {
            memcpy(step, g, 8 * n);
            if (g_pgo_sparse) chol_skyline_solve(H, n, first, step, 0); else chol_solve(H, n, step);
            for (int i = 0; i < n; i++) { if (!isfinite(step[i])) ls_fail = 1; step[i] = -step[i]; }
        }
        int valid = 0; double model_cost_change = 0;
        if (!ls_fail) {
            memset(mres, 0, 8 * P->nres); pg_right_mul(P, step, mres);
            double mc = 0; for (int i = 0; i < P->nres; i++) mc += mres[i] * (P->res[i] + mres[i] / 2.0);
            model_cost_change = -mc; valid = model_cost_change > 0.0;
        }
        if (!valid) {
            if (++invalid >= 5) { term = ls_fail ? ISV_TERM_LINEAR_SOLVER : ISV_TERM_INVALID_STEPS; break; }
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diag = 1;       /* StepIsInvalid == StepRejected(0) */
            if (it < ISV_MAX_TRACE) { out->trace_cost[it] = x_cost; out->trace_accepted[it] = 0; }
            continue;
        }
        invalid = 0;
        for (int i = 0; i < n; i++) delta[i] = step[i] * scale[i];
        pg_plus(P, P->pose, delta, P->cand);
        const double cand_cost = pg_evaluate(P, P->cand, 0);
        const double step_norm = sqrt(pg_norm2_diff(P, P->pose, P->cand));
        int stop = 0, accepted = 0;
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { term = ISV_TERM_PARAMETER_TOL; stop = 1; }
        else if (fabs(x_cost - cand_cost) <= 1e-6 * x_cost) { term = ISV_TERM_FUNCTION_TOL; stop = 1; }
        if (stop) { if (it < ISV_MAX_TRACE) { out->trace_cost[it] = x_cost; out->trace_accepted[it] = 0; } break; }
        const double rel = (x_cost - cand_cost) / model_cost_change;
        if (rel > 1e-3) {
            accepted = 1;
            memcpy(P->pose, P->cand, sizeof(double) * 7 * P->P1);
            x_norm = sqrt(pg_norm2_diff(P, P->pose, NULL));
            x_cost = pg_evaluate(P, P->pose, 1);
            pg_scale_cols(P, scale);
            GMAX(gmax);
            radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3.0));        /* StepAccepted */
            radius = fmin(1e16, radius); decrease_factor = 2.0; reuse_diag = 0;
            out->num_successful++;
        } else { radius /= decrease_factor; decrease_factor *= 2.0; reuse_diag = 1; }   /* StepRejected */
        if (it < ISV_MAX_TRACE) { out->trace_cost[it] = accepted ? x_cost : cand_cost; out->trace_accepted[it] = accepted; }
    }
    out->iterations = it; out->termination = term; out->final_cost = x_cost; out->status = ISV_OK;
    free(scale); free(diag); free(g); free(step); free(delta); free(H); free(mres); free(neg); free(pp); free(first);
}

/* PoseGraph::optimizeCS, one pass (pose_graph.cpp:246-409) */
int isvo_pgo_optimize(const isv_pgo_config_t *cfg, int32_t n, isv_pg_keyframe_t *kf, int32_t first_looped_index,
                      int32_t cur_index, isv_pgo_result_t *out) {
    memset(out, 0, sizeof(*out));
    pg_problem_t P; memset(&P, 0, sizeof(P));
    P.huber = cfg->huber_delta;
    /* parameter blocks: keyframes first_looped_index .. cur_index in list order (:277-306) */
    int *local = (int *)malloc(sizeof(int) * (n + 1));      /* list position -> local index, -1 if outside */
    P.kf = (const isv_pg_keyframe_t **)calloc(n + 1, sizeof(void *));
    int pi = 0, cur_pos = -1;
    for (int k = 0; k < n; k++) {
        local[k] = -1;
        if (kf[k].index < first_looped_index || cur_pos >= 0) continue;
        local[k] = pi; P.kf[pi] = &kf[k];
        if (kf[k].index == cur_index) { cur_pos = k; }
        pi++;
    }
    if (cur_pos < 0) { free(local); free((void *)P.kf); return ISV_ERR_INVALID_ARG; }
    P.P1 = pi;
    const int param_index = pi - 1;                         /* local index of cur */
    P.col = (int *)malloc(sizeof(int) * P.P1);
    P.pose = (double *)calloc(7 * P.P1, 8); P.cand = (double *)calloc(7 * P.P1, 8);
    for (int k = 0; k < n; k++) {
        const int li = local[k]; if (li < 0) continue;
        quat_t q = q_normalized(q_from_R(kf[k].vio_R_w_i));         /* tmp_q = tmp_r; tmp_q.normalize() */
        double *p = P.pose + 7 * li;
        memcpy(p, kf[k].vio_T_w_i, 24); p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
        const int constant = kf[k].index == first_looped_index || kf[k].sequence == 0;
        P.col[li] = constant ? -1 : 6 * P.nfree;
        if (!constant) P.nfree++;
    }
    P.ncols = 6 * P.nfree;
    /* residual blocks (:309-339), keyframes BEFORE cur only */
    P.edges = (pg_edge_t *)calloc(3 * (size_t)P.P1 + 1, sizeof(pg_edge_t));
    P.loopf = (isv_relpose_t *)calloc(P.P1 + 1, sizeof(isv_relpose_t));
    int ne = 0, roff = 0, nloop = 0;
    for (int k = 0; k < n; k++) {
        const int li = local[k]; if (li < 0 || k == cur_pos) continue;
        if (kf[k].has_rollpitch) { pg_edge_t *E = &P.edges[ne++]; E->kind = 0; E->a = li; E->b = li; E->dim = 2; E->roff = roff; roff += 2; }
        if (li + 1 <= param_index) { pg_edge_t *E = &P.edges[ne++]; E->kind = 1; E->a = li; E->b = li + 1; E->dim = 6; E->roff = roff; roff += 6; }
        if (kf[k].has_loop) {
            int conn = -1;
            for (int m = 0; m < n; m++) if (kf[m].index == kf[k].loop_index) conn = local[m];
            if (conn < 0) { free(local); return ISV_ERR_INVALID_ARG; }        /* assert(loop_index >= first_looped_index) */
            isv_relpose_t *f = &P.loopf[nloop];
            memset(f, 0, sizeof(*f));
            memcpy(f->delta_t, kf[k].loop_info, 24);
            quat_t q = {kf[k].loop_info[3], kf[k].loop_info[4], kf[k].loop_info[5], kf[k].loop_info[6]};
            q_to_R(q, f->delta_R);                                             /* getLoopRelativeQ().toRotationMatrix() */
            for (int d = 0; d < 6; d++) f->sqrt_info[d * 6 + d] = sqrt(kf[k].loop_weight);
            pg_edge_t *E = &P.edges[ne++]; E->kind = 2; E->a = conn; E->b = li; E->kf = nloop; E->robust = 1; E->dim = 6; E->roff = roff; roff += 6;
            nloop++;
        }
    }
    P.nedges = ne; P.nres = roff;
    P.res = (double *)calloc(roff + 1, 8); P.jac = (double *)calloc((size_t)ne * 72 + 1, 8);
    out->n_poses = P.P1; out->n_free = P.nfree; out->n_loop_edges = nloop;
    if (P.ncols > 0) pg_minimize(&P, cfg->max_iterations, out);
    else { out->initial_cost = out->final_cost = pg_evaluate(&P, P.pose, 0); out->termination = ISV_TERM_GRADIENT_TOL; }
    /* ceres::Covariance of every block before cur (:345-350): (J^T J)^-1 at the solution, tangent space, lifted by [I6; 0] */
    double *Sig = (double *)calloc((size_t)P.ncols * P.ncols + 1, 8);
    if (P.ncols > 0) {
        double *H = (double *)calloc((size_t)P.ncols * P.ncols + 1, 8);
        pg_evaluate(&P, P.pose, 1);
        if (g_pgo_sparse) {
            /* the diagonal blocks of H^-1 only: six envelope solves per pose block */
            int *first = (int *)calloc(P.ncols + 1, sizeof(int));
            double *e = (double *)calloc(P.ncols + 1, 8);
            pg_envelope(&P, first); pg_hessian_env(&P, H, first);
            if (chol_skyline(H, P.ncols, first) == 0) {
                for (int c = 0; c < P.ncols; c++) {
                    const int c0 = c - c % 6;
                    memset(e, 0, 8 * (size_t)P.ncols); e[c] = 1.0;
                    chol_skyline_solve(H, P.ncols, first, e, c);
                    for (int r = c0; r < c0 + 6; r++) Sig[(size_t)r * P.ncols + c] = e[r];
                }
            } else out->status = ISV_ERR_NONFINITE;
            free(first); free(e);
        } else {
        pg_hessian(&P, H);
        if (chol_lower(H, P.ncols) == 0) {
            for (int c = 0; c < P.ncols; c++) {              /* column c of the inverse */
                double *e = (double *)calloc(P.ncols, 8); e[c] = 1.0; chol_solve(H, P.ncols, e);
                for (int r = 0; r < P.ncols; r++) Sig[(size_t)r * P.ncols + c] = e[r];
                free(e);
            }
        } else out->status = ISV_ERR_NONFINITE;
        }
        free(H);
    }
    /* write back (:362-385): updatePose, updateCov, update() of the previous keyframe's relative-pose factor */
    isv_pg_keyframe_t *last = NULL; const double *last_pose = NULL;
    for (int k = 0; k < n; k++) {
        const int li = local[k]; if (li < 0) continue;
        const double *p = P.pose + 7 * li;
        memcpy(kf[k].T_w_i, p, 24); q_to_R(q_from_pose(p), kf[k].R_w_i);
        if (li < param_index) {
            double c7[49]; memset(c7, 0, sizeof(c7));
            if (P.col[li] >= 0) for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) c7[a * 7 + b] = Sig[(size_t)(P.col[li] + a) * P.ncols + P.col[li] + b];
            /* the first 36 doubles of the row-major 7x7, read as a column-major 6x6; stored row-major here */
            for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) kf[k].cov[a * 6 + b] = c7[a + 6 * b];
            kf[k].cov_computed = 1;
        }
        if (last) isvo_relpose_update(&last->relative_pose, last->T_w_i, last->R_w_i, kf[k].T_w_i, kf[k].R_w_i, last_pose, p);
        last = &kf[k]; last_pose = p;
    }
    /* drift (:387-393) and the keyframes after cur (:399-407) */
    {
        const isv_pg_keyframe_t *c = &kf[cur_pos];
        double yc[3], yv[3], vT[9], t[3];
        R2ypr(c->R_w_i, yc); R2ypr(c->vio_R_w_i, yv);
        out->yaw_drift = yc[0] - yv[0];
        m3_t(c->vio_R_w_i, vT); mm(c->R_w_i, vT, out->r_drift, 3, 3, 3);
        m3v(out->r_drift, c->vio_T_w_i, t);
        for (int k = 0; k < 3; k++) out->t_drift[k] = c->T_w_i[k] - t[k];
        for (int k = cur_pos + 1; k < n; k++) {
            double Pn[3], Rn[9];
            m3v(out->r_drift, kf[k].vio_T_w_i, Pn); for (int d = 0; d < 3; d++) Pn[d] += out->t_drift[d];
            mm(out->r_drift, kf[k].vio_R_w_i, Rn, 3, 3, 3);
            memcpy(kf[k].T_w_i, Pn, 24); memcpy(kf[k].R_w_i, Rn, 72);
        }
    }
    free(Sig); free(local); free((void *)P.kf); free(P.col); free(P.pose); free(P.cand); free(P.edges); free(P.loopf); free(P.res); free(P.jac);
    return ISV_OK;
}

