// isv_device_types.h -- device-resident layout of a batch of sliding windows (HBM) shared by the
// kernels and the host side of the C ABI.
//
// All windows of a handle share N (ALL_BUF_SIZE) and Nvo (Vo_SIZE); landmark and factor counts are
// ragged and addressed through per-window offsets (CSR over windows, then CSR over landmarks:
// a landmark's factors are contiguous, in FeatureManager traversal order -- the reference's
// "bit-identical indexing" contract, src/feature_tracker/feature_manager.cpp:27-31,188-204 and
// src/estimator.cpp:1057-1092).
#pragma once
#include <stdint.h>
#include "../../include/isvins_backend.h"

#define ISV_TILE 64            // reprojection factors per wavefront tile
#define ISV_MAX_FRAMES 32
#define ISV_MARG_WS 1408             // doubles per window: Lp @0 (441), Jr @448 (441), V @896 (441), eigenvalues @1344 (21)
#define ISV_SPLIT_MAX_GROUPS 32       // workgroups one window's rank-1 downdates are split over (k_schur_split)
#ifndef ISV_SPLIT_MIN_PASSES
#define ISV_SPLIT_MIN_PASSES 4        // ... when it has at least this many 64-landmark passes (> 192 landmarks; measured on ONE 18-frame
                                      // window of 300 landmarks: 2.66 ms with 8, 2.52 ms with 4; 11 frames: 1.94 either way)
#endif
#define ISV_FUSED_MAX_FACTORS 8192   // longest window (reprojection factors) the one-workgroup-per-window k_lin_gram takes
#define ISV_IMU_IN 64          // packed IMU record (doubles)
// offsets inside the packed IMU record
#define IMU_DP 0
#define IMU_DQ 3               // x y z w
#define IMU_DV 7
#define IMU_LBA 10
#define IMU_LBG 13
#define IMU_DT 16
#define IMU_DP_DBA 17
#define IMU_DP_DBG 26
#define IMU_DQ_DBG 35
#define IMU_DV_DBA 44
#define IMU_DV_DBG 53

// per-window prior strip layout (doubles): [se3: r6 J36][lin9: r9 J81][relpose k: r6 Ji36 Jj36]...[rollpitch m: r2 J12]
#define ISV_SWEEP_WAVES 8
#define PR_SE3 0
#define PR_LIN9 42
#define PR_REL0 132
#define PR_REL_SZ 78
#define PR_RP_SZ 14
// precomputed J^T J (lower-triangle pairs a>=b at a(a+1)/2+b) and J^T r of the non-visual factors
#define ISV_IMU_H 495          // 30*31/2 pairs + 30
#define PH_SE3 0               // 21 + 6
#define PH_LIN9 27             // 45 + 9
#define PH_REL0 81             // 78 + 12 each
#define PH_REL_SZ 90
#define PH_RP_SZ 27            // 21 + 6 each

#define ISV_SEQ_IDLE(d, w) ((d).seq_hdr != nullptr && (d).seq_hdr[(size_t)(w) * 8] < 0)
struct FactorRec { int32_t lm; int32_t ij; };   // global landmark index; frame_i | frame_j << 8

// solver scalars of one window (DoglegStrategy + TrustRegionMinimizer state, Ceres 2.0.0)
struct SolveState {
    double x_cost, initial_cost, radius, mu, alpha, dogleg_step_norm;
    double x_norm, gmax, step_norm, qT;
    int32_t iteration, termination, reuse, invalid, need_linearize, step_valid, num_successful, ls_fail;
    int32_t fresh, _pad;
};

struct DevBatch {
    int32_t B, N, Nvo, Ltot, Ftot, n_tiles, max_rp, np;   // np = 15 N
    int32_t prior_strip_sz, n_prior_slots, max_iter, _pad;
    double proj_sqrt_info[4];
    double G[3];
    double alpha_cut;
    double init_depth;                  // INIT_DEPTH (k_triangulate clamps to it)
    // Eigen-level state, in/out
    double *Ps, *Rs, *Vs, *Bas, *Bgs, *tic, *ric, *depth;
    int32_t *solve_flag;
    // para_* (current point x), candidate point
    double *pose, *sb, *ex, *lam;
    double *cpose, *csb, *clam;
    // structure
    int32_t *lm_off, *f_off;            // [B+1]
    int32_t *lm_host, *lm_k, *lm_f0;    // [Ltot]
    double *lm_pts_i;                   // [Ltot][3]
    FactorRec *f_rec;                   // [Ftot]
    double *f_pts_j;                    // [Ftot][2]
    double *f_pts_z;                        // [Ftot] third component of the observing view's point (triangulation only)
    int32_t *tile_win, *tile_f0, *tile_n;   // [n_tiles]
    int32_t *pg_sched, *pg_sched_off;       // balanced pair -> wavefront schedule of k_sweep_mfma: [B][NP] (h | j << 8 | p << 16), [B][ISV_SWEEP_WAVES + 1]
    int32_t *pg_perm, *pg_off;              // factors of a window sorted by (host, observer) frame pair: [Ftot] window-relative ids, [B][N(N-1)/2 + 1] group starts
    // IMU
    double *imu_in;                     // [B (N-1)][ISV_IMU_IN]
    double *imu_cov;                    // [B (N-1)][225]
    double *imu_sqrt;                   // [B (N-1)][225]
    int32_t *imu_skip;                  // [B (N-1)]  sum_dt > 10 (estimator.cpp:1043)
    // priors (the C ABI structs, verbatim)
    isv_se3_prior_t *se3; isv_linear9_t *lin9; isv_relpose_t *relpose; isv_rollpitch_t *rollpitch;
    int32_t *n_rp;
    // linearisation outputs
    double *strip;                      // [Ftot][28]  Jacobian strips, CSR factor order
    double *fcost;                      // [Ftot]      rho(s)/2 per factor
    double *imu_strip;                  // [B (N-1)][465]
    double *imu_raw;                    // [B (N-1) / 8][144][8] the raw (unweighted) IMU residual / Jacobian entries that are not structurally 0 / +-1, eight factors interleaved (k_imu_raw)
    double *imu_cost;                   // [B (N-1)]
    double *prior_strip;                // [B][prior_strip_sz]
    double *prior_cost;                 // [B][n_prior_slots]
    double *cost;                       // [B] total cost of the last linearisation
    SolveState *st;                     // [B]
    // ---- trust-region solve (isv_solver.hip) ----------------------------------------------
    // tangent vectors, pose/speed-bias part [B][np]; landmark part [Ltot]
    double *scale_p, *diag_p, *grad_p, *gn_p, *delta_p, *zp, *up;
    double *lmE, *lmG, *scale_l, *diag_l, *grad_l, *gn_l, *delta_l, *lm_aterm;
    double *Tglob;                      // [B][N(N+1)/2 * 225] reduced system when it does not fit LDS
    double *fcost_c, *imu_cost_c, *prior_cost_c, *cost_c;      // candidate-point costs
    double *fmodel, *imu_model, *prior_model, *model;          // (J d)^T (r + J d / 2) per block
    double *trace_cost, *trace_radius, *trace_step;            // [B][ISV_MAX_TRACE]
    int32_t *trace_acc;
    isv_marg_result_t *marg;            // [B]
    double *marg_scratch;               // [B][marg_scratch_sz]
    double *marg_ws;                    // [B][ISV_MARG_WS] MargBackward between its three kernels: Lp | Jr | V | eigenvalues
    int32_t *margin_old;                // [B]
    double *header0;                    // [B]
    double *dbg;                        // [B][64] diagnostic stamps (ISV_STAMP builds)
    double *imu_H;                      // [B (N-1)][ISV_IMU_H]
    double *prior_H;                    // [B][prior_H_sz]
    uint32_t *lm_meta;                  // [Ltot] host | k << 8 | (first factor - f_off[w]) << 16
    double *W;                          // [Ftot + Ltot][6]  w = J_pose^T J_lambda per observation (obs index = factor + landmark [+1])
    double *init_scratch, *init_kld;    // initFactorGraph scratch (per window init_per_window doubles) and its KLD output
    size_t init_per_window;
    int32_t *act;                       // [ISV_MAX_TRACE] windows that linearised / solved in iteration i (bench bookkeeping)
    double2 *lm_cg;                     // [Ltot]  {c_l = s_l^2 / (s_l^2 E_l + mu D_l^2), g_l}
    int32_t sw_global;                  // this launch keeps the pair partials in sw_part (set per launch: only batches that need the occupancy)
    double *sw_part;                    // [B][NP * 84] pair partials of k_sweep_mfma when they do not share a CU's LDS four ways (long windows); else null
    const int32_t *seq_hdr;             // device-resident sequences: the frame headers [B][8]; header word 0 < 0 = this window has NO FRAME this step (ISV_SEQ_IDLE):
                                        // nothing is slid, appended, solved or written back for it.  null on the upload path
    double *st_ws;                      // [B][build_solve_st_ws_doubles(N)] = [324 N] (N <= 11) or [324 N + ytot] k_build_solve_st: L_i^-1 | C_i' of the chain nodes between elimination and back-substitution | Y_i' (long windows only: their tile wavefronts read it) | the prescaled init blocks of every node; null: the handle runs k_build_solve_sb
    int32_t split_cap, _pad4;            // groups a window's rank-1 downdates may be split into on this handle (from max_batch at creation; 1: never split)
    double *cs_ws;                      // [B][sb_cs_doubles(N)] hand-over of the split solve (k_build_solve_sb MODE 1 -> MODE 2: isv_build_solve_sb.hip); null: the handle solves in one launch
    double *r1_part;                    // [split_cap_B][ISV_SPLIT_MAX_GROUPS][tiles * 256] raw accumulator tiles of the split rank-1 downdates (k_schur_split -> k_schur_fold); null: no split on this handle
    double *Tvis;                       // [B][tvis_sz] reprojection part of the reduced system (6x6 pose corners), hd, g, bs
    int32_t force_retry, init_mode;       // init_mode: this enqueue is Estimator::initFactorGraph (no update(), no marginalisation)            // test hook (env ISV_DEBUG_FORCE_RETRY): treat the first n factorisations of an iteration as failed
    int32_t prior_H_sz, tvis_sz, wd_ld, max_lm;   // wd_ld: panel width of k_rank1_mfma (6N + 1 rounded up to 16); max_lm: landmarks per window cap
    int32_t marg_scratch_sz, lds_T;     // lds_T: reduced system lives in LDS (15N <= 165)
    // test hooks (env, read at create; the oracle has the same two): ISV_DEBUG_FORCE_INVALID = treat the first n
    // trust-region steps as invalid; ISV_DEBUG_MIN_RADIUS overrides min_trust_region_radius (1e-32)
    int32_t force_invalid, _pad2;
    double min_radius;
    // fused linearise + Gram kernel (isv_visual.hip): factor records in (host, observer) pair order and the per-factor
    // landmark pieces it leaves for k_rank1_mfma's prologue
    int32_t *pg_rec;                    // [Ftot][2] the factor STREAM of k_lin_gram: per window the pair groups in schedule order (sweep wavefront, then pair): {global landmark index, CSR factor id within the window | host << 16 | observer << 24}
    double *pg_pts;                     // [Ftot][2] observing view's point, same order
    int32_t *pg_wstart;                 // [B][ISV_SWEEP_WAVES + 1] stream offsets (within the window) of the sweep wavefronts' slices
    double *flm;                        // [Ftot][8] {J_l^T J_l, J_l^T r, J_i^T J_l (6)} per factor, CSR factor order
    int32_t fused_visual;
    int32_t bs_split, _pad3;             // set per enqueue: k_backsub_split does the landmark back-substitution (k_dogleg skips it)
    int32_t ctl_stage_lm, dg_stage_ph;  // the step control stages its per-landmark gathers in LDS (set per enqueue from lg_lcap); k_dogleg stages the priors' J^T J record (set per enqueue: while the batch still fits one resident round)
    int32_t lg_lcap;                    // landmarks per window k_lin_gram stages in LDS for this upload (longest window, rounded up to 32)
    // ESTIMATE_EXTRINSIC = 1 (src/estimator.cpp:1028-1036): the extrinsic is one more 6-dof block coupled to every
    // reprojection factor.  On the device it rides as a PSEUDO-FRAME: N = Nr + 1 frames, frame Nr's pose block IS
    // para_Ex_Pose (same PoseLocalParameterization), its speed/bias block is a dummy (zero state, unit Hessian diagonal,
    // zero gradient -> zero step) and the IMU factor towards it is flagged skipped, so Plus / candidate / norms / the
    // reduced-system solve / the dogleg need no special case.  Only the reprojection kernels know about it.
    int32_t Nr, est_ex;                 // real frames (ALL_BUF_SIZE); N == Nr + est_ex
    double *Wex;                        // [Ltot][6]  w of the extrinsic block per landmark: sum over its factors of J_ex^T J_l
    double *flmx;                       // [Ftot][6]  J_ex^T J_l per factor (CSR factor order)
    double *ex_part;                    // [B][NP][114] per pair group: J_ex^T J_i (36) | J_ex^T J_j (36) | J_ex^T J_ex (36) | J_ex^T r (6)
    double *strip_ex;                   // [Ftot][12] J_ex of every factor (linearise API only)
};
