/*
 * isvins_backend.h -- C ABI of the MI355X sliding-window VIO backend.
 *
 * Drop-in boundary for ONE path of lyeemax/IS-VINS: the per-frame sliding-window solve
 *   Estimator::solveOdometry()        src/estimator.cpp:461-472
 *     -> Estimator::backendOptimization()   src/estimator.cpp:1541-1562
 *          vector2double()  :474-516, problemSolve() :1004-1146, double2vector() :518-594,
 *          MargForward() :1149-1352, MargBackward() :1354-1539
 * The reference has no FFI layer: the seam is that C++ member call.  Every
 * struct below mirrors the Estimator members that call reads and writes
 * (include/estimator.h:90-154); every entry point names the reference function it replaces.
 *
 * Conventions (identical to the reference):
 *   - all reals are IEEE float64, all indices int32
 *   - matrices are ROW-MAJOR here (Eigen's default is column-major: the C++ shim in
 *     include/isvins_estimator_shim.hpp converts)
 *   - pose block  = [px py pz qx qy qz qw]              (src/estimator.cpp:476-485)
 *   - speed-bias  = [vx vy vz bax bay baz bgx bgy bgz]  (src/estimator.cpp:487-497)
 *   - IMU residual/tangent order O_P=0 O_R=3 O_V=6 O_BA=9 O_BG=12 (include/parameters.h:89-96)
 *   - landmark parameter = inverse depth in its host (start) frame
 * No allocation crosses this ABI: the caller owns every buffer it passes; the handle owns
 * all device memory.  Functions return ISV_OK (0) or a negative isv_status_t; they never abort
 * (the reference asserts instead, src/estimator.cpp:773,1218,1386).
 */
#ifndef ISVINS_BACKEND_H
#define ISVINS_BACKEND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): isv_batch_pack_results' `stream` is ALWAYS the caller's hipStream_t since round 3 (NULL = the legacy default stream;
   ISV_STREAM_OF_HANDLE = the handle's own) -- version 1 callers passed NULL for the handle's stream --, the isv_backend_seq_* entry
   points and structs exist, and the launch variants whose sums differ in the last bits are chosen per HANDLE (max_batch). */
#define ISV_ABI_VERSION 2

typedef enum isv_status {
    ISV_OK = 0,
    ISV_ERR_INVALID_ARG = -1,   /* null pointer, bad size, index out of window          */
    ISV_ERR_CAPACITY = -2,      /* more landmarks / observations / windows than created  */
    ISV_ERR_NONFINITE = -3,     /* NaN/Inf in inputs or produced by the solve            */
    ISV_ERR_DEVICE = -4,        /* HIP runtime error (no GPU, OOM, launch failure)       */
    ISV_ERR_UNSUPPORTED = -5    /* e.g. estimate_extrinsic = 2, or = 1 with n_frames > 19 */
} isv_status_t;

/* ceres::TerminationType + the reason strings of TrustRegionMinimizer (Ceres 2.0.0) */
typedef enum isv_termination {
    ISV_TERM_RUNNING = 0,
    ISV_TERM_GRADIENT_TOL = 1,     /* CONVERGENCE: max_norm(gradient) <= 1e-10            */
    ISV_TERM_PARAMETER_TOL = 2,    /* CONVERGENCE: |step| <= 1e-8 (|x| + 1e-8)            */
    ISV_TERM_FUNCTION_TOL = 3,     /* CONVERGENCE: |dcost| <= 1e-6 cost                   */
    ISV_TERM_MAX_ITERATIONS = 4,   /* NO_CONVERGENCE: NUM_ITERATIONS reached              */
    ISV_TERM_MIN_RADIUS = 5,       /* CONVERGENCE: trust region radius <= 1e-32           */
    ISV_TERM_INVALID_STEPS = 6,    /* FAILURE: 5 consecutive invalid steps                */
    ISV_TERM_LINEAR_SOLVER = 7     /* FAILURE: Cholesky failed up to mu = 1               */
} isv_termination_t;

/* ---- configuration: include/parameters.h:35-40 + config/euroc_config.yaml ------------- */
typedef struct isv_config {
    int32_t n_frames;            /* ALL_BUF_SIZE (18 in the reference; 11 / 20 synthetic)      */
    int32_t n_vo;                /* Vo_SIZE (8 in the reference), 2 <= n_vo <= n_frames-1      */
    int32_t max_landmarks;       /* NUM_OF_F capacity per window                              */
    int32_t max_obs;             /* capacity: sum over landmarks of track length, per window  */
    int32_t max_rollpitch;       /* capacity of vioRollPitchEdges (<= n_vo + 1)               */
    int32_t max_batch;           /* windows the handle can hold at once                       */
    int32_t num_iterations;      /* NUM_ITERATIONS (yaml:50) -> max_num_iterations            */
    int32_t estimate_extrinsic;  /* ESTIMATE_EXTRINSIC: 0 = block constant, 1 = free (J_ex,
                                    src/estimator.cpp:1028-1036); 2 (initial calibration) is refused */
    double  proj_sqrt_info[4];   /* ProjectionFactor::sqrt_info = PIXEL_SQRT_INFO * I2        */
    double  gravity[3];          /* G (src/parameters.cpp), enters the IMU residual with '+'  */
    double  alpha;               /* ALPHA eigenvalue cut of the sparsification (yaml:86)      */
    double  init_depth;          /* INIT_DEPTH                                                */
} isv_config_t;

/* ---- IntegrationBase members read by IMUFactor (include/factor/integration_base.h:188-207) */
typedef struct isv_imu {
    double delta_p[3];
    double delta_q[4];           /* x y z w */
    double delta_v[3];
    double linearized_ba[3];
    double linearized_bg[3];
    double sum_dt;
    double jacobian[225];        /* 15x15 row-major */
    double covariance[225];      /* 15x15 row-major */
} isv_imu_t;

/* ---- prior factors: include/estimator.h:134-154 --------------------------------------- */
typedef struct isv_se3_prior {   /* SE3PriorFactor  include/factor/se3_prior_factor.h:135-140 */
    double t[3];
    double R[9];
    double sqrt_info[36];
    int32_t index;
    int32_t _pad;
} isv_se3_prior_t;

typedef struct isv_linear9 {     /* Linear9Factor   include/factor/linear9_factor.h:69-73      */
    double VB[9];
    double sqrt_info[81];
    int32_t index;
    int32_t _pad;
} isv_linear9_t;

typedef struct isv_relpose {     /* RelativePoseFactor include/factor/relative_pose_factor.h:190-195 */
    double delta_t[3];
    double delta_R[9];
    double sqrt_info[36];
    int32_t imu_i, imu_j;
} isv_relpose_t;

typedef struct isv_rollpitch {   /* RollPitchFactor include/factor/rollpitch_factor.h:132-136  */
    double R[9];
    double sqrt_info[4];
    int32_t index;
    int32_t _pad;
} isv_rollpitch_t;

/* ---- one sliding window = the Estimator members backendOptimization() touches --------- */
typedef struct isv_window {
    /* Eigen-level state, in/out (include/estimator.h:90-94,84-85) */
    double *Ps;                  /* [N][3]                                               */
    double *Rs;                  /* [N][9]  rotation matrices, row-major                 */
    double *Vs;                  /* [N][3]                                               */
    double *Bas;                 /* [N][3]                                               */
    double *Bgs;                 /* [N][3]                                               */
    double *tic;                 /* [3]                                                  */
    double *ric;                 /* [9]                                                  */
    /* FeatureManager view in IDsfeatures order restricted to goodFeature() landmarks
     * (feature_manager.cpp:27-31,188-204): landmark l has host frame lm_start_frame[l]
     * and observations obs_point[lm_obs_ptr[l] .. lm_obs_ptr[l+1]) in consecutive frames */
    int32_t n_landmarks;         /* L = getFeatureCount()                                */
    int32_t n_obs;               /* lm_obs_ptr[L] = F + L                                */
    const int32_t *lm_start_frame; /* [L]                                                */
    const int32_t *lm_obs_ptr;   /* [L+1]                                                */
    const double *obs_point;     /* [n_obs][3] normalised image points (x, y, 1)         */
    double *lm_depth;            /* [L] estimated_depth, in/out (getDepthVector/setDepth) */
    int32_t *lm_solve_flag;      /* [L] out: 1 ok, 2 depth<0 or >10 (feature_manager.cpp:156-161) */
    /* pre_integrations[1..N-1]: imu[i] links frame i -> i+1 */
    const isv_imu_t *imu;        /* [N-1]                                                */
    /* prior factors, in/out (update() shifts them, double2vector rotates two of them)   */
    isv_se3_prior_t *pose_prior; /* vioPosePriorEdge                                     */
    isv_linear9_t   *vb_prior;   /* vioVBPrior                                           */
    isv_relpose_t   *relpose;    /* [n_vo-1] = vioRelativePoseEdges[1..n_vo-1]           */
    isv_rollpitch_t *rollpitch;  /* [n_rollpitch] vioRollPitchEdges                      */
    int32_t n_rollpitch;
    int32_t margin_old;          /* marginalization_flag == MARGIN_OLD                   */
    double  header0;             /* Headers[0] (copied into CombinedFactors.ts)          */
    /* para_* arrays, out (include/estimator.h:122-126); may be NULL                      */
    double *para_Pose;           /* [N][7]                                               */
    double *para_SpeedBias;      /* [N][9]                                               */
    double *para_Ex_Pose;        /* [7]                                                  */
    double *para_Feature;        /* [L]                                                  */
} isv_window_t;

#define ISV_MAX_TRACE 64
typedef struct isv_summary {     /* ceres::Solver::Summary subset + per-iteration trace  */
    int32_t status;              /* isv_status_t of this window                          */
    int32_t termination;         /* isv_termination_t                                    */
    int32_t iterations;          /* trust-region iterations performed (<= num_iterations) */
    int32_t num_successful;
    double  initial_cost;
    double  final_cost;
    double  trace_cost[ISV_MAX_TRACE];    /* cost after iteration k (k=0: initial)        */
    double  trace_radius[ISV_MAX_TRACE];  /* trust-region radius after iteration k        */
    double  trace_step_norm[ISV_MAX_TRACE];
    int32_t trace_accepted[ISV_MAX_TRACE];
} isv_summary_t;

/* ---- outputs of MargForward / MargBackward ------------------------------------------- */
typedef struct isv_combined_factors {  /* CombinedFactors include/factor/pose_graph_factors.h:6-17 */
    isv_relpose_t relative_pose;       /* pgRaltivePoseFactor (src/estimator.cpp:1243-1255) */
    int32_t has_rollpitch;             /* vioRollPitchEdges[0]->index == 0                */
    int32_t _pad;
    isv_rollpitch_t rollpitch;
    double covRel[36];
    double covAbs[4];
    double distance;
    double ts;
    double Ri[9];
    double ti[3];
} isv_combined_factors_t;

typedef struct isv_marg_result {
    int32_t valid;                     /* 1 when margin_old was set and marg ran          */
    int32_t n_marg_landmarks;          /* MargPointIdx.size()                             */
    isv_combined_factors_t combined;   /* -> pose_graph_factors_buf                       */
    isv_se3_prior_t forward_pose_prior;      /* forwardPosePriorEdgeToAdd                 */
    isv_relpose_t   backward_relpose;        /* backwardRelativePoseEdgeToAdd             */
    isv_linear9_t   backward_vb;             /* backwardVBEdgeToAdd                       */
    isv_rollpitch_t backward_rollpitch;      /* pushed to vioRollPitchEdges, index n_vo-1 */
    double forward_kld;                /* the reference's "zero test" (estimator.cpp:1337-1343) */
    double backward_kld;               /* (estimator.cpp:1528-1533)                       */
} isv_marg_result_t;

/* ---- linearisation output (config "residual+Jacobian kernels only") ------------------ */
/* Jacobian strip of one reprojection factor, in CSR (landmark-major) factor order:
 *   [ r(2) | J_pose_i 2x6 row-major | J_pose_j 2x6 row-major | J_lambda 2x1 ] = 28 doubles,
 * already Cauchy-corrected (ceres Corrector) and with the local-parameterisation
 * Jacobian applied (first 6 of the 7 pose columns).                                      */
#define ISV_PROJ_STRIP 28
/* IMU factor strip: [ r(15) | 15x6 pose_i | 15x9 sb_i | 15x6 pose_j | 15x9 sb_j ] row-major
 * blocks = 15 + 450 = 465 doubles                                                        */
#define ISV_IMU_STRIP 465

typedef struct isv_backend isv_backend_t;

/* lifecycle ------------------------------------------------------------------------- */
int  isv_abi_version(void);
/* Estimator::Estimator()/setParameter(): src/estimator.cpp:16-38 */
int  isv_backend_create(const isv_config_t *cfg, isv_backend_t **out);
void isv_backend_destroy(isv_backend_t *h);
const char *isv_backend_last_error(const isv_backend_t *h);

/* Estimator::backendOptimization() NON_LINEAR branch (src/estimator.cpp:1549-1560):
 * vector2double, problemSolve, double2vector, and when w->margin_old MargForward+MargBackward.
 * marg may be NULL when margin_old == 0.                                                */
int  isv_backend_optimize(isv_backend_t *h, isv_window_t *w, isv_summary_t *summary,
                          isv_marg_result_t *marg);
/* the same over n independent windows (multi-sequence throughput entry) */
int  isv_backend_optimize_batch(isv_backend_t *h, int32_t n, isv_window_t *const *w,
                                isv_summary_t *summary, isv_marg_result_t *marg);

/* Stages, for tests and for the "kernels only" configuration --------------------------- */
/* One evaluation of every residual block at the window's current state, as
 * ceres::Problem::Evaluate would do for the problem problemSolve() builds
 * (src/estimator.cpp:1022-1117): ProjectionFactor::Evaluate (projection_factor.cpp:24-122),
 * IMUFactor::Evaluate (imu_factor.h:23-159), the four prior factors, CauchyLoss(1.0) corrector.
 * proj_strips [F][28], imu_strips [N-1][465], cost (1/2 sum rho) may each be NULL.       */
/* Estimator::initFactorGraph (src/estimator.cpp:667-1001), the one-time INITIAL_STRUCTURE -> NON_LINEAR step of
 * backendOptimization(): solve the window WITHOUT prior factors (IMU + reprojection factors, 3 x num_iterations dogleg
 * iterations; the reference's 1 s wall-clock cap is not restated), derive the first prior factors from the solved
 * estimate (relative poses (i, i+1) for i < n_vo - 1, SE3 prior on pose 0, Linear9 prior on speed/bias n_vo - 1:
 * marginal of the first n_vo - 1 IMU factors, eigen-truncated at alpha), then double2vector.
 * w->pose_prior, w->vb_prior, w->relpose[n_vo - 1] are OUTPUTS; w->n_rollpitch becomes 0.  *kld (may be NULL) receives
 * the Kullback-Leibler divergence of the recovered factors against the truncated marginal (:976-989).          */
int  isv_backend_init_factor_graph(isv_backend_t *h, isv_window_t *w, isv_summary_t *summary, double *kld);
/* the same for n windows at once (n sequences reaching their first solve together); summary [n], kld [n] or NULL */
int  isv_backend_init_factor_graph_batch(isv_backend_t *h, int32_t n, isv_window_t *const *w, isv_summary_t *summary, double *kld);
/* FeatureManager::triangulate (src/feature_tracker/feature_manager.cpp:206-258), the step solveOdometry() runs right
 * before backendOptimization(): every landmark of the n windows whose lm_depth is not positive gets the DLT depth over
 * all its views (smallest right singular vector, host-camera frame), replaced by INIT_DEPTH outside [0.1, 8].
 * lm_depth is updated in place; landmarks that already have a depth are left alone.           */
int  isv_backend_triangulate(isv_backend_t *h, int32_t n, isv_window_t *const *w);
/* Estimator::solveOdometry (src/estimator.cpp:461-472) = triangulate then backendOptimization, for n windows with one
 * upload and one download (the window manager's per-frame call in steady state).                */
int  isv_backend_solve_odometry_batch(isv_backend_t *h, int32_t n, isv_window_t *const *w,
                                      isv_summary_t *summary, isv_marg_result_t *marg);
int  isv_backend_linearize(isv_backend_t *h, const isv_window_t *w,
                           double *proj_strips, double *imu_strips, double *cost);

/* Device-resident batch: upload once, run many times (what bench.py times) -------------- */
int  isv_batch_upload(isv_backend_t *h, int32_t n, isv_window_t *const *w);
/* restore the uploaded initial state, then run backendOptimization on every resident window.
 * sync bit 0: wait for completion (otherwise asynchronous on the handle's stream);
 * sync bit 1: record HIP events around the dominant kernels of every iteration (isv_batch_last_timing);
 *             a profiling pass, a few percent slower than a plain one.                      */
int  isv_batch_optimize(isv_backend_t *h, int32_t sync);
/* only the factor-linearisation kernels over the resident batch */
int  isv_batch_linearize(isv_backend_t *h, int32_t sync);
int  isv_batch_download(isv_backend_t *h, int32_t n, isv_window_t *const *w,
                        isv_summary_t *summary, isv_marg_result_t *marg);
int  isv_batch_sync(isv_backend_t *h);
/* HIP-event timing of the last isv_batch_* launch sequence, milliseconds, per kernel family:
 * after isv_batch_linearize: out[0]=total, [1]=k_proj_linearize, [2]=imu+prior+reduce;
 * after isv_batch_optimize:  out[0]=total; after a profiling pass (sync bit 1) also the sums over the
 * iterations of [1]=k_proj_linearize<0>, [2]=k_sweep_mfma, [3]=k_rank1_mfma, [4]=k_build_solve*, [5]=k_dogleg,
 * [6]=k_step_control.
 * Events are recorded on the handle's own stream.                                          */
int  isv_batch_last_timing(isv_backend_t *h, double out_ms[8]);
/* Multi-GPU exchange step (the reference has no counterpart: it is one process): per-window RESULT RECORDS
 * of the resident batch, written into a caller-owned DEVICE buffer [n][isv_result_record_doubles()] so that the caller
 * can all-gather them over RCCL without a host copy.  record = [para_Pose 7N | para_SpeedBias 9N | inverse depths zero
 * padded to max_landmarks | final_cost initial_cost iterations termination num_successful radius header0 n_landmarks].
 * `stream` is the CALLER's hipStream_t -- NULL is a stream too (the legacy default stream, which is what
 * torch.cuda.current_stream().cuda_stream is for torch's default stream): the pack kernel waits (event, no host sync)
 * for everything enqueued on the handle's own non-blocking stream, runs on `stream`, and the handle's next launch waits
 * for it in turn, so a collective enqueued on `stream` afterwards reads finished records and the next solve does not
 * overwrite the states under the pack.  ISV_STREAM_OF_HANDLE asks for the handle's own stream instead (then the caller
 * orders its reads with isv_batch_sync).                                                                              */
#define ISV_STREAM_OF_HANDLE ((void *)(intptr_t)-1)
int64_t isv_result_record_doubles(const isv_backend_t *h);
int  isv_batch_pack_results(isv_backend_t *h, void *device_dst, void *stream);
/* ---- device-resident sequences ("keeps device-resident window state across frames so only new
 * observations / IMU deltas cross PCIe each frame") ----------------------------------------------------------------
 * Slot b of the handle (0 <= b < max_batch) holds ONE sequence's whole window on the device between frames: the states,
 * the IMU records and their sqrt_info, the prior factors, and the FeatureManager's tracks (every IDFeatures, good or not:
 * start_frame, observation ring, estimated_depth, solve_flag) in list order.  Per frame the caller hands over what is NEW --
 * the newest frame's propagated state, its feature observations, one IMU record (two after a MARGIN_SECOND_NEW slide) --
 * and the device does the rest of Estimator::slideWindow (src/estimator.cpp:1565-1698: state / IMU shift, the rotation of
 * the prior factors with the marginalisation outputs that never left the device, slideWindowOld ->
 * FeatureManager::removeBackShiftDepth feature_manager.cpp:275-313 with the depth re-hosting, slideWindowNew -> removeFront
 * :335-354, removeFailures :165-174), rebuilds the solver's view of the window (goodFeature() landmarks in list order, the
 * (host, observer) pair groups, their schedule and the factor stream: what isv_batch_upload builds on the host), and runs
 * solveOdometry (triangulate + backendOptimization).  The caller keeps the INTEGER side of the FeatureManager (ids,
 * start_frame, track lengths: which feature continues which track) and therefore knows every offset; it never needs the
 * points, depths or window states back -- only the newest frame's state (for processIMU), the oldest pose (pose_output.txt),
 * the solve summary and the landmarks' solve_flag (removeFailures).  Results are bitwise those of the re-upload path
 * (tests/test_gpu_resident.py), estimate_extrinsic = 1 included (round 4: the solved tic[0] / ric[0] stay on the device, k_seq_slide
 * makes them the next solve's extrinsic block, every result record carries them); windows the per-window kernels do not take (> 8192 factors) are refused with ISV_ERR_UNSUPPORTED
 * BEFORE anything is launched, and the caller (isv_estimator_step does) solves that frame through the upload path.        */
typedef struct isv_seq_track {       /* one IDFeatures of the seed: feature_manager.h:36-63 */
    int32_t start_frame, n_obs, solve_flag, slot;      /* slot: the caller's storage slot of the track (< tracks capacity), kept for its lifetime */
    double  depth;
} isv_seq_track_t;
typedef struct isv_seq_obs {         /* one observation of the newest frame */
    int32_t track;                   /* ordinal of its track in the list AFTER the slide (== n_tracks + k for the k-th new track) */
    int32_t slot;                    /* storage slot (only read for a new track) */
    double  point[3];
} isv_seq_obs_t;
typedef struct isv_seq_frame {       /* the hand-over of one frame of one sequence */
    int32_t prev_slide;              /* what Estimator::slideWindow did after the previous solve: 0 = nothing to apply (first frame after the seed), 1 = MARGIN_OLD, 2 = MARGIN_SECOND_NEW;
                                        -1 (round 4) = this sequence has NO image this step (System::getMeasurements pairs IMU and images per sequence,
                                        src/System.cpp:160-202): only n_tracks is read, nothing is slid, appended, solved or written back for it -- a pending
                                        slide stays pending -- and its isv_seq_result_t comes back zeroed */
    int32_t margin_old;              /* marginalization_flag of THIS solve */
    int32_t n_tracks;                /* tracks alive after the slide, before this frame's features are added (consistency check) */
    int32_t n_obs;                   /* observations of the newest frame */
    const isv_seq_obs_t *obs;        /* [n_obs] in std::map order of the reference's image (feature id ascending) */
    int32_t n_imu;                   /* 1: imu[0] = pre_integrations[N-1]; 2 (after MARGIN_SECOND_NEW): imu[0] = the merged pre_integrations[N-2], imu[1] = pre_integrations[N-1] */
    int32_t want_marg;               /* copy the isv_marg_result_t of this solve back (the CombinedFactors for the pose graph) */
    const isv_imu_t *imu;
    double Ps[3], Rs[9], Vs[3], Bas[3], Bgs[3];   /* newest frame (N-1) as processIMU propagated it */
    double header0;                  /* Headers[0] */
    /* the solver's view, which the caller knows from its integer bookkeeping (capacity checks, launch sizes) */
    int32_t n_landmarks, n_factors;  /* goodFeature() landmarks and their reprojection factors after this frame's features were added */
} isv_seq_frame_t;
typedef struct isv_seq_result {
    isv_summary_t summary;
    double Ps_new[3], Rs_new[9], Vs_new[3], Bas_new[3], Bgs_new[3];   /* frame N-1 after the solve */
    double Ps_old[3], Rs_old[9];                                       /* frame 0 after the solve (pose_output.txt row) */
    double Ps_second[3], Rs_second[9];                                 /* frame 1 (becomes frame 0 after a MARGIN_OLD slide) */
    int32_t marg_valid, n_failed_landmarks;
    double tic[3], ric[9];                                             /* the extrinsic after the solve (estimate_extrinsic = 1: tic[0] / ric[0] of double2vector) */
} isv_seq_result_t;
/* allocate the track store: tracks_per_window >= every track alive in a window (good or not) */
int  isv_backend_seq_enable(isv_backend_t *h, int32_t tracks_per_window);
/* make slots 0 .. n-1 resident from the caller's full state AFTER a slide: windows as for isv_batch_upload (imu[N-2], the
 * record of the frame still to come, is ignored), tracks[b][n_tracks[b]] in list order with their points
 * points[b][sum n_obs][3] (track-major, oldest observation first) */
int  isv_backend_seq_seed(isv_backend_t *h, int32_t n, isv_window_t *const *w, const int32_t *n_tracks,
                          const isv_seq_track_t *const *tracks, const double *const *points);
/* one frame of the n resident sequences: slide, append, solveOdometry.  solve_flags[b] (may be NULL) receives the
 * lm_solve_flag of the n_landmarks goodFeature() landmarks in list order; marg [n] is filled where want_marg is set */
int  isv_backend_seq_frame(isv_backend_t *h, int32_t n, const isv_seq_frame_t *frames, isv_seq_result_t *results,
                           int32_t *const *solve_flags, isv_marg_result_t *marg);
/* the resident state of slot b back into caller buffers (a sequence leaving the resident mode, tests): the window as
 * isv_batch_download fills it restricted to states and prior factors (w->tic / w->ric too when they are non-NULL: the extrinsic as
 * the last solve left it), and every track's depth / solve_flag in list order */
int  isv_backend_seq_download(isv_backend_t *h, int32_t slot, isv_window_t *w, int32_t n_tracks, double *track_depth, int32_t *track_flag);
/* the slide of the previous solve is applied by the NEXT isv_backend_seq_frame; a caller that has slid its own side and
 * wants the resident state back first lets the device catch up: prev_slide [n] as in isv_seq_frame_t, n_tracks [n] the
 * caller's track counts after the slide (consistency check) */
int  isv_backend_seq_flush(isv_backend_t *h, int32_t n, const int32_t *prev_slide, const int32_t *n_tracks);
/* the marginalisation outputs of slot b's last solve (they stay on the device unless want_marg was set) */
int  isv_backend_seq_marg(isv_backend_t *h, int32_t slot, isv_marg_result_t *out);

/* last optimize: [0] k_lin_gram (or k_proj_linearize<0>) launches, [1] k_build_solve* launches, [2] k_rank1_mfma launches,
 * [3] window-iterations that were linearised and solved (windows gated out of an iteration do no work),
 * [4] 1 when the fused k_lin_gram ran (no Jacobian strips), [5] 1 when k_dogleg<true> carried the step control */
int  isv_batch_last_counts(isv_backend_t *h, int64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* ISVINS_BACKEND_H */
