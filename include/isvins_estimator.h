/* isvins_estimator.h -- host-side window manager around the MI355X backend.
 *
 * What the reference does between two calls of backendOptimization(), restated natively so that whole sequences run
 * without the reference binary, for S independent sequences in lock step (one batched backend call per frame):
 *
 *   Estimator::processIMU      src/estimator.cpp:91-124     -> isv_estimator_process_imu
 *   IntegrationBase::push_back / propagate / midPointIntegration   include/factor/integration_base.h:31-158
 *   Estimator::processImage    src/estimator.cpp:126-215    -> isv_estimator_push_image + isv_estimator_step
 *   Estimator::solveOdometry   src/estimator.cpp:461-472    (triangulate -> backendOptimization, on the backend)
 *   Estimator::slideWindow     src/estimator.cpp:1565-1724  (MARGIN_OLD / MARGIN_NEW, rotation of the prior factors)
 *   FeatureManager::addFeatureAndCheckParallax / compensatedParallax2 / removeBackShiftDepth / removeBack /
 *   removeFront / removeFailures / goodFeature   src/feature_tracker/feature_manager.cpp:27-31,52-101,262-390
 *   the trajectory row of System::ProcessBackEnd   src/System.cpp:401-410   -> isv_estimator_trajectory
 *
 * NOT here: the visual-inertial initialisation (Estimator::initialStructure, SfM + alignment; out of scope, DESIGN.md
 * section 8).  When a sequence's window first fills, the caller provides the window states
 * (isv_estimator_set_bootstrap) at the point where the reference switches to INITIAL_STRUCTURE
 * (src/estimator.cpp:176-181); from there on everything is the reference's own sequence of steps.
 *
 * All arithmetic of the hot path (triangulate, initFactorGraph, backendOptimization) runs on the GPU through
 * include/isvins_backend.h; isv_estimator_create fails when there is no GPU.  There is no CPU solver in the library.
 */
#ifndef ISVINS_ESTIMATOR_H
#define ISVINS_ESTIMATOR_H

#include "isvins_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct isv_estimator_params {
    isv_config_t cfg;            /* window shape, capacities, solver constants; cfg.max_batch is set to n_sequences.
                                  * cfg.estimate_extrinsic = 1 (ESTIMATE_EXTRINSIC, src/estimator.cpp:1033): the extrinsic is a free
                                  * block of every solve and tic[0] / ric[0] carry over from frame to frame (double2vector :575-583,
                                  * slideWindowOld :1714-1719); = 0: the configured ric / tic below, every frame.  (2, the online
                                  * initial calibration of src/initial/, is outside this path.)                                  */
    double ric[9];               /* RIC[0], row-major  (config yaml extrinsicRotation)                            */
    double tic[3];               /* TIC[0]                                                                       */
    double acc_n, gyr_n;         /* ACC_N, GYR_N  (yaml acc_n / gyr_n)  -> IntegrationBase::noise                */
    double acc_w, gyr_w;         /* ACC_W, GYR_W                                                                 */
    double min_parallax;         /* MIN_PARALLAX = keyframe_parallax / FOCAL_LENGTH  (src/parameters.cpp:82-83)   */
} isv_estimator_params_t;

/* The three backend entry points the window manager calls.  isv_estimator_create binds them to the HIP backend;
 * isv_estimator_create_with_solver exists so that the host logic can be unit-tested on a machine without a GPU by
 * injecting a checker (tests/ inject the CPU oracle).  The library itself never provides an implementation other
 * than the HIP backend. */
typedef struct isv_solver_vtbl {
    void *ctx;
    int (*triangulate)(void *ctx, int32_t n, isv_window_t *const *w);
    int (*init_factor_graph)(void *ctx, isv_window_t *w, isv_summary_t *summary, double *kld);
    int (*optimize_batch)(void *ctx, int32_t n, isv_window_t *const *w, isv_summary_t *summary, isv_marg_result_t *marg);
    /* optional (may be NULL): init_factor_graph for n windows at once */
    int (*init_factor_graph_batch)(void *ctx, int32_t n, isv_window_t *const *w, isv_summary_t *summary, double *kld);
    /* optional (may be NULL): triangulate + optimize_batch in one hand-over; used when no sequence is at its first solve */
    int (*solve_odometry_batch)(void *ctx, int32_t n, isv_window_t *const *w, isv_summary_t *summary, isv_marg_result_t *marg);
} isv_solver_vtbl_t;

typedef struct isv_estimator isv_estimator_t;

/* Estimator::Estimator() + setParameter() for n_sequences independent estimators sharing one backend handle */
int  isv_estimator_create(const isv_estimator_params_t *p, int32_t n_sequences, isv_estimator_t **out);
int  isv_estimator_create_with_solver(const isv_estimator_params_t *p, int32_t n_sequences, const isv_solver_vtbl_t *solver,
                                      isv_estimator_t **out);
void isv_estimator_destroy(isv_estimator_t *e);
const char *isv_estimator_last_error(const isv_estimator_t *e);

/* Threading: a handle is driven by one thread, except that isv_estimator_process_imu(_n), _push_image and
 * _set_bootstrap touch only their own sequence's state and may be called concurrently for DIFFERENT sequences
 * (tools/isv_replay feeds the sequences of a group from several host threads).  Distinct handles are independent. */
/* Estimator::processIMU(dt, linear_acceleration, angular_velocity)  src/estimator.cpp:91-124 */
int  isv_estimator_process_imu(isv_estimator_t *e, int32_t seq, double dt, const double acc[3], const double gyr[3]);

/* n consecutive processIMU calls: dt [n], acc [n][3], gyr [n][3] */
int  isv_estimator_process_imu_n(isv_estimator_t *e, int32_t seq, int32_t n, const double *dt, const double *acc, const double *gyr);

/* The image argument of Estimator::processImage (feature id -> normalised point (x, y, z = 1)); staged until
 * isv_estimator_step.  Ids may come in any order (the reference iterates a std::map, i.e. ascending id). */
int  isv_estimator_push_image(isv_estimator_t *e, int32_t seq, double header, int32_t n, const int32_t *feature_id,
                              const double *point /* [n][3] */);

/* Window states handed over in place of initialStructure(): Ps [N][3], Rs [N][9] row-major, Vs [N][3].  Must be set
 * before the step that processes the sequence's N-th image. */
int  isv_estimator_set_bootstrap(isv_estimator_t *e, int32_t seq, const double *Ps, const double *Rs, const double *Vs);

/* Estimator::processImage on every sequence that has a staged image: addFeatureAndCheckParallax, then for the
 * sequences whose window is full ONE batched triangulate, initFactorGraph where a sequence is at its first solve,
 * ONE batched backendOptimization, slideWindow, removeFailures.  Returns the number of sequences solved (>= 0) or a
 * negative isv_status_t. */
int  isv_estimator_step(isv_estimator_t *e);

/* wall-clock milliseconds of the last isv_estimator_step: [0] whole step, [1] addFeatureAndCheckParallax + window
 * packing, [2] triangulate (backend call), [3] initFactorGraph calls, [4] backendOptimization (backend call),
 * [5] read-back + slideWindow + removeFailures */
int  isv_estimator_last_step_ms(const isv_estimator_t *e, double out[6]);

/* out[0] solver_flag (0 INITIAL, 1 NON_LINEAR), [1] frame_count, [2] marginalization_flag of the last image
 * (1 MARGIN_OLD, 0 MARGIN_SECOND_NEW), [3] tracks in the feature manager, [4] landmarks in the last solve,
 * [5] roll/pitch factors held, [6] solves so far, [7] iterations of the last solve */
int  isv_estimator_status(const isv_estimator_t *e, int32_t seq, int32_t out[8]);
/* the window states; any pointer may be NULL */
int  isv_estimator_get_window(const isv_estimator_t *e, int32_t seq, double *Ps, double *Rs, double *Vs, double *Bas,
                              double *Bgs, double *Headers);
/* tic[0] / ric[0] (row-major): the configured extrinsic, or with cfg.estimate_extrinsic = 1 the one the last solve left
 * (double2vector, src/estimator.cpp:575-583); valid in the resident mode too.  Either pointer may be NULL */
int  isv_estimator_get_extrinsic(const isv_estimator_t *e, int32_t seq, double *tic, double *ric);
/* pre_integrations[frame] of the sequence (IntegrationBase members the IMU factor reads), 1 <= frame <= frame_count */
int  isv_estimator_get_preintegration(const isv_estimator_t *e, int32_t seq, int32_t frame, isv_imu_t *out);
int  isv_estimator_last_summary(const isv_estimator_t *e, int32_t seq, isv_summary_t *out);
/* number of solves of this sequence whose result was not finite (isv_summary_t::status == ISV_ERR_NONFINITE): such a
 * result is NOT copied into the window -- the sequence keeps its pre-solve states and depths, slides, and REBUILDS its
 * prior factors with initFactorGraph at its next solve (without this frame's marginalisation outputs the old priors
 * could not follow the slide).  The reference has no guard here, src/estimator.cpp:1541-1562: a NaN would spread
 * through every later frame. */
int  isv_estimator_failed_solves(const isv_estimator_t *e, int32_t seq);
/* which = 0: the rows the reference appends to pose_output.txt after every solve, 8 doubles per row
 *            (Headers[0], Ps[0], Quaterniond(Rs[0]) as w x y z)            src/System.cpp:401-410
 * which = 1: the newest frame after every solve, 13 doubles per row (header, P, R row-major)
 * Copies at most max_rows rows into out (may be NULL) and returns the number of rows recorded so far. */
int  isv_estimator_trajectory(const isv_estimator_t *e, int32_t seq, int32_t which, double *out, int32_t max_rows);
/* Device-resident windows (include/isvins_backend.h "device-resident sequences"): once every sequence
 * has made its first solves on the host path, the windows -- states, IMU records, prior factors, every track with its
 * points and depth -- stay on the MI355X between frames.  Per frame only the newest frame's propagated state, its feature
 * observations and one (two) IMU record(s) go to the device; the newest / oldest poses, the solve summary and the
 * landmarks' solve_flag come back.  Estimator::slideWindow (src/estimator.cpp:1565-1698) then runs on the device; the host
 * keeps the integer side of the FeatureManager (ids, track lengths) and the IMU pre-integration.  Results are bitwise those
 * of the re-upload path (estimate_extrinsic = 1 included, round 4: the solved tic[0] / ric[0] stay on the device and are the
 * next frame's extrinsic block).  Needs the HIP backend; a sequence without an image in a step idles on the device; a
 * non-finite solve brings every window back to the host (isv_estimator_failed_solves).
 * isv_estimator_get_window is refused while resident; set_resident(e, 0) downloads the windows again (only valid right
 * after isv_estimator_create or ... a frame boundary is handled internally).                                          */
int  isv_estimator_set_resident(isv_estimator_t *e, int32_t on);
int64_t isv_estimator_resident_frames(const isv_estimator_t *e);      /* frames solved through the resident path so far */
/* frames the resident path handed back to the re-upload path so far (a window beyond the resident store's or the per-window
 * kernels' limits, a sequence that left the steady state): each costs a flush, the downloads and a re-seed -- more than a frame
 * of the re-upload path -- so a count that keeps growing says the handle is too tight for resident mode.
 * SEEDING (the first time, and again after every fall-back or failed solve) needs ONE frame in which EVERY sequence of the
 * estimator solves and is in steady state; image streams that never meet in one step stay on the re-upload path.      */
int64_t isv_estimator_resident_fallbacks(const isv_estimator_t *e);

#ifdef __cplusplus
}
#endif
#endif
