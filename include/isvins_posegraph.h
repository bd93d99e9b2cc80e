/*
 * isvins_posegraph.h -- C ABI of the pose-graph optimisation that consumes the CombinedFactors the sliding-window
 * backend emits:
 *   CombinedFactors::operator+            include/factor/pose_graph_factors.h:27-51   (edge composition, host)
 *   PoseGraph::optimizeCS (one pass)      src/pose_graph/pose_graph.cpp:234-428       (MI355X)
 *   loop_pose_output.txt                  src/pose_graph/pose_graph.cpp:412-423       (host)
 * Loop DETECTION (DBoW query, BRIEF matching, PnP-RANSAC: keyframe.cpp, pose_graph.cpp:123-223) stays in the reference:
 * its result arrives here as the keyframe's has_loop / loop_index / loop_info / loop_weight members.
 *
 * What optimizeCS solves (Ceres 2.0.0, external): poses of the keyframes first_looped_index .. cur_index, 7-parameter
 * blocks with PoseLocalParameterization; the first one (and every keyframe of sequence 0) constant; residual blocks, in
 * this order per keyframe BEFORE cur: its RollPitchFactor (no loss), its RelativePoseFactor to the next keyframe (no
 * loss), and, if it closed a loop, a RelativePoseFactor (sqrt_info = sqrt(loop_weight) I6) to the matched keyframe
 * under HuberLoss(0.1).  (cur_index's own factors are NOT added: the edge loop breaks before them, :314.)
 * Solver::Options: SPARSE_NORMAL_CHOLESKY, max_num_iterations = 10, everything else default -> LEVENBERG_MARQUARDT trust
 * region.  Then ceres::Covariance of every pose block before cur.
 * Conventions: as include/isvins_backend.h (row-major matrices, pose = [p, qx qy qz qw]).
 */
#ifndef ISVINS_POSEGRAPH_H
#define ISVINS_POSEGRAPH_H

#include "isvins_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* KeyFrame members optimizeCS reads and writes (include/pose_graph/keyframe.h:84-117) */
typedef struct isv_pg_keyframe {
    double  time_stamp;
    int32_t index;                 /* global keyframe index                                              */
    int32_t sequence;              /* keyframes of sequence 0 are held constant (pose_graph.cpp:299-301) */
    int32_t has_loop, loop_index;  /* loop closure found by the reference's detector, matched keyframe    */
    double  loop_info[8];          /* relative t (3), relative q as w x y z (4), relative yaw (keyframe.h:110) */
    double  loop_weight;           /* keyframe.cpp:224                                                   */
    double  vio_T_w_i[3], vio_R_w_i[9];   /* getVioPose                                                  */
    double  T_w_i[3], R_w_i[9];           /* getPose / updatePose: in/out                                */
    double  cov[36];               /* updateCov: out, row-major image of the Eigen 6x6 the reference stores */
    int32_t cov_computed;          /* out                                                                */
    int32_t has_rollpitch;         /* keyfactor->rollPitchFactor != nullptr                              */
    isv_relpose_t   relative_pose; /* keyfactor->relativePoseFactor: edge to the NEXT keyframe; in/out (update()) */
    isv_rollpitch_t rollpitch;     /* keyfactor->rollPitchFactor                                         */
} isv_pg_keyframe_t;

typedef struct isv_pgo_config {
    int32_t max_keyframes;         /* capacity per graph                                                 */
    int32_t max_graphs;            /* graphs per batch call                                              */
    int32_t max_loop_blocks;       /* capacity: sum over loop edges of (later - earlier) keyframes, per graph */
    int32_t max_iterations;        /* options.max_num_iterations = 10 (pose_graph.cpp:268)               */
    double  huber_delta;           /* HuberLoss(0.1) (pose_graph.cpp:271)                                */
} isv_pgo_config_t;

typedef struct isv_pgo_result {
    int32_t status;                /* isv_status_t                                                       */
    int32_t termination;           /* isv_termination_t                                                  */
    int32_t iterations, num_successful;
    int32_t n_poses, n_free;       /* parameter blocks in the problem, non-constant ones                 */
    int32_t n_loop_edges, _pad;
    double  initial_cost, final_cost;
    double  yaw_drift, r_drift[9], t_drift[3];     /* pose_graph.cpp:387-393                             */
    double  trace_cost[ISV_MAX_TRACE];
    int32_t trace_accepted[ISV_MAX_TRACE];
} isv_pgo_result_t;

typedef struct isv_pgo isv_pgo_t;

/* CombinedFactors::operator+ (include/factor/pose_graph_factors.h:27-51): acc <- acc + other.  Host arithmetic (6x6
 * inverses, SE(3) adjoint); `length` and `vio_index` are the CombinedFactors members of the same name, kept by the caller. */
int  isv_combined_factors_add(isv_combined_factors_t *acc, int32_t *acc_length, int64_t *acc_vio_index,
                              const isv_combined_factors_t *other, int64_t other_vio_index);

int  isv_pgo_create(const isv_pgo_config_t *cfg, isv_pgo_t **out);
void isv_pgo_destroy(isv_pgo_t *h);
const char *isv_pgo_last_error(const isv_pgo_t *h);
/* measurement: duration of the pose-graph kernel(s) of the last optimize call -- a batch of >= 64 graphs goes to the device in
 * four chunks on four streams (ISV_PGO_CHUNKS overrides): from the first chunk's kernel start to the last chunk's kernel end,
 * HIP events on the chunks' streams -- and the number of 6x6 skyline blocks its graphs held */
int  isv_pgo_last_kernel_ms(isv_pgo_t *h, double *ms, double *skyline_blocks);
/* graphs whose STRUCTURE analysis (parameter blocks, adjacency, skyline, column patterns) was reused from the previous call on the same
   batch slot since the handle was created: a graph that is optimised again with the same keyframe list (same indices, sequences,
   roll/pitch and loop flags) only refreshes its numbers.  Results are identical either way. */
int64_t isv_pgo_structure_cache_hits(const isv_pgo_t *h);

/* One pass of PoseGraph::optimizeCS (src/pose_graph/pose_graph.cpp:246-409) over the keyframe list kf[0..n) (list
 * order, indices increasing): solve, covariances, updatePose / updateCov, the relative-pose update() calls, drift, and
 * the drift correction of the keyframes after cur_index.  Keyframes are updated in place.                        */
int  isv_pgo_optimize(isv_pgo_t *h, int32_t n, isv_pg_keyframe_t *kf, int32_t first_looped_index, int32_t cur_index,
                      isv_pgo_result_t *result);
/* the same for n_graphs independent pose graphs (one per sequence) in one launch.  Every graph is written back; when a
 * graph's covariance factorisation fails (results[g].status == ISV_ERR_NONFINITE) its poses are written, its keyframes'
 * cov / cov_computed are left untouched, and the call returns that status (the first failing graph's).            */
int  isv_pgo_optimize_batch(isv_pgo_t *h, int32_t n_graphs, const int32_t *n, isv_pg_keyframe_t *const *kf,
                            const int32_t *first_looped_index, const int32_t *cur_index, isv_pgo_result_t *results);

/* ./loop_pose_output.txt (src/pose_graph/pose_graph.cpp:412-423): `fixed` rows  stamp px py pz qw qx qy qz  of getPose()
 * for every keyframe, file truncated first.                                                                      */
int  isv_pgo_write_loop_pose_output(const char *path, int32_t n, const isv_pg_keyframe_t *kf);

#ifdef __cplusplus
}
#endif
#endif /* ISVINS_POSEGRAPH_H */
