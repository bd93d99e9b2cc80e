// isv_kernels.h -- kernel and solver-stage declarations shared by the host translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "isv_device_types.h"

// per-wave LDS of k_proj_linearize<MODE>: R,P per frame + extrinsic; MODE 0 adds a 64 x 15 transpose buffer, MODE 1 the poses at x and the tangent step
__host__ __device__ inline size_t proj_lds_doubles_per_wave(int N, int mode) { return (size_t)N * 12 + 12 + (mode == 0 ? 64 * 15 : (size_t)N * 18); }

__global__ void k_vector2double(DevBatch d);
__global__ void k_imu_prep(DevBatch d, const int32_t *sel);     // sel: factor index per workgroup (null: identity; < 0: skip)
template <int MODE> __global__ void k_proj_linearize(DevBatch d, const double *pose_src, const double *lam_src, double *fcost_out, int gate);
template <bool JAC> __global__ void k_imu_linearize(DevBatch d, const double *pose_src, const double *sb_src, double *cost_out, int gate);
template <bool JAC> __global__ void k_prior_linearize(DevBatch d, const double *pose_src, const double *sb_src, double *cost_out, int gate);
__global__ void k_cost_reduce(DevBatch d, const double *fcost, const double *imu_cost, const double *prior_cost, double *out, int gate);

// solver stage (isv_solver.hip)
// host-side launch parameters of a handle: device figures and environment hooks, read ONCE in isv_solver_alloc (the
// kernel VARIANT of a launch is then chosen from these and the uploaded batch: see isv_solver_enqueue)
struct SolverHost {
    int n_cus = 0;                    // compute units of the handle's device
    size_t cap_batch = 0;             // the handle's max_batch: kernel VARIANTS whose results differ in the last bits (k_build_solve_st, the split
                                      // elimination, the one-launch MargBackward) are chosen from it, never from the uploaded batch size, so a
                                      // window gives the same bits alone and inside any batch of the same handle
    int dogleg_per_cu_regs = 0;       // workgroups of k_dogleg<true, EX> per CU by registers (the LDS bound is applied per enqueue)
    bool one_stream = false;          // ISV_ONE_STREAM: diagnostics, everything on one stream
    bool split_control = false;       // ISV_SPLIT_CONTROL: k_dogleg<false> + k_step_control
    bool generic_n = false;           // ISV_GENERIC_N: run-time-N instantiation of k_build_solve_sb for every N
    bool lg_batch_waves = false;      // ISV_LG_BATCH_WAVES: never the eight-wavefront k_lin_gram
    bool debug_sw_global = false;     // ISV_DEBUG_SW_GLOBAL: pair partials in the global scratch for every launch
    bool legacy_visual = false;       // ISV_LEGACY_VISUAL: the unfused k_proj_linearize<0> + k_sweep_mfma pair
    bool no_persistent = false;       // ISV_NO_PERSISTENT: never the one-launch solve of small batches
    bool sw_global_ok = false;        // the handle's windows are long enough (or ISV_DEBUG_SW_GLOBAL) for the pair partials in global memory
    bool solve_st = false;            // this handle runs k_build_solve_st (four windows per CU; chosen from the handle's max_batch, ISV_SOLVE_ST=0/1 forces)
    bool chain_split = false;         // this handle splits the reduced-system solve into the chain kernel (side stream) and the pose kernel (k_build_solve_sb MODE 1 / 2):
                                      // handles whose batches leave CUs idle (max_batch <= CUs); ISV_CHAIN_SPLIT=0/1 forces
    bool no_pose_dogleg = false;      // ISV_NO_POSE_DOGLEG (A/B and test hook): the pose half, the dogleg and the step control as separate launches (same bits)
    bool no_split = false;            // ISV_NO_SPLIT: never spread one window's landmark elimination over several workgroups (k_schur_split)
    bool marg_one_kernel = false;     // ISV_MARG_ONE_KERNEL / ISV_MARG_SPLIT force MargBackward as one launch (k_marg_bwd<2>) or as build / k_marg_jacobi /
    bool marg_split = false;          // project, whatever the batch size (default: split up to n_cus windows); the two are bitwise equal (tested)
    bool no_update = false;           // ISV_DEBUG_NO_UPDATE (sensitivity study, tests/test_sequence_long.py; the oracle has the same
                                      // hook): skip the update() of the prior factors' pseudo-measurements after the solve
                                      // (src/estimator.cpp:1133-1144).  NOT the reference's behaviour.
};
int isv_solver_alloc(DevBatch &d, SolverHost &hc, size_t B, size_t L, size_t F, std::vector<void *> &allocs, std::string &err);
int isv_solver_enqueue(DevBatch &d, const SolverHost &hc, hipStream_t st, hipStream_t st2, hipEvent_t *fj, int64_t *counts, hipEvent_t *prof_ev, std::string &err);
// (jac = false: the residual-only evaluation, see prior_linearize_body)
// + the window's prior factor records staged in LDS (round 3): SE3 49 + Linear9 91 + 49 per relative-pose / 14 per roll-pitch words
__host__ __device__ static inline size_t prior_lds_bytes(int slots, bool jac = true) { return ((size_t)slots * (jac ? 82 + 90 : 10 + 10) + 140 + (size_t)(slots > 2 ? slots - 2 : 0) * 49) * sizeof(double); }
__global__ void k_triangulate(DevBatch d);
// wavefronts of k_lin_gram per window: LG_WAVES for batches (three workgroups per CU), LG_WAVES_SMALL while the batch leaves
// every window a CU of its own (each wavefront takes the pair groups of ISV_SWEEP_WAVES / LGW sweep wavefronts)
#define LG_WAVES 4
#define LG_WAVES_SMALL 8
template <bool EX, int LGW> __global__ void k_lin_gram(DevBatch d);     // EX: the extrinsic is estimated (J_ex, one more block row)
size_t lin_gram_lds_bytes(int N, bool partials_in_lds, bool ex, int waves, int lcap);
#define ISV_LDS_PER_CU ((size_t)160 * 1024)
template <int NT, int TPW> __global__ void k_schur_split(DevBatch d, int Gs, int GrMax);
template <int NT, int TPW> __global__ void k_rank1_split(DevBatch d, int GrMax);
__global__ void k_schur_fold(DevBatch d, int GrMax, int NT, int from_partials);
__global__ void k_imu_raw(DevBatch d, const double *pose_src, const double *sb_src, int gate);
__global__ void k_imu_weight(DevBatch d, double *cost_out, int gate);
#define ISV_PROF_FAMILIES 6      // 0 = k_proj_linearize<0>, 1 = k_sweep_mfma, 2 = k_rank1_mfma, 3 = k_build_solve*, 4 = k_dogleg, 5 = k_step_control
// pinned staging for the result records (capacity max_batch): asynchronous device-to-host copies into pageable
// memory go through the runtime's own staging and may complete lazily, which stalled the NEXT upload by 13-30 ms for
// batches above ~1 MB of records
struct SolverStage { SolveState *st; double *tc, *tr, *ts; int32_t *ta; isv_marg_result_t *marg; };
// isv_batch_upload's device half (isv_sequence.hip): raw CSR -> lm_* / f_* / pg_* arrays
size_t upload_build_lds_bytes(int N, int lcap);
int isv_upload_build_enqueue(DevBatch &d, const int32_t *optr, const double *obs_raw, int lcap, hipStream_t st);
int isv_solver_download(DevBatch &d, hipStream_t st, int n, const SolverStage &stage, isv_summary_t *summary, isv_marg_result_t *marg, std::string &err, bool staged = false);
void isv_solver_unpack_window(const SolverStage &stage, int b, isv_summary_t *summary, isv_marg_result_t *marg);
int isv_solver_debug_read(DevBatch &d, hipStream_t st, int what, double *out, int64_t count, std::string &err);
