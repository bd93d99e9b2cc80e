// isv_solver.hip -- on-device trust-region loop around k_build_solve: Ceres-Solver 2.0.0's
// TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG) as problemSolve() configures them
// (reference src/estimator.cpp:1119-1128: DENSE_SCHUR, DOGLEG, max_num_iterations=NUM_ITERATIONS),
// then the pseudo-measurement update (:1133-1144) and double2vector (:518-594).
//
// No host round trip inside the solve: every kernel reads the per-window SolveState and skips
// windows that have terminated; the host enqueues a fixed schedule of max_iter iteration slots.
//   slot:  [linearize if need_linearize] -> k_sweep_mfma -> k_rank1_mfma -> k_build_solve_sb -> k_dogleg (incl. back-substitution)
//          -> candidate evaluation (cost + model cost change) -> k_step_control
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "isv_kernels.h"
#include "isv_device_math.h"
#include "isv_proj_factor.h"
#include "isv_prior_factor.h"
#include "isv_imu_factor.h"

extern size_t build_solve_lds_bytes(int N, bool lds_T);
template <bool BIG, int NC, int MODE> __global__ void k_build_solve_sb(DevBatch d);
extern size_t build_solve_sb_bytes(int N, int prior_H_sz);
extern size_t build_solve_cs_doubles(int N);
template <bool EX, bool BIG, int NC, int NT> __global__ void k_lin_gram_chain(DevBatch d);
__global__ void k_front(DevBatch d);
template <bool BIG, int NC, bool EX> __global__ void k_pose_dogleg(DevBatch d);
extern size_t front_lds_bytes(int N, int slots);
template <bool BIG, int NC> __global__ void k_build_solve_st(DevBatch d);
extern size_t build_solve_st_bytes(int N, int prior_H_sz);
extern size_t build_solve_st_ws_doubles(int N);
__global__ void k_model_imu_prior(DevBatch d);
__global__ void k_backsub_split(DevBatch d);
__global__ void k_init_priors(DevBatch d, double *scratch, size_t per_window, double *kld_out);
__global__ void k_imu_raw(DevBatch d, const double *pose_src, const double *sb_src, int gate);
__global__ void k_imu_weight(DevBatch d, double *cost_out, int gate);
__global__ void k_sweep_mfma(DevBatch d);
template <int NT, int TPW, int R1_CHUNK, int MINW, bool EX> __global__ void k_rank1_mfma(DevBatch d);
__global__ void k_marg_clear(DevBatch d);
template <int MTF> __global__ void k_marg_fwd(DevBatch d);
template <int PART> __global__ void k_marg_bwd(DevBatch d);
template <int NC> __global__ void k_marg_jacobi(DevBatch d);
template <bool LDS_T> __global__ void k_build_solve(DevBatch d);

// ------------------------------------------------------------------------------------------
__global__ void k_init_state(DevBatch d) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= d.B) return;
    SolveState s;
    memset(&s, 0, sizeof(s));
    s.radius = 1e4; s.mu = 1e-8; s.need_linearize = 1; s.termination = ISV_TERM_RUNNING;
    if (ISV_SEQ_IDLE(d, w)) { s.termination = ISV_TERM_MAX_ITERATIONS; s.need_linearize = 0; }      // (a resident sequence without a frame this step: every solver kernel skips it)
    d.st[w] = s;
    for (int k = 0; k < ISV_MAX_TRACE; k++) {
        d.trace_cost[(size_t)w * ISV_MAX_TRACE + k] = 0; d.trace_radius[(size_t)w * ISV_MAX_TRACE + k] = 0;
        d.trace_step[(size_t)w * ISV_MAX_TRACE + k] = 0; d.trace_acc[(size_t)w * ISV_MAX_TRACE + k] = 0;
    }
}

#include "isv_dogleg.h"

// CONTROL: the candidate evaluation + TrustRegionMinimizer step control of this window follow in the same workgroup
// (one launch and one round of workgroups less per iteration than k_dogleg -> k_step_control<true>)
// EX: estimate_extrinsic = 1 (its own instantiation: the extra code costs the ordinary kernels registers otherwise)
template <bool CONTROL, bool EX>
__global__ __launch_bounds__(256, 4) void k_dogleg(DevBatch d) {
    __shared__ double red[256];
    __shared__ int s_accept;
    const int w = blockIdx.x, t = threadIdx.x;
#ifdef ISV_STAMP
    const unsigned long long tk0 = wall_clock64();
#endif
    dogleg_body<EX>(d, w, t, red);
#ifdef ISV_STAMP
    const unsigned long long tk1 = wall_clock64();
#endif
    if (CONTROL) {
        extern __shared__ __align__(16) double dync[];
        __syncthreads();                               // candidate states, per-factor costs and model pieces of this window are written
        step_control_body<true, EX>(d, w, t, dync, red, s_accept);
    }
#ifdef ISV_STAMP
    if (t == 0) { d.dbg[(size_t)w * 64 + 54] += (double)(tk1 - tk0); d.dbg[(size_t)w * 64 + 55] += (double)(wall_clock64() - tk1); }
#endif
}
template __global__ void k_dogleg<false, false>(DevBatch);
template __global__ void k_dogleg<true, false>(DevBatch);
template __global__ void k_dogleg<false, true>(DevBatch);
template __global__ void k_dogleg<true, true>(DevBatch);

// (round 4) the landmark back-substitution of a LONG window on many CUs (grid (B, groups of 128 landmarks)): what k_dogleg's
// first phase does with one workgroup -- per landmark the same function, so the same bits.  Launched between
// k_build_solve_sb and k_dogleg when the elimination is split (d.bs_split).
__global__ __launch_bounds__(128) void k_backsub_split(DevBatch d) {
    extern __shared__ __align__(16) double zsus[];
    const int w = blockIdx.x, t = threadIdx.x;
    const SolveState &st = d.st[w];
    if (st.termination != ISV_TERM_RUNNING || st.ls_fail || !st.fresh) return;
    const int n = d.np, l0 = d.lm_off[w], l1 = d.lm_off[w + 1];
    const int l = l0 + (int)blockIdx.y * 128 + t;
    if (l0 + (int)blockIdx.y * 128 >= l1) return;
    double *zs = zsus, *us = zsus + n;
    for (int i = t; i < n; i += 128) { zs[i] = d.zp[(size_t)w * n + i]; us[i] = d.up[(size_t)w * n + i]; }
    __syncthreads();
    if (l < l1) backsub_landmark<false>(d, l, d.f_off[w], zs, us, st.mu);
}


template <bool FUSED, bool EX>
__global__ __launch_bounds__(256) void k_step_control(DevBatch d) {
    extern __shared__ __align__(16) double cl[];
    __shared__ double red[256];
    __shared__ int s_accept;
    step_control_body<FUSED, EX>(d, blockIdx.x, threadIdx.x, cl, red, s_accept);
}

// ------------------------------------------------------------------------------------------
// after the solve: update() of every prior (estimator.cpp:1133-1144), then double2vector (:518-594)
__global__ __launch_bounds__(64) void k_finalize(DevBatch d, int do_update) {
    // one wavefront per window: lane roles for the prior updates, lane per frame for double2vector, lanes over
    // the landmarks for the depths
    const int w = blockIdx.x, lane = threadIdx.x;
    if (ISV_SEQ_IDLE(d, w)) return;
    const int N = d.N, v = d.Nvo - 1;
    const double *pose = d.pose + (size_t)w * N * 7, *sb = d.sb + (size_t)w * N * 9;
    double *Ps = d.Ps + (size_t)w * N * 3, *Rs = d.Rs + (size_t)w * N * 9, *Vs = d.Vs + (size_t)w * N * 3;
    double *Bas = d.Bas + (size_t)w * N * 3, *Bgs = d.Bgs + (size_t)w * N * 3;
    // ---- double2vector, part 1: the yaw re-anchoring rotation from the OLD Rs[0] / Ps[0] (every lane) ----
    double origin_R0[3], origin_P0[3], origin_R00[3], R00[9], rot_diff[9], Rs0[9];
    for (int k = 0; k < 9; k++) Rs0[k] = Rs[k];
    R2ypr(Rs0, origin_R0);
    for (int k = 0; k < 3; k++) origin_P0[k] = Ps[k];
    q_to_R(q_from_pose(pose), R00);
    R2ypr(R00, origin_R00);
    const double y_diff = origin_R0[0] - origin_R00[0];
    double ypr[3] = {y_diff, 0, 0};
    ypr2R(ypr, rot_diff);
    if (fabs(fabs(origin_R0[1]) - 90) < 1.0 || fabs(fabs(origin_R00[1]) - 90) < 1.0) m3_mul_nt(Rs0, R00, rot_diff);
    // ---- update() of the prior factors (pseudo-measurement shift), one lane each; they read the old Ps / Rs ----
    // (initFactorGraph creates its priors at the estimate and calls double2vector() only: do_update == 0)
    if (!do_update) {
    } else if (lane == 0) {
        // Linear9Factor::update  linear9_factor.h:60-68
        isv_linear9_t &f = d.lin9[w];
        for (int k = 0; k < 3; k++) {
            f.VB[k] += sb[9 * v + k] - Vs[3 * v + k];
            f.VB[3 + k] += sb[9 * v + 3 + k] - Bas[3 * v + k];
            f.VB[6 + k] += sb[9 * v + 6 + k] - Bgs[3 * v + k];
        }
    } else if (lane == 1) {
        // SE3PriorFactor::update  se3_prior_factor.h:73-81
        isv_se3_prior_t &f = d.se3[w];
        Quat R0 = q_from_R(Rs), R1 = q_normalized(q_from_pose(pose));
        double dR[3], E[9], Rn[9];
        so3_log(so3_mul(q_conj(R1), R0), dR);
        for (int k = 0; k < 3; k++) f.t[k] += pose[k] - Ps[k];
        q_to_R(so3_exp(dR), E); m3_mul(f.R, E, Rn);
        for (int k = 0; k < 9; k++) f.R[k] = Rn[k];
    } else if (lane < 32) {
        // RelativePoseFactor::update (solver overload)  relative_pose_factor.h:103-117
        for (int i = lane - 2; i < d.Nvo - 1; i += 30) {
            isv_relpose_t &f = d.relpose[(size_t)w * (d.Nvo - 1) + i];
            const double *PSi = pose + 7 * i, *PSj = pose + 7 * (i + 1);
            const double *ti = Ps + 3 * i, *tj = Ps + 3 * (i + 1), *Ri = Rs + 9 * i, *Rj = Rs + 9 * (i + 1);
            Quat Qi = q_from_pose(PSi), Qj = q_from_pose(PSj);
            double d_tj[3], d_ti[3], A[9], Bm[9], Qm[9];
            for (int k = 0; k < 3; k++) { d_tj[k] = PSj[k] - tj[k]; d_ti[k] = PSi[k] - ti[k]; }
            q_to_R(q_inv(Qj), Qm); m3_mul(Qm, Rj, A); Quat d_Rj = q_from_R(A);
            q_to_R(q_inv(Qi), Qm); m3_mul(Qm, Ri, Bm); Quat d_Ri = q_from_R(Bm);
            double lgi[3], lgj[3], v1[3], v2[3], S[9], v3[3];
            so3_log(d_Ri, lgi); so3_log(d_Rj, lgj);
            m3tv(Ri, d_tj, v1); m3tv(Ri, d_ti, v2);
            skew3(f.delta_t, S); m3v(S, lgi, v3);
            for (int k = 0; k < 3; k++) f.delta_t[k] += v1[k] - v2[k] + v3[k];
            double Ji[9], ww[3], E[9], T[9];
            q_to_R(q_mul(q_inv(Qj), Qi), Ji);
            for (int k = 0; k < 9; k++) Ji[k] = -Ji[k];
            m3v(Ji, lgi, ww);
            q_to_R(so3_exp(ww), E); m3_mul(f.delta_R, E, T); for (int k = 0; k < 9; k++) f.delta_R[k] = T[k];
            q_to_R(so3_exp(lgj), E); m3_mul(f.delta_R, E, T); for (int k = 0; k < 9; k++) f.delta_R[k] = T[k];
        }
    } else {
        // RollPitchFactor::update  rollpitch_factor.h:78-83
        for (int m = lane - 32; m < d.n_rp[w]; m += 32) {
            isv_rollpitch_t &f = d.rollpitch[(size_t)w * d.max_rp + m];
            const int idx = f.index;
            Quat R0 = q_from_R(Rs + 9 * idx), R1 = q_normalized(q_from_pose(pose + 7 * idx));
            double dR[3], E[9], T[9];
            so3_log(so3_mul(q_conj(R1), R0), dR);
            q_to_R(so3_exp(dR), E); m3_mul(f.R, E, T); for (int k = 0; k < 9; k++) f.R[k] = T[k];
        }
    }
    __syncthreads();                       // every update has read the old window states and written its prior
    // ---- double2vector, part 2 ----------------------------------------------------------------
    if (lane == 0) {
        double tt[3];
        m3v(rot_diff, d.lin9[w].VB + 6, tt); for (int k = 0; k < 3; k++) d.lin9[w].VB[6 + k] = tt[k];     // :549 (gyro-bias slot)
    } else if (lane == 1) {
        double Rn[9];
        m3_mul(rot_diff, d.se3[w].R, Rn); for (int k = 0; k < 9; k++) d.se3[w].R[k] = Rn[k];               // :550
    } else if (lane == 2) {
        // tic / ric <- para_Ex_Pose (:577-586); when the extrinsic is estimated the solved block is the pseudo-frame's
        // pose, which also becomes d.ex (MargForward linearises at it, and the caller reads para_Ex_Pose from it)
        double *exw = d.ex + (size_t)w * 7;
        if (d.est_ex) for (int k = 0; k < 7; k++) exw[k] = pose[7 * d.Nr + k];
        for (int k = 0; k < 3; k++) d.tic[(size_t)w * 3 + k] = exw[k];
        q_to_R(q_from_pose(exw), d.ric + (size_t)w * 9);
    }
    const double p0[3] = {pose[0], pose[1], pose[2]};
    for (int i = lane; i < d.Nr; i += 64) {            // (the real frames: the pseudo-frame is not a body pose)
        double Ri[9], dd[3], tt[3], Ro[9];
        q_to_R(q_normalized(q_from_pose(pose + 7 * i)), Ri);
        m3_mul(rot_diff, Ri, Ro);
        for (int k = 0; k < 9; k++) Rs[9 * i + k] = Ro[k];
        for (int k = 0; k < 3; k++) dd[k] = pose[7 * i + k] - p0[k];
        m3v(rot_diff, dd, tt);
        for (int k = 0; k < 3; k++) Ps[3 * i + k] = tt[k] + origin_P0[k];
        m3v(rot_diff, sb + 9 * i, tt);
        for (int k = 0; k < 3; k++) { Vs[3 * i + k] = tt[k]; Bas[3 * i + k] = sb[9 * i + 3 + k]; Bgs[3 * i + k] = sb[9 * i + 6 + k]; }
    }
    for (int l = d.lm_off[w] + lane; l < d.lm_off[w + 1]; l += 64) {        // FeatureManager::setDepth :145-163
        const double dep = 1.0 / d.lam[l];
        d.depth[l] = dep;
        d.solve_flag[l] = (dep < 0 || dep > 10) ? 2 : 1;
    }
}

// ------------------------------------------------------------------------------------------
// host side
#define HCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return ISV_ERR_DEVICE; } } while (0)

template <typename T>
static int dal(T **p, size_t n, std::vector<void *> &allocs, std::string &err) {
    void *q = nullptr;
    HCHK(hipMalloc(&q, (n ? n : 1) * sizeof(T)));
    allocs.push_back(q);
    *p = (T *)q;
    return ISV_OK;
}
#define TRYA(x) do { int rc_ = (x); if (rc_ != ISV_OK) return rc_; } while (0)
static std::mutex g_lds_attr_mutex;
// dynamic LDS of k_dogleg (candidate-point IMU / prior evaluation on the LDS path + the two tangent vectors) and of the step control
static size_t dogleg_lds_bytes(const DevBatch &d) {
    // + two 496-double staging rows for the IMU J^T J records of the model pieces and the tangent step (np)
    // + the window's prior J^T J record (prior_H_sz) for the priors' model pieces
    return (d.lds_T ? ((size_t)(d.N - 1) * 48 + (size_t)d.n_prior_slots * 16 + 992 + (size_t)d.np + (d.dg_stage_ph ? (size_t)d.prior_H_sz : 0)) * sizeof(double) + prior_lds_bytes(d.n_prior_slots, false) : 0) + 2 * (size_t)d.np * sizeof(double);
}
// the step control: candidate / current poses, the tangent step, and (when they fit) the per-landmark data its factor loop gathers
// -- host point, candidate and current inverse depth, the landmark's step: 6 doubles each (d.ctl_stage_lm, set per enqueue)
static size_t step_control_lds_bytes(const DevBatch &d) { return ((size_t)30 * d.N + 12 + (d.ctl_stage_lm ? 6 * (size_t)d.lg_lcap : 0)) * sizeof(double); }
#define ISV_CTL_STAGE_MAX_BYTES ((size_t)36 * 1024)      // staged only while four workgroups still share a CU

// workgroups of kernel `fn` (threads per workgroup, dynamic LDS bytes) the current device holds at once
static size_t resident_workgroups(const void *fn, int threads, size_t lds) {
    int per_cu = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return (size_t)per_cu * (size_t)cus;
}

int isv_solver_alloc(DevBatch &d, SolverHost &hc, size_t B, size_t L, size_t F, std::vector<void *> &allocs, std::string &err) {
    const size_t n = d.np, NI = B * (d.N - 1);
    TRYA(dal(&d.scale_p, B * n, allocs, err)); TRYA(dal(&d.diag_p, B * n, allocs, err)); TRYA(dal(&d.grad_p, B * n, allocs, err));
    TRYA(dal(&d.gn_p, B * n, allocs, err)); TRYA(dal(&d.delta_p, B * n, allocs, err)); TRYA(dal(&d.zp, B * n, allocs, err)); TRYA(dal(&d.up, B * n, allocs, err));
    TRYA(dal(&d.lmE, L, allocs, err)); TRYA(dal(&d.lmG, L, allocs, err)); TRYA(dal(&d.scale_l, L, allocs, err)); TRYA(dal(&d.diag_l, L, allocs, err));
    TRYA(dal(&d.grad_l, L, allocs, err)); TRYA(dal(&d.gn_l, L, allocs, err)); TRYA(dal(&d.delta_l, L, allocs, err)); TRYA(dal(&d.lm_aterm, L, allocs, err));
    TRYA(dal(&d.fcost_c, F, allocs, err)); TRYA(dal(&d.imu_cost_c, NI, allocs, err)); TRYA(dal(&d.prior_cost_c, B * (size_t)d.n_prior_slots, allocs, err)); TRYA(dal(&d.cost_c, B, allocs, err));
    TRYA(dal(&d.fmodel, F, allocs, err)); TRYA(dal(&d.imu_model, NI, allocs, err)); TRYA(dal(&d.prior_model, B * (size_t)d.n_prior_slots, allocs, err)); TRYA(dal(&d.model, B, allocs, err));
    TRYA(dal(&d.trace_cost, B * ISV_MAX_TRACE, allocs, err)); TRYA(dal(&d.trace_radius, B * ISV_MAX_TRACE, allocs, err));
    TRYA(dal(&d.trace_step, B * ISV_MAX_TRACE, allocs, err)); TRYA(dal(&d.trace_acc, B * ISV_MAX_TRACE, allocs, err));
    d.tvis_sz = 36 * (d.N * (d.N + 1) / 2) + 18 * d.N;
    TRYA(dal(&d.W, (F + L) * 6, allocs, err)); TRYA(dal(&d.lm_cg, L, allocs, err));
    d.wd_ld = 16 * ((6 * d.N + 16) / 16);          // 6N pose columns + the g_l column, rounded up to 16
    TRYA(dal(&d.Tvis, B * (size_t)d.tvis_sz, allocs, err));
    TRYA(dal(&d.dbg, B * 64, allocs, err)); TRYA(dal(&d.act, ISV_MAX_TRACE, allocs, err));
    HCHK(hipMemset(d.dbg, 0, B * 64 * sizeof(double)));
    TRYA(dal(&d.marg, B, allocs, err)); TRYA(dal(&d.margin_old, B, allocs, err)); TRYA(dal(&d.header0, B, allocs, err));
    const size_t nblkT = (size_t)d.N * (d.N + 1) / 2 * 225;
    // LDS-resident fast path: the structure-aware solve (two windows per CU up to N = 11, one beyond), the pair
    // partials of k_sweep_mfma and the panel + landmark tables of k_rank1_mfma must fit the 160 KB of a CU
    const size_t lds_r1 = (64 * (size_t)(d.wd_ld + 4) + (size_t)d.max_lm * 3 + 2) * sizeof(double);
    const size_t n_pairs = (size_t)d.N * (d.N - 1) / 2;
    const bool sw_global = n_pairs * 84 * sizeof(double) > 40 * 1024;      // partials too big to share a CU's LDS four ways
    const size_t lds_sw = (n_pairs * 84 + (n_pairs + 2) / 2 + 1) * sizeof(double);       // (the LDS variant's size; the global one needs less)
    const size_t lds_sb = build_solve_sb_bytes(d.N, d.prior_H_sz);
    d.lds_T = (d.N <= 20 && d.wd_ld <= 128 && d.prior_H_sz <= 1024 && lds_sb <= (d.N <= 11 ? 80u : 160u) * 1024 &&
               lds_r1 <= 160 * 1024 && lds_sw <= 160 * 1024) ? 1 : 0;
    if (getenv("ISV_DEBUG_PATH")) fprintf(stderr, "isv: N=%d wd_ld=%d prior_H_sz=%d lds_sb=%zu lds_r1=%zu lds_sw=%zu -> lds_T=%d\n", d.N, d.wd_ld, d.prior_H_sz, lds_sb, lds_r1, lds_sw, d.lds_T);
    TRYA(dal(&d.Tglob, d.lds_T ? 1 : B * nblkT, allocs, err));
    d.sw_part = nullptr; d.sw_global = 0;
    // ISV_DEBUG_SW_GLOBAL=1 (test hook): give every LDS-path handle the global pair-partial scratch and use it for
    // every launch, so that small batches exercise the variant large N >= 12 batches run
    hc.one_stream = getenv("ISV_ONE_STREAM") != nullptr; hc.split_control = getenv("ISV_SPLIT_CONTROL") != nullptr;
    hc.generic_n = getenv("ISV_GENERIC_N") != nullptr; hc.lg_batch_waves = getenv("ISV_LG_BATCH_WAVES") != nullptr;
    hc.debug_sw_global = getenv("ISV_DEBUG_SW_GLOBAL") != nullptr; hc.legacy_visual = getenv("ISV_LEGACY_VISUAL") != nullptr;
    hc.no_persistent = getenv("ISV_NO_PERSISTENT") != nullptr; hc.no_split = getenv("ISV_NO_SPLIT") != nullptr;
    hc.no_update = getenv("ISV_DEBUG_NO_UPDATE") != nullptr; hc.no_pose_dogleg = getenv("ISV_NO_POSE_DOGLEG") != nullptr;
    hc.marg_one_kernel = getenv("ISV_MARG_ONE_KERNEL") != nullptr; hc.marg_split = getenv("ISV_MARG_SPLIT") != nullptr;
    int dev0_ = 0;
    HCHK(hipGetDevice(&dev0_));
    if (hipDeviceGetAttribute(&hc.n_cus, hipDeviceAttributeMultiprocessorCount, dev0_) != hipSuccess) { (void)hipGetLastError(); hc.n_cus = 0; }
    // one long window over many CUs (k_schur_split): possible on this handle when a window can have ISV_SPLIT_MIN_PASSES passes
    const bool can_split = d.lds_T && !d.est_ex && !hc.no_split && ((size_t)d.max_lm + 63) / 64 >= ISV_SPLIT_MIN_PASSES && hc.n_cus > ISV_SPLIT_MIN_PASSES;
    if (d.lds_T && (sw_global || hc.debug_sw_global || can_split)) TRYA(dal(&d.sw_part, B * n_pairs * 84, allocs, err));
    hc.sw_global_ok = sw_global || hc.debug_sw_global;
    hc.cap_batch = B;
    // k_build_solve_st (isv_build_solve_st.hip): a quarter (N <= 11) / half of a CU's LDS per window, 256 threads.  It wins when
    // the batch fills the GPU more than twice over with k_build_solve_sb's two (one) windows per CU; a window alone on a CU is
    // faster with the 512-thread kernel.  The choice is per HANDLE (its max_batch): a window gives the same bits alone and in any
    // batch of the same handle.
    d.st_ws = nullptr;
    {
        const size_t lds_st = build_solve_st_bytes(d.N, d.prior_H_sz);
        const bool fits = d.lds_T && (d.N <= 11 ? lds_st <= 40 * 1024 : lds_st <= 80 * 1024);
        const char *e = getenv("ISV_SOLVE_ST");
        const size_t per_cu_sb = d.N <= 11 ? 2 : 1;
        // (measured: N = 11, 1024 windows: 180 -> 146 us per launch, 182 -> 190-193 k windows/s, 512 windows: 3.55 against 3.22 ms;
        //  N = 18 (512 threads, two windows per CU), 1024 windows: 11.09 -> 10.33 ms per step, 512: 5.99 -> 5.61, 256: 3.27 -> 3.68:
        //  only handles whose batches exceed the resident windows of k_build_solve_sb take it)
        hc.solve_st = fits && (e ? atoi(e) != 0 : B > per_cu_sb * (size_t)hc.n_cus);
        if (getenv("ISV_DEBUG_PATH")) fprintf(stderr, "isv: k_build_solve_st lds=%zu fits=%d -> solve_st=%d\n", lds_st, (int)fits, (int)hc.solve_st);
        if (hc.solve_st) TRYA(dal(&d.st_ws, B * build_solve_st_ws_doubles(d.N), allocs, err));
    }
    // the split solve (isv_build_solve_sb.hip, MODE 1 / 2): handles that run k_build_solve_sb and whose batches leave CUs idle -- the chain
    // kernel needs a CU of its own beside k_lin_gram / k_rank1_mfma to take the speed/bias elimination off the critical path
    d.cs_ws = nullptr;
    {
        const char *e = getenv("ISV_CHAIN_SPLIT");
        hc.chain_split = d.lds_T && !hc.solve_st && (e ? atoi(e) != 0 : B <= (size_t)hc.n_cus);
        if (hc.chain_split) { TRYA(dal(&d.cs_ws, B * build_solve_cs_doubles(d.N), allocs, err)); HCHK(hipMemset(d.cs_ws, 0, B * build_solve_cs_doubles(d.N) * sizeof(double))); }     // (entries above the diagonal of a diagonal block are never written: finite)
    }
    d.r1_part = nullptr;
    // groups per window (DevBatch::split_cap), PER HANDLE (its max_batch and max_landmarks): a window is split the same way alone and inside
    // any batch of the handle.  All or nothing: the split pays while every group of every window of a full batch finds a CU of its own
    // (measured, round 5: with two groups' workgroups per CU -- 128 windows x 4 groups, 256 x 2 -- the partial tiles' traffic, 123 KB per
    // window and group against 20 KB of Tvis, and the fold launch cost what the shorter passes save: 1.89 -> 1.90 ms, 2.35 -> 2.45 ms).
    d.split_cap = 1;
    if (can_split) {
        const size_t pm = ((size_t)d.max_lm + 63) / 64, gmH = pm < ISV_SPLIT_MAX_GROUPS ? pm : ISV_SPLIT_MAX_GROUPS;
        if (B * (gmH + 1) <= (size_t)hc.n_cus) d.split_cap = ISV_SPLIT_MAX_GROUPS;
    }
    if (can_split && d.split_cap >= 2) {
        const size_t nt_ = d.wd_ld / 16, pm = ((size_t)d.max_lm + 63) / 64, gm = pm < (size_t)d.split_cap ? pm : (size_t)d.split_cap;
        TRYA(dal(&d.r1_part, B * gm * (nt_ * (nt_ + 1) / 2) * 256, allocs, err));
    }
    d.marg_scratch_sz = 26;
    TRYA(dal(&d.marg_scratch, L * 26, allocs, err));
    TRYA(dal(&d.marg_ws, B * ISV_MARG_WS, allocs, err));
    int dev_ = 0;
    HCHK(hipGetDevice(&dev_));
    if (d.lds_T) {
        // the attribute is per kernel (and device) and process-wide: handles of different shapes must not lower each other's value
#define SETLDS(K, BYTES) do { static size_t cur_[64] = {}; std::lock_guard<std::mutex> lk_(g_lds_attr_mutex); if ((size_t)(BYTES) > cur_[dev_ & 63]) { HCHK(hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES))); cur_[dev_ & 63] = (size_t)(BYTES); } } while (0)
        SETLDS((k_rank1_mfma<1, 1, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<1, 1, 64, 1, true>), lds_r1); SETLDS((k_rank1_mfma<2, 1, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<2, 1, 64, 1, true>), lds_r1); SETLDS((k_rank1_mfma<3, 1, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<3, 1, 64, 1, true>), lds_r1);
        SETLDS((k_rank1_mfma<4, 1, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<4, 1, 64, 1, true>), lds_r1); SETLDS((k_rank1_mfma<5, 1, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<5, 1, 64, 1, true>), lds_r1); SETLDS((k_rank1_mfma<6, 2, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<6, 2, 64, 1, true>), lds_r1);
        SETLDS((k_rank1_mfma<7, 2, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<7, 2, 64, 1, true>), lds_r1); SETLDS((k_rank1_mfma<8, 3, 64, 1, false>), lds_r1); SETLDS((k_rank1_mfma<8, 3, 64, 1, true>), lds_r1);
        SETLDS(k_sweep_mfma, lds_sw);
        if (d.r1_part) {
            const size_t lds_sp = lds_r1 > (2 * n_pairs + 2) * sizeof(int) ? lds_r1 : (2 * n_pairs + 2) * sizeof(int);
            SETLDS((k_schur_split<1, 1>), lds_sp); SETLDS((k_schur_split<2, 1>), lds_sp); SETLDS((k_schur_split<3, 1>), lds_sp); SETLDS((k_schur_split<4, 1>), lds_sp);
            SETLDS((k_schur_split<5, 1>), lds_sp); SETLDS((k_schur_split<6, 2>), lds_sp); SETLDS((k_schur_split<7, 2>), lds_sp); SETLDS((k_schur_split<8, 3>), lds_sp);
            SETLDS((k_rank1_split<1, 1>), lds_r1); SETLDS((k_rank1_split<2, 1>), lds_r1); SETLDS((k_rank1_split<3, 1>), lds_r1); SETLDS((k_rank1_split<4, 1>), lds_r1);
            SETLDS((k_rank1_split<5, 1>), lds_r1); SETLDS((k_rank1_split<6, 2>), lds_r1); SETLDS((k_rank1_split<7, 2>), lds_r1); SETLDS((k_rank1_split<8, 3>), lds_r1);
        }
        {   // k_lin_gram: up to the whole CU (the launch sizes its LDS from the uploaded windows: isv_batch_upload)
            auto cap = [&](int waves) { const size_t b = lin_gram_lds_bytes(d.Nr, true, d.est_ex != 0, waves, d.max_lm); return b < ISV_LDS_PER_CU ? b : ISV_LDS_PER_CU; };
            if (d.est_ex) { SETLDS((k_lin_gram<true, LG_WAVES>), cap(LG_WAVES)); SETLDS((k_lin_gram<true, LG_WAVES_SMALL>), cap(LG_WAVES_SMALL)); }
            else { SETLDS((k_lin_gram<false, LG_WAVES>), cap(LG_WAVES)); SETLDS((k_lin_gram<false, LG_WAVES_SMALL>), cap(LG_WAVES_SMALL)); }
        }
        if (hc.solve_st) {
            const size_t lds_st = build_solve_st_bytes(d.N, d.prior_H_sz);
            if (d.N == 11) SETLDS((k_build_solve_st<false, 11>), lds_st);
            if (d.N <= 11) SETLDS((k_build_solve_st<false, 0>), lds_st);
            if (d.N > 11) SETLDS((k_build_solve_st<true, 0>), lds_st);
            if (d.N == 18) SETLDS((k_build_solve_st<true, 18>), lds_st);
        }
        if (d.N == 11) SETLDS((k_build_solve_sb<false, 11, 0>), lds_sb);
        if (d.N <= 11) SETLDS((k_build_solve_sb<false, 0, 0>), lds_sb);
        if (d.N > 11) SETLDS((k_build_solve_sb<true, 0, 0>), lds_sb);
        if (hc.chain_split) {
            SETLDS(k_front, front_lds_bytes(d.N, d.n_prior_slots));
            {   // k_lin_gram_chain: the larger of its two roles' needs
                const size_t lg = lin_gram_lds_bytes(d.Nr, true, d.est_ex != 0, LG_WAVES_SMALL, d.max_lm);
                size_t both = lg > lds_sb ? lg : lds_sb;
                if (both < lds_r1) both = lds_r1;
                if (both > ISV_LDS_PER_CU) both = ISV_LDS_PER_CU;
                if (d.est_ex) { if (d.N <= 11) SETLDS((k_lin_gram_chain<true, false, 0, 0>), both); else SETLDS((k_lin_gram_chain<true, true, 0, 0>), both); }
                else if (d.N == 11) { SETLDS((k_lin_gram_chain<false, false, 11, 0>), both); SETLDS((k_lin_gram_chain<false, false, 0, 0>), both); }
                else if (d.N < 11) SETLDS((k_lin_gram_chain<false, false, 0, 0>), both);
                else { SETLDS((k_lin_gram_chain<false, true, 0, 0>), both); SETLDS((k_lin_gram_chain<false, true, 0, 7>), both); }
            }
            {   // k_pose_dogleg: the pose half's LDS or the dogleg / step control's, whichever is larger (the latter is sized per enqueue: up to a CU)
                const size_t pd = ISV_LDS_PER_CU - 4096;
                if (d.est_ex) { if (d.N <= 11) SETLDS((k_pose_dogleg<false, 0, true>), pd); else SETLDS((k_pose_dogleg<true, 0, true>), pd); }
                else if (d.N == 11) { SETLDS((k_pose_dogleg<false, 11, false>), pd); SETLDS((k_pose_dogleg<false, 0, false>), pd); }
                else if (d.N < 11) SETLDS((k_pose_dogleg<false, 0, false>), pd);
                else SETLDS((k_pose_dogleg<true, 0, false>), pd);
            }
            if (d.N == 11) { SETLDS((k_build_solve_sb<false, 11, 1>), lds_sb); SETLDS((k_build_solve_sb<false, 11, 2>), lds_sb); }
            if (d.N <= 11) { SETLDS((k_build_solve_sb<false, 0, 1>), lds_sb); SETLDS((k_build_solve_sb<false, 0, 2>), lds_sb); }
            if (d.N > 11) { SETLDS((k_build_solve_sb<true, 0, 1>), lds_sb); SETLDS((k_build_solve_sb<true, 0, 2>), lds_sb); }
        }
#undef SETLDS
    }
    else {
        static size_t cur_bs[64] = {};
        std::lock_guard<std::mutex> lk(g_lds_attr_mutex);
        const size_t want = build_solve_lds_bytes(d.N, false);
        if (want > cur_bs[dev_ & 63]) { HCHK(hipFuncSetAttribute((const void *)k_build_solve<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want)); cur_bs[dev_ & 63] = want; }
    }
    // device figures the per-launch variant choice needs (once per handle, not per isv_batch_optimize): hc.n_cus (above) and
    {
        // k_dogleg<true, EX> by registers alone (a small dynamic LDS request); the LDS bound is applied per enqueue
        int per_cu = 0;
        if (d.lds_T && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, d.est_ex ? (const void *)k_dogleg<true, true> : (const void *)k_dogleg<true, false>, 256, 1024) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        hc.dogleg_per_cu_regs = per_cu;
    }
    return ISV_OK;
}

// prof_ev (optional): [max_iter][ISV_PROF_FAMILIES][2] events recorded around the three dominant
// kernel families on the solve stream, so bench.py can report per-kernel durations measured live.
int isv_solver_enqueue(DevBatch &d, const SolverHost &hc, hipStream_t st, hipStream_t st2, hipEvent_t *fj, int64_t *counts, hipEvent_t *prof_ev, std::string &err) {
    // st2 / fj[4]: side stream + fork/join events: the IMU and prior factor kernels are latency-bound and
    // independent of the reprojection pipeline, so they run beside it and are joined before their consumers
#define PROF(slot, fam, which) do { if (prof_ev) (void)hipEventRecord(prof_ev[((slot) * ISV_PROF_FAMILIES + (fam)) * 2 + (which)], st); } while (0)
    const size_t NI = (size_t)d.B * (d.N - 1);
    const size_t lds_proj = 4 * proj_lds_doubles_per_wave(d.N, 0) * sizeof(double), lds_proj1 = 4 * proj_lds_doubles_per_wave(d.N, 1) * sizeof(double);
    if (hc.one_stream) st2 = st;            // diagnostics: serialise everything on one stream (per-kernel timelines)
    const size_t lds_bs = build_solve_lds_bytes(d.N, false);
    d.ctl_stage_lm = (6 * (size_t)d.lg_lcap * sizeof(double) <= ISV_CTL_STAGE_MAX_BYTES) ? 1 : 0;
    // the priors' J^T J record is staged in k_dogleg's LDS while that keeps the batch in one resident round (else: read from global)
    d.dg_stage_ph = 1;
    if (ISV_LDS_PER_CU / ((dogleg_lds_bytes(d) > step_control_lds_bytes(d) ? dogleg_lds_bytes(d) : step_control_lds_bytes(d)) + 2304) * (size_t)hc.n_cus < (size_t)d.B) d.dg_stage_ph = 0;
    const size_t lds_dg = dogleg_lds_bytes(d), lds_sc = step_control_lds_bytes(d);
    const size_t lds_dgc = lds_dg > lds_sc ? lds_dg : lds_sc;
    // workgroups of k_dogleg<true> a CU holds: registers (hc) and LDS (dynamic + the 2 KB of static reduction space)
    const size_t dg_by_lds = ISV_LDS_PER_CU / (lds_dgc + 2304);
    const size_t res_dogleg_ctl = (size_t)hc.n_cus * ((size_t)hc.dogleg_per_cu_regs < dg_by_lds ? (size_t)hc.dogleg_per_cu_regs : dg_by_lds);
    // dogleg + step control in one kernel while the batch fits ONE resident round of it (it holds fewer workgroups per CU
    // than k_dogleg<false>: at 2048 windows the fused kernel needs a second round and the step is 21 % slower, measured);
    // same arithmetic either way (bitwise, tests/test_gpu_branches.py)
    // (a free extrinsic is only evaluated by the per-window kernels: k_proj_linearize<1> reads the constant one)
    const bool control_in_wg = d.lds_T && ((size_t)d.Ftot <= (size_t)4096 * d.B || d.est_ex);
    const bool fuse_control = control_in_wg && !hc.split_control && (size_t)d.B <= res_dogleg_ctl;
    const int n_cus = hc.n_cus;
    hipLaunchKernelGGL(k_init_state, dim3((d.B + 63) / 64), dim3(64), 0, st, d);
    for (int slot = 0; slot < d.max_iter; slot++) {
        // linearise where needed (k_*_linearize skip windows whose need_linearize == 0 via the tile/window flags)
        const bool generic_n = hc.generic_n;             // (A/B / test hook: the run-time-N instantiations for every N)
        // Small-batch handles (hc.chain_split: every workgroup of every kernel finds a CU of its own) run the whole iteration on ONE
        // stream: an event record costs the critical stream ~7 us and a wait that really blocks ~13 us on this GPU (measured, rocprofv3
        // kernel trace of one window), more than the kernels they would overlap.  Instead, independent work shares a LAUNCH:
        // k_front = {IMU factors | prior factors} of every window, k_lin_gram_chain = {k_lin_gram | chain half of the split solve}.
        const bool cs = d.lds_T && hc.chain_split;
        bool chain_done = false, rank1_done = false, dogleg_done = false;
        // (what the visual part of this upload will run, needed here already: k_lin_gram_chain or not)
        const bool fused = d.lds_T && d.fused_visual;
        d.sw_global = (d.sw_part && hc.sw_global_ok && (d.B > 256 || hc.debug_sw_global)) ? 1 : 0;     // more than one workgroup per CU: trade LDS for occupancy
        const int lgw = (fused && d.B <= n_cus && !hc.lg_batch_waves && lin_gram_lds_bytes(d.Nr, !d.sw_global, d.est_ex != 0, LG_WAVES_SMALL, d.lg_lcap) <= ISV_LDS_PER_CU) ? LG_WAVES_SMALL : LG_WAVES;
        bool chain_on_side = false;
        if (cs) {
            hipLaunchKernelGGL(k_front, dim3(d.B, 2), dim3(512), front_lds_bytes(d.N, d.n_prior_slots), st, d);
            if (!(fused && lgw == LG_WAVES_SMALL)) {
                // an upload k_lin_gram_chain does not take (windows beyond the fused kernel's limits: BASELINE config 5's 30 000 factors):
                // the chain half (~75 us at 20 frames) on the side stream beside the linearisation and the split elimination -- worth its
                // fork / join events here (on the solve stream behind them it made the one-launch solve's 142 us into 75 + 75)
                HCHK(hipEventRecord(fj[0], st)); HCHK(hipStreamWaitEvent(st2, fj[0], 0));
                const size_t lds_sb = build_solve_sb_bytes(d.N, d.prior_H_sz);
                if (d.N == 11 && !generic_n) hipLaunchKernelGGL((k_build_solve_sb<false, 11, 1>), dim3(d.B), dim3(512), lds_sb, st2, d);
                else if (d.N <= 11) hipLaunchKernelGGL((k_build_solve_sb<false, 0, 1>), dim3(d.B), dim3(512), lds_sb, st2, d);
                else hipLaunchKernelGGL((k_build_solve_sb<true, 0, 1>), dim3(d.B), dim3(512), lds_sb, st2, d);
                HCHK(hipEventRecord(fj[1], st2));
                chain_on_side = true; chain_done = true;
            }
        } else {
            HCHK(hipEventRecord(fj[0], st)); HCHK(hipStreamWaitEvent(st2, fj[0], 0));
            if (NI) {
                hipLaunchKernelGGL(k_imu_raw, dim3((unsigned)(NI + 63) / 64), dim3(256), 0, st2, d, d.pose, d.sb, 1);
                hipLaunchKernelGGL(k_imu_weight, dim3((unsigned)(NI + 7) / 8), dim3(256), 0, st2, d, d.imu_cost, 1);
            }
            hipLaunchKernelGGL(k_prior_linearize<true>, dim3(d.B), dim3(64), prior_lds_bytes(d.n_prior_slots), st2, d, d.pose, d.sb, d.prior_cost, 1);
            HCHK(hipEventRecord(fj[1], st2));
        }
        // a SMALL batch with LONG windows: one window's elimination over many CUs (k_schur_split + k_schur_fold, isv_sweep.hip)
        // WHETHER a window's downdates are split is decided by the HANDLE (its max_batch and max_landmarks: ADVICE r4 -- with the groups
        // counted from the longest window of the upload, a 320-landmark window was split alone and unsplit beside a 2048-landmark one);
        // into how many groups, by the window (schur_split_groups).  An upload without a window of ISV_SPLIT_MIN_PASSES passes skips
        // the split launch: its windows are one group each, the unsplit sums.
        const int Pmax = (d.lg_lcap + 63) / 64, PmaxH = (d.max_lm + 63) / 64, GrMax = PmaxH < d.split_cap ? PmaxH : d.split_cap;
        const bool split = d.lds_T && d.r1_part && !hc.no_split && Pmax >= ISV_SPLIT_MIN_PASSES && d.split_cap >= 2;
        // the landmark back-substitution on its own workgroups (k_backsub_split) pays for LONG windows only: per landmark it is the same
        // routine as k_dogleg's first phase (same bits), which takes ~2.5 us for 300 landmarks against 6-8 us for the extra launch
        const bool bsub_split = split && d.lg_lcap > 1024;
        d.bs_split = bsub_split ? 1 : 0;
        PROF(slot, 0, 0);
        if (fused) {
            // linearisation fused with the Gram products: no Jacobian strip goes to HBM (isv_visual.hip)
            // (eight wavefronts per window while the batch leaves every window a CU of its own: 41 -> 27 us per launch for one window)
            const bool ex = d.est_ex != 0;
            const size_t lds_lg = lin_gram_lds_bytes(d.Nr, !d.sw_global, ex, lgw, d.lg_lcap);
            if (cs && lgw == LG_WAVES_SMALL) {
                const size_t lds_sb = build_solve_sb_bytes(d.N, d.prior_H_sz), lds_r1c = (64 * (d.wd_ld + 4) + (size_t)d.max_lm * 3 + 2) * sizeof(double);
                size_t both = lds_lg > lds_sb ? lds_lg : lds_sb;
                const dim3 g2(d.B, 2);
                // the rank-1 downdates in the same workgroup (handles that do not split the elimination; the shapes instantiated for it)
                const int nt = d.wd_ld / 16;
                const bool r1_here = !split && !ex && !generic_n && d.N > 11 && nt == 7 && lds_r1c <= ISV_LDS_PER_CU;
                if (r1_here && both < lds_r1c) both = lds_r1c;
                if (ex) { if (d.N <= 11) hipLaunchKernelGGL((k_lin_gram_chain<true, false, 0, 0>), g2, dim3(512), both, st, d); else hipLaunchKernelGGL((k_lin_gram_chain<true, true, 0, 0>), g2, dim3(512), both, st, d); }
                else if (r1_here) hipLaunchKernelGGL((k_lin_gram_chain<false, true, 0, 7>), g2, dim3(512), both, st, d);
                else if (d.N == 11 && !generic_n) hipLaunchKernelGGL((k_lin_gram_chain<false, false, 11, 0>), g2, dim3(512), both, st, d);
                else if (d.N <= 11) hipLaunchKernelGGL((k_lin_gram_chain<false, false, 0, 0>), g2, dim3(512), both, st, d);
                else hipLaunchKernelGGL((k_lin_gram_chain<false, true, 0, 0>), g2, dim3(512), both, st, d);
                chain_done = true; rank1_done = r1_here;
            } else if (lgw == LG_WAVES_SMALL) {
                if (ex) hipLaunchKernelGGL((k_lin_gram<true, LG_WAVES_SMALL>), dim3(d.B), dim3(64 * LG_WAVES_SMALL), lds_lg, st, d);
                else hipLaunchKernelGGL((k_lin_gram<false, LG_WAVES_SMALL>), dim3(d.B), dim3(64 * LG_WAVES_SMALL), lds_lg, st, d);
            } else if (ex) hipLaunchKernelGGL((k_lin_gram<true, LG_WAVES>), dim3(d.B), dim3(64 * LG_WAVES), lds_lg, st, d);
            else hipLaunchKernelGGL((k_lin_gram<false, LG_WAVES>), dim3(d.B), dim3(64 * LG_WAVES), lds_lg, st, d);
            counts[0]++; counts[4] = 1;
        } else if (d.n_tiles > 0) { hipLaunchKernelGGL(k_proj_linearize<0>, dim3((d.n_tiles + 3) / 4), dim3(256), lds_proj, st, d, d.pose, d.lam, d.fcost, 1); counts[0]++; }
        PROF(slot, 0, 1);
        if (split) {
            int Gs = 0;
            if (!fused) { Gs = n_cus / d.B - GrMax; if (Gs > 16) Gs = 16; if (Gs < 1) Gs = 1; }
            const int nt = d.wd_ld / 16;
            const size_t n_pairs = (size_t)d.N * (d.N - 1) / 2;
            size_t lds_sp = (64 * (d.wd_ld + 4) + (size_t)d.max_lm * 3 + 2) * sizeof(double);
            const size_t lds_r1s = lds_sp;
            if (lds_sp < (2 * n_pairs + 2) * sizeof(int)) lds_sp = (2 * n_pairs + 2) * sizeof(int);
            const dim3 grid(d.B, Gs + GrMax);
            PROF(slot, 1, 0);
            if (Gs == 0) switch (nt) {          // the downdates alone: the lean kernel (two workgroups per CU)
            case 1: hipLaunchKernelGGL((k_rank1_split<1, 1>), grid, dim3(64 * 1), lds_r1s, st, d, GrMax); break;
            case 2: hipLaunchKernelGGL((k_rank1_split<2, 1>), grid, dim3(64 * 3), lds_r1s, st, d, GrMax); break;
            case 3: hipLaunchKernelGGL((k_rank1_split<3, 1>), grid, dim3(64 * 6), lds_r1s, st, d, GrMax); break;
            case 4: hipLaunchKernelGGL((k_rank1_split<4, 1>), grid, dim3(64 * 10), lds_r1s, st, d, GrMax); break;
            case 5: hipLaunchKernelGGL((k_rank1_split<5, 1>), grid, dim3(64 * 15), lds_r1s, st, d, GrMax); break;
            case 6: hipLaunchKernelGGL((k_rank1_split<6, 2>), grid, dim3(64 * 11), lds_r1s, st, d, GrMax); break;
            case 7: hipLaunchKernelGGL((k_rank1_split<7, 2>), grid, dim3(64 * 14), lds_r1s, st, d, GrMax); break;
            default: hipLaunchKernelGGL((k_rank1_split<8, 3>), grid, dim3(64 * 12), lds_r1s, st, d, GrMax); break;
            }
            else switch (nt) {
            case 1: hipLaunchKernelGGL((k_schur_split<1, 1>), grid, dim3(64 * 1), lds_sp, st, d, Gs, GrMax); break;
            case 2: hipLaunchKernelGGL((k_schur_split<2, 1>), grid, dim3(64 * 3), lds_sp, st, d, Gs, GrMax); break;
            case 3: hipLaunchKernelGGL((k_schur_split<3, 1>), grid, dim3(64 * 6), lds_sp, st, d, Gs, GrMax); break;
            case 4: hipLaunchKernelGGL((k_schur_split<4, 1>), grid, dim3(64 * 10), lds_sp, st, d, Gs, GrMax); break;
            case 5: hipLaunchKernelGGL((k_schur_split<5, 1>), grid, dim3(64 * 15), lds_sp, st, d, Gs, GrMax); break;
            case 6: hipLaunchKernelGGL((k_schur_split<6, 2>), grid, dim3(64 * 11), lds_sp, st, d, Gs, GrMax); break;
            case 7: hipLaunchKernelGGL((k_schur_split<7, 2>), grid, dim3(64 * 14), lds_sp, st, d, Gs, GrMax); break;
            default: hipLaunchKernelGGL((k_schur_split<8, 3>), grid, dim3(64 * 12), lds_sp, st, d, Gs, GrMax); break;
            }
            PROF(slot, 1, 1);
            PROF(slot, 2, 0);
            hipLaunchKernelGGL(k_schur_fold, dim3(d.B, nt * (nt + 1) / 2), dim3(256), 0, st, d, GrMax, nt, Gs > 0 ? 1 : 0);      // one accumulator-tile entry per thread
            PROF(slot, 2, 1);
            counts[2]++; counts[7] = GrMax;
        } else if (d.lds_T) {
            // landmark elimination: Gram products of the pose Jacobians, then the rank-1 downdates (both FP64 MFMA)
            PROF(slot, 1, 0);
            if (!fused) {
                const size_t n_pairs = (size_t)d.N * (d.N - 1) / 2;
                hipLaunchKernelGGL(k_sweep_mfma, dim3(d.B), dim3(64 * ISV_SWEEP_WAVES), ((d.sw_global ? 0 : n_pairs * 84) + (n_pairs + 2) / 2 + 1) * sizeof(double), st, d);
            }
            counts[2]++;
            PROF(slot, 1, 1);
            const int nt = d.wd_ld / 16;
            // one workgroup per window: nt(nt+1)/2 tile wavefronts, w vectors expanded to panel rows in LDS
            const size_t lds_r1 = (64 * (d.wd_ld + 4) + (size_t)d.max_lm * 3 + 2) * sizeof(double);
            PROF(slot, 2, 0);
            if (!rank1_done) switch (nt) {
            case 1: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<1, 1, 64, 1, true>), dim3(d.B), dim3(64 * 1), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<1, 1, 64, 1, false>), dim3(d.B), dim3(64 * 1), lds_r1, st, d); break;
            case 2: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<2, 1, 64, 1, true>), dim3(d.B), dim3(64 * 3), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<2, 1, 64, 1, false>), dim3(d.B), dim3(64 * 3), lds_r1, st, d); break;
            case 3: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<3, 1, 64, 1, true>), dim3(d.B), dim3(64 * 6), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<3, 1, 64, 1, false>), dim3(d.B), dim3(64 * 6), lds_r1, st, d); break;
            case 4: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<4, 1, 64, 1, true>), dim3(d.B), dim3(64 * 10), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<4, 1, 64, 1, false>), dim3(d.B), dim3(64 * 10), lds_r1, st, d); break;
            case 5: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<5, 1, 64, 1, true>), dim3(d.B), dim3(64 * 15), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<5, 1, 64, 1, false>), dim3(d.B), dim3(64 * 15), lds_r1, st, d); break;
            case 6: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<6, 2, 64, 1, true>), dim3(d.B), dim3(64 * 11), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<6, 2, 64, 1, false>), dim3(d.B), dim3(64 * 11), lds_r1, st, d); break;
            case 7: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<7, 2, 64, 1, true>), dim3(d.B), dim3(64 * 14), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<7, 2, 64, 1, false>), dim3(d.B), dim3(64 * 14), lds_r1, st, d); break;
            default: if (d.est_ex) hipLaunchKernelGGL((k_rank1_mfma<8, 3, 64, 1, true>), dim3(d.B), dim3(64 * 12), lds_r1, st, d); else hipLaunchKernelGGL((k_rank1_mfma<8, 3, 64, 1, false>), dim3(d.B), dim3(64 * 12), lds_r1, st, d); break;
            }
            PROF(slot, 2, 1);
        }
        if (!cs || chain_on_side) HCHK(hipStreamWaitEvent(st, fj[1], 0));
        if (cs && !chain_done) {                          // (an upload k_lin_gram_chain does not take: the chain half alone; same bits)
            const size_t lds_sb = build_solve_sb_bytes(d.N, d.prior_H_sz);
            if (d.N == 11 && !generic_n) hipLaunchKernelGGL((k_build_solve_sb<false, 11, 1>), dim3(d.B), dim3(512), lds_sb, st, d);
            else if (d.N <= 11) hipLaunchKernelGGL((k_build_solve_sb<false, 0, 1>), dim3(d.B), dim3(512), lds_sb, st, d);
            else hipLaunchKernelGGL((k_build_solve_sb<true, 0, 1>), dim3(d.B), dim3(512), lds_sb, st, d);
        }
        if (!d.lds_T) hipLaunchKernelGGL(k_cost_reduce, dim3(d.B), dim3(256), 0, st, d, d.fcost, d.imu_cost, d.prior_cost, d.cost, 1);   // (k_build_solve_sb sums the cost itself)
        PROF(slot, 3, 0);
        // (the window length as a compile-time constant for the benchmark's 11 frames: isv_build_solve_sb.hip)
        if (d.lds_T && hc.solve_st) {
            counts[6] = 1;
            const size_t lds_st = build_solve_st_bytes(d.N, d.prior_H_sz);
            if (d.N == 11 && !generic_n) hipLaunchKernelGGL((k_build_solve_st<false, 11>), dim3(d.B), dim3(256), lds_st, st, d);
            else if (d.N <= 11) hipLaunchKernelGGL((k_build_solve_st<false, 0>), dim3(d.B), dim3(256), lds_st, st, d);
            // (the reference's ALL_BUF_SIZE as a compile-time constant: 404 -> 357 us per launch, 1024 windows 10.27 -> 9.87 ms.  The same for
            //  k_build_solve_sb<true, 18> spills 62 registers and is SLOWER for the single window it serves: 2.54 against 2.48 ms; not instantiated)
            else if (d.N == 18 && !generic_n) hipLaunchKernelGGL((k_build_solve_st<true, 18>), dim3(d.B), dim3(512), lds_st, st, d);     // (the reference's ALL_BUF_SIZE)
            else hipLaunchKernelGGL((k_build_solve_st<true, 0>), dim3(d.B), dim3(512), lds_st, st, d);       // (long windows: eight wavefronts, two windows per CU)
        } else if (d.lds_T && hc.chain_split && fuse_control && !bsub_split && !hc.no_pose_dogleg && d.N <= 11 &&      // (measured: 128 windows of 11 frames 1.92 -> 1.86 ms; the 18-frame
                   // form LOSES -- 117 us against 70 + 37 for the two launches, rocprofv3 -- and is not used)
                   (build_solve_sb_bytes(d.N, d.prior_H_sz) > lds_dgc ? build_solve_sb_bytes(d.N, d.prior_H_sz) : lds_dgc) + 2304 <= ISV_LDS_PER_CU) {       // (+ the 2 KB of static reduction space)
            // the pose half of the split solve + k_dogleg<true> (dogleg and step control) in one launch: same routines, same bits
            const size_t lds_sb = build_solve_sb_bytes(d.N, d.prior_H_sz), lds_pd = lds_sb > lds_dgc ? lds_sb : lds_dgc;
            if (d.est_ex) { if (d.N <= 11) hipLaunchKernelGGL((k_pose_dogleg<false, 0, true>), dim3(d.B), dim3(512), lds_pd, st, d); else hipLaunchKernelGGL((k_pose_dogleg<true, 0, true>), dim3(d.B), dim3(512), lds_pd, st, d); }
            else if (d.N == 11 && !generic_n) hipLaunchKernelGGL((k_pose_dogleg<false, 11, false>), dim3(d.B), dim3(512), lds_pd, st, d);
            else if (d.N <= 11) hipLaunchKernelGGL((k_pose_dogleg<false, 0, false>), dim3(d.B), dim3(512), lds_pd, st, d);
            else hipLaunchKernelGGL((k_pose_dogleg<true, 0, false>), dim3(d.B), dim3(512), lds_pd, st, d);
            dogleg_done = true; counts[5] = 1;
        } else if (d.lds_T && hc.chain_split) {           // the pose half of the split solve (the chain half ran beside k_lin_gram)
            const size_t lds_sb = build_solve_sb_bytes(d.N, d.prior_H_sz);
            if (d.N == 11 && !generic_n) hipLaunchKernelGGL((k_build_solve_sb<false, 11, 2>), dim3(d.B), dim3(512), lds_sb, st, d);
            else if (d.N <= 11) hipLaunchKernelGGL((k_build_solve_sb<false, 0, 2>), dim3(d.B), dim3(512), lds_sb, st, d);
            else hipLaunchKernelGGL((k_build_solve_sb<true, 0, 2>), dim3(d.B), dim3(512), lds_sb, st, d);
        } else if (d.lds_T && generic_n) {
            if (d.N <= 11) hipLaunchKernelGGL((k_build_solve_sb<false, 0, 0>), dim3(d.B), dim3(512), build_solve_sb_bytes(d.N, d.prior_H_sz), st, d);
            else hipLaunchKernelGGL((k_build_solve_sb<true, 0, 0>), dim3(d.B), dim3(512), build_solve_sb_bytes(d.N, d.prior_H_sz), st, d);
        } else if (d.lds_T && d.N == 11) hipLaunchKernelGGL((k_build_solve_sb<false, 11, 0>), dim3(d.B), dim3(512), build_solve_sb_bytes(d.N, d.prior_H_sz), st, d);
        else if (d.lds_T && d.N < 11) hipLaunchKernelGGL((k_build_solve_sb<false, 0, 0>), dim3(d.B), dim3(512), build_solve_sb_bytes(d.N, d.prior_H_sz), st, d);
        else if (d.lds_T) hipLaunchKernelGGL((k_build_solve_sb<true, 0, 0>), dim3(d.B), dim3(512), build_solve_sb_bytes(d.N, d.prior_H_sz), st, d);
        else hipLaunchKernelGGL(k_build_solve<false>, dim3(d.B), dim3(512), lds_bs, st, d);
        counts[1]++;
        PROF(slot, 3, 1);
        PROF(slot, 4, 0);
        if (bsub_split) hipLaunchKernelGGL(k_backsub_split, dim3(d.B, (d.lg_lcap + 127) / 128), dim3(128), 2 * (size_t)d.np * sizeof(double), st, d);
        if (dogleg_done) {
        } else if (fuse_control) {
            if (d.est_ex) hipLaunchKernelGGL((k_dogleg<true, true>), dim3(d.B), dim3(256), lds_dgc, st, d);
            else hipLaunchKernelGGL((k_dogleg<true, false>), dim3(d.B), dim3(256), lds_dgc, st, d);
            counts[5] = 1;
        } else if (d.est_ex) hipLaunchKernelGGL((k_dogleg<false, true>), dim3(d.B), dim3(256), lds_dg, st, d);
        else hipLaunchKernelGGL((k_dogleg<false, false>), dim3(d.B), dim3(256), lds_dg, st, d);
        PROF(slot, 4, 1);
        if (!d.lds_T) {                    // (the LDS path evaluates these inside k_dogleg)
            HCHK(hipEventRecord(fj[2], st)); HCHK(hipStreamWaitEvent(st2, fj[2], 0));
            if (NI) hipLaunchKernelGGL(k_imu_linearize<false>, dim3((unsigned)NI), dim3(64), 0, st2, d, d.cpose, d.csb, d.imu_cost_c, 2);
            hipLaunchKernelGGL(k_prior_linearize<false>, dim3(d.B), dim3(64), prior_lds_bytes(d.n_prior_slots, false), st2, d, d.cpose, d.csb, d.prior_cost_c, 2);
            hipLaunchKernelGGL(k_model_imu_prior, dim3(d.B * d.N), dim3(64), 0, st2, d);
            HCHK(hipEventRecord(fj[3], st2));
        }
        // candidate evaluation of the reprojection factors: inside the per-window control kernel, unless the windows are
        // so large that one workgroup per window would serialise it (config 5: 30 000 factors in one window)
        PROF(slot, 5, 0);
        if (fuse_control) {
        } else if (control_in_wg) {
            if (d.est_ex) hipLaunchKernelGGL((k_step_control<true, true>), dim3(d.B), dim3(256), lds_sc, st, d);
            else hipLaunchKernelGGL((k_step_control<true, false>), dim3(d.B), dim3(256), lds_sc, st, d);
        }
        else {
            if (d.n_tiles > 0) hipLaunchKernelGGL(k_proj_linearize<1>, dim3((d.n_tiles + 3) / 4), dim3(256), lds_proj1, st, d, d.cpose, d.clam, d.fcost_c, 2);
            if (!d.lds_T) HCHK(hipStreamWaitEvent(st, fj[3], 0));
            hipLaunchKernelGGL((k_step_control<false, false>), dim3(d.B), dim3(256), 0, st, d);
        }
        PROF(slot, 5, 1);
    }
    if (d.init_mode) {                     // Estimator::initFactorGraph: first priors from the solved estimate, then double2vector
        hipLaunchKernelGGL(k_init_priors, dim3(d.B), dim3(64), 0, st, d, d.init_scratch, d.init_per_window, d.init_kld);
        hipLaunchKernelGGL(k_finalize, dim3(d.B), dim3(64), 0, st, d, 0);
        HCHK(hipGetLastError());
        return ISV_OK;
    }
    hipLaunchKernelGGL(k_finalize, dim3(d.B), dim3(64), 0, st, d, hc.no_update ? 0 : 1);
    // MargForward and MargBackward are independent: run them side by side.  The longer one (backward, 270 us per window against
    // 110) goes on the main stream so that it is dispatched FIRST: four workgroups of each per CU do not fit the LDS together
    // (4 x 27 KB + 4 x 19 KB > 160 KB), and whichever kernel arrives second runs its last workgroups after the first one's.
    hipLaunchKernelGGL(k_marg_clear, dim3(d.B), dim3(64), 0, st, d);
    HCHK(hipEventRecord(fj[0], st)); HCHK(hipStreamWaitEvent(st2, fj[0], 0));
    // a batch that leaves SIMDs idle (one wavefront per window: B < 4 n_cus) is latency bound and takes the four-wavefront
    // eigen-decomposition (B = 1 / 64 / 256 / 512: 1.95 / 2.27 / 2.47 / 3.16 ms against 1.98 / 2.35 / 2.55 / 3.25); once
    // every SIMD holds a window the phase is instruction-issue bound and the two forms tie (B = 1024: 5.75 ms either way);
    // beyond that the one launch is kept
    const bool marg_one = hc.marg_one_kernel || (!hc.marg_split && hc.cap_batch > 3 * (size_t)hc.n_cus);     // (per handle: the two forms sum their convergence test in different orders)
    if (marg_one) hipLaunchKernelGGL(k_marg_bwd<2>, dim3(d.B), dim3(64), 0, st, d);
    else {
        hipLaunchKernelGGL(k_marg_bwd<0>, dim3(d.B), dim3(64), 0, st, d);
        hipLaunchKernelGGL(k_marg_jacobi<21>, dim3(d.B), dim3(256), 0, st, d);
        hipLaunchKernelGGL(k_marg_bwd<1>, dim3(d.B), dim3(64), 0, st, d);
    }
    // (four wavefronts for the landmark phases while the batch leaves SIMDs idle; same bits either way)
    if (d.B <= 2 * n_cus) hipLaunchKernelGGL(k_marg_fwd<256>, dim3(d.B), dim3(256), 0, st2, d);
    else hipLaunchKernelGGL(k_marg_fwd<64>, dim3(d.B), dim3(64), 0, st2, d);
    HCHK(hipEventRecord(fj[1], st2));
    HCHK(hipStreamWaitEvent(st, fj[1], 0));
    HCHK(hipGetLastError());
    return ISV_OK;
}

// staged: the caller has brought the records into g already (isv_batch_download's two-copy form) and synchronised
int isv_solver_download(DevBatch &d, hipStream_t st, int n, const SolverStage &g, isv_summary_t *summary, isv_marg_result_t *marg, std::string &err, bool staged) {
    if (!summary && !marg) return ISV_OK;
    const size_t nt = (size_t)n * ISV_MAX_TRACE;
    if (!staged) {
    if (summary) {
        HCHK(hipMemcpyAsync(g.st, d.st, sizeof(SolveState) * n, hipMemcpyDeviceToHost, st));
        HCHK(hipMemcpyAsync(g.tc, d.trace_cost, sizeof(double) * nt, hipMemcpyDeviceToHost, st));
        HCHK(hipMemcpyAsync(g.tr, d.trace_radius, sizeof(double) * nt, hipMemcpyDeviceToHost, st));
        HCHK(hipMemcpyAsync(g.ts, d.trace_step, sizeof(double) * nt, hipMemcpyDeviceToHost, st));
        HCHK(hipMemcpyAsync(g.ta, d.trace_acc, sizeof(int32_t) * nt, hipMemcpyDeviceToHost, st));
    }
    if (marg) HCHK(hipMemcpyAsync(g.marg, d.marg, sizeof(isv_marg_result_t) * n, hipMemcpyDeviceToHost, st));
    HCHK(hipStreamSynchronize(st));
    }
    for (int b = 0; b < n; b++) isv_solver_unpack_window(g, b, summary ? &summary[b] : nullptr, marg ? &marg[b] : nullptr);
    return ISV_OK;
}

// window b's records out of the staging area (isv_batch_download's host threads call it per window; isv_solver_download for all)
void isv_solver_unpack_window(const SolverStage &g, int b, isv_summary_t *summary, isv_marg_result_t *marg) {
    if (marg) *marg = g.marg[b];
    if (summary) {
        isv_summary_t &s = *summary;
        memset(&s, 0, sizeof(s));
        s.status = ISV_OK; s.termination = g.st[b].termination; s.iterations = g.st[b].iteration; s.num_successful = g.st[b].num_successful;
        s.initial_cost = g.st[b].initial_cost; s.final_cost = g.st[b].x_cost;
        memcpy(s.trace_cost, &g.tc[(size_t)b * ISV_MAX_TRACE], sizeof(s.trace_cost));
        memcpy(s.trace_radius, &g.tr[(size_t)b * ISV_MAX_TRACE], sizeof(s.trace_radius));
        memcpy(s.trace_step_norm, &g.ts[(size_t)b * ISV_MAX_TRACE], sizeof(s.trace_step_norm));
        memcpy(s.trace_accepted, &g.ta[(size_t)b * ISV_MAX_TRACE], sizeof(s.trace_accepted));
        if (!(s.final_cost - s.final_cost == 0.0)) s.status = ISV_ERR_NONFINITE;
    }
}

int isv_solver_debug_read(DevBatch &d, hipStream_t st, int what, double *out, int64_t count, std::string &err) {
    const double *src = nullptr;
    switch (what) {
    case 10: src = d.gn_p; break;
    case 11: src = d.gn_l; break;
    case 12: src = d.grad_p; break;
    case 13: src = d.grad_l; break;
    case 14: src = d.scale_p; break;
    case 15: src = d.scale_l; break;
    case 16: src = d.diag_p; break;
    case 17: src = d.delta_p; break;
    case 18: src = d.delta_l; break;
    case 19: src = d.cost_c; break;
    case 20: src = d.model; break;
    case 21: src = d.dbg; break;
    default: return ISV_ERR_INVALID_ARG;
    }
    HCHK(hipMemcpyAsync(out, src, sizeof(double) * count, hipMemcpyDeviceToHost, st));
    HCHK(hipStreamSynchronize(st));
    return ISV_OK;
}
