// isvins_estimator_shim.hpp -- reference-side binding of the C ABI (include/isvins_backend.h).
//
// Drop-in replacement bodies for
//     void Estimator::backendOptimization()   lyeemax/IS-VINS src/estimator.cpp:1541-1562  (NON_LINEAR branch)
//     void Estimator::initFactorGraph()       src/estimator.cpp:667-1001                    (INITIAL_STRUCTURE branch)
// A maintainer includes this header at the end of src/estimator.cpp (after renaming the original members to
// backendOptimizationCeres / initFactorGraphCeres), adds ONE member `isv_backend_t *isv_handle` to class Estimator and
// links libisvins_hip.so; System, FeatureTracker, FeatureManager, slideWindow() and PoseGraphBuilder are untouched.
// INTEGRATION.md lists every Estimator member this file reads and writes, with the reference's file:line.
//
// This file needs the reference's own headers (Eigen, estimator.h, the factor classes), which are absent from this
// repository's build image; it is TYPE-CHECKED here against declarations transcribed from those headers and a minimal
// Eigen-API stub (tests/test_shim_typecheck.py, `g++ -fsyntax-only`: type-check only, pins nothing).
// Everything it does is data marshalling: Eigen (column-major) <-> the ABI's row-major PODs.
#pragma once
#include <iostream>
#include <vector>
#include "isvins_backend.h"

namespace isvins {

inline void to_row_major(const Eigen::Matrix3d &M, double *o) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o[r * 3 + c] = M(r, c); }
inline Eigen::Matrix3d from_row_major3(const double *o) { Eigen::Matrix3d M; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M(r, c) = o[r * 3 + c]; return M; }
inline void to_row_major(const Eigen::MatrixXd &M, double *o) { for (int r = 0; r < M.rows(); r++) for (int c = 0; c < M.cols(); c++) o[r * M.cols() + c] = M(r, c); }
inline Eigen::MatrixXd from_row_major(const double *o, int rows, int cols) { Eigen::MatrixXd M(rows, cols); for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) M(r, c) = o[r * cols + c]; return M; }

// One handle per Estimator, created once (Estimator::setParameter time, src/estimator.cpp:21-38).
inline isv_backend_t *create_backend() {
    isv_config_t cfg{};
    cfg.n_frames = ALL_BUF_SIZE; cfg.n_vo = Vo_SIZE; cfg.max_landmarks = NUM_OF_F;
    cfg.max_obs = NUM_OF_F * ALL_BUF_SIZE; cfg.max_rollpitch = Vo_SIZE + 1; cfg.max_batch = 1;
    cfg.num_iterations = NUM_ITERATIONS; cfg.estimate_extrinsic = ESTIMATE_EXTRINSIC;
    cfg.proj_sqrt_info[0] = ProjectionFactor::sqrt_info(0, 0); cfg.proj_sqrt_info[1] = ProjectionFactor::sqrt_info(0, 1);
    cfg.proj_sqrt_info[2] = ProjectionFactor::sqrt_info(1, 0); cfg.proj_sqrt_info[3] = ProjectionFactor::sqrt_info(1, 1);
    cfg.gravity[0] = G.x(); cfg.gravity[1] = G.y(); cfg.gravity[2] = G.z();
    cfg.alpha = ALPHA; cfg.init_depth = INIT_DEPTH;
    isv_backend_t *h = nullptr;
    if (isv_backend_create(&cfg, &h) != ISV_OK) return nullptr;
    return h;
}

// The buffers one isv_window_t points into, filled from the Estimator members (the ONE marshalling routine both
// replacement members share).
struct WindowPack {
    std::vector<double> Ps, Rs, Vs, Bas, Bgs, tic, ric, obs, depth;
    std::vector<int32_t> start, ptr, flag;
    std::vector<IDFeatures *> good;           // the landmarks of the window, in IDsfeatures order (feature_manager.cpp:27-31)
    std::vector<isv_imu_t> imu;
    isv_se3_prior_t pp; isv_linear9_t vb;
    std::vector<isv_relpose_t> rel; std::vector<isv_rollpitch_t> rp;
    isv_window_t w;
};

// Estimator -> window.  with_priors = false: initFactorGraph (the prior factors do not exist yet; they are outputs).
inline void pack_window(Estimator &e, bool with_priors, WindowPack &k) {
    const int N = ALL_BUF_SIZE;
    k.Ps.resize(N * 3); k.Rs.resize(N * 9); k.Vs.resize(N * 3); k.Bas.resize(N * 3); k.Bgs.resize(N * 3); k.tic.resize(3); k.ric.resize(9);
    for (int i = 0; i < N; i++) {
        for (int c = 0; c < 3; c++) { k.Ps[3 * i + c] = e.Ps[i](c); k.Vs[3 * i + c] = e.Vs[i](c); k.Bas[3 * i + c] = e.Bas[i](c); k.Bgs[3 * i + c] = e.Bgs[i](c); }
        to_row_major(e.Rs[i], &k.Rs[9 * i]);
    }
    for (int c = 0; c < 3; c++) k.tic[c] = e.tic[0](c);
    to_row_major(e.ric[0], k.ric.data());
    // FeatureManager view in IDsfeatures order, goodFeature() only: the traversal of problemSolve (src/estimator.cpp:1057-1063)
    k.start.clear(); k.ptr.assign(1, 0); k.obs.clear(); k.depth.clear(); k.good.clear();
    for (auto &f : e.f_manager.IDsfeatures) {
        f.used_num = (int)f.idfeatures.size();
        if (!e.f_manager.goodFeature(f)) continue;
        k.good.push_back(&f);
        k.start.push_back(f.start_frame);
        for (auto &o : f.idfeatures) { k.obs.push_back(o.point.x()); k.obs.push_back(o.point.y()); k.obs.push_back(o.point.z()); }
        k.ptr.push_back((int32_t)(k.obs.size() / 3));
        k.depth.push_back(f.estimated_depth);
    }
    k.flag.assign(k.good.size(), 0);
    // pre_integrations[1..N-1] (include/factor/integration_base.h:188-207)
    k.imu.resize(N - 1);
    for (int j = 1; j < N; j++) {
        const IntegrationBase &p = *e.pre_integrations[j]; isv_imu_t &o = k.imu[j - 1];
        for (int c = 0; c < 3; c++) { o.delta_p[c] = p.delta_p(c); o.delta_v[c] = p.delta_v(c); o.linearized_ba[c] = p.linearized_ba(c); o.linearized_bg[c] = p.linearized_bg(c); }
        o.delta_q[0] = p.delta_q.x(); o.delta_q[1] = p.delta_q.y(); o.delta_q[2] = p.delta_q.z(); o.delta_q[3] = p.delta_q.w();
        o.sum_dt = p.sum_dt;
        for (int r = 0; r < 15; r++) for (int c = 0; c < 15; c++) { o.jacobian[r * 15 + c] = p.jacobian(r, c); o.covariance[r * 15 + c] = p.covariance(r, c); }
    }
    // prior factors (include/estimator.h:134-138)
    k.pp = isv_se3_prior_t(); k.vb = isv_linear9_t();
    k.rel.assign(Vo_SIZE - 1, isv_relpose_t()); k.rp.clear();
    if (with_priors) {
        for (int c = 0; c < 3; c++) k.pp.t[c] = e.vioPosePriorEdge->t(c);
        to_row_major(e.vioPosePriorEdge->R, k.pp.R); to_row_major(e.vioPosePriorEdge->sqrt_info, k.pp.sqrt_info); k.pp.index = 0;
        for (int c = 0; c < 9; c++) k.vb.VB[c] = e.vioVBPrior->VB(c);
        to_row_major(e.vioVBPrior->sqrt_info, k.vb.sqrt_info); k.vb.index = Vo_SIZE - 1;
        for (int i = 0; i < Vo_SIZE - 1; i++) {
            RelativePoseFactor *f = e.vioRelativePoseEdges[i + 1];
            for (int c = 0; c < 3; c++) k.rel[i].delta_t[c] = f->delta_t(c);
            to_row_major(f->delta_R, k.rel[i].delta_R); to_row_major(f->sqrt_info, k.rel[i].sqrt_info); k.rel[i].imu_i = i; k.rel[i].imu_j = i + 1;
        }
        k.rp.resize(e.vioRollPitchEdges.size());
        for (size_t i = 0; i < k.rp.size(); i++) { RollPitchFactor *f = e.vioRollPitchEdges[i]; to_row_major(f->R, k.rp[i].R); to_row_major(f->sqrt_info, k.rp[i].sqrt_info); k.rp[i].index = f->index; }
    }
    isv_window_t &w = k.w;
    w = isv_window_t();
    w.Ps = k.Ps.data(); w.Rs = k.Rs.data(); w.Vs = k.Vs.data(); w.Bas = k.Bas.data(); w.Bgs = k.Bgs.data(); w.tic = k.tic.data(); w.ric = k.ric.data();
    w.n_landmarks = (int32_t)k.good.size(); w.n_obs = k.ptr.back();
    w.lm_start_frame = k.start.data(); w.lm_obs_ptr = k.ptr.data(); w.obs_point = k.obs.data(); w.lm_depth = k.depth.data(); w.lm_solve_flag = k.flag.data();
    w.imu = k.imu.data(); w.pose_prior = &k.pp; w.vb_prior = &k.vb; w.relpose = k.rel.data();
    w.rollpitch = k.rp.empty() ? nullptr : k.rp.data(); w.n_rollpitch = (int32_t)k.rp.size();
    w.margin_old = with_priors && (e.marginalization_flag == Estimator::MARGIN_OLD); w.header0 = e.Headers[0];
    w.para_Pose = &e.para_Pose[0][0]; w.para_SpeedBias = &e.para_SpeedBias[0][0]; w.para_Ex_Pose = &e.para_Ex_Pose[0][0]; w.para_Feature = &e.para_Feature[0][0];
}

// window -> Estimator: double2vector's outputs (src/estimator.cpp:518-594) incl. FeatureManager::setDepth
inline void unpack_states(Estimator &e, const WindowPack &k) {
    const int N = ALL_BUF_SIZE;
    for (int i = 0; i < N; i++) {
        for (int c = 0; c < 3; c++) { e.Ps[i](c) = k.Ps[3 * i + c]; e.Vs[i](c) = k.Vs[3 * i + c]; e.Bas[i](c) = k.Bas[3 * i + c]; e.Bgs[i](c) = k.Bgs[3 * i + c]; }
        e.Rs[i] = from_row_major3(&k.Rs[9 * i]);
    }
    for (int c = 0; c < 3; c++) e.tic[0](c) = k.tic[c];
    e.ric[0] = from_row_major3(k.ric.data());
    for (size_t l = 0; l < k.good.size(); l++) { k.good[l]->estimated_depth = k.depth[l]; k.good[l]->solve_flag = k.flag[l]; }
}

}  // namespace isvins

// ---- Estimator::backendOptimization(), NON_LINEAR branch -------------------------------------------------------------
inline void Estimator_backendOptimization_isv(Estimator &e) {
    using namespace isvins;
    WindowPack k;
    pack_window(e, true, k);
    isv_window_t &w = k.w;
    isv_summary_t sum; isv_marg_result_t mg;
#ifdef ISVINS_DEVICE_TRIANGULATE
    // FeatureManager::triangulate on the device as well (feature_manager.cpp:206-258): landmarks without a depth get the
    // DLT depth over their views, then the solve, with ONE hand-over (isv_backend_solve_odometry_batch).  Define this and
    // drop the f_manager.triangulate(Ps, tic, ric) call in Estimator::solveOdometry() (src/estimator.cpp:466); setDepth()
    // below writes the depths back like the solve's.
    { isv_window_t *wp = &w; if (isv_backend_solve_odometry_batch(e.isv_handle, 1, &wp, &sum, &mg) != ISV_OK) { std::cerr << "isv_backend_solve_odometry_batch: " << isv_backend_last_error(e.isv_handle) << std::endl; return; } }
#else
    if (isv_backend_optimize(e.isv_handle, &w, &sum, &mg) != ISV_OK) { std::cerr << "isv_backend_optimize: " << isv_backend_last_error(e.isv_handle) << std::endl; return; }
#endif
    unpack_states(e, k);
    // the shifted / rotated prior measurements (update() :1133-1144, double2vector :549-550)
    for (int c = 0; c < 3; c++) e.vioPosePriorEdge->t(c) = k.pp.t[c];
    e.vioPosePriorEdge->R = from_row_major3(k.pp.R);
    for (int c = 0; c < 9; c++) e.vioVBPrior->VB(c) = k.vb.VB[c];
    for (int i = 0; i < Vo_SIZE - 1; i++) { RelativePoseFactor *f = e.vioRelativePoseEdges[i + 1]; for (int c = 0; c < 3; c++) f->delta_t(c) = k.rel[i].delta_t[c]; f->delta_R = from_row_major3(k.rel[i].delta_R); }
    for (size_t i = 0; i < k.rp.size(); i++) e.vioRollPitchEdges[i]->R = from_row_major3(k.rp[i].R);
    if (w.margin_old && mg.valid) {
        // MargForward outputs (src/estimator.cpp:1243-1283, 1349-1351)
        RelativePoseFactor *pg = new RelativePoseFactor(Eigen::Vector3d(mg.combined.relative_pose.delta_t), from_row_major3(mg.combined.relative_pose.delta_R));
        pg->sqrt_info = from_row_major(mg.combined.relative_pose.sqrt_info, 6, 6);
        CombinedFactors *cmb = new CombinedFactors();
        delete cmb->relativePoseFactor; cmb->relativePoseFactor = pg;       // (the reference leaks the constructor's placeholder, :1264)
        if (!e.vioRollPitchEdges.empty()) {
            if (mg.combined.has_rollpitch) { cmb->rollPitchFactor = e.vioRollPitchEdges[0]; cmb->covAbs = from_row_major(mg.combined.covAbs, 2, 2); }
            else cmb->rollPitchFactor = nullptr;
        }
        // PoseGraphFactorCount is a file-scope static of include/estimator.h:28, not a member
        cmb->vio_index = PoseGraphFactorCount; cmb->distance = mg.combined.distance; cmb->covRel = from_row_major(mg.combined.covRel, 6, 6);
        cmb->ts = mg.combined.ts; cmb->Ri = from_row_major3(mg.combined.Ri); cmb->ti = Eigen::Vector3d(mg.combined.ti);
        PoseGraphFactorCount++;
        e.m_pose_graph_buf.lock(); e.pose_graph_factors_buf.push(cmb); e.m_pose_graph_buf.unlock();
        SE3PriorFactor *se3 = new SE3PriorFactor(Eigen::Vector3d(mg.forward_pose_prior.t), Eigen::Quaterniond(from_row_major3(mg.forward_pose_prior.R)));
        se3->sqrt_info = from_row_major(mg.forward_pose_prior.sqrt_info, 6, 6);
        e.forwardPosePriorEdgeToAdd = se3;
        // MargBackward outputs (src/estimator.cpp:1536-1538)
        RelativePoseFactor *brp = new RelativePoseFactor(Eigen::Vector3d(mg.backward_relpose.delta_t), from_row_major3(mg.backward_relpose.delta_R));
        brp->sqrt_info = from_row_major(mg.backward_relpose.sqrt_info, 6, 6);
        Eigen::Matrix<double, 9, 1> vbv; for (int c = 0; c < 9; c++) vbv(c) = mg.backward_vb.VB[c];
        Linear9Factor *bvb = new Linear9Factor(vbv); bvb->sqrt_info = from_row_major(mg.backward_vb.sqrt_info, 9, 9);
        RollPitchFactor *brl = new RollPitchFactor(Eigen::Quaterniond(from_row_major3(mg.backward_rollpitch.R)));
        brl->sqrt_info = from_row_major(mg.backward_rollpitch.sqrt_info, 2, 2); brl->setIndex(Vo_SIZE - 1);
        e.vioRollPitchEdges.push_back(brl); e.backwardVBEdgeToAdd = bvb; e.backwardRelativePoseEdgeToAdd = brp;
    }
    e.MargPointIdx.clear(); e.features2Marg.clear();
}

// ---- Estimator::initFactorGraph() (src/estimator.cpp:667-1001), the INITIAL_STRUCTURE branch -------------------------
// The backend solves the prior-free window, derives the first prior factors and runs double2vector; the reference's
// factor objects are created here from the returned PODs.
inline void Estimator_initFactorGraph_isv(Estimator &e) {
    using namespace isvins;
    WindowPack k;
    pack_window(e, false, k);
    isv_summary_t sum; double kld = 0;
    if (isv_backend_init_factor_graph(e.isv_handle, &k.w, &sum, &kld) != ISV_OK) { std::cerr << "isv_backend_init_factor_graph: " << isv_backend_last_error(e.isv_handle) << std::endl; return; }
    unpack_states(e, k);
    // the first prior factors (src/estimator.cpp:821-864, 944-974)
    if (e.vioRelativePoseEdges.size() < (size_t)Vo_SIZE) e.vioRelativePoseEdges.resize(Vo_SIZE, nullptr);
    e.vioRelativePoseEdges[0] = nullptr;
    for (int i = 0; i < Vo_SIZE - 1; i++) {
        RelativePoseFactor *f = new RelativePoseFactor(Eigen::Vector3d(k.rel[i].delta_t), from_row_major3(k.rel[i].delta_R));
        f->setIndex(i, i + 1); f->sqrt_info = from_row_major(k.rel[i].sqrt_info, 6, 6);
        e.vioRelativePoseEdges[i + 1] = f;
    }
    SE3PriorFactor *se3 = new SE3PriorFactor(Eigen::Vector3d(k.pp.t), Eigen::Quaterniond(from_row_major3(k.pp.R)));
    se3->setIndex(0); se3->sqrt_info = from_row_major(k.pp.sqrt_info, 6, 6); e.vioPosePriorEdge = se3;
    Eigen::Matrix<double, 9, 1> vbv; for (int c = 0; c < 9; c++) vbv(c) = k.vb.VB[c];
    Linear9Factor *l9 = new Linear9Factor(vbv); l9->setIndex(Vo_SIZE - 1); l9->sqrt_info = from_row_major(k.vb.sqrt_info, 9, 9); e.vioVBPrior = l9;
}
