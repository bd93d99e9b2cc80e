// isvins_estimator_shim.hpp -- reference-side binding of the C ABI (include/isvins_backend.h).
//
// Drop-in replacement body for  void Estimator::backendOptimization()
// (lyeemax/IS-VINS src/estimator.cpp:1541-1562, NON_LINEAR branch).  A maintainer includes this
// header at the end of src/estimator.cpp (after renaming the original member to
// backendOptimizationCeres) and links libisvins_hip.so; System, FeatureTracker, FeatureManager,
// slideWindow() and PoseGraphBuilder are untouched.  See INTEGRATION.md.
//
// This file needs the reference's own headers (Eigen, estimator.h, the factor classes); it is NOT
// compiled in this repository (those dependencies are absent from the build image).  Everything it
// does is data marshalling: Eigen (column-major) <-> the ABI's row-major PODs.
#pragma once
#include <vector>
#include "isvins_backend.h"

namespace isvins {

inline void to_row_major(const Eigen::Matrix3d &M, double *o) { for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o[r * 3 + c] = M(r, c); }
inline Eigen::Matrix3d from_row_major3(const double *o) { Eigen::Matrix3d M; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M(r, c) = o[r * 3 + c]; return M; }
inline void to_row_major(const Eigen::MatrixXd &M, double *o) { for (int r = 0; r < M.rows(); r++) for (int c = 0; c < M.cols(); c++) o[r * M.cols() + c] = M(r, c); }
inline Eigen::MatrixXd from_row_major(const double *o, int rows, int cols) { Eigen::MatrixXd M(rows, cols); for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) M(r, c) = o[r * cols + c]; return M; }

// One handle per Estimator, created once (Estimator::setParameter time).
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

}  // namespace isvins

// ---- the replacement member --------------------------------------------------------------------
// Requires a member  isv_backend_t *isv_handle  in class Estimator (include/estimator.h), set from
// isvins::create_backend() in setParameter().  INITIAL_STRUCTURE (initFactorGraph) stays on the
// reference's Ceres path.
inline void Estimator_backendOptimization_isv(Estimator &e) {
    using namespace isvins;
    const int N = ALL_BUF_SIZE;
    // Eigen-level state
    std::vector<double> Ps(N * 3), Rs(N * 9), Vs(N * 3), Bas(N * 3), Bgs(N * 3), tic(3), ric(9);
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 3; k++) { Ps[3 * i + k] = e.Ps[i](k); Vs[3 * i + k] = e.Vs[i](k); Bas[3 * i + k] = e.Bas[i](k); Bgs[3 * i + k] = e.Bgs[i](k); }
        to_row_major(e.Rs[i], &Rs[9 * i]);
    }
    for (int k = 0; k < 3; k++) tic[k] = e.tic[0](k);
    to_row_major(e.ric[0], ric.data());
    // FeatureManager view in IDsfeatures order, goodFeature() only (feature_manager.cpp:27-31)
    std::vector<int32_t> start, ptr(1, 0), flag;
    std::vector<double> obs, depth;
    std::vector<IDFeatures *> good;
    for (auto &f : e.f_manager.IDsfeatures) {
        f.used_num = f.idfeatures.size();
        if (!e.f_manager.goodFeature(f)) continue;
        good.push_back(&f);
        start.push_back(f.start_frame);
        for (auto &o : f.idfeatures) { obs.push_back(o.point.x()); obs.push_back(o.point.y()); obs.push_back(o.point.z()); }
        ptr.push_back((int32_t)(obs.size() / 3));
        depth.push_back(f.estimated_depth);
    }
    flag.assign(good.size(), 0);
    // pre-integrations 1..N-1
    std::vector<isv_imu_t> imu(N - 1);
    for (int j = 1; j < N; j++) {
        const IntegrationBase &p = *e.pre_integrations[j]; isv_imu_t &o = imu[j - 1];
        for (int k = 0; k < 3; k++) { o.delta_p[k] = p.delta_p(k); o.delta_v[k] = p.delta_v(k); o.linearized_ba[k] = p.linearized_ba(k); o.linearized_bg[k] = p.linearized_bg(k); }
        o.delta_q[0] = p.delta_q.x(); o.delta_q[1] = p.delta_q.y(); o.delta_q[2] = p.delta_q.z(); o.delta_q[3] = p.delta_q.w();
        o.sum_dt = p.sum_dt;
        for (int r = 0; r < 15; r++) for (int c = 0; c < 15; c++) { o.jacobian[r * 15 + c] = p.jacobian(r, c); o.covariance[r * 15 + c] = p.covariance(r, c); }
    }
    // prior factors
    isv_se3_prior_t pp{}; isv_linear9_t vb{};
    std::vector<isv_relpose_t> rel(Vo_SIZE - 1); std::vector<isv_rollpitch_t> rp(e.vioRollPitchEdges.size());
    for (int k = 0; k < 3; k++) pp.t[k] = e.vioPosePriorEdge->t(k);
    to_row_major(e.vioPosePriorEdge->R, pp.R); to_row_major(e.vioPosePriorEdge->sqrt_info, pp.sqrt_info); pp.index = 0;
    for (int k = 0; k < 9; k++) vb.VB[k] = e.vioVBPrior->VB(k);
    to_row_major(e.vioVBPrior->sqrt_info, vb.sqrt_info); vb.index = Vo_SIZE - 1;
    for (int i = 0; i < Vo_SIZE - 1; i++) {
        auto *f = e.vioRelativePoseEdges[i + 1];
        for (int k = 0; k < 3; k++) rel[i].delta_t[k] = f->delta_t(k);
        to_row_major(f->delta_R, rel[i].delta_R); to_row_major(f->sqrt_info, rel[i].sqrt_info); rel[i].imu_i = i; rel[i].imu_j = i + 1;
    }
    for (size_t i = 0; i < rp.size(); i++) { auto *f = e.vioRollPitchEdges[i]; to_row_major(f->R, rp[i].R); to_row_major(f->sqrt_info, rp[i].sqrt_info); rp[i].index = f->index; }

    isv_window_t w{};
    w.Ps = Ps.data(); w.Rs = Rs.data(); w.Vs = Vs.data(); w.Bas = Bas.data(); w.Bgs = Bgs.data(); w.tic = tic.data(); w.ric = ric.data();
    w.n_landmarks = (int32_t)good.size(); w.n_obs = ptr.back();
    w.lm_start_frame = start.data(); w.lm_obs_ptr = ptr.data(); w.obs_point = obs.data(); w.lm_depth = depth.data(); w.lm_solve_flag = flag.data();
    w.imu = imu.data(); w.pose_prior = &pp; w.vb_prior = &vb; w.relpose = rel.data(); w.rollpitch = rp.data(); w.n_rollpitch = (int32_t)rp.size();
    w.margin_old = (e.marginalization_flag == Estimator::MARGIN_OLD); w.header0 = e.Headers[0];
    w.para_Pose = &e.para_Pose[0][0]; w.para_SpeedBias = &e.para_SpeedBias[0][0]; w.para_Ex_Pose = &e.para_Ex_Pose[0][0]; w.para_Feature = &e.para_Feature[0][0];

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

    // write back (double2vector's outputs and the shifted / rotated priors)
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 3; k++) { e.Ps[i](k) = Ps[3 * i + k]; e.Vs[i](k) = Vs[3 * i + k]; e.Bas[i](k) = Bas[3 * i + k]; e.Bgs[i](k) = Bgs[3 * i + k]; }
        e.Rs[i] = from_row_major3(&Rs[9 * i]);
    }
    for (int k = 0; k < 3; k++) e.tic[0](k) = tic[k];
    e.ric[0] = from_row_major3(ric.data());
    for (size_t l = 0; l < good.size(); l++) { good[l]->estimated_depth = depth[l]; good[l]->solve_flag = flag[l]; }
    for (int k = 0; k < 3; k++) e.vioPosePriorEdge->t(k) = pp.t[k];
    e.vioPosePriorEdge->R = from_row_major3(pp.R);
    for (int k = 0; k < 9; k++) e.vioVBPrior->VB(k) = vb.VB[k];
    for (int i = 0; i < Vo_SIZE - 1; i++) { auto *f = e.vioRelativePoseEdges[i + 1]; for (int k = 0; k < 3; k++) f->delta_t(k) = rel[i].delta_t[k]; f->delta_R = from_row_major3(rel[i].delta_R); }
    for (size_t i = 0; i < rp.size(); i++) e.vioRollPitchEdges[i]->R = from_row_major3(rp[i].R);
    if (w.margin_old && mg.valid) {
        // MargForward outputs (estimator.cpp:1243-1283, 1349-1351)
        auto *pg = new RelativePoseFactor(Eigen::Vector3d(mg.combined.relative_pose.delta_t), from_row_major3(mg.combined.relative_pose.delta_R));
        pg->sqrt_info = from_row_major(mg.combined.relative_pose.sqrt_info, 6, 6);
        CombinedFactors *cmb = new CombinedFactors();
        delete cmb->relativePoseFactor; cmb->relativePoseFactor = pg;
        if (!e.vioRollPitchEdges.empty()) {
            if (mg.combined.has_rollpitch) { cmb->rollPitchFactor = e.vioRollPitchEdges[0]; cmb->covAbs = from_row_major(mg.combined.covAbs, 2, 2); }
            else cmb->rollPitchFactor = nullptr;
        }
        cmb->vio_index = e.PoseGraphFactorCount++; cmb->distance = mg.combined.distance; cmb->covRel = from_row_major(mg.combined.covRel, 6, 6);
        cmb->ts = mg.combined.ts; cmb->Ri = from_row_major3(mg.combined.Ri); cmb->ti = Eigen::Vector3d(mg.combined.ti);
        e.m_pose_graph_buf.lock(); e.pose_graph_factors_buf.push(cmb); e.m_pose_graph_buf.unlock();
        auto *se3 = new SE3PriorFactor(Eigen::Vector3d(mg.forward_pose_prior.t), Eigen::Quaterniond(from_row_major3(mg.forward_pose_prior.R)));
        se3->sqrt_info = from_row_major(mg.forward_pose_prior.sqrt_info, 6, 6);
        e.forwardPosePriorEdgeToAdd = se3;
        // MargBackward outputs (estimator.cpp:1536-1538)
        auto *brp = new RelativePoseFactor(Eigen::Vector3d(mg.backward_relpose.delta_t), from_row_major3(mg.backward_relpose.delta_R));
        brp->sqrt_info = from_row_major(mg.backward_relpose.sqrt_info, 6, 6);
        Eigen::Matrix<double, 9, 1> vbv; for (int k = 0; k < 9; k++) vbv(k) = mg.backward_vb.VB[k];
        auto *bvb = new Linear9Factor(vbv); bvb->sqrt_info = from_row_major(mg.backward_vb.sqrt_info, 9, 9);
        auto *brl = new RollPitchFactor(Eigen::Quaterniond(from_row_major3(mg.backward_rollpitch.R)));
        brl->sqrt_info = from_row_major(mg.backward_rollpitch.sqrt_info, 2, 2); brl->setIndex(Vo_SIZE - 1);
        e.vioRollPitchEdges.push_back(brl); e.backwardVBEdgeToAdd = bvb; e.backwardRelativePoseEdgeToAdd = brp;
    }
    e.MargPointIdx.clear(); e.features2Marg.clear();
}

// ---- Estimator::initFactorGraph() (src/estimator.cpp:667-1001), the INITIAL_STRUCTURE branch -------------------
// Same marshalling as above without prior factors; the backend solves the prior-free window, derives the first
// prior factors and runs double2vector.  The reference's factor objects are created here from the returned PODs.
inline void Estimator_initFactorGraph_isv(Estimator &e) {
    using namespace isvins;
    const int N = ALL_BUF_SIZE;
    std::vector<double> Ps(N * 3), Rs(N * 9), Vs(N * 3), Bas(N * 3), Bgs(N * 3), tic(3), ric(9);
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 3; k++) { Ps[3 * i + k] = e.Ps[i](k); Vs[3 * i + k] = e.Vs[i](k); Bas[3 * i + k] = e.Bas[i](k); Bgs[3 * i + k] = e.Bgs[i](k); }
        to_row_major(e.Rs[i], &Rs[9 * i]);
    }
    for (int k = 0; k < 3; k++) tic[k] = e.tic[0](k);
    to_row_major(e.ric[0], ric.data());
    std::vector<int32_t> start, ptr(1, 0), flag;
    std::vector<double> obs, depth;
    std::vector<IDFeatures *> good;
    for (auto &f : e.f_manager.IDsfeatures) {
        f.used_num = f.idfeatures.size();
        if (!e.f_manager.goodFeature(f)) continue;
        good.push_back(&f); start.push_back(f.start_frame);
        for (auto &o : f.idfeatures) { obs.push_back(o.point.x()); obs.push_back(o.point.y()); obs.push_back(o.point.z()); }
        ptr.push_back((int32_t)(obs.size() / 3)); depth.push_back(f.estimated_depth);
    }
    flag.assign(good.size(), 0);
    std::vector<isv_imu_t> imu(N - 1);
    for (int j = 1; j < N; j++) {
        const IntegrationBase &p = *e.pre_integrations[j]; isv_imu_t &o = imu[j - 1];
        for (int k = 0; k < 3; k++) { o.delta_p[k] = p.delta_p(k); o.delta_v[k] = p.delta_v(k); o.linearized_ba[k] = p.linearized_ba(k); o.linearized_bg[k] = p.linearized_bg(k); }
        o.delta_q[0] = p.delta_q.x(); o.delta_q[1] = p.delta_q.y(); o.delta_q[2] = p.delta_q.z(); o.delta_q[3] = p.delta_q.w();
        o.sum_dt = p.sum_dt; to_row_major(Eigen::MatrixXd(p.jacobian), o.jacobian); to_row_major(Eigen::MatrixXd(p.covariance), o.covariance);
    }
    isv_se3_prior_t pp{}; isv_linear9_t vb{}; std::vector<isv_relpose_t> rel(Vo_SIZE - 1);
    isv_window_t w{};
    w.Ps = Ps.data(); w.Rs = Rs.data(); w.Vs = Vs.data(); w.Bas = Bas.data(); w.Bgs = Bgs.data(); w.tic = tic.data(); w.ric = ric.data();
    w.n_landmarks = (int32_t)good.size(); w.n_obs = ptr.back();
    w.lm_start_frame = start.data(); w.lm_obs_ptr = ptr.data(); w.obs_point = obs.data(); w.lm_depth = depth.data(); w.lm_solve_flag = flag.data();
    w.imu = imu.data(); w.pose_prior = &pp; w.vb_prior = &vb; w.relpose = rel.data(); w.rollpitch = nullptr; w.n_rollpitch = 0;
    w.header0 = e.Headers[0];
    w.para_Pose = &e.para_Pose[0][0]; w.para_SpeedBias = &e.para_SpeedBias[0][0]; w.para_Ex_Pose = &e.para_Ex_Pose[0][0]; w.para_Feature = &e.para_Feature[0][0];
    isv_summary_t sum; double kld = 0;
    if (isv_backend_init_factor_graph(e.isv_handle, &w, &sum, &kld) != ISV_OK) { std::cerr << "isv_backend_init_factor_graph: " << isv_backend_last_error(e.isv_handle) << std::endl; return; }
    for (int i = 0; i < N; i++) {
        for (int k = 0; k < 3; k++) { e.Ps[i](k) = Ps[3 * i + k]; e.Vs[i](k) = Vs[3 * i + k]; e.Bas[i](k) = Bas[3 * i + k]; e.Bgs[i](k) = Bgs[3 * i + k]; }
        e.Rs[i] = from_row_major3(&Rs[9 * i]);
    }
    for (int k = 0; k < 3; k++) e.tic[0](k) = tic[k];
    e.ric[0] = from_row_major3(ric.data());
    for (size_t l = 0; l < good.size(); l++) { good[l]->estimated_depth = depth[l]; good[l]->solve_flag = flag[l]; }
    // the first prior factors (src/estimator.cpp:821-864, 944-974)
    e.vioRelativePoseEdges[0] = nullptr;
    for (int i = 0; i < Vo_SIZE - 1; i++) {
        auto *f = new RelativePoseFactor(Eigen::Vector3d(rel[i].delta_t), from_row_major3(rel[i].delta_R));
        f->setIndex(i, i + 1); f->sqrt_info = from_row_major(rel[i].sqrt_info, 6, 6);
        e.vioRelativePoseEdges[i + 1] = f;
    }
    auto *se3 = new SE3PriorFactor(Eigen::Vector3d(pp.t), Eigen::Quaterniond(from_row_major3(pp.R)));
    se3->setIndex(0); se3->sqrt_info = from_row_major(pp.sqrt_info, 6, 6); e.vioPosePriorEdge = se3;
    Eigen::Matrix<double, 9, 1> vbv; for (int k = 0; k < 9; k++) vbv(k) = vb.VB[k];
    auto *l9 = new Linear9Factor(vbv); l9->setIndex(Vo_SIZE - 1); l9->sqrt_info = from_row_major(vb.sqrt_info, 9, 9); e.vioVBPrior = l9;
}

