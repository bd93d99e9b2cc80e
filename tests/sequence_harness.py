"""Host-side window manager of the reference estimator, restated for whole-sequence tests (TEST INFRASTRUCTURE).

Mirrors, with the reference's control flow and quirks:
  Estimator::processIMU        src/estimator.cpp:91-124     (state propagation + pre-integration feed)
  Estimator::processImage      src/estimator.cpp:126-215    (NON_LINEAR branch; see `bootstrap` for INITIAL)
  Estimator::solveOdometry     src/estimator.cpp:461-472    (triangulate -> backendOptimization)
  Estimator::slideWindow       src/estimator.cpp:1565-1724  (MARGIN_OLD / MARGIN_NEW, prior-factor rotation)
  FeatureManager::addFeatureAndCheckParallax / compensatedParallax2 / removeBackShiftDepth / removeFront /
  removeFailures               src/feature_tracker/feature_manager.cpp:52-101, 356-390, 275-313, 335-354, 262-273
The arithmetic of the hot path itself (triangulate, initFactorGraph, backendOptimization) is NOT here: it is delegated
to a `solver` object, either the CPU oracle or the MI355X backend, so the same stream of IMU samples and feature
observations can be pushed through both and the two trajectories compared (ATE).

Not restated: the visual-inertial initialisation (initialStructure: SfM + alignment, out of scope).  `bootstrap`
replaces it: when the window first fills, the states are set from a (perturbed) reference trajectory, exactly the
hand-over point at which the reference switches to INITIAL_STRUCTURE (src/estimator.cpp:176-181).
"""
import ctypes as C

import numpy as np

from isvins_amd import abi, synth

ACC_N, GYR_N, ACC_W, GYR_W = synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W
MIN_PARALLAX = 10.0 / 460.0            # keyframe_parallax / FOCAL_LENGTH (config/euroc_config.yaml:43, parameters.cpp:82-83)


def _delta_q_R(theta):
    """Utility::deltaQ(theta).toRotationMatrix(): q = [1, theta/2] NOT normalised, Eigen's toRotationMatrix formula"""
    w, x, y, z = 1.0, theta[0] / 2, theta[1] / 2, theta[2] / 2
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def _quat_from_R(R):
    """Eigen::Quaterniond(R): (w, x, y, z) by the trace / largest-diagonal branches"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        return np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
    s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
    q = np.zeros(4)
    q[1 + i] = 0.25 * s; q[0] = (R[k, j] - R[j, k]) / s
    q[1 + j] = (R[j, i] + R[i, j]) / s; q[1 + k] = (R[k, i] + R[i, k]) / s
    return q


class PreInt:
    """IntegrationBase (include/factor/integration_base.h): the POD the factors read + the raw sample buffers;
    propagation by the oracle's restated midpoint rule (isvo_x_preint_step)."""

    def __init__(self, lib, acc_0, gyr_0, ba, bg):
        self.lib = lib
        self.pod = abi.isv_imu_t()
        self.acc_0, self.gyr_0 = np.array(acc_0, float), np.array(gyr_0, float)
        lib.isvo_x_preint_init(C.byref(self.pod), abi._p(np.array(ba, float)), abi._p(np.array(bg, float)))
        self.noise = np.array([ACC_N, GYR_N, ACC_W, GYR_W])

    def push_back(self, dt, acc, gyr):
        acc, gyr = np.array(acc, float), np.array(gyr, float)
        self.lib.isvo_x_preint_step(C.byref(self.pod), C.c_double(float(dt)), abi._p(self.acc_0), abi._p(self.gyr_0), abi._p(acc), abi._p(gyr),
                                    abi._p(self.noise))
        self.acc_0, self.gyr_0 = acc, gyr


class Track:
    """IDFeatures: one landmark track (feature_manager.h:44-63)"""

    def __init__(self, fid, start_frame):
        self.id, self.start_frame = fid, start_frame
        self.points = []               # Feature::point per frame from start_frame on
        self.depth = -1.0              # estimated_depth
        self.solve_flag = 0

    def end_frame(self):
        return self.start_frame + len(self.points) - 1


class OracleSolver:
    def __init__(self, lib, cfg):
        self.lib, self.cfg = lib, cfg

    def triangulate(self, w):
        assert self.lib.isvo_triangulate(C.byref(self.cfg), C.byref(w.c())) == 0

    def init_factor_graph(self, w):
        s = abi.isv_summary_t(); kld = np.zeros(1)
        assert self.lib.isvo_init_factor_graph(C.byref(self.cfg), C.byref(w.c()), C.byref(s), abi._p(kld)) == 0
        w.n_rollpitch = 0
        return s

    def optimize(self, w):
        s = abi.isv_summary_t(); m = abi.isv_marg_result_t()
        assert self.lib.isvo_optimize(C.byref(self.cfg), C.byref(w.c()), C.byref(s), C.byref(m)) == 0
        return s, m


class DeviceSolver:
    def __init__(self, backend):
        self.be, self.cfg = backend, backend.cfg

    def triangulate(self, w):
        self.be.triangulate([w])

    def init_factor_graph(self, w):
        s, _ = self.be.init_factor_graph(w)
        return s

    def optimize(self, w):
        return self.be.optimize(w)


class Estimator:
    def __init__(self, solver, lib, N, Nvo, g_norm=9.81007):
        self.solver, self.lib, self.N, self.Nvo = solver, lib, N, Nvo
        self.g = np.array([0, 0, g_norm])
        self.ric, self.tic = synth.RIC.copy(), synth.TIC.copy()
        self.Ps, self.Vs = np.zeros((N, 3)), np.zeros((N, 3))
        self.Rs = np.tile(np.eye(3), (N, 1, 1))
        self.Bas, self.Bgs = np.zeros((N, 3)), np.zeros((N, 3))
        self.Headers = np.zeros(N)
        self.pre = [None] * N
        self.bufs = [[] for _ in range(N)]           # (dt, acc, gyr) per frame
        self.frame_count, self.first_imu = 0, True
        self.acc_0, self.gyr_0 = np.zeros(3), np.zeros(3)
        self.solver_flag = "INITIAL"
        self.tracks = []                             # IDsfeatures, insertion order
        self.pose_prior, self.vb_prior = abi.isv_se3_prior_t(), abi.isv_linear9_t()
        self.relpose = [abi.isv_relpose_t() for _ in range(Nvo - 1)]      # edge (i, i+1) = vioRelativePoseEdges[i+1]
        self.rollpitch = []                          # vioRollPitchEdges
        self.margin_old = True
        self.to_add = None                           # (forward pose prior, backward relpose, backward vb) from the last MARGIN_OLD solve
        self.pose_output = []
        self.margin_history = []
        self.trajectory = []                         # (header, P, R) of the newest frame after every solve
        self.summaries = []

    # ---- Estimator::processIMU  src/estimator.cpp:91-124 ---------------------------------------
    def process_imu(self, dt, acc, gyr):
        acc, gyr = np.array(acc, float), np.array(gyr, float)
        if self.first_imu:
            self.first_imu = False
            self.acc_0, self.gyr_0 = acc, gyr
        j = self.frame_count
        if self.pre[j] is None:
            self.pre[j] = PreInt(self.lib, self.acc_0, self.gyr_0, self.Bas[j], self.Bgs[j])
        if j != 0:
            self.pre[j].push_back(dt, acc, gyr)
            self.bufs[j].append((dt, acc, gyr))
            un_acc_0 = self.Rs[j] @ (self.acc_0 - self.Bas[j]) - self.g
            un_gyr = 0.5 * (self.gyr_0 + gyr) - self.Bgs[j]
            self.Rs[j] = self.Rs[j] @ _delta_q_R(un_gyr * dt)
            un_acc_1 = self.Rs[j] @ (acc - self.Bas[j]) - self.g
            un_acc = 0.5 * (un_acc_0 + un_acc_1)
            self.Ps[j] = self.Ps[j] + dt * self.Vs[j] + 0.5 * dt * dt * un_acc
            self.Vs[j] = self.Vs[j] + dt * un_acc
        self.acc_0, self.gyr_0 = acc, gyr

    # ---- FeatureManager::addFeatureAndCheckParallax  feature_manager.cpp:52-101 -----------------
    def _add_features(self, image):
        fc = self.frame_count
        by_id = {t.id: t for t in self.tracks}
        last_track_num = 0
        for fid in sorted(image):                    # std::map iteration order
            pt = np.array(image[fid], float)
            t = by_id.get(fid)
            if t is None:
                t = Track(fid, fc); self.tracks.append(t); by_id[fid] = t
                t.points.append(pt)
            else:
                t.points.append(pt); last_track_num += 1
        if fc < 2 or last_track_num < 20:
            return True
        psum, pnum = 0.0, 0
        for t in self.tracks:
            if t.start_frame <= fc - 2 and t.start_frame + len(t.points) - 1 >= fc - 1:
                p2, p1 = t.points[fc - 2 - t.start_frame], t.points[fc - 1 - t.start_frame]      # compensatedParallax2 :356-390
                du, dv = p2[0] / p2[2] - p1[0], p2[1] / p2[2] - p1[1]
                psum += max(0.0, np.sqrt(du * du + dv * dv)); pnum += 1
        return True if pnum == 0 else psum / pnum >= MIN_PARALLAX

    def _good(self):
        """goodFeature (feature_manager.cpp:27-31): used_num >= 2 and start_frame < Vo_SIZE, in IDsfeatures order"""
        return [t for t in self.tracks if len(t.points) >= 2 and t.start_frame < self.Nvo]

    # ---- pack the Estimator members backendOptimization() touches into the ABI window -----------
    def _window(self):
        good = self._good()
        n_obs = sum(len(t.points) for t in good)
        w = abi.Window(self.N, self.Nvo, len(good), n_obs, len(self.rollpitch))
        w.Ps[...] = self.Ps.reshape(w.Ps.shape); w.Rs[...] = self.Rs.reshape(w.Rs.shape); w.Vs[...] = self.Vs.reshape(w.Vs.shape)
        w.Bas[...] = self.Bas.reshape(w.Bas.shape); w.Bgs[...] = self.Bgs.reshape(w.Bgs.shape)
        w.tic[...] = self.tic.reshape(w.tic.shape); w.ric[...] = self.ric.reshape(w.ric.shape)
        ptr, obs = [0], []
        for l, t in enumerate(good):
            w.lm_start_frame[l] = t.start_frame
            obs.extend(t.points); ptr.append(len(obs))
            w.lm_depth[l] = t.depth
        w.lm_obs_ptr[: len(ptr)] = ptr
        if obs:
            w.obs_point.reshape(-1, 3)[: len(obs)] = np.array(obs)
        for j in range(1, self.N):
            C.memmove(C.byref(w.imu[j - 1]), C.byref(self.pre[j].pod), C.sizeof(abi.isv_imu_t))
        C.memmove(C.byref(w.pose_prior), C.byref(self.pose_prior), C.sizeof(abi.isv_se3_prior_t))
        C.memmove(C.byref(w.vb_prior), C.byref(self.vb_prior), C.sizeof(abi.isv_linear9_t))
        for i in range(self.Nvo - 1):
            C.memmove(C.byref(w.relpose[i]), C.byref(self.relpose[i]), C.sizeof(abi.isv_relpose_t))
        for i, f in enumerate(self.rollpitch):
            C.memmove(C.byref(w.rollpitch[i]), C.byref(f), C.sizeof(abi.isv_rollpitch_t))
        w.margin_old = 1 if self.margin_old else 0
        w.header0 = float(self.Headers[0])
        return w, good

    def _read_back(self, w, good):
        self.Ps = w.Ps.reshape(self.N, 3).copy(); self.Rs = w.Rs.reshape(self.N, 3, 3).copy(); self.Vs = w.Vs.reshape(self.N, 3).copy()
        self.Bas = w.Bas.reshape(self.N, 3).copy(); self.Bgs = w.Bgs.reshape(self.N, 3).copy()
        for l, t in enumerate(good):
            t.depth = float(w.lm_depth[l]); t.solve_flag = int(w.lm_solve_flag[l])
        C.memmove(C.byref(self.pose_prior), C.byref(w.pose_prior), C.sizeof(abi.isv_se3_prior_t))
        C.memmove(C.byref(self.vb_prior), C.byref(w.vb_prior), C.sizeof(abi.isv_linear9_t))
        for i in range(self.Nvo - 1):
            C.memmove(C.byref(self.relpose[i]), C.byref(w.relpose[i]), C.sizeof(abi.isv_relpose_t))
        for i in range(len(self.rollpitch)):
            C.memmove(C.byref(self.rollpitch[i]), C.byref(w.rollpitch[i]), C.sizeof(abi.isv_rollpitch_t))
        if getattr(self.solver, "cfg", None) is not None and self.solver.cfg.estimate_extrinsic:
            # double2vector: tic[0] / ric[0] from para_Ex_Pose (src/estimator.cpp:575-583) -- the next window and slideWindowOld use them
            self.tic = np.asarray(w.tic, dtype=np.float64).reshape(3).copy()
            self.ric = np.asarray(w.ric, dtype=np.float64).reshape(3, 3).copy()

    # ---- solveOdometry  src/estimator.cpp:461-472 + backendOptimization :1541-1562 ------------------
    @staticmethod
    def _copy(src, T):
        o = T()
        C.memmove(C.byref(o), C.byref(src), C.sizeof(T))
        return o

    def _solve_odometry(self):
        w, good = self._window()
        self.solver.triangulate(w)                   # f_manager.triangulate(Ps, tic, ric)
        for l, t in enumerate(good):
            t.depth = float(w.lm_depth[l])
        if self.solver_flag == "INITIAL_STRUCTURE":  # vector2double(); initFactorGraph(); solver_flag = NON_LINEAR
            s0 = self.solver.init_factor_graph(w)
            self.rollpitch = []
            self._read_back(w, good)
            self.solver_flag = "NON_LINEAR"
            self.to_add = None
            self.summaries.append(s0)
            w, good = self._window()                 # the NON_LINEAR branch runs in the same call (two `if`s, not else-if)
        s, m = self.solver.optimize(w)
        self._read_back(w, good)
        if self.margin_old and m.valid:
            self.to_add = (self._copy(m.forward_pose_prior, abi.isv_se3_prior_t), self._copy(m.backward_relpose, abi.isv_relpose_t),
                           self._copy(m.backward_vb, abi.isv_linear9_t))
            brp = self._copy(m.backward_rollpitch, abi.isv_rollpitch_t); brp.index = self.Nvo - 1
            self.rollpitch.append(brp)               # vioRollPitchEdges.push_back (MargBackward :1536-1538)
        self.summaries.append(s)

    # ---- slideWindow  src/estimator.cpp:1565-1724 --------------------------------------------------
    def _slide_window(self):
        N, Nvo = self.N, self.Nvo
        if self.margin_old:
            back_R0, back_P0 = self.Rs[0].copy(), self.Ps[0].copy()
            if self.frame_count != N - 1:
                return
            for a in (self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs):
                a[: N - 1] = a[1:].copy()            # the swaps; the last slot is overwritten below anyway
            self.Headers[: N - 1] = self.Headers[1:].copy()
            self.pre = self.pre[1:] + [None]; self.bufs = self.bufs[1:] + [[]]
            self.Headers[N - 1] = self.Headers[N - 2]
            for a in (self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs):
                a[N - 1] = a[N - 2]
            self.pre[N - 1] = PreInt(self.lib, self.acc_0, self.gyr_0, self.Bas[N - 1], self.Bgs[N - 1])
            shift_depth = self.solver_flag == "NON_LINEAR"
            if shift_depth and self.to_add is not None:
                fwd_prior, bwd_rel, bwd_vb = self.to_add
                for f in self.relpose:
                    f.imu_i -= 1; f.imu_j -= 1       # RelativePoseFactor::shift()
                self.relpose = self.relpose[1:] + [bwd_rel]
                bwd_rel.imu_i, bwd_rel.imu_j = Nvo - 2, Nvo - 1
                kept = []
                for f in self.rollpitch:
                    f.index -= 1                     # RollPitchFactor::shift()
                    if f.index >= 0:
                        kept.append(f)
                self.rollpitch = kept
                fwd_prior.index = 0; self.pose_prior = fwd_prior
                bwd_vb.index = Nvo - 1; self.vb_prior = bwd_vb
                self.to_add = None
            # slideWindowOld
            if shift_depth:
                R0, R1 = back_R0 @ self.ric, self.Rs[0] @ self.ric
                P0, P1 = back_P0 + back_R0 @ self.tic, self.Ps[0] + self.Rs[0] @ self.tic
                keep = []
                for t in self.tracks:                # removeBackShiftDepth  feature_manager.cpp:275-313
                    if t.start_frame != 0:
                        t.start_frame -= 1; keep.append(t); continue
                    uv = t.points.pop(0)
                    if len(t.points) < 2:
                        continue
                    pj = R1.T @ (R0 @ (uv * t.depth) + P0 - P1)
                    t.depth = float(pj[2]) if pj[2] > 0 else self.solver.cfg.init_depth
                    keep.append(t)
                self.tracks = keep
            else:
                keep = []
                for t in self.tracks:                # removeBack :315-332
                    if t.start_frame != 0:
                        t.start_frame -= 1; keep.append(t)
                    else:
                        t.points.pop(0)
                        if t.points:
                            keep.append(t)
                self.tracks = keep
        else:
            fc = self.frame_count
            if fc != N - 1:
                return
            for (dt, a, g) in self.bufs[fc]:
                self.pre[fc - 1].push_back(dt, a, g)
                self.bufs[fc - 1].append((dt, a, g))
            self.Headers[fc - 1] = self.Headers[fc]
            for arr in (self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs):
                arr[fc - 1] = arr[fc]
            self.pre[N - 1] = PreInt(self.lib, self.acc_0, self.gyr_0, self.Bas[N - 1], self.Bgs[N - 1])
            self.bufs[N - 1] = []
            keep = []
            for t in self.tracks:                    # removeFront  feature_manager.cpp:335-354
                if t.start_frame == fc:
                    t.start_frame -= 1; keep.append(t); continue
                if t.end_frame() < fc - 1:
                    keep.append(t); continue
                del t.points[N - 1 - 1 - t.start_frame]
                if t.points:
                    keep.append(t)
            self.tracks = keep

    # ---- processImage  src/estimator.cpp:126-215 ---------------------------------------------------
    def process_image(self, image, header, bootstrap=None):
        self.margin_old = self._add_features(image)
        self.margin_history.append(bool(self.margin_old))
        self.Headers[self.frame_count] = header
        if self.solver_flag == "INITIAL":
            if self.frame_count == self.N - 1:
                # initialStructure() is out of scope: the states come from `bootstrap` (see module docstring)
                P, R, V = bootstrap
                self.Ps, self.Rs, self.Vs = P.copy(), R.copy(), V.copy()
                self.solver_flag = "INITIAL_STRUCTURE"
                self._solve_odometry()
                self._slide_window()
                self.tracks = [t for t in self.tracks if t.solve_flag != 2]      # removeFailures
            else:
                self.frame_count += 1
                return
        else:
            self._solve_odometry()
            self._slide_window()
            self.tracks = [t for t in self.tracks if t.solve_flag != 2]
        self.trajectory.append((header, self.Ps[self.N - 1].copy(), self.Rs[self.N - 1].copy()))
        # the row System::ProcessBackEnd appends to pose_output.txt once NON_LINEAR: the OLDEST frame  src/System.cpp:401-410
        self.pose_output.append((self.Headers[0], self.Ps[0].copy(), self.Rs[0].copy()))

    def write_pose_output(self, path):
        """pose_output.txt as the reference writes it (src/System.cpp:408-409): `fixed` stream formatting, so six
        decimals: stamp px py pz qw qx qy qz"""
        with open(path, "w") as f:
            for (t, p, R) in self.pose_output:
                q = _quat_from_R(R)
                f.write("%.6f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n" % (t, p[0], p[1], p[2], q[0], q[1], q[2], q[3]))


class FigureEight:
    """6-DoF stand-in for a EuRoC machine-hall flight: a figure-eight position path
    (lemniscate of Gerono, 8 m x 4 m, 12 s per lap: up to 2.1 m/s and 1.1 m/s^2, enough excitation for the monocular
    scale to be observable) with a 0.5 m vertical wave, sinusoidal yaw (+-1 rad), roll and pitch (+-8 deg); analytic
    velocity / acceleration / body rates, same interface as synth.Trajectory"""

    def __init__(self, A=4.0, B=2.0, Cz=0.5, period=12.0):
        self.A, self.B, self.Cz = A, B, Cz
        self.w = 2 * np.pi / period
        self.wz = 1.7 * self.w
        self.ya, self.wy = 1.0, 0.6 * self.w * 3
        self.amp = np.deg2rad(8.0)

    def p(self, t):
        return np.array([self.A * np.sin(self.w * t), 0.5 * self.B * np.sin(2 * self.w * t), self.Cz * np.sin(self.wz * t)])

    def vel(self, t):
        return np.array([self.A * self.w * np.cos(self.w * t), self.B * self.w * np.cos(2 * self.w * t), self.Cz * self.wz * np.cos(self.wz * t)])

    def acc(self, t):
        return np.array([-self.A * self.w ** 2 * np.sin(self.w * t), -2 * self.B * self.w ** 2 * np.sin(2 * self.w * t), -self.Cz * self.wz ** 2 * np.sin(self.wz * t)])

    def ypr(self, t):
        return (self.ya * np.sin(self.wy * t), self.amp * np.sin(2 * np.pi * 0.5 * t), self.amp * np.sin(2 * np.pi * 0.3 * t + 1.0))

    def ypr_dot(self, t):
        return (self.ya * self.wy * np.cos(self.wy * t), self.amp * 2 * np.pi * 0.5 * np.cos(2 * np.pi * 0.5 * t),
                self.amp * 2 * np.pi * 0.3 * np.cos(2 * np.pi * 0.3 * t + 1.0))

    def R(self, t):
        return synth._rot_zyx(*self.ypr(t))

    def gyro(self, t):
        y, p, r = self.ypr(t)
        yd, pd, rd = self.ypr_dot(t)
        return np.array([rd - yd * np.sin(p), pd * np.cos(r) + yd * np.sin(r) * np.cos(p), -pd * np.sin(r) + yd * np.cos(r) * np.cos(p)])


class Simulator:
    """a camera-IMU rig on synth.Trajectory: feature observations at 10 Hz with pixel noise 1/460, IMU at 200 Hz.
    `euroc_like=True`: the SURVEY 8d stand-in for the EuRoC runs instead -- FigureEight, 20 Hz frames, 200 Hz IMU,
    ~150 features per frame (max_cnt of config/euroc_config.yaml:40)."""

    def __init__(self, seed=0, n_points=1500, frame_dt=0.1, imu_per_frame=20, euroc_like=False):
        self.rng = np.random.default_rng(seed)
        self.traj = synth.Trajectory(0.3)
        # the yaml extrinsic points the camera's optical axis along the body z axis: the landmarks form a ceiling
        if euroc_like:
            self.traj = FigureEight()
            n_points, frame_dt, imu_per_frame = 1900, 0.05, 10
            self.points = np.stack([self.rng.uniform(-13.0, 13.0, n_points), self.rng.uniform(-10.0, 10.0, n_points), self.rng.uniform(3.5, 8.0, n_points)], 1)
        else:
            self.points = np.stack([self.rng.uniform(-8.0, 8.0, n_points), self.rng.uniform(-8.0, 8.0, n_points), self.rng.uniform(3.0, 8.0, n_points)], 1)
        self.frame_dt, self.k = frame_dt, imu_per_frame
        self.ba, self.bg = self.rng.normal(0, 0.02, 3), self.rng.normal(0, 0.002, 3)

    def frame(self, i):
        t = i * self.frame_dt
        R, P = self.traj.R(t), self.traj.p(t)
        Rc, tc = R @ synth.RIC, P + R @ synth.TIC
        pc = (self.points - tc) @ Rc                 # points in the camera frame
        vis = (pc[:, 2] > 0.5) & (np.abs(pc[:, 0] / pc[:, 2]) < 0.7) & (np.abs(pc[:, 1] / pc[:, 2]) < 0.5)
        image = {}
        for fid in np.nonzero(vis)[0]:
            xy = pc[fid, :2] / pc[fid, 2] + self.rng.normal(0, 1.0 / 460.0, 2)
            image[int(fid)] = (xy[0], xy[1], 1.0)
        return t, image

    def imu_between(self, i):
        """samples in (t_{i-1}, t_i]"""
        out = []
        dt = self.frame_dt / self.k
        G = np.array([0, 0, 9.81007])
        for s in range(1, self.k + 1):
            t = (i - 1) * self.frame_dt + s * dt
            acc = self.traj.R(t).T @ (self.traj.acc(t) + G) + self.ba + ACC_N * self.rng.normal(0, 1, 3) * 0.05
            gyr = self.traj.gyro(t) + self.bg + GYR_N * self.rng.normal(0, 1, 3) * 0.05
            out.append((dt, acc, gyr))
        return out

    def truth_window(self, i_last, N):
        ts = [(i_last - (N - 1) + k) * self.frame_dt for k in range(N)]
        P = np.array([self.traj.p(t) for t in ts]); R = np.array([self.traj.R(t) for t in ts]); V = np.array([self.traj.vel(t) for t in ts])
        return P, R, V


def run_sequence(solver, lib, N, Nvo, n_frames, seed=0, euroc_like=False, progress=None):
    sim = Simulator(seed, euroc_like=euroc_like)
    est = Estimator(solver, lib, N, Nvo)
    for i in range(n_frames):
        if i > 0:
            for (dt, a, g) in sim.imu_between(i):
                est.process_imu(dt, a, g)
        else:
            est.process_imu(sim.frame_dt / sim.k, sim.traj.R(0).T @ (sim.traj.acc(0) + np.array([0, 0, 9.81007])) + sim.ba, sim.traj.gyro(0) + sim.bg)
        t, image = sim.frame(i)
        boot = None
        if est.solver_flag == "INITIAL" and est.frame_count == N - 1:
            P, R, V = sim.truth_window(i, N)
            nrng = np.random.default_rng(1000 + seed)
            P = P + nrng.normal(0, 0.01, P.shape); V = V + nrng.normal(0, 0.02, V.shape)
            boot = (P, R, V)
        est.process_image(image, t, bootstrap=boot)
        if progress and i % 200 == 0:
            progress(i)
    return est, sim


# ---- the native window manager (include/isvins_estimator.h) driven by the same simulated streams ------------------
def oracle_vtbl(oracle, cfg, fused=False):
    """isv_solver_vtbl_t whose three entry points are the CPU oracle: the test seam of isv_estimator_create_with_solver,
    so that the C++ host logic can be checked on a machine without a GPU"""
    from isvins_amd import estimator as E

    def tri(ctx, n, ws):
        for i in range(n):
            if oracle.isvo_triangulate(C.byref(cfg), ws[i]) != 0:
                return -1
        return 0

    def init(ctx, w, s, kld):
        rc = oracle.isvo_init_factor_graph(C.byref(cfg), w, s, kld)
        w.contents.n_rollpitch = 0
        return rc

    def opt(ctx, n, ws, sums, margs):
        for i in range(n):
            if oracle.isvo_optimize(C.byref(cfg), ws[i], C.byref(sums[i]), C.byref(margs[i])) != 0:
                return -1
        return 0
    def solve_odometry(ctx, n, ws, sums, margs):
        return tri(ctx, n, ws) or opt(ctx, n, ws, sums, margs)
    if fused:
        def init_batch(ctx, n, ws, sums, klds):
            for i in range(n):
                if oracle.isvo_init_factor_graph(C.byref(cfg), ws[i], C.byref(sums[i]), None) != 0:
                    return -1
                ws[i].contents.n_rollpitch = 0
            return 0
        return E.isv_solver_vtbl_t(None, E.TRIANGULATE_FN(tri), E.INIT_FN(init), E.OPTIMIZE_FN(opt), E.INIT_BATCH_FN(init_batch), E.OPTIMIZE_FN(solve_odometry))
    return E.isv_solver_vtbl_t(None, E.TRIANGULATE_FN(tri), E.INIT_FN(init), E.OPTIMIZE_FN(opt))


def estimator_params(cfg):
    from isvins_amd import estimator as E
    return E.make_params(cfg, synth.RIC, synth.TIC, ACC_N, GYR_N, ACC_W, GYR_W, MIN_PARALLAX)


def run_sequences_native(est, N, n_frames, seeds, euroc_like=False):
    """push len(seeds) simulated streams (Simulator(seed)) through a SequenceEstimator in lock step"""
    sims = [Simulator(sd, euroc_like=euroc_like) for sd in seeds]
    for i in range(n_frames):
        for s, (sim, sd) in enumerate(zip(sims, seeds)):
            if i > 0:
                for (dt, a, g) in sim.imu_between(i):
                    est.process_imu(s, dt, a, g)
            else:
                est.process_imu(s, sim.frame_dt / sim.k, sim.traj.R(0).T @ (sim.traj.acc(0) + np.array([0, 0, 9.81007])) + sim.ba, sim.traj.gyro(0) + sim.bg)
            t, image = sim.frame(i)
            st = est.status(s)
            if st["solver_flag"] == 0 and st["frame_count"] == N - 1:
                P, R, V = sim.truth_window(i, N)
                nrng = np.random.default_rng(1000 + sd)
                P = P + nrng.normal(0, 0.01, P.shape); V = V + nrng.normal(0, 0.02, V.shape)
                est.set_bootstrap(s, P, R, V)
            ids = np.array(list(image.keys()), np.int32)[::-1]                 # any order: the estimator sorts by id
            pts = np.array([image[int(k)] for k in ids], float).reshape(-1, 3)
            est.push_image(s, t, ids, pts)
        est.step()
    return sims


def write_stream(path, N, Nvo, n_frames, seed=0, max_landmarks=800, num_iterations=10):
    """the simulated stream of run_sequence(seed=...) in the text form tools/isv_replay reads (floats with 17
    significant digits, so the replayed values are the simulator's bit for bit)"""
    sim = Simulator(seed)
    r = lambda v: " ".join(repr(float(x)) for x in np.asarray(v, float).ravel())
    with open(path, "w") as f:
        f.write("# isv-stream 1: simulated camera-IMU rig (tests/sequence_harness.Simulator), seed %d\n" % seed)
        f.write("config %d %d %d %d %r %r %r %r %r %r %r %r %r\n" % (N, Nvo, max_landmarks, num_iterations, 460.0, 9.81007, 0.1, 5.0,
                                                               ACC_N, GYR_N, ACC_W, GYR_W, MIN_PARALLAX))
        f.write("ric %s\ntic %s\n" % (r(synth.RIC), r(synth.TIC)))
        for i in range(n_frames):
            if i > 0:
                for (dt, a, g) in sim.imu_between(i):
                    f.write("imu %r %s %s\n" % (float(dt), r(a), r(g)))
            else:
                f.write("imu %r %s %s\n" % (sim.frame_dt / sim.k, r(sim.traj.R(0).T @ (sim.traj.acc(0) + np.array([0, 0, 9.81007])) + sim.ba), r(sim.traj.gyro(0) + sim.bg)))
            t, image = sim.frame(i)
            if i == N - 1:
                P, R, V = sim.truth_window(i, N)
                nrng = np.random.default_rng(1000 + seed)
                P = P + nrng.normal(0, 0.01, P.shape); V = V + nrng.normal(0, 0.02, V.shape)
                f.write("boot\n")
                for k in range(N):
                    f.write("%s %s %s\n" % (r(P[k]), r(R[k]), r(V[k])))
            f.write("frame %r %d\n" % (float(t), len(image)))
            for fid in image:
                f.write("%d %s\n" % (fid, r(image[fid])))


def write_euroc_like(root, N, Nvo, n_frames, seed=0, max_landmarks=800, num_iterations=10, t0_ns=1_000_000_000):
    """the same simulated stream as write_stream(seed=...), in the files the reference's test/run_euroc.cpp reads
    plus a feature-track table in place of the images:
      root/mav0/imu0/data.csv   `#timestamp [ns],w_x,w_y,w_z,a_x,a_y,a_z`  (200 Hz, one sample before the first frame)
      root/tracks.csv           `#timestamp [ns],id,x,y,z`  one row per observation, frames in time order
      root/config.txt           the `config` / `ric` / `tic` lines of the stream format + the `boot` rows
    Timestamps are integer nanoseconds; frames fall on IMU sample times.  (With a real EuRoC epoch, ~1.4e18 ns, the
    reference's `stamp / 1e9` in double precision quantises the sample spacing to 2.4e-7 s; t0_ns = 1 s keeps the
    spacing exact so that this stream reproduces write_stream's.)"""
    import os
    sim = Simulator(seed)
    r = lambda v: ",".join(repr(float(x)) for x in np.asarray(v, float).ravel())
    rs = lambda v: " ".join(repr(float(x)) for x in np.asarray(v, float).ravel())
    os.makedirs(os.path.join(root, "mav0", "imu0"), exist_ok=True)
    G = np.array([0, 0, 9.81007])
    dt_ns = int(round(sim.frame_dt / sim.k * 1e9)); frame_ns = dt_ns * sim.k
    a0 = sim.traj.R(0).T @ (sim.traj.acc(0) + G) + sim.ba; g0 = sim.traj.gyro(0) + sim.bg
    imu_rows = ["%d,%s,%s" % (t0_ns - dt_ns, r(g0), r(a0)),            # a sample before the first image (else the image is thrown)
                "%d,%s,%s" % (t0_ns, r(g0), r(a0))]
    track_rows, boot_rows = [], []
    for i in range(n_frames + 1):                                       # same generator order as write_stream: IMU of frame i, then frame i
        if i > 0:                                                       # (+ one more frame of IMU than of images: "wait for imu")
            for s_, (dt, a, g) in enumerate(sim.imu_between(i), 1):
                imu_rows.append("%d,%s,%s" % (t0_ns + (i - 1) * frame_ns + s_ * dt_ns, r(g), r(a)))
        if i == n_frames:
            break
        t, image = sim.frame(i)
        if i == N - 1:
            P, R, V = sim.truth_window(i, N)
            nrng = np.random.default_rng(1000 + seed)
            P = P + nrng.normal(0, 0.01, P.shape); V = V + nrng.normal(0, 0.02, V.shape)
            boot_rows = ["%s %s %s" % (rs(P[k]), rs(R[k]), rs(V[k])) for k in range(N)]
        for fid in image:
            track_rows.append("%d,%d,%s" % (t0_ns + i * frame_ns, fid, r(image[fid])))
    with open(os.path.join(root, "mav0", "imu0", "data.csv"), "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_RS_S_y [rad s^-1],w_RS_S_z [rad s^-1],a_RS_S_x [m s^-2],a_RS_S_y [m s^-2],a_RS_S_z [m s^-2]\n")
        f.write("\n".join(imu_rows) + "\n")
    with open(os.path.join(root, "tracks.csv"), "w") as f:
        f.write("#timestamp [ns],id,x,y,z\n" + "\n".join(track_rows) + "\n")
    with open(os.path.join(root, "config.txt"), "w") as c:
        c.write("config %d %d %d %d %r %r %r %r %r %r %r %r %r\n" % (N, Nvo, max_landmarks, num_iterations, 460.0, 9.81007, 0.1, 5.0,
                                                               ACC_N, GYR_N, ACC_W, GYR_W, MIN_PARALLAX))
        c.write("ric %s\ntic %s\n" % (rs(synth.RIC), rs(synth.TIC)))
        c.write("boot\n" + "\n".join(boot_rows) + "\n")
