"""ctypes mirror of include/isvins_backend.h (the C ABI of the backend).

Only data-layout code lives here: structures, and `Window`, a numpy-backed owner of the buffers
an `isv_window_t` points to.  The field names are the reference's own member names
(include/estimator.h:90-154 of lyeemax/IS-VINS) so tests read like the reference.
"""
import ctypes as C
import numpy as np

ISV_MAX_TRACE = 64
ISV_PROJ_STRIP = 28
ISV_IMU_STRIP = 465
c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class isv_config_t(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_vo", C.c_int32), ("max_landmarks", C.c_int32),
                ("max_obs", C.c_int32), ("max_rollpitch", C.c_int32), ("max_batch", C.c_int32),
                ("num_iterations", C.c_int32), ("estimate_extrinsic", C.c_int32),
                ("proj_sqrt_info", C.c_double * 4), ("gravity", C.c_double * 3),
                ("alpha", C.c_double), ("init_depth", C.c_double)]


class isv_imu_t(C.Structure):
    _fields_ = [("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
                ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3),
                ("sum_dt", C.c_double), ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225)]


class isv_se3_prior_t(C.Structure):
    _fields_ = [("t", C.c_double * 3), ("R", C.c_double * 9), ("sqrt_info", C.c_double * 36),
                ("index", C.c_int32), ("_pad", C.c_int32)]


class isv_linear9_t(C.Structure):
    _fields_ = [("VB", C.c_double * 9), ("sqrt_info", C.c_double * 81),
                ("index", C.c_int32), ("_pad", C.c_int32)]


class isv_relpose_t(C.Structure):
    _fields_ = [("delta_t", C.c_double * 3), ("delta_R", C.c_double * 9), ("sqrt_info", C.c_double * 36),
                ("imu_i", C.c_int32), ("imu_j", C.c_int32)]


class isv_rollpitch_t(C.Structure):
    _fields_ = [("R", C.c_double * 9), ("sqrt_info", C.c_double * 4),
                ("index", C.c_int32), ("_pad", C.c_int32)]


class isv_window_t(C.Structure):
    _fields_ = [("Ps", c_double_p), ("Rs", c_double_p), ("Vs", c_double_p), ("Bas", c_double_p),
                ("Bgs", c_double_p), ("tic", c_double_p), ("ric", c_double_p),
                ("n_landmarks", C.c_int32), ("n_obs", C.c_int32),
                ("lm_start_frame", c_int32_p), ("lm_obs_ptr", c_int32_p), ("obs_point", c_double_p),
                ("lm_depth", c_double_p), ("lm_solve_flag", c_int32_p),
                ("imu", C.POINTER(isv_imu_t)),
                ("pose_prior", C.POINTER(isv_se3_prior_t)), ("vb_prior", C.POINTER(isv_linear9_t)),
                ("relpose", C.POINTER(isv_relpose_t)), ("rollpitch", C.POINTER(isv_rollpitch_t)),
                ("n_rollpitch", C.c_int32), ("margin_old", C.c_int32), ("header0", C.c_double),
                ("para_Pose", c_double_p), ("para_SpeedBias", c_double_p),
                ("para_Ex_Pose", c_double_p), ("para_Feature", c_double_p)]


class isv_summary_t(C.Structure):
    _fields_ = [("status", C.c_int32), ("termination", C.c_int32), ("iterations", C.c_int32),
                ("num_successful", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("trace_cost", C.c_double * ISV_MAX_TRACE), ("trace_radius", C.c_double * ISV_MAX_TRACE),
                ("trace_step_norm", C.c_double * ISV_MAX_TRACE), ("trace_accepted", C.c_int32 * ISV_MAX_TRACE)]


class isv_combined_factors_t(C.Structure):
    _fields_ = [("relative_pose", isv_relpose_t), ("has_rollpitch", C.c_int32), ("_pad", C.c_int32),
                ("rollpitch", isv_rollpitch_t), ("covRel", C.c_double * 36), ("covAbs", C.c_double * 4),
                ("distance", C.c_double), ("ts", C.c_double), ("Ri", C.c_double * 9), ("ti", C.c_double * 3)]


class isv_marg_result_t(C.Structure):
    _fields_ = [("valid", C.c_int32), ("n_marg_landmarks", C.c_int32),
                ("combined", isv_combined_factors_t), ("forward_pose_prior", isv_se3_prior_t),
                ("backward_relpose", isv_relpose_t), ("backward_vb", isv_linear9_t),
                ("backward_rollpitch", isv_rollpitch_t),
                ("forward_kld", C.c_double), ("backward_kld", C.c_double)]


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def arr(cfield, shape=None):
    """numpy copy of a ctypes array field"""
    a = np.ctypeslib.as_array(cfield).copy()
    return a.reshape(shape) if shape is not None else a


def make_config(n_frames, n_vo, max_landmarks=1000, max_obs=None, max_batch=1, num_iterations=10,
                pixel_sqrt_info=460.0, g_norm=9.81007, alpha=0.1, init_depth=5.0, estimate_extrinsic=0):
    """isv_config_t with the reference's yaml defaults (config/euroc_config.yaml:50,61,83,86)."""
    cfg = isv_config_t()
    cfg.n_frames, cfg.n_vo = n_frames, n_vo
    cfg.max_landmarks = max_landmarks
    cfg.max_obs = max_obs if max_obs is not None else max_landmarks * n_frames
    cfg.max_rollpitch = n_vo + 1
    cfg.max_batch = max_batch
    cfg.num_iterations = num_iterations
    cfg.estimate_extrinsic = estimate_extrinsic
    cfg.proj_sqrt_info[:] = [pixel_sqrt_info, 0.0, 0.0, pixel_sqrt_info]
    cfg.gravity[:] = [0.0, 0.0, g_norm]
    cfg.alpha = alpha
    cfg.init_depth = init_depth
    return cfg


class Window:
    """Owns the numpy buffers of one sliding window and exposes them as an isv_window_t.

    Attributes carry the reference's member names: Ps, Rs, Vs, Bas, Bgs, tic, ric,
    lm_start_frame / lm_obs_ptr / obs_point (FeatureManager view), lm_depth (estimated_depth),
    imu (pre_integrations[1..N-1]), pose_prior (vioPosePriorEdge), vb_prior (vioVBPrior),
    relpose (vioRelativePoseEdges[1..]), rollpitch (vioRollPitchEdges)."""

    def __init__(self, n_frames, n_vo, n_landmarks, n_obs, n_rollpitch):
        N, L = n_frames, n_landmarks
        self.N, self.Nvo, self.L = N, n_vo, L
        self.Ps = np.zeros((N, 3)); self.Rs = np.tile(np.eye(3), (N, 1, 1)); self.Vs = np.zeros((N, 3))
        self.Bas = np.zeros((N, 3)); self.Bgs = np.zeros((N, 3))
        self.tic = np.zeros(3); self.ric = np.eye(3)
        self.lm_start_frame = np.zeros(max(L, 1), np.int32)
        self.lm_obs_ptr = np.zeros(L + 1, np.int32)
        self.obs_point = np.zeros((max(n_obs, 1), 3))
        self.lm_depth = np.zeros(max(L, 1)); self.lm_solve_flag = np.zeros(max(L, 1), np.int32)
        self.imu = (isv_imu_t * max(N - 1, 1))()
        self.pose_prior = isv_se3_prior_t(); self.vb_prior = isv_linear9_t()
        self.relpose = (isv_relpose_t * max(n_vo - 1, 1))()
        self.rollpitch = (isv_rollpitch_t * max(n_rollpitch, 1))()
        self.n_rollpitch = n_rollpitch
        self.n_obs = n_obs
        self.margin_old = 0
        self.header0 = 0.0
        self.para_Pose = np.zeros((N, 7)); self.para_SpeedBias = np.zeros((N, 9))
        self.para_Ex_Pose = np.zeros(7); self.para_Feature = np.zeros(max(L, 1))
        self.truth = None
        self._c = None

    @property
    def n_factors(self):
        return int(self.n_obs - self.L)

    def c(self):
        """the isv_window_t view (pointers into this object's buffers)"""
        w = isv_window_t()
        for name in ("Ps", "Rs", "Vs", "Bas", "Bgs", "tic", "ric", "obs_point", "lm_depth",
                     "para_Pose", "para_SpeedBias", "para_Ex_Pose", "para_Feature"):
            a = getattr(self, name)
            assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], name
            setattr(w, name, _p(a))
        w.lm_start_frame = _p(self.lm_start_frame, C.c_int32)
        w.lm_obs_ptr = _p(self.lm_obs_ptr, C.c_int32)
        w.lm_solve_flag = _p(self.lm_solve_flag, C.c_int32)
        w.n_landmarks, w.n_obs = self.L, self.n_obs
        w.imu = C.cast(self.imu, C.POINTER(isv_imu_t))
        w.pose_prior = C.pointer(self.pose_prior); w.vb_prior = C.pointer(self.vb_prior)
        w.relpose = C.cast(self.relpose, C.POINTER(isv_relpose_t))
        w.rollpitch = C.cast(self.rollpitch, C.POINTER(isv_rollpitch_t))
        w.n_rollpitch, w.margin_old, w.header0 = self.n_rollpitch, self.margin_old, self.header0
        self._c = w
        return w

    def clone(self):
        o = Window(self.N, self.Nvo, self.L, self.n_obs, self.n_rollpitch)
        for name in ("Ps", "Rs", "Vs", "Bas", "Bgs", "tic", "ric", "lm_start_frame", "lm_obs_ptr",
                     "obs_point", "lm_depth", "lm_solve_flag", "para_Pose", "para_SpeedBias",
                     "para_Ex_Pose", "para_Feature"):
            getattr(o, name)[...] = getattr(self, name)
        C.memmove(o.imu, self.imu, C.sizeof(self.imu))
        C.memmove(C.byref(o.pose_prior), C.byref(self.pose_prior), C.sizeof(isv_se3_prior_t))
        C.memmove(C.byref(o.vb_prior), C.byref(self.vb_prior), C.sizeof(isv_linear9_t))
        C.memmove(o.relpose, self.relpose, C.sizeof(self.relpose))
        C.memmove(o.rollpitch, self.rollpitch, C.sizeof(self.rollpitch))
        o.margin_old, o.header0, o.truth = self.margin_old, self.header0, self.truth
        return o

    def state_vector(self):
        """all in/out state as one flat vector (for parity comparisons)"""
        return np.concatenate([self.Ps.ravel(), self.Rs.ravel(), self.Vs.ravel(), self.Bas.ravel(),
                               self.Bgs.ravel(), self.lm_depth[: self.L]])

    def priors_vector(self):
        out = [arr(self.pose_prior.t), arr(self.pose_prior.R), arr(self.vb_prior.VB)]
        for i in range(self.Nvo - 1):
            out += [arr(self.relpose[i].delta_t), arr(self.relpose[i].delta_R)]
        for i in range(self.n_rollpitch):
            out += [arr(self.rollpitch[i].R)]
        return np.concatenate(out)
