"""Host-side mirror of include/isvins_estimator.h: the reference's window manager (processIMU / processImage /
slideWindow / FeatureManager bookkeeping, src/estimator.cpp:91-215, 1565-1724) for S sequences in lock step, every
solve on the MI355X backend.  `SequenceEstimator(...)` raises when the HIP extension is missing or there is no GPU;
`solver=` takes an isv_solver_vtbl_t and exists for the CPU unit tests of the host logic (tests/ inject the oracle).
"""
import ctypes as C

import numpy as np

from . import abi, backend


class isv_estimator_params_t(C.Structure):
    _fields_ = [("cfg", abi.isv_config_t), ("ric", C.c_double * 9), ("tic", C.c_double * 3),
                ("acc_n", C.c_double), ("gyr_n", C.c_double), ("acc_w", C.c_double), ("gyr_w", C.c_double),
                ("min_parallax", C.c_double)]


_wpp = C.POINTER(C.POINTER(abi.isv_window_t))
TRIANGULATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, _wpp)
INIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(abi.isv_window_t), C.POINTER(abi.isv_summary_t), C.POINTER(C.c_double))
INIT_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, _wpp, C.POINTER(abi.isv_summary_t), C.POINTER(C.c_double))
OPTIMIZE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, _wpp, C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t))


class isv_solver_vtbl_t(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("triangulate", TRIANGULATE_FN), ("init_factor_graph", INIT_FN), ("optimize_batch", OPTIMIZE_FN),
                ("init_factor_graph_batch", INIT_BATCH_FN), ("solve_odometry_batch", OPTIMIZE_FN)]


EXPORTS = ["isv_estimator_create", "isv_estimator_create_with_solver", "isv_estimator_destroy", "isv_estimator_last_error",
           "isv_estimator_process_imu", "isv_estimator_process_imu_n", "isv_estimator_last_step_ms", "isv_estimator_push_image", "isv_estimator_set_bootstrap", "isv_estimator_step",
           "isv_estimator_status", "isv_estimator_get_window", "isv_estimator_get_extrinsic", "isv_estimator_get_preintegration", "isv_estimator_last_summary", "isv_estimator_trajectory", "isv_estimator_failed_solves",
           "isv_estimator_set_resident", "isv_estimator_resident_frames", "isv_estimator_resident_fallbacks"]

_bound = False


def _bind(lib):
    global _bound
    if _bound:
        return
    vp, dp, ip = C.c_void_p, abi.c_double_p, C.POINTER(C.c_int32)
    lib.isv_estimator_create.argtypes = [C.POINTER(isv_estimator_params_t), C.c_int32, C.POINTER(vp)]
    lib.isv_estimator_create_with_solver.argtypes = [C.POINTER(isv_estimator_params_t), C.c_int32, C.POINTER(isv_solver_vtbl_t), C.POINTER(vp)]
    lib.isv_estimator_destroy.argtypes = [vp]; lib.isv_estimator_destroy.restype = None
    lib.isv_estimator_last_error.argtypes = [vp]; lib.isv_estimator_last_error.restype = C.c_char_p
    lib.isv_estimator_process_imu.argtypes = [vp, C.c_int32, C.c_double, dp, dp]
    lib.isv_estimator_process_imu_n.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp]
    lib.isv_estimator_last_step_ms.argtypes = [vp, dp]
    lib.isv_estimator_push_image.argtypes = [vp, C.c_int32, C.c_double, C.c_int32, ip, dp]
    lib.isv_estimator_set_bootstrap.argtypes = [vp, C.c_int32, dp, dp, dp]
    lib.isv_estimator_step.argtypes = [vp]
    lib.isv_estimator_status.argtypes = [vp, C.c_int32, ip]
    lib.isv_estimator_get_window.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, dp]
    lib.isv_estimator_get_extrinsic.argtypes = [vp, C.c_int32, dp, dp]
    lib.isv_estimator_get_preintegration.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(abi.isv_imu_t)]
    lib.isv_estimator_last_summary.argtypes = [vp, C.c_int32, C.POINTER(abi.isv_summary_t)]
    lib.isv_estimator_trajectory.argtypes = [vp, C.c_int32, C.c_int32, dp, C.c_int32]
    lib.isv_estimator_failed_solves.argtypes = [vp, C.c_int32]
    lib.isv_estimator_set_resident.argtypes = [vp, C.c_int32]
    lib.isv_estimator_resident_frames.argtypes = [vp]; lib.isv_estimator_resident_frames.restype = C.c_int64
    lib.isv_estimator_resident_fallbacks.argtypes = [vp]; lib.isv_estimator_resident_fallbacks.restype = C.c_int64
    _bound = True


def make_params(cfg, ric, tic, acc_n, gyr_n, acc_w, gyr_w, min_parallax):
    p = isv_estimator_params_t()
    C.memmove(C.byref(p.cfg), C.byref(cfg), C.sizeof(abi.isv_config_t))
    p.ric[:] = list(np.asarray(ric, float).ravel()); p.tic[:] = list(np.asarray(tic, float).ravel())
    p.acc_n, p.gyr_n, p.acc_w, p.gyr_w, p.min_parallax = acc_n, gyr_n, acc_w, gyr_w, min_parallax
    return p


class SequenceEstimator:
    STATUS_FIELDS = ("solver_flag", "frame_count", "margin_old", "n_tracks", "n_landmarks", "n_rollpitch", "n_solves", "iterations")

    def __init__(self, params, n_sequences=1, solver=None):
        self.lib = backend.load_library()
        _bind(self.lib)
        self.N = params.cfg.n_frames
        self.h = C.c_void_p()
        self._solver = solver            # keeps the callbacks alive
        if solver is None:
            rc = self.lib.isv_estimator_create(C.byref(params), n_sequences, C.byref(self.h))
        else:
            rc = self.lib.isv_estimator_create_with_solver(C.byref(params), n_sequences, C.byref(solver), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise backend.BackendError(f"isv_estimator_create: {backend.STATUS.get(rc, rc)} (a GPU and the HIP extension are required)")

    def close(self):
        if getattr(self, "h", None):
            self.lib.isv_estimator_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            msg = self.lib.isv_estimator_last_error(self.h)
            raise backend.BackendError(f"{what}: {backend.STATUS.get(rc, rc)} {msg.decode() if msg else ''}")
        return rc

    def process_imu(self, seq, dt, acc, gyr):
        a = np.ascontiguousarray(acc, float); g = np.ascontiguousarray(gyr, float)
        self._check(self.lib.isv_estimator_process_imu(self.h, seq, float(dt), abi._p(a), abi._p(g)), "process_imu")

    def process_imu_n(self, seq, dts, accs, gyrs):
        d = np.ascontiguousarray(dts, float); a = np.ascontiguousarray(accs, float).reshape(-1, 3); g = np.ascontiguousarray(gyrs, float).reshape(-1, 3)
        self._check(self.lib.isv_estimator_process_imu_n(self.h, seq, len(d), abi._p(d), abi._p(a), abi._p(g)), "process_imu_n")

    def last_step_ms(self):
        out = np.zeros(6)
        self._check(self.lib.isv_estimator_last_step_ms(self.h, abi._p(out)), "last_step_ms")
        return dict(zip(("step", "features_and_packing", "triangulate", "init_factor_graph", "optimize", "readback_and_slide"), out))

    def push_image(self, seq, header, ids, points):
        ids = np.ascontiguousarray(ids, np.int32); pts = np.ascontiguousarray(points, float).reshape(-1, 3)
        self._check(self.lib.isv_estimator_push_image(self.h, seq, float(header), len(ids), abi._p(ids, C.c_int32), abi._p(pts)), "push_image")

    def set_bootstrap(self, seq, Ps, Rs, Vs):
        P = np.ascontiguousarray(Ps, float); R = np.ascontiguousarray(Rs, float); V = np.ascontiguousarray(Vs, float)
        self._check(self.lib.isv_estimator_set_bootstrap(self.h, seq, abi._p(P), abi._p(R), abi._p(V)), "set_bootstrap")

    def step(self):
        return self._check(self.lib.isv_estimator_step(self.h), "step")

    def status(self, seq):
        out = np.zeros(8, np.int32)
        self._check(self.lib.isv_estimator_status(self.h, seq, abi._p(out, C.c_int32)), "status")
        return dict(zip(self.STATUS_FIELDS, (int(x) for x in out)))

    def window(self, seq):
        N = self.N
        Ps, Rs, Vs, Bas, Bgs, H = np.zeros((N, 3)), np.zeros((N, 3, 3)), np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3)), np.zeros(N)
        self._check(self.lib.isv_estimator_get_window(self.h, seq, abi._p(Ps), abi._p(Rs), abi._p(Vs), abi._p(Bas), abi._p(Bgs), abi._p(H)), "get_window")
        return dict(Ps=Ps, Rs=Rs, Vs=Vs, Bas=Bas, Bgs=Bgs, Headers=H)

    def extrinsic(self, seq):
        """(tic[0], ric[0]): configured, or as the last solve left them with cfg.estimate_extrinsic = 1"""
        tic, ric = np.zeros(3), np.zeros((3, 3))
        self._check(self.lib.isv_estimator_get_extrinsic(self.h, seq, abi._p(tic), abi._p(ric)), "get_extrinsic")
        return tic, ric

    def preintegration(self, seq, frame):
        out = abi.isv_imu_t()
        self._check(self.lib.isv_estimator_get_preintegration(self.h, seq, frame, C.byref(out)), "get_preintegration")
        return out

    def last_summary(self, seq):
        s = abi.isv_summary_t()
        self._check(self.lib.isv_estimator_last_summary(self.h, seq, C.byref(s)), "last_summary")
        return s

    def set_resident(self, on=True):
        """keep the windows on the device between frames (include/isvins_estimator.h)"""
        self._check(self.lib.isv_estimator_set_resident(self.h, 1 if on else 0), "set_resident")

    def resident_frames(self):
        return int(self.lib.isv_estimator_resident_frames(self.h))

    def resident_fallbacks(self):
        return int(self.lib.isv_estimator_resident_fallbacks(self.h))

    def failed_solves(self, seq):
        return self._check(self.lib.isv_estimator_failed_solves(self.h, seq), "failed_solves")

    def trajectory(self, seq, which=0):
        """which=0: pose_output.txt rows [n][8] (stamp p qw qx qy qz of the oldest frame); 1: newest frame [n][13]"""
        cols = 8 if which == 0 else 13
        n = self._check(self.lib.isv_estimator_trajectory(self.h, seq, which, None, 0), "trajectory")
        out = np.zeros((max(n, 1), cols))
        self._check(self.lib.isv_estimator_trajectory(self.h, seq, which, abi._p(out), n), "trajectory")
        return out[:n]
