"""Host-side mirror of the reference call surface over the HIP C-ABI library.

`Backend.optimize(window)` is `Estimator::backendOptimization()` (reference
src/estimator.cpp:1541-1562, NON_LINEAR branch) for one window; `optimize_batch` the
multi-sequence form; `linearize` one `ceres::Problem::Evaluate`-equivalent pass of the factor
kernels.  There is NO CPU fallback: if the HIP extension is missing or no GPU is present the
constructor raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import abi

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# (ISVINS_LIB: A/B measurement hook -- another build of the same library, e.g. one compiled with -DISV_STAMP)
LIB_PATH = os.environ.get("ISVINS_LIB") or os.path.join(CSRC, "libisvins_hip.so")
_lib = None

STATUS = {0: "ISV_OK", -1: "ISV_ERR_INVALID_ARG", -2: "ISV_ERR_CAPACITY", -3: "ISV_ERR_NONFINITE",
          -4: "ISV_ERR_DEVICE", -5: "ISV_ERR_UNSUPPORTED"}
EXPORTS = ["isv_abi_version", "isv_backend_create", "isv_backend_destroy", "isv_backend_last_error",
           "isv_backend_optimize", "isv_backend_optimize_batch", "isv_backend_init_factor_graph", "isv_backend_init_factor_graph_batch", "isv_backend_triangulate", "isv_backend_solve_odometry_batch", "isv_backend_linearize",
           "isv_batch_upload", "isv_batch_optimize", "isv_batch_linearize", "isv_batch_download",
           "isv_batch_sync", "isv_batch_last_timing", "isv_batch_last_counts",
           "isv_result_record_doubles", "isv_batch_pack_results",
           "isv_backend_seq_enable", "isv_backend_seq_seed", "isv_backend_seq_frame", "isv_backend_seq_download", "isv_backend_seq_flush", "isv_backend_seq_marg"]


class BackendError(RuntimeError):
    pass


def build(force=False):
    """compile every HIP source for gfx950 into csrc/libisvins_hip.so (hipcc cross-compiles)"""
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:      # ranks of one node may call this at the same time
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force:
                subprocess.check_call(["make", "-s", "-C", CSRC, "clean"])
            subprocess.check_call(["make", "-s", "-C", CSRC])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(f"HIP extension missing: {LIB_PATH} (run __graft_entry__.build())")
    # PyTorch-ROCm bundles its own libamdhip64.  If this process is going to use torch as well (bench.py, smoke()),
    # torch's copy has to be mapped FIRST so that libisvins_hip.so binds to the same HIP runtime: with the system
    # runtime mapped first, the process ends up with two runtimes and the second one to initialise sees no device.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    wpp = C.POINTER(C.POINTER(abi.isv_window_t))
    dp = abi.c_double_p
    lib.isv_abi_version.restype = C.c_int
    lib.isv_backend_create.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(vp)]
    lib.isv_backend_destroy.argtypes = [vp]; lib.isv_backend_destroy.restype = None
    lib.isv_backend_last_error.argtypes = [vp]; lib.isv_backend_last_error.restype = C.c_char_p
    lib.isv_backend_optimize.argtypes = [vp, C.POINTER(abi.isv_window_t), C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t)]
    lib.isv_backend_optimize_batch.argtypes = [vp, C.c_int32, wpp, C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t)]
    lib.isv_backend_init_factor_graph.argtypes = [vp, C.POINTER(abi.isv_window_t), C.POINTER(abi.isv_summary_t), dp]
    lib.isv_backend_init_factor_graph_batch.argtypes = [vp, C.c_int32, wpp, C.POINTER(abi.isv_summary_t), dp]
    lib.isv_backend_triangulate.argtypes = [vp, C.c_int32, wpp]
    lib.isv_backend_solve_odometry_batch.argtypes = [vp, C.c_int32, wpp, C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t)]
    lib.isv_backend_linearize.argtypes = [vp, C.POINTER(abi.isv_window_t), dp, dp, dp]
    lib.isv_batch_upload.argtypes = [vp, C.c_int32, wpp]
    lib.isv_batch_optimize.argtypes = [vp, C.c_int32]
    lib.isv_batch_linearize.argtypes = [vp, C.c_int32]
    lib.isv_batch_download.argtypes = [vp, C.c_int32, wpp, C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t)]
    lib.isv_batch_sync.argtypes = [vp]
    lib.isv_batch_last_timing.argtypes = [vp, dp]
    lib.isv_batch_last_counts.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.isv_debug_read.argtypes = [vp, C.c_int32, dp, C.c_int64]
    lib.isv_result_record_doubles.argtypes = [vp]; lib.isv_result_record_doubles.restype = C.c_int64
    lib.isv_batch_pack_results.argtypes = [vp, vp, vp]
    _lib = lib
    return lib


class Backend:
    def __init__(self, n_frames=11, n_vo=5, max_landmarks=1000, max_obs=None, max_batch=1, **kw):
        self.lib = load_library()
        self.cfg = abi.make_config(n_frames, n_vo, max_landmarks=max_landmarks, max_obs=max_obs, max_batch=max_batch, **kw)
        self.h = C.c_void_p()
        rc = self.lib.isv_backend_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise BackendError(f"isv_backend_create failed: {STATUS.get(rc, rc)} (a MI355X and the HIP extension are required; there is no CPU path)")
        self._keep = None

    def close(self):
        if self.h:
            self.lib.isv_backend_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.isv_backend_last_error(self.h)
            raise BackendError(f"{what}: {STATUS.get(rc, rc)} {msg.decode() if msg else ''}")

    def _ptrs(self, windows):
        cs = [w.c() for w in windows]
        arr = (C.POINTER(abi.isv_window_t) * len(cs))(*[C.pointer(c) for c in cs])
        self._keep = (cs, arr, windows)
        return arr

    # ---- reference-surface calls ----------------------------------------------------------
    def optimize(self, window):
        """Estimator::backendOptimization() on one window (in place). -> (summary, marg)"""
        s, m = self.optimize_batch([window])
        return s[0], m[0]

    def optimize_batch(self, windows):
        n = len(windows)
        sums = (abi.isv_summary_t * n)(); margs = (abi.isv_marg_result_t * n)()
        self._check(self.lib.isv_backend_optimize_batch(self.h, n, self._ptrs(windows), sums, margs), "optimize_batch")
        return list(sums), list(margs)

    def linearize(self, window):
        """one evaluation of every residual block -> (proj_strips [F,28], imu_strips [N-1,465], cost)"""
        F, N = window.n_factors, window.N
        ps = np.zeros((max(F, 1), abi.ISV_PROJ_STRIP)); im = np.zeros((N - 1, abi.ISV_IMU_STRIP)); cost = np.zeros(1)
        cw = window.c()
        self._check(self.lib.isv_backend_linearize(self.h, C.byref(cw), abi._p(ps), abi._p(im), abi._p(cost)), "linearize")
        return ps[:F], im, float(cost[0])

    def init_factor_graph(self, window):
        """Estimator::initFactorGraph: prior-free solve + first prior factors (written into the window); returns (summary, kld)"""
        s = abi.isv_summary_t(); kld = np.zeros(1)
        self._check(self.lib.isv_backend_init_factor_graph(self.h, C.byref(window.c()), C.byref(s), abi._p(kld)), "init_factor_graph")
        window.n_rollpitch = 0; window.margin_old = 0        # (the C side reset them in its view of the window)
        return s, float(kld[0])

    def solve_odometry_batch(self, windows):
        """Estimator::solveOdometry: triangulate + backendOptimization with one hand-over. -> (summaries, margs)"""
        n = len(windows)
        sums = (abi.isv_summary_t * n)(); margs = (abi.isv_marg_result_t * n)()
        self._check(self.lib.isv_backend_solve_odometry_batch(self.h, n, self._ptrs(windows), sums, margs), "solve_odometry_batch")
        return list(sums), list(margs)

    def init_factor_graph_batch(self, windows):
        """Estimator::initFactorGraph for several windows at once. -> (summaries, klds)"""
        n = len(windows)
        sums = (abi.isv_summary_t * n)(); kld = np.zeros(n)
        self._check(self.lib.isv_backend_init_factor_graph_batch(self.h, n, self._ptrs(windows), sums, abi._p(kld)), "init_factor_graph_batch")
        for w in windows:
            w.n_rollpitch = 0; w.margin_old = 0              # (the C side reset them in its view of the windows)
        return list(sums), kld

    def triangulate(self, windows):
        """FeatureManager::triangulate for the landmarks without a positive depth (lm_depth updated in place)"""
        self._check(self.lib.isv_backend_triangulate(self.h, len(windows), self._ptrs(windows)), "triangulate")

    # ---- device-resident batch (bench) -----------------------------------------------------
    def marshal(self, windows):
        """the isv_window_t* array for `windows` (ctypes marshalling costs ~75 us per window in Python; a C++ caller
        has these pointers for free) - pass it as ptrs= to upload()/download() to keep that out of a timed region"""
        return self._ptrs(windows)

    def upload(self, windows, ptrs=None):
        self._check(self.lib.isv_batch_upload(self.h, len(windows), ptrs if ptrs is not None else self._ptrs(windows)), "upload")
        self._n = len(windows)

    def run_optimize(self, sync=True, profile=False):
        """profile=True: record HIP events around the dominant kernels (read with last_timing())"""
        self._check(self.lib.isv_batch_optimize(self.h, (1 if sync else 0) | (2 if profile else 0)), "batch_optimize")

    def run_linearize(self, sync=True):
        self._check(self.lib.isv_batch_linearize(self.h, 1 if sync else 0), "batch_linearize")

    def sync(self):
        self._check(self.lib.isv_batch_sync(self.h), "sync")

    def download(self, windows, ptrs=None, as_list=True):
        n = len(windows)
        sums = (abi.isv_summary_t * n)(); margs = (abi.isv_marg_result_t * n)()
        self._check(self.lib.isv_batch_download(self.h, n, ptrs if ptrs is not None else self._ptrs(windows), sums, margs), "download")
        return (list(sums), list(margs)) if as_list else (sums, margs)

    def record_doubles(self):
        return int(self.lib.isv_result_record_doubles(self.h))

    def pack_results(self, device_ptr, stream=None):
        """per-window result records of the resident batch into a caller-owned DEVICE buffer [n][record_doubles()],
        ordered after the handle's stream and enqueued on `stream`: a hipStream_t VALUE of the caller -- 0 is the legacy
        default stream (torch's default stream), a stream like any other; None = the handle's own stream
        (ISV_STREAM_OF_HANDLE), after which the caller orders its reads with sync()"""
        s = C.c_void_p(-1) if stream is None else C.c_void_p(int(stream))
        self._check(self.lib.isv_batch_pack_results(self.h, C.c_void_p(device_ptr), s), "pack_results")

    def last_timing(self):
        out = np.zeros(8)
        self._check(self.lib.isv_batch_last_timing(self.h, abi._p(out)), "last_timing")
        return out

    def last_counts(self):
        out = (C.c_int64 * 8)()
        self._check(self.lib.isv_batch_last_counts(self.h, out), "last_counts")
        return list(out)

    def debug_read(self, what, count):
        out = np.zeros(count)
        self._check(self.lib.isv_debug_read(self.h, what, abi._p(out), count), "debug_read")
        return out
