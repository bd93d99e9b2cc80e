"""Host-side mirror of include/isvins_posegraph.h: the pose-graph optimisation that consumes the backend's
CombinedFactors (reference src/pose_graph/pose_graph.cpp:234-428 `PoseGraph::optimizeCS`,
include/factor/pose_graph_factors.h:27-51 `CombinedFactors::operator+`, `loop_pose_output.txt` :412-423).
`PoseGraphOptimizer(...)` raises when the HIP extension is missing or there is no GPU: no CPU path.

`make_pose_graph` builds deterministic synthetic keyframe lists (a drifting VIO trajectory with loop closures) for the
tests and the bench; loop DETECTION is not part of this package (it stays in the reference)."""
import ctypes as C

import numpy as np

from . import abi, backend, synth

ISV_MAX_TRACE = abi.ISV_MAX_TRACE


class isv_pg_keyframe_t(C.Structure):
    _fields_ = [("time_stamp", C.c_double), ("index", C.c_int32), ("sequence", C.c_int32), ("has_loop", C.c_int32), ("loop_index", C.c_int32),
                ("loop_info", C.c_double * 8), ("loop_weight", C.c_double),
                ("vio_T_w_i", C.c_double * 3), ("vio_R_w_i", C.c_double * 9), ("T_w_i", C.c_double * 3), ("R_w_i", C.c_double * 9),
                ("cov", C.c_double * 36), ("cov_computed", C.c_int32), ("has_rollpitch", C.c_int32),
                ("relative_pose", abi.isv_relpose_t), ("rollpitch", abi.isv_rollpitch_t)]


class isv_pgo_config_t(C.Structure):
    _fields_ = [("max_keyframes", C.c_int32), ("max_graphs", C.c_int32), ("max_loop_blocks", C.c_int32), ("max_iterations", C.c_int32),
                ("huber_delta", C.c_double)]


class isv_pgo_result_t(C.Structure):
    _fields_ = [("status", C.c_int32), ("termination", C.c_int32), ("iterations", C.c_int32), ("num_successful", C.c_int32),
                ("n_poses", C.c_int32), ("n_free", C.c_int32), ("n_loop_edges", C.c_int32), ("_pad", C.c_int32),
                ("initial_cost", C.c_double), ("final_cost", C.c_double), ("yaw_drift", C.c_double), ("r_drift", C.c_double * 9),
                ("t_drift", C.c_double * 3), ("trace_cost", C.c_double * ISV_MAX_TRACE), ("trace_accepted", C.c_int32 * ISV_MAX_TRACE)]


EXPORTS = ["isv_combined_factors_add", "isv_pgo_create", "isv_pgo_destroy", "isv_pgo_last_error", "isv_pgo_last_kernel_ms", "isv_pgo_structure_cache_hits", "isv_pgo_optimize",
           "isv_pgo_optimize_batch", "isv_pgo_write_loop_pose_output"]


def make_config(max_keyframes=1024, max_graphs=1, max_loop_blocks=None, max_iterations=10, huber_delta=0.1):
    c = isv_pgo_config_t()
    c.max_keyframes, c.max_graphs = max_keyframes, max_graphs
    c.max_loop_blocks = max_loop_blocks if max_loop_blocks is not None else 8 * max_keyframes
    c.max_iterations, c.huber_delta = max_iterations, huber_delta
    return c


_bound = False


def _bind(lib):
    global _bound
    if _bound:
        return
    vp = C.c_void_p
    kfp = C.POINTER(isv_pg_keyframe_t)
    lib.isv_combined_factors_add.argtypes = [C.POINTER(abi.isv_combined_factors_t), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                             C.POINTER(abi.isv_combined_factors_t), C.c_int64]
    lib.isv_pgo_create.argtypes = [C.POINTER(isv_pgo_config_t), C.POINTER(vp)]
    lib.isv_pgo_destroy.argtypes = [vp]; lib.isv_pgo_destroy.restype = None
    lib.isv_pgo_last_error.argtypes = [vp]; lib.isv_pgo_last_error.restype = C.c_char_p
    lib.isv_pgo_optimize.argtypes = [vp, C.c_int32, kfp, C.c_int32, C.c_int32, C.POINTER(isv_pgo_result_t)]
    lib.isv_pgo_optimize_batch.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(kfp), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                           C.POINTER(isv_pgo_result_t)]
    lib.isv_pgo_write_loop_pose_output.argtypes = [C.c_char_p, C.c_int32, kfp]
    _bound = True


class PoseGraphOptimizer:
    """PoseGraph::optimizeCS on the MI355X: one workgroup per pose graph"""

    def __init__(self, max_keyframes=1024, max_graphs=1, **kw):
        self.lib = backend.load_library()
        _bind(self.lib)
        self.cfg = make_config(max_keyframes, max_graphs, **kw)
        self.h = C.c_void_p()
        rc = self.lib.isv_pgo_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise backend.BackendError(f"isv_pgo_create: {backend.STATUS.get(rc, rc)} (a MI355X and the HIP extension are required; there is no CPU path)")

    def close(self):
        if getattr(self, "h", None):
            self.lib.isv_pgo_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.isv_pgo_last_error(self.h)
            raise backend.BackendError(f"{what}: {backend.STATUS.get(rc, rc)} {msg.decode() if msg else ''}")

    def optimize(self, kf, first_looped_index, cur_index):
        """kf: ctypes array of isv_pg_keyframe_t (updated in place) -> isv_pgo_result_t"""
        r = isv_pgo_result_t()
        self._check(self.lib.isv_pgo_optimize(self.h, len(kf), kf, first_looped_index, cur_index, C.byref(r)), "pgo_optimize")
        return r

    def last_kernel_ms(self):
        """(ms of k_pgo in the last optimize call, 6x6 skyline blocks of its graphs)"""
        ms, nb = C.c_double(), C.c_double()
        self.lib.isv_pgo_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        self._check(self.lib.isv_pgo_last_kernel_ms(self.h, C.byref(ms), C.byref(nb)), "pgo_last_kernel_ms")
        return ms.value, nb.value

    def structure_cache_hits(self):
        """graphs whose structure analysis was reused from the previous call on the same batch slot (since create)"""
        self.lib.isv_pgo_structure_cache_hits.argtypes = [C.c_void_p]; self.lib.isv_pgo_structure_cache_hits.restype = C.c_int64
        return int(self.lib.isv_pgo_structure_cache_hits(self.h))

    def optimize_batch(self, graphs, firsts, curs):
        n = len(graphs)
        ns = (C.c_int32 * n)(*[len(g) for g in graphs])
        ptrs = (C.POINTER(isv_pg_keyframe_t) * n)(*[C.cast(g, C.POINTER(isv_pg_keyframe_t)) for g in graphs])
        res = (isv_pgo_result_t * n)()
        self._check(self.lib.isv_pgo_optimize_batch(self.h, n, ns, ptrs, (C.c_int32 * n)(*firsts), (C.c_int32 * n)(*curs), res), "pgo_optimize_batch")
        return list(res)

    def write_loop_pose_output(self, path, kf):
        self._check(self.lib.isv_pgo_write_loop_pose_output(str(path).encode(), len(kf), kf), "write_loop_pose_output")


def combined_factors_add(acc, length, vio_index, other, other_vio_index, lib=None):
    """CombinedFactors::operator+ through the product library (host arithmetic): returns (length, vio_index)"""
    lib = lib or backend.load_library()
    _bind(lib)
    ln, vi = C.c_int32(length), C.c_int64(vio_index)
    rc = lib.isv_combined_factors_add(C.byref(acc), C.byref(ln), C.byref(vi), C.byref(other), other_vio_index)
    if rc != 0:
        raise backend.BackendError(f"isv_combined_factors_add: {backend.STATUS.get(rc, rc)}")
    return ln.value, vi.value


# ---------------------------------------------------------------------------------------------------------------------
def clone_keyframes(kf):
    out = (isv_pg_keyframe_t * len(kf))()
    C.memmove(out, kf, C.sizeof(kf))
    return out


def make_pose_graph(seed, n_keyframes=120, n_loops=3, drift=0.002, loop_noise=0.005, sequence=1, rollpitch_every=1):
    """A keyframe list as PoseGraphBuilder / PoseGraph::addKeyFrame leave it: truth on a closed figure-eight (so that
    places are revisited), VIO poses = the truth composed with an accumulating drift, keyfactor->relativePoseFactor =
    the relative VIO pose to the next keyframe with a dense upper-triangular sqrt_info, a RollPitchFactor per keyframe
    (gravity is observable: the VIO roll / pitch are nearly the truth), and n_loops loop closures whose relative pose
    comes from the truth + loop_noise.  Returns (ctypes keyframe array, truth positions [K,3], first_looped_index)."""
    rng = synth.SplitMix64(0x9E0_0000_0000 + int(seed))
    K = n_keyframes
    s = np.arange(K) / K * 2 * np.pi * 2                     # two laps: the second lap revisits the first
    P = np.stack([4.0 * np.sin(s), 2.0 * np.sin(2 * s), 0.3 * np.sin(3 * s)], 1)
    yaw = 0.8 * np.sin(s) + 0.3
    Rt = [synth._rot_zyx(yaw[k], 0.1 * np.sin(2 * s[k]), 0.08 * np.cos(s[k])) for k in range(K)]
    # drift: a slowly turning similarity-free error, integrated along the chain (4-dof dominant: yaw + translation)
    e = rng.normal(6 * K).reshape(K, 6)
    vioP, vioR = [P[0].copy()], [Rt[0].copy()]
    for k in range(1, K):
        dR = Rt[k - 1].T @ Rt[k]; dt = Rt[k - 1].T @ (P[k] - P[k - 1])
        dRn = dR @ synth._exp_so3(drift * np.array([0.1 * e[k, 3], 0.1 * e[k, 4], e[k, 5]]))
        dtn = dt + drift * 5.0 * e[k, :3] * np.linalg.norm(dt)
        vioP.append(vioP[-1] + vioR[-1] @ dtn); vioR.append(vioR[-1] @ dRn)
    kf = (isv_pg_keyframe_t * K)()
    z = rng.normal(40 * K).reshape(K, 40)

    def tri(n, d, zz):
        e2 = zz[: n * n].reshape(n, n)
        d = np.asarray(d, float)
        return (np.diag(d) + 0.05 * np.triu(e2, 1) * d[:, None]).ravel()

    for k in range(K):
        f = kf[k]
        f.time_stamp = 0.25 * k; f.index = k; f.sequence = sequence; f.has_loop = 0; f.loop_index = -1
        f.vio_T_w_i[:] = vioP[k]; f.vio_R_w_i[:] = vioR[k].ravel()
        f.T_w_i[:] = vioP[k]; f.R_w_i[:] = vioR[k].ravel()
        if k + 1 < K:
            rp = f.relative_pose
            rp.delta_t[:] = vioR[k].T @ (vioP[k + 1] - vioP[k]); rp.delta_R[:] = (vioR[k].T @ vioR[k + 1]).ravel()
            rp.sqrt_info[:] = tri(6, [200.0] * 3 + [500.0] * 3, z[k]); rp.imu_i, rp.imu_j = 0, 1
        else:
            f.relative_pose.delta_R[:] = np.eye(3).ravel()
        f.has_rollpitch = 1 if (k % rollpitch_every == 0) else 0
        # roll / pitch measured against gravity: the truth's, with the VIO yaw (the factor is blind to yaw)
        Rm = synth._rot_zyx(0.0, 0.1 * np.sin(2 * s[k]), 0.08 * np.cos(s[k]))
        f.rollpitch.R[:] = Rm.ravel(); f.rollpitch.sqrt_info[:] = tri(2, [300.0, 300.0], z[k, 36:]); f.rollpitch.index = 0
    # loop closures: keyframe j (second lap) re-observes keyframe i = j - K/2 (same place on the figure-eight)
    first = K
    u = rng.uniform(n_loops); ln = rng.normal(6 * max(n_loops, 1)).reshape(-1, 6)
    for m in range(n_loops):
        j = int(K // 2 + 2 + u[m] * (K // 2 - 6)); i = j - K // 2
        f = kf[j]
        f.has_loop = 1; f.loop_index = i
        rel_t = Rt[i].T @ (P[j] - P[i]) + loop_noise * ln[m, :3]
        rel_R = Rt[i].T @ Rt[j] @ synth._exp_so3(loop_noise * ln[m, 3:])
        q = _quat_wxyz(rel_R)
        f.loop_info[:] = [rel_t[0], rel_t[1], rel_t[2], q[0], q[1], q[2], q[3], 0.0]
        f.loop_weight = 40.0 + 20.0 * u[m]
        first = min(first, i)
    return kf, P, (first if n_loops else 0)


def _quat_wxyz(R):
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
