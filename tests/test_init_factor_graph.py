"""Estimator::initFactorGraph (src/estimator.cpp:667-1001): invariants of the oracle restatement (CPU) and the HIP
path against the oracle through the C ABI (GPU)."""
import ctypes as C

import numpy as np
import pytest

from isvins_amd import abi, synth


def _info(U, n):
    U = np.asarray(U).reshape(n, n)
    return U.T @ U


def _oracle_init(oracle, cfg, w):
    s = abi.isv_summary_t(); kld = np.zeros(1)
    assert oracle.isvo_init_factor_graph(C.byref(cfg), C.byref(w.c()), C.byref(s), abi._p(kld)) == 0
    w.n_rollpitch = 0; w.margin_old = 0
    return s, float(kld[0])


def test_oracle_init_factor_graph_invariants(oracle):
    w = synth.make_window(9, n_landmarks=80)
    cfg = abi.make_config(w.N, w.Nvo, max_landmarks=w.L, max_obs=w.n_obs, max_batch=1)
    w0 = w.clone()
    s, kld = _oracle_init(oracle, cfg, w)
    # the prior-free solve runs 3 x NUM_ITERATIONS at most and reduces the cost
    assert 1 <= s.iterations <= 3 * cfg.num_iterations and s.final_cost < 1e-3 * s.initial_cost
    # recovered factors: upper-triangular sqrt_info with positive diagonal (LLT(...).matrixL().transpose())
    for U, n in [(w.pose_prior.sqrt_info, 6), (w.vb_prior.sqrt_info, 9)] + [(w.relpose[i].sqrt_info, 6) for i in range(w.Nvo - 1)]:
        U = np.asarray(abi.arr(U)).reshape(n, n)
        assert np.allclose(np.tril(U, -1), 0) and (np.diag(U) > 0).all() and np.isfinite(U).all()
    # measurements are the solved estimate: zero residual for every recovered factor (modulo the yaw re-anchoring
    # of double2vector, which rotates the pose prior together with the states)
    assert w.pose_prior.index == 0 and w.vb_prior.index == w.Nvo - 1
    assert np.allclose(abi.arr(w.pose_prior.t), w.para_Pose.reshape(-1, 7)[0, :3], atol=1e-12)
    assert np.allclose(abi.arr(w.vb_prior.VB)[:6], w.para_SpeedBias.reshape(-1, 9)[w.Nvo - 1, :6], atol=1e-12)
    for i in range(w.Nvo - 1):
        assert (w.relpose[i].imu_i, w.relpose[i].imu_j) == (i, i + 1)
        R = np.asarray(abi.arr(w.relpose[i].delta_R)).reshape(3, 3)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12)
    # sparsification quality: the KLD against the truncated marginal is finite, non-negative and small
    assert np.isfinite(kld) and -1e-6 < kld < 5.0
    # inputs that are not outputs stay untouched
    assert np.array_equal(w.obs_point, w0.obs_point)


@pytest.mark.gpu
@pytest.mark.parametrize("n_frames,n_vo,n_lm", [(11, 5, 120), (18, 8, 200)])
def test_gpu_init_factor_graph_matches_oracle(oracle, n_frames, n_vo, n_lm):
    from isvins_amd import backend
    backend.build()
    w = synth.make_window(70, n_frames=n_frames, n_vo=n_vo, n_landmarks=n_lm)
    b = backend.Backend(n_frames, n_vo, max_landmarks=n_lm, max_obs=n_lm * n_frames, max_batch=1)
    try:
        o = w.clone()
        so, kld_o = _oracle_init(oracle, b.cfg, o)
        g = w.clone()
        sg, kld_g = b.init_factor_graph(g)
        n = so.iterations
        assert sg.iterations == n and sg.termination == so.termination
        assert list(sg.trace_accepted[: n + 1]) == list(so.trace_accepted[: n + 1])
        # the prior-free problem has gauge freedom (only the LM diagonal fixes it): 1e-6 on the cost trace
        assert np.allclose(np.array(sg.trace_cost[: n + 1]), np.array(so.trace_cost[: n + 1]), rtol=1e-6)
        for name in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
            assert np.abs(getattr(g, name) - getattr(o, name)).max() < 1e-6, name
        for (Ug, Uo, m) in [(g.pose_prior.sqrt_info, o.pose_prior.sqrt_info, 6), (g.vb_prior.sqrt_info, o.vb_prior.sqrt_info, 9)] + \
                [(g.relpose[i].sqrt_info, o.relpose[i].sqrt_info, 6) for i in range(n_vo - 1)]:
            a, e = _info(abi.arr(Ug), m), _info(abi.arr(Uo), m)
            assert np.abs(a - e).max() < 1e-5 * np.abs(e).max(), np.abs(a - e).max() / np.abs(e).max()
        assert np.abs(g.priors_vector() - o.priors_vector()).max() < 1e-6
        assert abs(kld_g - kld_o) < 1e-4 * max(1.0, abs(kld_o))
    finally:
        b.close()


@pytest.mark.gpu
def test_gpu_init_factor_graph_batch_equals_single():
    """isv_backend_init_factor_graph_batch (many sequences reaching their first solve in the same frame) gives every
    window the bits of the single-window call"""
    from isvins_amd import backend
    backend.build()
    ws = [synth.make_window(90 + i, n_landmarks=60 + 25 * i) for i in range(5)]
    be = backend.Backend(11, 5, max_landmarks=200, max_obs=max(w.n_obs for w in ws), max_batch=len(ws))
    try:
        a = [w.clone() for w in ws]
        sums, klds = be.init_factor_graph_batch(a)
        for i, w in enumerate(ws):
            b = w.clone()
            s1, k1 = be.init_factor_graph(b)
            assert np.array_equal(a[i].state_vector(), b.state_vector()) and np.array_equal(a[i].priors_vector(), b.priors_vector())
            assert np.array_equal(abi.arr(a[i].pose_prior.sqrt_info), abi.arr(b.pose_prior.sqrt_info))
            assert np.array_equal(abi.arr(a[i].vb_prior.sqrt_info), abi.arr(b.vb_prior.sqrt_info))
            assert sums[i].iterations == s1.iterations and sums[i].final_cost == s1.final_cost and klds[i] == k1
    finally:
        be.close()
