"""GPU parity of the full on-device solve (problemSolve + update() + double2vector) against the
CPU oracle, through the C ABI.  Summation order differs (owner-computes tree on the GPU vs the
oracle's sequential Ceres order), so results agree to rounding-amplified-by-conditioning, not bitwise:
  - per-iteration cost trace: 1e-7 relative (measured 7e-9 on the first, steepest step)
  - accept/reject pattern, iteration count, termination: identical
  - final states (Ps, Rs, Vs, Bas, Bgs, depths, priors): 1e-7 absolute on O(1) quantities
    (north_star asks ATE within 1e-6 m)"""
import ctypes as C

import numpy as np
import pytest

from isvins_amd import abi, backend, synth

pytestmark = pytest.mark.gpu


def oracle_run(oracle, cfg, w):
    o = w.clone()
    s = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
    assert oracle.isvo_optimize(C.byref(cfg), C.byref(o.c()), C.byref(s), C.byref(mg)) == 0
    return o, s, mg


@pytest.fixture(scope="module")
def be():
    backend.build()
    b = backend.Backend(11, 5, max_landmarks=400, max_obs=4400, max_batch=16)
    yield b
    b.close()


def check_window(o, so, g, sg, tol_state=1e-7, tol_cost=1e-7, tol_final=1e-9):
    n = so.iterations
    assert sg.iterations == n, (sg.iterations, n)
    assert sg.termination == so.termination
    assert list(sg.trace_accepted[: n + 1]) == list(so.trace_accepted[: n + 1])
    tc_o, tc_g = np.array(so.trace_cost[: n + 1]), np.array(sg.trace_cost[: n + 1])
    assert np.allclose(tc_g, tc_o, rtol=tol_cost), np.abs(tc_g / tc_o - 1).max()
    assert np.allclose(np.array(sg.trace_radius[: n + 1]), np.array(so.trace_radius[: n + 1]), rtol=1e-7)
    assert abs(sg.final_cost - so.final_cost) < tol_final * so.final_cost, abs(sg.final_cost / so.final_cost - 1)
    for name in ("Ps", "Rs", "Vs", "Bas", "Bgs", "para_Pose", "para_SpeedBias"):
        a, b = getattr(g, name), getattr(o, name)
        assert np.abs(a - b).max() < tol_state, (name, np.abs(a - b).max())
    if g.L:
        assert np.abs(g.lm_depth[: g.L] - o.lm_depth[: o.L]).max() < 1e-5 * max(1.0, np.abs(o.lm_depth[: o.L]).max())
    assert np.array_equal(g.lm_solve_flag[: g.L], o.lm_solve_flag[: o.L])
    assert np.abs(g.priors_vector() - o.priors_vector()).max() < tol_state


@pytest.mark.parametrize("wid", [0, 3])
def test_optimize_matches_oracle(oracle, be, wid):
    w = synth.make_window(wid)
    o, so, _ = oracle_run(oracle, be.cfg, w)
    g = w.clone()
    sg, _ = be.optimize(g)
    check_window(o, so, g, sg)


def test_batch_is_bitwise_single(oracle, be):
    """a window solved inside a ragged batch gives bitwise the result of solving it alone"""
    ws = synth.make_windows([10, 11], n_landmarks=150) + [synth.make_window(12, n_landmarks=60)]
    singles = []
    for w in ws:
        g = w.clone(); be.optimize(g); singles.append(g)
    batch = [w.clone() for w in ws]
    sums, _ = be.optimize_batch(batch)
    for a, b in zip(batch, singles):
        assert np.array_equal(a.state_vector(), b.state_vector())
    for w, g, s in zip(ws, batch, sums):
        o, so, _ = oracle_run(oracle, be.cfg, w)
        check_window(o, so, g, s)


def test_landmark_free_window(oracle, be):
    w = synth.make_window(5, n_landmarks=40)
    w0 = abi.Window(w.N, w.Nvo, 0, 0, w.n_rollpitch)
    for name in ("Ps", "Rs", "Vs", "Bas", "Bgs", "tic", "ric"):
        getattr(w0, name)[...] = getattr(w, name)
    C.memmove(w0.imu, w.imu, C.sizeof(w.imu)); C.memmove(w0.relpose, w.relpose, C.sizeof(w.relpose))
    C.memmove(w0.rollpitch, w.rollpitch, C.sizeof(w.rollpitch))
    C.memmove(C.byref(w0.pose_prior), C.byref(w.pose_prior), C.sizeof(w.pose_prior))
    C.memmove(C.byref(w0.vb_prior), C.byref(w.vb_prior), C.sizeof(w.vb_prior))
    o, so, _ = oracle_run(oracle, be.cfg, w0)
    g = w0.clone(); sg, _ = be.optimize(g)
    check_window(o, so, g, sg)


def _info(U, n):
    U = np.asarray(U).reshape(n, n)
    return U.T @ U


def check_marg(mo, mg, Nvo):
    """MargForward / MargBackward outputs.  The recovered factors' information matrices (sqrt_info^T
    sqrt_info) are compared, relative to their largest entry: 1e-6 (they go through an
    eigen-decomposition and several small inverses; the GPU uses Jacobi sweeps + Gauss-Jordan)."""
    assert mg.valid == 1 and mg.n_marg_landmarks == mo.n_marg_landmarks
    for name, n in (("forward_pose_prior", 6), ("backward_relpose", 6), ("backward_vb", 9), ("backward_rollpitch", 2)):
        a, b = _info(getattr(mg, name).sqrt_info, n), _info(getattr(mo, name).sqrt_info, n)
        assert np.abs(a - b).max() < 1e-6 * np.abs(b).max(), (name, np.abs(a - b).max() / np.abs(b).max())
        U = np.asarray(getattr(mg, name).sqrt_info).reshape(n, n)
        assert np.allclose(np.tril(U, -1), 0)
    a, b = _info(mg.combined.relative_pose.sqrt_info, 6), _info(mo.combined.relative_pose.sqrt_info, 6)
    assert np.abs(a - b).max() < 1e-6 * np.abs(b).max()
    for f in ("t", "R"):
        assert np.allclose(abi.arr(getattr(mg.forward_pose_prior, f)), abi.arr(getattr(mo.forward_pose_prior, f)), atol=1e-7)
    for obj_g, obj_o in ((mg.backward_relpose, mo.backward_relpose), (mg.combined.relative_pose, mo.combined.relative_pose)):
        assert np.allclose(abi.arr(obj_g.delta_t), abi.arr(obj_o.delta_t), atol=1e-7)
        assert np.allclose(abi.arr(obj_g.delta_R), abi.arr(obj_o.delta_R), atol=1e-7)
        assert (obj_g.imu_i, obj_g.imu_j) == (obj_o.imu_i, obj_o.imu_j)
    assert np.allclose(abi.arr(mg.backward_vb.VB), abi.arr(mo.backward_vb.VB), atol=1e-7)
    assert np.allclose(abi.arr(mg.backward_rollpitch.R), abi.arr(mo.backward_rollpitch.R), atol=1e-7)
    assert mg.backward_rollpitch.index == Nvo - 1 and mg.backward_vb.index == mo.backward_vb.index
    assert np.allclose(abi.arr(mg.combined.covRel), abi.arr(mo.combined.covRel), rtol=1e-5, atol=1e-6 * np.abs(abi.arr(mo.combined.covRel)).max())
    assert mg.combined.has_rollpitch == mo.combined.has_rollpitch
    assert np.allclose(abi.arr(mg.combined.covAbs), abi.arr(mo.combined.covAbs), rtol=1e-6)
    assert abs(mg.combined.distance - mo.combined.distance) < 1e-7 and mg.combined.ts == mo.combined.ts
    assert np.allclose(abi.arr(mg.combined.Ri), abi.arr(mo.combined.Ri), atol=1e-7) and np.allclose(abi.arr(mg.combined.ti), abi.arr(mo.combined.ti), atol=1e-7)
    assert abs(mg.forward_kld - mo.forward_kld) < 1e-6
    assert abs(mg.backward_kld - mo.backward_kld) < 1e-5 * max(1.0, abs(mo.backward_kld))


@pytest.mark.parametrize("wid", [0, 21])
def test_marginalisation_matches_oracle(oracle, be, wid):
    w = synth.make_window(wid)                      # margin_old = 1
    o, so, mo = oracle_run(oracle, be.cfg, w)
    g = w.clone()
    sg, mg = be.optimize(g)
    check_window(o, so, g, sg)
    check_marg(mo, mg, w.Nvo)


def test_margin_new_skips_marginalisation(oracle, be):
    w = synth.make_window(22, margin_old=0)
    g = w.clone()
    sg, mg = be.optimize(g)
    assert mg.valid == 0 and mg.n_marg_landmarks == 0


@pytest.mark.parametrize("n_frames,n_vo,n_lm", [(4, 2, 40), (6, 3, 80), (7, 4, 90), (8, 4, 100), (10, 6, 120), (11, 8, 150)])
def test_window_geometries_match_oracle(oracle, n_frames, n_vo, n_lm):
    """other window shapes on the LDS path: odd / even N (the two speed/bias chains have equal / unequal
    length), small N (fewer MFMA panel tiles, k_rank1_mfma<NT> variants), Nvo != 5 (more / fewer prior slots)"""
    b = backend.Backend(n_frames, n_vo, max_landmarks=n_lm, max_obs=n_lm * n_frames, max_batch=4)
    try:
        ws = synth.make_windows([40, 41], n_frames=n_frames, n_vo=n_vo, n_landmarks=n_lm)
        batch = [w.clone() for w in ws]
        sums, margs = b.optimize_batch(batch)
        for w, g, s, m in zip(ws, batch, sums, margs):
            o, so, mo = oracle_run(oracle, b.cfg, w)
            check_window(o, so, g, s)
            check_marg(mo, m, w.Nvo)
    finally:
        b.close()


@pytest.mark.parametrize("n_forced", [1, 2])
def test_forced_mu_retry_matches_oracle(oracle, monkeypatch, n_forced):
    """the mu x 10 retry of the Gauss-Newton solve (a failed factorisation) is never reached by well-posed windows,
    so both sides get the same fault injected: the first n factorisations of every iteration count as failed.  On
    the GPU the retry corrects the landmark part of the reduced system in place (T' = T - sum dc_l w_l w_l^T)
    instead of re-eliminating; the oracle re-eliminates from scratch like ceres."""
    monkeypatch.setenv("ISV_DEBUG_FORCE_RETRY", str(n_forced))
    b = backend.Backend(11, 5, max_landmarks=300, max_obs=3300, max_batch=2)      # the hook is read at create
    monkeypatch.delenv("ISV_DEBUG_FORCE_RETRY")
    oracle.isvo_debug_force_retry(n_forced)
    try:
        w = synth.make_window(7)
        o, so, _ = oracle_run(oracle, b.cfg, w)
        g = w.clone()
        sg, _ = b.optimize(g)
        check_window(o, so, g, sg)
    finally:
        oracle.isvo_debug_force_retry(0)
        b.close()


@pytest.mark.parametrize("n_frames,n_vo,n_lm", [(12, 5, 150), (15, 7, 200), (18, 8, 300), (19, 8, 150), (20, 8, 120)])
def test_long_windows_match_oracle(oracle, n_frames, n_vo, n_lm):
    """N > 11 on the LDS path (k_build_solve_sb<true>, one window per CU; k_rank1_mfma with several tiles per
    wavefront): the reference itself is compiled for ALL_BUF_SIZE = 18, Vo_SIZE = 8 (include/parameters.h:35-40)"""
    b = backend.Backend(n_frames, n_vo, max_landmarks=n_lm, max_obs=n_lm * n_frames, max_batch=4)
    try:
        ws = synth.make_windows([50, 51], n_frames=n_frames, n_vo=n_vo, n_landmarks=n_lm)
        batch = [w.clone() for w in ws]
        sums, margs = b.optimize_batch(batch)
        for w, g, s, m in zip(ws, batch, sums, margs):
            o, so, mo = oracle_run(oracle, b.cfg, w)
            check_window(o, so, g, s)
            check_marg(mo, m, w.Nvo)
    finally:
        b.close()


def test_ragged_batch_with_edge_cases_matches_oracle(oracle, be):
    """one batch of unequal windows, each carrying an edge the reference's backend has a branch for:
      - an IMU factor whose pre-integration spans more than 10 s is left out (src/estimator.cpp:1043)
      - a window without any roll/pitch factor (the first frames after initialisation)
      - a landmark whose optimised depth ends up negative gets solve_flag 2 (double2vector, src/estimator.cpp:806-822)
      - a window with a handful of landmarks next to full ones (ragged CSR offsets)"""
    ws = [synth.make_window(40 + i, n_landmarks=n) for i, n in enumerate((300, 7, 120, 301, 64))]
    ws[0].imu[4].sum_dt = 10.5
    ws[2].n_rollpitch = 0; ws[2].rollpitch = (abi.isv_rollpitch_t * 1)()
    # a landmark observed with the wrong sign of parallax: its depth optimises to the other side of the camera
    w = ws[3]; l = 5; o0 = w.lm_obs_ptr[l]; k = w.lm_obs_ptr[l + 1] - o0
    for o in range(1, k):
        w.obs_point[o0 + o, :2] = w.obs_point[o0, :2] + 3.0 * (w.obs_point[o0, :2] - w.obs_point[o0 + o, :2])
    w.lm_depth[l] = 0.3
    outs_o = [oracle_run(oracle, be.cfg, x) for x in ws]
    gs = [x.clone() for x in ws]
    sums, _ = be.optimize_batch(gs)
    for (o, so, _), g, sg in zip(outs_o, gs, sums):
        check_window(o, so, g, sg)
    assert outs_o[0][1].iterations >= 1
    flags = np.concatenate([g.lm_solve_flag[: g.L] for g in gs])
    assert set(np.unique(flags)) <= {1, 2}
    print("solve_flag==2 landmarks per window:", [int((g.lm_solve_flag[: g.L] == 2).sum()) for g in gs])


@pytest.mark.parametrize("n_frames,n_vo,n_lm", [(3, 2, 25), (5, 2, 60), (11, 10, 200), (11, 5, 1000)])
def test_extreme_shapes_match_oracle(oracle, n_frames, n_vo, n_lm):
    """the smallest window the ABI accepts (3 frames, Vo = 2), Vo at both ends of its range, and the reference's
    landmark capacity NUM_OF_F = 1000 (include/parameters.h:40) in one window"""
    w = synth.make_window(70 + n_frames + n_lm, n_frames=n_frames, n_vo=n_vo, n_landmarks=n_lm)
    backend.build()
    b = backend.Backend(n_frames, n_vo, max_landmarks=n_lm, max_obs=w.n_obs, max_batch=1)
    try:
        o, so, mo = oracle_run(oracle, b.cfg, w)
        g = w.clone(); sg, mg = b.optimize(g)
        check_window(o, so, g, sg)
        assert mg.valid == mo.valid and mg.n_marg_landmarks == mo.n_marg_landmarks
    finally:
        b.close()


@pytest.mark.parametrize("n_frames,n_vo,count", [(11, 5, 48), (18, 8, 16)])
def test_parity_sweep_over_many_windows(oracle, n_frames, n_vo, count):
    """many windows of mixed sizes in one batch, every one compared with the oracle: iteration counts, accept / reject
    patterns, terminations, cost traces, solved states, depth flags and prior updates (a wider net than the named
    cases above for rounding-induced accept / reject flips)"""
    rng = np.random.default_rng(11)
    sizes = [int(x) for x in rng.integers(12, 320, count)]
    ws = [synth.make_window(500 + i, n_frames=n_frames, n_vo=n_vo, n_landmarks=n) for i, n in enumerate(sizes)]
    backend.build()
    b = backend.Backend(n_frames, n_vo, max_landmarks=320, max_obs=max(w.n_obs for w in ws), max_batch=len(ws))
    try:
        gs = [w.clone() for w in ws]
        sums, margs = b.optimize_batch(gs)
        n_rejected = 0
        for w, g, sg, mg in zip(ws, gs, sums, margs):
            o, so, mo = oracle_run(oracle, b.cfg, w)
            check_window(o, so, g, sg)
            assert mg.valid == mo.valid and mg.n_marg_landmarks == mo.n_marg_landmarks
            n_rejected += sum(1 for k in range(1, so.iterations + 1) if so.trace_accepted[k] == 0)
        print("rejected steps across the sweep:", n_rejected)
    finally:
        b.close()
