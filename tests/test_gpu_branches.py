"""GPU parity (HIP path vs the CPU oracle, through the C ABI) of the reference branches and kernel variants
that ordinary synthetic windows never reach:

  - double2vector near pitch +-90 deg (src/estimator.cpp:537-547: rot_diff = Rs[0] R0^T instead of a pure yaw)
  - MargForward with a rank-deficient Lamda_prior (src/estimator.cpp:1304-1331)
  - the LINEAR_SOLVER / INVALID_STEPS / MIN_RADIUS terminations of ceres' TrustRegionMinimizer, with the SAME
    fault injected on both sides (the hooks are ceres Solver::Options / failure points, not changes of the algorithm)
  - k_build_solve<false>, the generic kernel windows of N = 21..32 frames run on
  - k_sweep_mfma with more than 64 (host, observer) pair groups on one wavefront, and its global-scratch variant
  - the prior factors' JACOBIANS, compared block by block (not only through the solve trace)
"""
import ctypes as C

import numpy as np
import pytest

from isvins_amd import abi, backend, synth
from test_gpu_solve import check_marg, check_window, oracle_run

pytestmark = pytest.mark.gpu
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


def _solve_pair(oracle, b, ws, marg=True):
    gs = [w.clone() for w in ws]
    sums, margs = b.optimize_batch(gs)
    for w, g, s, m in zip(ws, gs, sums, margs):
        o, so, mo = oracle_run(oracle, b.cfg, w)
        check_window(o, so, g, s)
        if marg:
            check_marg(mo, m, w.Nvo)
    return gs, sums, margs


# ---- k_sweep_mfma: more than 64 pair groups on one wavefront -------------------------------------
@pytest.mark.parametrize("n_frames,n_vo,n_lm", [(18, 8, 150), (18, 8, 300), (20, 8, 200)])
def test_sweep_with_more_than_64_groups_per_wavefront(oracle, n_frames, n_vo, n_lm):
    """every landmark hosted in frame 0 and tracked for at most 7 frames: 6 heavy (host, observer) pairs and ~150 empty
    ones, so the longest-first schedule piles > 64 groups onto single wavefronts (85 at N = 18 with 150 landmarks)"""
    ws = synth.make_windows([60, 61], n_frames=n_frames, n_vo=n_vo, n_landmarks=n_lm, host_frames=(0, 1), max_track=7)
    assert all((w.lm_start_frame[: w.L] == 0).all() for w in ws)
    b = backend.Backend(n_frames, n_vo, max_landmarks=n_lm, max_obs=max(w.n_obs for w in ws), max_batch=2)
    try:
        _solve_pair(oracle, b, ws)
    finally:
        b.close()


def test_sweep_global_scratch_variant_is_bitwise_the_lds_one(oracle, monkeypatch):
    """the variant of k_sweep_mfma that keeps its pair partials in a global scratch slice (N >= 12 and more than 256
    windows) against the LDS variant on the same windows: bitwise; and against the oracle.  ISV_DEBUG_SW_GLOBAL forces
    the variant on a small batch; the next test reaches it the natural way."""
    ws = synth.make_windows([70, 71, 72], n_frames=18, n_vo=8, n_landmarks=120)
    cap = dict(max_landmarks=120, max_obs=max(w.n_obs for w in ws), max_batch=3)
    monkeypatch.setenv("ISV_DEBUG_SW_GLOBAL", "1")
    bg = backend.Backend(18, 8, **cap)
    monkeypatch.delenv("ISV_DEBUG_SW_GLOBAL")
    bl = backend.Backend(18, 8, **cap)
    try:
        monkeypatch.setenv("ISV_DEBUG_SW_GLOBAL", "1")
        g_glob, _, _ = _solve_pair(oracle, bg, ws)
        monkeypatch.delenv("ISV_DEBUG_SW_GLOBAL")
        g_lds = [w.clone() for w in ws]
        bl.optimize_batch(g_lds)
        for a, c in zip(g_glob, g_lds):
            assert np.array_equal(a.state_vector(), c.state_vector())
    finally:
        bg.close(); bl.close()


def test_batch_of_more_than_256_long_windows(oracle):
    """260 windows of the reference's own shape (N = 18, Vo = 8) in one batch: the launch configuration large N >= 12
    batches get (sw_global, several workgroups per CU).  A sample is compared with the oracle, and every window is
    bitwise the same window solved in a batch of four on the same handle."""
    ids = list(range(800, 1060))
    ws = synth.make_windows(ids, n_frames=18, n_vo=8, n_landmarks=24)
    cap = dict(max_landmarks=24, max_obs=max(w.n_obs for w in ws))
    big = backend.Backend(18, 8, max_batch=len(ws), **cap)
    small = backend.Backend(18, 8, max_batch=4, **cap)
    try:
        gs = [w.clone() for w in ws]
        sums, margs = big.optimize_batch(gs)
        for k in (0, 1, 130, 259):
            o, so, mo = oracle_run(oracle, big.cfg, ws[k])
            check_window(o, so, gs[k], sums[k])
            check_marg(mo, margs[k], 8)
        for k0 in (0, 128, 256):
            # bitwise on the SAME handle (round 4: a handle of more than 256 long windows runs k_build_solve_st -- chosen per handle);
            # a four-window handle (k_build_solve_sb) agrees to rounding with identical control flow
            ref = [w.clone() for w in ws[k0: k0 + 4]]
            big.optimize_batch(ref)
            assert big.last_counts()[6] == 1
            for a, c in zip(gs[k0: k0 + 4], ref):
                assert np.array_equal(a.state_vector(), c.state_vector())
            ref2 = [w.clone() for w in ws[k0: k0 + 4]]
            s2, _ = small.optimize_batch(ref2)
            assert small.last_counts()[6] == 0
            for k, (a, c) in enumerate(zip(gs[k0: k0 + 4], ref2)):
                assert s2[k].iterations == sums[k0 + k].iterations
                assert np.abs(a.state_vector() - c.state_vector()).max() < 1e-7
    finally:
        big.close(); small.close()


def test_batch_beyond_one_resident_round_of_the_control_kernel(oracle):
    """the library picks kernel VARIANTS from the batch size -- eight-wavefront k_lin_gram while B <= CUs, the step
    control inside k_dogleg<true> while B fits one resident round of it (4 per CU), k_dogleg<false> + k_step_control beyond.
    1100 short windows of the benchmark's shape cross both thresholds: every sampled window is bitwise the same window
    solved in a batch of four ON THE SAME HANDLE (the small-batch variants), and a sample is compared with the oracle."""
    ids = list(range(2000, 3100))
    ws = synth.make_windows(ids, n_frames=11, n_vo=5, n_landmarks=16)
    cap = dict(max_landmarks=16, max_obs=max(w.n_obs for w in ws))
    big = backend.Backend(11, 5, max_batch=len(ws), **cap)
    small = backend.Backend(11, 5, max_batch=4, **cap)
    try:
        gs = [w.clone() for w in ws]
        sums, margs = big.optimize_batch(gs)
        assert big.last_counts()[5] == 0                  # the split control ran (more than one resident round)
        for k in (0, 555, 1099):
            o, so, mo = oracle_run(oracle, big.cfg, ws[k])
            check_window(o, so, gs[k], sums[k])
            check_marg(mo, margs[k], 5)
        for k0 in (0, 300, 1096):
            # the same HANDLE, a batch of four: the small-batch launch variants (eight-wavefront k_lin_gram, the step control inside
            # k_dogleg<true>) -- bitwise.  (Round 4: kernels whose sums differ in the last bits -- k_build_solve_st, the one-launch
            # MargBackward -- are chosen per handle from its max_batch, never from the uploaded batch size.)
            ref = [w.clone() for w in ws[k0: k0 + 4]]
            big.optimize_batch(ref)
            assert big.last_counts()[5] == 1 and big.last_counts()[6] == 1
            for a, c in zip(gs[k0: k0 + 4], ref):
                assert np.array_equal(a.state_vector(), c.state_vector())
            # a four-window handle runs k_build_solve_sb: same control flow, states to rounding
            ref2 = [w.clone() for w in ws[k0: k0 + 4]]
            s2, _ = small.optimize_batch(ref2)
            assert small.last_counts()[5] == 1 and small.last_counts()[6] == 0
            for k, (a, c) in enumerate(zip(gs[k0: k0 + 4], ref2)):
                assert s2[k].iterations == sums[k0 + k].iterations
                # (relative for entries beyond 1: a window of 16 landmarks can hold a depth of -634 m, whose two values differ by 3.7e-10 relative)
                va, vc = a.state_vector(), c.state_vector()
                assert (np.abs(va - vc) <= 1e-7 * np.maximum(1.0, np.abs(vc))).all(), np.abs(va - vc).max()
    finally:
        big.close(); small.close()


@pytest.mark.parametrize("env", [{"ISV_LEGACY_VISUAL": "1"}, {"ISV_SPLIT_CONTROL": "1"},
                                 {"ISV_LEGACY_VISUAL": "1", "ISV_SPLIT_CONTROL": "1"}, {"ISV_GENERIC_N": "1"},
                                 {"ISV_LG_BATCH_WAVES": "1"}])
def test_unfused_kernel_variants_against_oracle_and_fused(oracle, monkeypatch, env):
    """the round-1 kernels the library still ships (k_proj_linearize<0> + k_sweep_mfma instead of k_lin_gram, k_dogleg<false> +
    k_step_control instead of k_dogleg<true>; selected by environment for A/B measurements, and what N > 20 handles use),
    the run-time-N instantiation of k_build_solve_sb in place of the one compiled for 11 frames, and the four-wavefront
    k_lin_gram large batches use in place of the eight-wavefront one of small batches (bitwise: which wavefront sums a pair
    group does not change the sums):
    against the oracle, and against the fused kernels on the same windows (cost trace and states to rounding: the landmark
    sums are formed in a different association, so not bitwise)."""
    ws = synth.make_windows([90, 91, 92], n_frames=11, n_vo=5, n_landmarks=150)
    cap = dict(max_landmarks=150, max_obs=max(w.n_obs for w in ws), max_batch=3)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    bu = backend.Backend(11, 5, **cap)
    try:
        g_unf, s_unf, _ = _solve_pair(oracle, bu, ws)
        cnt = bu.last_counts()
        assert cnt[4] == (0 if "ISV_LEGACY_VISUAL" in env else 1)
        assert cnt[5] == (0 if "ISV_SPLIT_CONTROL" in env else 1)
    finally:
        bu.close()
    for k in env:
        monkeypatch.delenv(k)
    bf = backend.Backend(11, 5, **cap)
    try:
        g_fus = [w.clone() for w in ws]
        s_fus, _ = bf.optimize_batch(g_fus)
        assert tuple(bf.last_counts()[4:6]) == (1, 1)
        for a, c, sa, sc in zip(g_unf, g_fus, s_unf, s_fus):
            assert sa.iterations == sc.iterations and sa.termination == sc.termination
            if "ISV_LEGACY_VISUAL" in env or "ISV_GENERIC_N" in env:
                assert np.allclose(a.state_vector(), c.state_vector(), rtol=0, atol=1e-8)
            else:       # the step control is the same arithmetic in either kernel: the library switches by batch size
                assert np.array_equal(a.state_vector(), c.state_vector())
    finally:
        bf.close()


@pytest.mark.parametrize("N,Nvo", [(18, 8), (11, 5)])
def test_build_solve_st_compiled_for_18_frames_is_bitwise_the_runtime_n_kernel(monkeypatch, N, Nvo):
    """k_build_solve_st<true, 18> (the reference's ALL_BUF_SIZE as a compile-time constant: every loop bound and index a constant;
    404 -> 357 us per 1024-window launch) against k_build_solve_st<true, 0> (ISV_GENERIC_N=1) on a handle whose capacity selects
    the streamed kernel: the same sums in the same order, BITWISE equal states, summaries and marginalisation records.  The same for
    k_build_solve_st<false, 11> against <false, 0> (round 5: the two instantiations run the chain recurrence of the
    back-substitution in two loop forms -- isv_build_solve_st.hip -- with the same arithmetic per node)."""
    import ctypes
    ws = [synth.make_window(300 + i, n_frames=N, n_vo=Nvo, n_landmarks=120, margin_old=i % 2) for i in range(6)]
    cap = dict(max_landmarks=140, max_obs=max(w.n_obs for w in ws), max_batch=512)
    monkeypatch.setenv("ISV_SOLVE_ST", "1")
    out = {}
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("ISV_GENERIC_N", "1")
        b = backend.Backend(N, Nvo, **cap)
        try:
            g = [w.clone() for w in ws]
            sums, margs = b.optimize_batch(g)
            assert b.last_counts()[6] == 1          # k_build_solve_st ran
            out[generic] = ([w.state_vector() for w in g], [bytes(ctypes.string_at(ctypes.addressof(x), ctypes.sizeof(x))) for x in list(sums) + list(margs)])
        finally:
            b.close()
        if generic:
            monkeypatch.delenv("ISV_GENERIC_N")
    for a, c in zip(out[False][0], out[True][0]):
        assert np.array_equal(a, c)
    assert out[False][1] == out[True][1]


def test_marg_backward_one_launch_against_three(oracle, monkeypatch):
    """MargBackward as one kernel (k_marg_bwd<2>: the window's one wavefront runs the 21 x 21 Jacobi sweeps, what batches
    beyond n_cus windows take) against build / k_marg_jacobi<21> (four wavefronts, one item per lane) / project (small
    batches): the same rotations in the same order with the same arithmetic per item, so the records are BYTEWISE equal;
    and either against the oracle."""
    import ctypes
    ws = synth.make_windows([70, 71, 72, 73, 74], n_frames=11, n_vo=5, n_landmarks=120)
    cap = dict(max_landmarks=120, max_obs=max(w.n_obs for w in ws), max_batch=5)
    recs = {}
    for env in ("ISV_MARG_ONE_KERNEL", "ISV_MARG_SPLIT"):
        monkeypatch.setenv(env, "1")
        b = backend.Backend(11, 5, **cap)
        try:
            _, _, margs = _solve_pair(oracle, b, ws)
            recs[env] = [bytes(ctypes.string_at(ctypes.addressof(m), ctypes.sizeof(m))) for m in margs]
        finally:
            b.close()
        monkeypatch.delenv(env)
    assert recs["ISV_MARG_ONE_KERNEL"] == recs["ISV_MARG_SPLIT"]


# ---- double2vector near pitch +-90 deg ------------------------------------------------------------------------------
def _pitch_deg(R):
    """Utility::R2ypr (include/utility/utility.h:66-81), pitch in degrees"""
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = np.arctan2(n[1], n[0])
    return np.degrees(np.arctan2(-n[2], n[0] * np.cos(y) + n[1] * np.sin(y)))


@pytest.mark.parametrize("pitch0,n_frames,n_vo", [(89.7, 11, 5), (-89.6, 11, 5), (89.8, 18, 8)])
def test_double2vector_near_pitch_90(oracle, pitch0, n_frames, n_vo):
    """a body that looks straight up / down: |pitch(Rs[0])| within 1 deg of 90, where double2vector re-anchors with the
    full rotation Rs[0] R0^T instead of the yaw difference (src/estimator.cpp:537-547)"""
    ws = synth.make_windows([90, 91, 92, 93], n_frames=n_frames, n_vo=n_vo, n_landmarks=100, pitch0_deg=pitch0, pitch_amp_deg=0.0)
    # (the generator perturbs the initial attitude by ~0.5 deg: keep the windows whose Rs[0] is inside the 1 deg band)
    ws = [w for w in ws if abs(abs(_pitch_deg(w.Rs[0])) - 90.0) < 0.9][:2]
    assert len(ws) >= 1
    b = backend.Backend(n_frames, n_vo, max_landmarks=100, max_obs=max(w.n_obs for w in ws), max_batch=2)
    try:
        gs, _, _ = _solve_pair(oracle, b, ws)
        # the branch really is a different rotation: a pure-yaw re-anchoring would leave another Rs[0]
        for w, g in zip(ws, gs):
            assert np.abs(g.Rs[0] - w.Rs[0]).max() < 1e-9          # full re-anchoring restores Rs[0] exactly
    finally:
        b.close()


# ---- MargForward, rank-deficient Lamda_prior ------------------------------------------------------------------------
def test_marg_forward_rank_deficient_prior(oracle):
    """No landmark is hosted in frame 0 and the relative-pose edge (0, 1) carries no rotation information, so the
    marginal on pose 1 has rank 3 and both sides take the eigen-truncation branch (src/estimator.cpp:1311-1331).
    In that branch Sigma = (J U) D^-1 (J U)^T is singular BY CONSTRUCTION and the reference's own next line inverts it
    (`covi.inverse()`, :1332): its sqrt_info is not finite there.  Parity is therefore: the same branch (the
    zero-test / KLD is only computed on the full-rank branch), the same non-finite pattern in the recovered prior,
    and every other output of the marginalisation equal as usual."""
    ws = synth.make_windows([95, 96], n_frames=11, n_vo=5, n_landmarks=80, host_frames=(1, 5))
    for w in ws:
        s = abi.arr(w.relpose[0].sqrt_info, (6, 6)); s[3:, :] = 0.0; s[:, 3:] = 0.0
        w.relpose[0].sqrt_info[:] = s.ravel()
    b = backend.Backend(11, 5, max_landmarks=80, max_obs=max(w.n_obs for w in ws), max_batch=2)
    try:
        gs = [w.clone() for w in ws]
        sums, margs = b.optimize_batch(gs)
        for w, g, s, mg in zip(ws, gs, sums, margs):
            o, so, mo = oracle_run(oracle, b.cfg, w)
            check_window(o, so, g, s)
            assert mg.valid == 1 and mo.valid == 1 and mg.n_marg_landmarks == mo.n_marg_landmarks == 0
            assert mg.forward_kld == 0.0 and mo.forward_kld == 0.0          # the rank < 6 branch on both sides
            fo, fg = abi.arr(mo.forward_pose_prior.sqrt_info), abi.arr(mg.forward_pose_prior.sqrt_info)
            assert not np.isfinite(fo).all() and not np.isfinite(fg).all()
            assert np.allclose(abi.arr(mg.forward_pose_prior.t), abi.arr(mo.forward_pose_prior.t), atol=1e-7)
            assert np.allclose(abi.arr(mg.forward_pose_prior.R), abi.arr(mo.forward_pose_prior.R), atol=1e-7)
            # MargBackward does not depend on it
            for name, n in (("backward_relpose", 6), ("backward_vb", 9), ("backward_rollpitch", 2)):
                A = abi.arr(getattr(mg, name).sqrt_info, (n, n)); Bm = abi.arr(getattr(mo, name).sqrt_info, (n, n))
                assert np.abs(A.T @ A - Bm.T @ Bm).max() < 1e-6 * np.abs(Bm.T @ Bm).max()
    finally:
        b.close()


# ---- terminations ---------------------------------------------------------------------------------------------------
def _hooked_backend(monkeypatch, env, **kw):
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    b = backend.Backend(11, 5, max_landmarks=80, max_obs=880, max_batch=2, **kw)      # the hooks are read at create
    for k in env:
        monkeypatch.delenv(k)
    return b


def test_termination_linear_solver_failure(oracle, monkeypatch):
    """every factorisation fails (fault injected on both sides): mu climbs to max_mu = 1, the step is invalid, and the
    fifth consecutive invalid step ends the solve with FAILURE (trust_region_minimizer.cc HandleInvalidStep)"""
    b = _hooked_backend(monkeypatch, {"ISV_DEBUG_FORCE_RETRY": 20})
    oracle.isvo_debug_force_retry(20)
    try:
        w = synth.make_window(7, n_landmarks=80)
        o, so, _ = oracle_run(oracle, b.cfg, w)
        g = w.clone(); sg, _ = b.optimize(g)
        assert so.termination == 7 and sg.termination == 7 and sg.iterations == so.iterations == 5
        assert sg.num_successful == 0
        check_window(o, so, g, sg)
        assert np.abs(g.Ps - w.Ps).max() < 1e-12                    # nothing was accepted: the state is the input
    finally:
        oracle.isvo_debug_force_retry(0); b.close()


@pytest.mark.parametrize("n_forced,term", [(5, 6), (2, None)])
def test_termination_invalid_steps(oracle, monkeypatch, n_forced, term):
    """the first n trust-region steps count as invalid (model_cost_change <= 0): five in a row end the solve
    (INVALID_STEPS); two make DoglegStrategy::StepIsInvalid raise mu twice, then the solve goes on as usual"""
    b = _hooked_backend(monkeypatch, {"ISV_DEBUG_FORCE_INVALID": n_forced})
    oracle.isvo_debug_force_invalid(n_forced)
    try:
        w = synth.make_window(9, n_landmarks=80)
        o, so, _ = oracle_run(oracle, b.cfg, w)
        g = w.clone(); sg, _ = b.optimize(g)
        if term is not None:
            assert so.termination == term and sg.termination == term and sg.iterations == 5
        else:
            assert so.num_successful > 0 and list(so.trace_accepted[1:3]) == [0, 0]
        check_window(o, so, g, sg)
    finally:
        oracle.isvo_debug_force_invalid(0); b.close()


def test_termination_min_radius(oracle, monkeypatch):
    """min_trust_region_radius raised from 1e-32 to 8000 on both sides: window 5 rejects its steps 3 and 4
    (radius 30000 -> 15000 -> 7500) and stops with 'trust region radius too small' after the fourth iteration"""
    b = _hooked_backend(monkeypatch, {"ISV_DEBUG_MIN_RADIUS": 8000.0})
    oracle.isvo_debug_min_radius(8000.0)
    try:
        w = synth.make_window(5, n_landmarks=80)
        o, so, _ = oracle_run(oracle, b.cfg, w)
        g = w.clone(); sg, _ = b.optimize(g)
        assert so.termination == 5 and sg.termination == 5 and so.iterations == sg.iterations == 4
        # the solve is cut off in mid-descent, so its "final" cost carries the sensitivity of an intermediate trace entry (2.9e-8
        # relative on this window) instead of a converged minimum's: measured 8.4e-10 with the one-launch reduced-system solve and
        # 1.06e-9 with the split solve of small-batch handles (scripts/split_accuracy.py: the two agree with the oracle equally
        # well on every other figure -- 5.8e-10 / 5.4e-10 on converged final costs, 6.3e-10 / 6.4e-10 on the states)
        check_window(o, so, g, sg, tol_final=5e-9)
    finally:
        oracle.isvo_debug_min_radius(0.0); b.close()


def test_a_window_has_the_same_bits_alone_and_beside_a_32_pass_window(oracle):
    """whether a window's rank-1 downdates are split over workgroups is decided by the HANDLE (max_batch x groups of
    max_landmarks against the CUs), never by the longest window of the upload: a 320-landmark window (5 passes of 64) solved
    alone and beside a 2048-landmark window (32 passes) on one handle gives the same bits; and both agree with the oracle."""
    w_small = synth.make_window(310, n_frames=11, n_vo=5, n_landmarks=320)
    w_long = synth.make_window(311, n_frames=11, n_vo=5, n_landmarks=2048, max_track=4)
    assert w_long.n_factors <= 8192
    b = backend.Backend(11, 5, max_landmarks=2048, max_obs=max(w_small.n_obs, w_long.n_obs), max_batch=2)
    try:
        alone = w_small.clone(); b.optimize_batch([alone])
        assert b.last_counts()[7] > 0                       # the split elimination ran
        pair = [w_small.clone(), w_long.clone()]
        sums, margs = b.optimize_batch(pair)
        assert b.last_counts()[7] > 0
        assert np.array_equal(alone.state_vector(), pair[0].state_vector())
        for w, g, s_, m in zip((w_small, w_long), pair, sums, margs):
            o, so, mo = oracle_run(oracle, b.cfg, w)
            check_window(o, so, g, s_)
    finally:
        b.close()


# ---- the generic kernel ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_frames,n_vo,n_lm", [(24, 8, 120), (21, 10, 60), (32, 8, 40)])
def test_generic_kernel_for_windows_longer_than_20_frames(oracle, n_frames, n_vo, n_lm):
    """N = 21..32 (ISV_MAX_FRAMES): k_build_solve<false> with the reduced system in a global scratch"""
    ws = synth.make_windows([110, 111], n_frames=n_frames, n_vo=n_vo, n_landmarks=n_lm)
    b = backend.Backend(n_frames, n_vo, max_landmarks=n_lm, max_obs=max(w.n_obs for w in ws), max_batch=2)
    try:
        _solve_pair(oracle, b, ws)
    finally:
        b.close()


# ---- prior Jacobians, block by block --------------------------------------------------------------------------------
def _corrected(r, Js):
    """ceres Corrector for CauchyLoss(1.0): residual and Jacobians scaled by sqrt(rho') = 1 / sqrt(1 + |r|^2)"""
    sc = 1.0 / np.sqrt(1.0 + float(r @ r))
    return r * sc, [J * sc for J in Js]


@pytest.mark.parametrize("n_frames,n_vo,perturb", [(11, 5, False), (11, 5, True), (18, 8, True)])
def test_prior_jacobians_match_oracle(oracle, n_frames, n_vo, perturb):
    """prior_strip = [se3: r6 J 6x6][lin9: r9 J 9x9][relpose k: r6 Ji 6x6 Jj 6x6]...[rollpitch m: r2 J 2x6] against
    SE3PriorFactor / Linear9Factor / RelativePoseFactor / RollPitchFactor::Evaluate of the oracle (weighted by sqrt_info,
    local parameterisation = first 6 of the 7 pose columns, Cauchy corrector)"""
    w = synth.make_window(120, n_frames=n_frames, n_vo=n_vo, n_landmarks=40)
    if perturb:                      # away from the prior measurements: non-zero residuals, J_r^-1 != I
        rng = np.random.default_rng(3)
        from scipy.spatial.transform import Rotation as Rot
        w.Ps += 0.03 * rng.normal(size=w.Ps.shape); w.Vs += 0.03 * rng.normal(size=w.Vs.shape)
        w.Bas += 0.01 * rng.normal(size=w.Bas.shape)
        for i in range(w.N):
            w.Rs[i] = w.Rs[i] @ Rot.from_rotvec(0.02 * rng.normal(size=3)).as_matrix()
    b = backend.Backend(n_frames, n_vo, max_landmarks=40, max_obs=w.n_obs, max_batch=1)
    try:
        b.linearize(w)
        strip = b.debug_read(0, 132 + 78 * (n_vo - 1) + 14 * b.cfg.max_rollpitch)
        pose = np.zeros((w.N, 7)); sb = np.zeros((w.N, 9))
        for i in range(w.N):
            q = np.zeros(4); oracle.isvo_x_R2q(P(np.ascontiguousarray(w.Rs[i])), P(q))
            pose[i, :3] = w.Ps[i]; pose[i, 3:] = q
            sb[i] = np.concatenate([w.Vs[i], w.Bas[i], w.Bgs[i]])

        def close(got, want, what):
            sc = max(1.0, np.abs(want).max())
            assert np.abs(got - want).max() < 1e-9 * sc, (what, np.abs(got - want).max() / sc)

        r = np.zeros(6); J = np.zeros((6, 7))
        oracle.isvo_x_se3prior(C.byref(w.pose_prior), 1, P(pose[0]), P(r), P(J))
        r, (J,) = _corrected(r, [J[:, :6]])
        close(strip[0:6], r, "se3 r"); close(strip[6:42].reshape(6, 6), J, "se3 J")
        r = np.zeros(9); J = np.zeros((9, 9))
        oracle.isvo_x_linear9(C.byref(w.vb_prior), 1, P(sb[n_vo - 1]), P(r), P(J))
        r, (J,) = _corrected(r, [J])
        close(strip[42:51], r, "lin9 r"); close(strip[51:132].reshape(9, 9), J, "lin9 J")
        for k in range(n_vo - 1):
            r = np.zeros(6); Ji = np.zeros((6, 7)); Jj = np.zeros((6, 7))
            oracle.isvo_x_relpose(C.byref(w.relpose[k]), 1, P(pose[k]), P(pose[k + 1]), P(r), P(Ji), P(Jj))
            r, (Ji, Jj) = _corrected(r, [Ji[:, :6], Jj[:, :6]])
            o0 = 132 + 78 * k
            close(strip[o0: o0 + 6], r, f"relpose {k} r")
            close(strip[o0 + 6: o0 + 42].reshape(6, 6), Ji, f"relpose {k} Ji")
            close(strip[o0 + 42: o0 + 78].reshape(6, 6), Jj, f"relpose {k} Jj")
        base = 132 + 78 * (n_vo - 1)
        for m in range(w.n_rollpitch):
            r = np.zeros(2); J = np.zeros((2, 7))
            oracle.isvo_x_rollpitch(C.byref(w.rollpitch[m]), 1, P(pose[w.rollpitch[m].index]), P(r), P(J))
            r, (J,) = _corrected(r, [J[:, :6]])
            close(strip[base + 14 * m: base + 14 * m + 2], r, f"rollpitch {m} r")
            close(strip[base + 14 * m + 2: base + 14 * m + 14].reshape(2, 6), J, f"rollpitch {m} J")
        if perturb:
            assert np.abs(strip[0:6]).max() > 1e-3
    finally:
        b.close()


# ---- one long window over many compute units (round 4: k_schur_split + k_schur_fold) ----------------------------------
@pytest.mark.parametrize("shape", ["unfused", "fused", "batch_mixed"])
def test_split_landmark_elimination_against_oracle_and_one_workgroup(oracle, monkeypatch, shape):
    """A small batch of LONG windows spreads each window's landmark elimination over Gs + Gr workgroups (src/estimator.cpp:1057-1092
    + DENSE_SCHUR :1121 for one window): `unfused` = more than ISV_FUSED_MAX_FACTORS factors (the direct part is split over pair
    groups too, bit for bit the one-workgroup partials), `fused` = k_lin_gram + split rank-1 downdates, `batch_mixed` = a long and
    a short window in one batch (the short one is ONE group: the unsplit bits).  Against the oracle (trace, accept pattern,
    states 1e-7) and against the same handle shape with ISV_NO_SPLIT=1 (the one-workgroup kernels: same control flow, states
    within 1e-9 -- only the downdate's partial sums differ)."""
    from test_gpu_solve import check_marg, check_window, oracle_run
    if shape == "unfused":
        ws = [synth.make_window(21, n_frames=14, n_vo=6, n_landmarks=900, target_factors=9000)]
        N, Nvo = 14, 6
    elif shape == "fused":
        ws = [synth.make_window(22, n_frames=11, n_vo=5, n_landmarks=640)]
        N, Nvo = 11, 5
    else:
        ws = [synth.make_window(23, n_frames=11, n_vo=5, n_landmarks=520), synth.make_window(24, n_frames=11, n_vo=5, n_landmarks=90)]
        N, Nvo = 11, 5
    L = max(w.L for w in ws); nobs = max(w.n_obs for w in ws)
    be = backend.Backend(N, Nvo, max_landmarks=L, max_obs=nobs, max_batch=len(ws))
    monkeypatch.setenv("ISV_NO_SPLIT", "1")
    one = backend.Backend(N, Nvo, max_landmarks=L, max_obs=nobs, max_batch=len(ws))
    monkeypatch.delenv("ISV_NO_SPLIT")
    try:
        gs = [w.clone() for w in ws]; sums, margs = be.optimize_batch(gs)
        cnt = be.last_counts()
        assert cnt[7] >= 8, cnt                                    # split groups of the longest window (>= 512 landmarks here)
        assert (cnt[4] == 0) == (shape == "unfused"), cnt
        g1 = [w.clone() for w in ws]; sums1, _ = one.optimize_batch(g1)
        assert one.last_counts()[7] == 0
        for w, g, s, mg, h, s1 in zip(ws, gs, sums, margs, g1, sums1):
            o, so, mo = oracle_run(oracle, be.cfg, w)
            check_window(o, so, g, s); check_marg(mo, mg, Nvo)
            assert s.iterations == s1.iterations and list(s.trace_accepted[: s.iterations + 1]) == list(s1.trace_accepted[: s1.iterations + 1])
            if w.L <= 192:                                         # fewer than four 64-landmark passes: one group, the one-workgroup sums bit for bit
                assert np.array_equal(g.state_vector(), h.state_vector())
            else:
                assert np.abs(g.state_vector() - h.state_vector()).max() < 1e-8          # (depths of O(10) included)
        # run-to-run reproducible (fixed-order fold, no atomics)
        again = [w.clone() for w in ws]; be.optimize_batch(again)
        for a, g in zip(again, gs):
            assert np.array_equal(a.state_vector(), g.state_vector())
    finally:
        be.close(); one.close()


# ---- k_build_solve_st: the four-windows-per-CU linear solve (round 4) -------------------------------------------------
@pytest.mark.parametrize("N,Nvo,L,ex", [(11, 5, 120, 0), (6, 3, 60, 0), (18, 8, 150, 0), (16, 7, 200, 0), (11, 5, 100, 1), (9, 4, 80, 0)])
def test_streamed_linear_solve_against_oracle_and_k_build_solve_sb(oracle, monkeypatch, N, Nvo, L, ex):
    """k_build_solve_st (isv_build_solve_st.hip; DENSE_SCHUR linear solve of src/estimator.cpp:1119-1128) streams the speed/bias
    chain blocks through small LDS rings and downdates the pose system node by node, so that four windows share a CU.  It is
    chosen per HANDLE (max_batch > 2 n_cus at N <= 11); here it is forced on a small handle (ISV_SOLVE_ST=1) and compared with the
    oracle (trace, accept pattern, states 1e-7, marginalisation) and with k_build_solve_sb on the same windows (ISV_SOLVE_ST=0:
    identical control flow, states within 1e-8 -- the two kernels sum in different orders)."""
    from test_gpu_solve import check_marg, check_window, oracle_run
    ws = synth.make_windows([31, 32, 33], n_frames=N, n_vo=Nvo, n_landmarks=L)
    kw = dict(max_landmarks=L, max_obs=max(w.n_obs for w in ws), max_batch=3, estimate_extrinsic=ex)
    monkeypatch.setenv("ISV_SOLVE_ST", "1")
    be = backend.Backend(N, Nvo, **kw)
    monkeypatch.setenv("ISV_SOLVE_ST", "0")
    sb = backend.Backend(N, Nvo, **kw)
    monkeypatch.delenv("ISV_SOLVE_ST")
    try:
        gs = [w.clone() for w in ws]; sums, margs = be.optimize_batch(gs)
        assert be.last_counts()[6] == 1
        hs = [w.clone() for w in ws]; sums1, _ = sb.optimize_batch(hs)
        assert sb.last_counts()[6] == 0
        for w, g, s, mg, h, s1 in zip(ws, gs, sums, margs, hs, sums1):
            o, so, mo = oracle_run(oracle, be.cfg, w)
            check_window(o, so, g, s); check_marg(mo, mg, Nvo)
            assert s.iterations == s1.iterations and list(s.trace_accepted[: s.iterations + 1]) == list(s1.trace_accepted[: s1.iterations + 1])
            assert np.abs(g.state_vector() - h.state_vector()).max() < 1e-7          # (depths of O(10) included: 1.5e-8 measured)
        # a window alone on the same handle: the same bits as inside the batch
        a = ws[1].clone(); be.optimize(a)
        assert np.array_equal(a.state_vector(), gs[1].state_vector())
    finally:
        be.close(); sb.close()


def test_streamed_linear_solve_forced_mu_retry(oracle, monkeypatch):
    """the retry branch of k_build_solve_st (a failed factorisation retries with mu x 10, the landmark part corrected in place)
    with the same fault injected into the oracle"""
    from test_gpu_solve import check_window, oracle_run
    monkeypatch.setenv("ISV_SOLVE_ST", "1"); monkeypatch.setenv("ISV_DEBUG_FORCE_RETRY", "2")
    be = backend.Backend(11, 5, max_landmarks=100, max_obs=1100, max_batch=2)
    monkeypatch.delenv("ISV_SOLVE_ST"); monkeypatch.delenv("ISV_DEBUG_FORCE_RETRY")
    oracle.isvo_debug_force_retry(2)
    try:
        for wid in (41, 42):
            w = synth.make_window(wid, n_landmarks=100)
            o, so, _ = oracle_run(oracle, be.cfg, w)
            g = w.clone(); s, _ = be.optimize(g)
            assert be.last_counts()[6] == 1
            check_window(o, so, g, s)
    finally:
        oracle.isvo_debug_force_retry(0); be.close()
