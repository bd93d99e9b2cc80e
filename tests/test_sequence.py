"""Whole-sequence run: the same stream of IMU samples and feature observations goes through the reference's window
manager (tests/sequence_harness.py) once with the CPU oracle and once with the MI355X backend doing triangulate /
initFactorGraph / backendOptimization.  north_star: ATE within 1e-6 m between the two solves of the same input."""
import numpy as np
import pytest

from isvins_amd import abi, backend, synth

import sequence_harness as sh


def _traj(est):
    return np.array([p for (_, p, _) in est.trajectory]), np.array([r for (_, _, r) in est.trajectory])


def test_sequence_harness_runs_on_the_oracle(oracle):
    N, Nvo = 11, 5
    cfg = abi.make_config(N, Nvo, max_landmarks=600, max_obs=6600, max_batch=1)
    est, sim = sh.run_sequence(sh.OracleSolver(oracle, cfg), oracle, N, Nvo, n_frames=16)
    assert len(est.trajectory) == 6 and est.solver_flag == "NON_LINEAR"
    P, _ = _traj(est)
    truth = np.array([sim.traj.p(t) for (t, _, _) in est.trajectory])
    assert np.isfinite(P).all() and np.linalg.norm(P - truth, axis=1).max() < 1.0      # it tracks; accuracy is not the point here
    assert 200 < len(est._good()) <= 600 and len(est.rollpitch) >= 1


def test_pose_output_rows_follow_the_reference_format(oracle, tmp_path):
    """pose_output.txt (src/System.cpp:401-410): one row per solved frame, `stamp px py pz qw qx qy qz` of the OLDEST
    window frame, readable by the usual TUM-style evaluation scripts"""
    N, Nvo = 11, 5
    cfg = abi.make_config(N, Nvo, max_landmarks=600, max_obs=6600, max_batch=1)
    est, sim = sh.run_sequence(sh.OracleSolver(oracle, cfg), oracle, N, Nvo, n_frames=14)
    path = tmp_path / "pose_output.txt"
    est.write_pose_output(path)
    rows = np.loadtxt(path)
    assert rows.shape == (4, 8)
    assert np.allclose(np.linalg.norm(rows[:, 4:], axis=1), 1.0, atol=1e-5)
    assert np.all(np.diff(rows[:, 0]) > 0)
    for row, (t, p, R) in zip(rows, est.pose_output):
        w, x, y, z = row[4:]
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert abs(row[0] - t) < 1e-6 and np.abs(row[1:4] - p).max() < 1e-6 and np.abs(Rq - R).max() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("N,Nvo,n_frames", [(11, 5, 41), (18, 8, 40)])
def test_sequence_ate_gpu_vs_oracle(oracle, N, Nvo, n_frames):
    from isvins_amd import backend
    backend.build()
    cfg = abi.make_config(N, Nvo, max_landmarks=800, max_obs=800 * N, max_batch=1)
    eo, _ = sh.run_sequence(sh.OracleSolver(oracle, cfg), oracle, N, Nvo, n_frames)
    b = backend.Backend(N, Nvo, max_landmarks=800, max_obs=800 * N, max_batch=1)
    try:
        eg, _ = sh.run_sequence(sh.DeviceSolver(b), oracle, N, Nvo, n_frames)
    finally:
        b.close()
    assert len(eg.trajectory) == len(eo.trajectory) == n_frames - (N - 1)
    assert [s.iterations for s in eg.summaries] == [s.iterations for s in eo.summaries]
    Pg, Rg = _traj(eg); Po, Ro = _traj(eo)
    ate = np.sqrt(np.mean(np.sum((Pg - Po) ** 2, axis=1)))
    rot = max(np.linalg.norm(a @ b_.T - np.eye(3)) for a, b_ in zip(Rg, Ro))
    print(f"N={N}: {len(Pg)} solved frames, ATE(GPU vs oracle) = {ate:.3e} m, max rotation difference = {rot:.3e}")
    assert ate < 1e-6 and rot < 1e-6


def _cmp_native_vs_harness(est, s, eo, tol):
    """the native window manager against the Python restatement after the same stream: trajectory rows, window states,
    bookkeeping"""
    rows = est.trajectory(s, 1); po = est.trajectory(s, 0)
    Po, Ro = _traj(eo)
    assert len(rows) == len(Po) == len(eo.pose_output)
    assert np.abs(rows[:, 1:4] - Po).max() < tol and np.abs(rows[:, 4:].reshape(-1, 3, 3) - Ro).max() < tol
    assert np.abs(po[:, 0] - np.array([t for (t, _, _) in eo.pose_output])).max() == 0
    assert np.abs(po[:, 1:4] - np.array([p for (_, p, _) in eo.pose_output])).max() < tol
    qo = np.array([sh._quat_from_R(R) for (_, _, R) in eo.pose_output])
    assert np.abs(po[:, 4:] - qo).max() < tol
    w = est.window(s)
    assert np.abs(w["Ps"] - eo.Ps).max() < tol and np.abs(w["Rs"] - eo.Rs).max() < tol and np.abs(w["Vs"] - eo.Vs).max() < tol
    assert np.abs(w["Bas"] - eo.Bas).max() < tol and np.abs(w["Bgs"] - eo.Bgs).max() < tol and np.array_equal(w["Headers"], eo.Headers)
    st = est.status(s)
    assert st["solver_flag"] == 1 and st["frame_count"] == eo.frame_count and st["n_tracks"] == len(eo.tracks)
    assert st["n_rollpitch"] == len(eo.rollpitch) and st["margin_old"] == int(eo.margin_old)
    assert st["iterations"] == eo.summaries[-1].iterations


@pytest.mark.parametrize("fused", [False, True])
def test_native_window_manager_matches_the_restatement(oracle, monkeypatch, fused):
    """include/isvins_estimator.h (C++: processIMU, pre-integration, addFeatureAndCheckParallax, slideWindow, prior
    rotation, removeBackShiftDepth / removeFront / removeFailures) against tests/sequence_harness.py on the same
    simulated streams, both with the CPU oracle as the solver (injected through isv_estimator_create_with_solver);
    two sequences in lock step.  Tolerance 1e-9: the two pre-integrations round differently in the last bits."""
    from isvins_amd import estimator as E
    monkeypatch.setenv("ISV_HOST_THREADS", "2")            # the per-sequence host work on two threads
    N, Nvo, n_frames, seeds = 11, 5, 19, (0, 3)
    cfg = abi.make_config(N, Nvo, max_landmarks=600, max_obs=6600, max_batch=len(seeds))
    vt = sh.oracle_vtbl(oracle, cfg, fused=fused)      # fused: the one-hand-over solveOdometry entry in steady state
    est = E.SequenceEstimator(sh.estimator_params(cfg), len(seeds), solver=vt)
    sh.run_sequences_native(est, N, n_frames, seeds)
    margin_flags = set()
    for s, sd in enumerate(seeds):
        eo, _ = sh.run_sequence(sh.OracleSolver(oracle, cfg), oracle, N, Nvo, n_frames, seed=sd)
        _cmp_native_vs_harness(est, s, eo, 1e-9)
        margin_flags |= set(eo.margin_history[N - 1:])
    assert margin_flags == {True, False}           # both slideWindow branches were taken after the first solve
    est.close()


def test_native_window_manager_carries_the_estimated_extrinsic(oracle):
    """cfg.estimate_extrinsic = 1: every solve moves para_Ex_Pose, double2vector writes it to tic[0] / ric[0]
    (src/estimator.cpp:575-583), and the NEXT frame's triangulation, window and slideWindowOld (:1714-1719) use those.  The
    native window manager against the restatement, both on the CPU oracle; and the extrinsic really moved (the same stream
    with estimate_extrinsic = 0 ends elsewhere)."""
    from isvins_amd import estimator as E
    N, Nvo, n_frames, seed = 11, 5, 19, 2
    cfg1 = abi.make_config(N, Nvo, max_landmarks=600, max_obs=6600, max_batch=1, estimate_extrinsic=1)
    est = E.SequenceEstimator(sh.estimator_params(cfg1), 1, solver=sh.oracle_vtbl(oracle, cfg1))
    sh.run_sequences_native(est, N, n_frames, (seed,))
    eo, _ = sh.run_sequence(sh.OracleSolver(oracle, cfg1), oracle, N, Nvo, n_frames, seed=seed)
    _cmp_native_vs_harness(est, 0, eo, 1e-9)
    assert np.abs(eo.tic - synth.TIC).max() > 1e-6 or np.abs(eo.ric - synth.RIC).max() > 1e-6
    tic, ric = est.extrinsic(0)                     # isv_estimator_get_extrinsic: tic[0] / ric[0] as the last solve left them
    assert np.abs(tic - eo.tic).max() < 1e-9 and np.abs(ric - eo.ric).max() < 1e-9
    cfg0 = abi.make_config(N, Nvo, max_landmarks=600, max_obs=6600, max_batch=1)
    e0, _ = sh.run_sequence(sh.OracleSolver(oracle, cfg0), oracle, N, Nvo, n_frames, seed=seed)
    assert np.array_equal(e0.tic, synth.TIC) and np.abs(e0.Ps - eo.Ps).max() > 1e-9
    est.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,Nvo,n_frames,n_seq,est_ex", [(11, 5, 31, 4, 0), (18, 8, 30, 2, 0), (11, 5, 27, 2, 1)])
def test_native_sequences_on_gpu_vs_oracle(oracle, N, Nvo, n_frames, n_seq, est_ex):
    """S sequences in lock step through the native window manager with every solve on the MI355X (one batched
    triangulate + backendOptimization per frame) against the Python restatement with the CPU oracle: ATE <= 1e-6 m.
    est_ex = 1: the extrinsic is a free block of every solve (k_lin_gram<true>, k_dogleg<.., true>) and is carried from
    frame to frame by the window manager."""
    from isvins_amd import estimator as E
    seeds = tuple(range(n_seq))
    cfg = abi.make_config(N, Nvo, max_landmarks=800, max_obs=800 * N, max_batch=n_seq, estimate_extrinsic=est_ex)
    est = E.SequenceEstimator(sh.estimator_params(cfg), n_seq)
    sh.run_sequences_native(est, N, n_frames, seeds)
    for s, sd in enumerate(seeds):
        eo, _ = sh.run_sequence(sh.OracleSolver(oracle, cfg), oracle, N, Nvo, n_frames, seed=sd)
        rows = est.trajectory(s, 1)
        Po, Ro = _traj(eo)
        assert len(rows) == len(Po) == n_frames - (N - 1)
        ate = np.sqrt(np.mean(np.sum((rows[:, 1:4] - Po) ** 2, axis=1)))
        rot = np.abs(rows[:, 4:].reshape(-1, 3, 3) - Ro).max()
        print(f"N={N} seq {s}: {len(Po)} solved frames, ATE(native+GPU vs restatement+oracle) = {ate:.3e} m, rotation {rot:.3e}")
        assert ate < 1e-6 and rot < 1e-6
        assert est.status(s)["n_tracks"] == len(eo.tracks)
    est.close()


@pytest.mark.gpu
def test_replay_tool_writes_the_reference_trajectory_file(oracle, tmp_path):
    """tools/isv_replay (native executable: stream file -> window manager -> MI355X -> pose_output_<s>.txt) on a
    recorded simulated stream, 6 copies in 2 groups (two estimators on two host threads), against the Python
    restatement with the oracle: the rows of pose_output.txt (6 decimals, src/System.cpp:408-409) agree to the last
    printed digit (+-1e-6) and all copies are identical"""
    import json, os, subprocess
    from isvins_amd import backend
    backend.build()
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isv_replay")
    N, Nvo, n_frames = 11, 5, 28
    stream = tmp_path / "stream.txt"
    sh.write_stream(stream, N, Nvo, n_frames, seed=2)
    out = subprocess.run([tool, str(stream), "--sequences", "6", "--groups", "2", "--out", str(tmp_path), "--write", "6"],
                         check=True, capture_output=True, text=True, timeout=300)
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["sequences"] == 6 and rep["groups"] == 2 and rep["frames_in_stream"] == n_frames and rep["frames_per_second"] > 0
    cfg = abi.make_config(N, Nvo, max_landmarks=800, max_obs=800 * N, max_batch=1)
    eo, _ = sh.run_sequence(sh.OracleSolver(oracle, cfg), oracle, N, Nvo, n_frames, seed=2)
    ref_path = tmp_path / "pose_output_oracle.txt"
    eo.write_pose_output(ref_path)
    ref = np.loadtxt(ref_path)
    first = np.loadtxt(tmp_path / "pose_output_0.txt")
    assert first.shape == ref.shape == (n_frames - (N - 1), 8)
    assert np.abs(first - ref).max() <= 1.000001e-6
    for s in range(1, 6):
        assert np.array_equal(np.loadtxt(tmp_path / f"pose_output_{s}.txt"), first)
    print("isv_replay:", rep)


@pytest.mark.gpu
def test_replay_tool_reads_euroc_format_files(tmp_path):
    """tools/isv_replay --euroc: the reference's imu0/data.csv format (test/run_euroc.cpp:26-50) + a feature-track
    table, paired the way System::getMeasurements / ProcessBackEnd pair them (src/System.cpp:160-202, 262-296),
    against the same simulated stream replayed from the pre-digested stream file: the same trajectory (the only
    difference is the reference's repeated boundary sample with dt = 0, which changes nothing but roundings)"""
    import json, os, subprocess
    from isvins_amd import backend
    backend.build()
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isv_replay")
    N, Nvo, n_frames = 11, 5, 26
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    sh.write_stream(tmp_path / "stream.txt", N, Nvo, n_frames, seed=5)
    sh.write_euroc_like(str(tmp_path / "ds"), N, Nvo, n_frames, seed=5)
    subprocess.run([tool, str(tmp_path / "stream.txt"), "--out", str(tmp_path / "a")], check=True, capture_output=True, text=True, timeout=300)
    out = subprocess.run([tool, "--euroc", str(tmp_path / "ds" / "mav0"), "--tracks", str(tmp_path / "ds" / "tracks.csv"),
                          "--config", str(tmp_path / "ds" / "config.txt"), "--out", str(tmp_path / "b")],
                         check=True, capture_output=True, text=True, timeout=300)
    rep = json.loads(out.stdout.strip().splitlines()[-1])
    assert rep["frames_in_stream"] == n_frames
    a = np.loadtxt(tmp_path / "a" / "pose_output_0.txt"); b = np.loadtxt(tmp_path / "b" / "pose_output_0.txt")
    assert a.shape == b.shape == (n_frames - (N - 1), 8)
    assert np.abs((b[:, 0] - 1.0) - a[:, 0]).max() < 2e-6                # stamps: offset by the 1 s epoch of the CSV files
    assert np.abs(a[:, 1:] - b[:, 1:]).max() <= 1.000001e-6              # positions / quaternions to the last printed digit


def test_native_preintegration_matches_the_oracle(oracle):
    """IntegrationBase::push_back (include/factor/integration_base.h:31-158) in the window manager (block-sparse
    products, AVX2 clone) against the oracle's dense restatement over 400 random samples: delta_p / q / v, the 15x15
    Jacobian and the covariance"""
    from isvins_amd import estimator as E
    cfg = abi.make_config(11, 5, max_landmarks=100, max_obs=1100, max_batch=1)
    est = E.SequenceEstimator(sh.estimator_params(cfg), 1, solver=sh.oracle_vtbl(oracle, cfg))
    rng = np.random.default_rng(5)
    # frame 0 takes no IMU (frame_count == 0); one image moves on to frame 1, whose pre-integration is then fed
    est.process_imu(0, 0.005, [0.1, -0.2, 9.7], [0.01, 0.02, -0.01])
    est.push_image(0, 0.0, np.arange(30, dtype=np.int32), np.tile([0.1, 0.2, 1.0], (30, 1))); est.step()
    acc0, gyr0 = np.array([0.1, -0.2, 9.7]), np.array([0.01, 0.02, -0.01])
    ref = sh.PreInt(oracle, acc0, gyr0, np.zeros(3), np.zeros(3))
    for _ in range(400):
        dt = float(rng.uniform(0.002, 0.01)); a = rng.normal(0, 2.0, 3) + [0, 0, 9.8]; g = rng.normal(0, 0.5, 3)
        est.process_imu(0, dt, a, g); ref.push_back(dt, a, g)
    got = est.preintegration(0, 1)
    for name in ("delta_p", "delta_q", "delta_v", "linearized_ba", "linearized_bg", "jacobian", "covariance"):
        a, b = abi.arr(getattr(got, name)), abi.arr(getattr(ref.pod, name))
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), name
    assert abs(got.sum_dt - ref.pod.sum_dt) < 1e-12
    est.close()


def test_failed_solve_reinitialises_the_priors(oracle):
    """a solve whose result is not finite is not copied into the window, but the window still slides; with
    MARGIN_OLD and no marginalisation outputs the prior factors would then sit one frame off for every later solve.  The
    window manager instead sends the sequence back through initFactorGraph at its next solve.  The failure is injected
    through the solver seam (status = ISV_ERR_NONFINITE on one MARGIN_OLD solve); afterwards every window handed to the
    solver carries its priors on the right frames, and the run stays close to the run without the failure."""
    from isvins_amd import estimator as E
    N, Nvo, n_frames, seed = 11, 5, 26, 0
    cfg = abi.make_config(N, Nvo, max_landmarks=600, max_obs=6600, max_batch=1)
    base = sh.oracle_vtbl(oracle, cfg)
    log = dict(opt=0, init=0, failed_at=None, init_after_failure=None, windows=[])
    ERR_NONFINITE = [k for k, v in backend.STATUS.items() if v == "ISV_ERR_NONFINITE"][0]

    def opt(ctx, n, ws, sums, margs):
        rc = base.optimize_batch(ctx, n, ws, sums, margs)
        for i in range(n):
            w = ws[i].contents
            log["windows"].append(dict(pp=w.pose_prior.contents.index, vb=w.vb_prior.contents.index,
                                       rel=[(w.relpose[k].imu_i, w.relpose[k].imu_j) for k in range(Nvo - 1)],
                                       rp=[w.rollpitch[k].index for k in range(w.n_rollpitch)], margin_old=w.margin_old))
            if log["failed_at"] is None and log["opt"] >= 6 and w.margin_old:
                sums[i].status = ERR_NONFINITE
                log["failed_at"] = log["opt"]
        log["opt"] += 1
        return rc

    def init(ctx, w, s, kld):
        if log["failed_at"] is not None and log["init_after_failure"] is None:
            log["init_after_failure"] = log["opt"]
        log["init"] += 1
        return base.init_factor_graph(ctx, w, s, kld)

    vt = E.isv_solver_vtbl_t(None, base.triangulate, E.INIT_FN(init), E.OPTIMIZE_FN(opt))
    est = E.SequenceEstimator(sh.estimator_params(cfg), 1, solver=vt)
    sh.run_sequences_native(est, N, n_frames, (seed,))
    ref = E.SequenceEstimator(sh.estimator_params(cfg), 1, solver=sh.oracle_vtbl(oracle, cfg))
    sh.run_sequences_native(ref, N, n_frames, (seed,))
    assert log["failed_at"] is not None and est.failed_solves(0) == 1 and ref.failed_solves(0) == 0
    assert log["init"] == 2 and log["init_after_failure"] == log["failed_at"] + 1       # the very next solve re-initialises
    for k, w in enumerate(log["windows"]):
        assert w["pp"] == 0 and w["vb"] == Nvo - 1 and w["rel"] == [(i, i + 1) for i in range(Nvo - 1)], (k, w)
        assert all(0 <= i < Nvo for i in w["rp"]) and len(set(w["rp"])) == len(w["rp"]), (k, w)
    # the window right after the re-initialisation starts without roll/pitch factors, like the first one ever solved
    assert log["windows"][log["failed_at"] + 1]["rp"] == []
    a, b = est.trajectory(0, 1), ref.trajectory(0, 1)
    assert len(a) == len(b) == n_frames - (N - 1) and np.isfinite(a).all()
    assert np.abs(a[: log["failed_at"], 1:4] - b[: log["failed_at"], 1:4]).max() == 0.0     # identical up to the failure
    # afterwards: the re-initialised run is a different (gauge re-anchored, scale re-estimated) but equally valid
    # estimate -- judged against the simulator's ground truth, beside the run that never failed
    sim = sh.Simulator(seed)
    truth = np.array([sim.traj.p(h) for h in a[:, 0]])
    err_a, err_b = np.linalg.norm(a[:, 1:4] - truth, axis=1).max(), np.linalg.norm(b[:, 1:4] - truth, axis=1).max()
    print(f"max position error against ground truth: {err_a:.3f} m with the injected failure, {err_b:.3f} m without")
    assert err_a < max(3.0 * err_b, 0.25)
    assert est.status(0)["solver_flag"] == 1
    est.close(); ref.close()
