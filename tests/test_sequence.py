"""Whole-sequence run: the same stream of IMU samples and feature observations goes through the reference's window
manager (tests/sequence_harness.py) once with the CPU oracle and once with the MI355X backend doing triangulate /
initFactorGraph / backendOptimization.  north_star: ATE within 1e-6 m between the two solves of the same input."""
import numpy as np
import pytest

from isvins_amd import abi

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
