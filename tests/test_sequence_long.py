"""BASELINE configs 1 / 3 (EuRoC MH_01 / MH_05 through ./run_euroc) at the size SURVEY 8d specifies for their stand-in:
the dataset, OpenCV and Ceres are absent from the image, so a simulated 6-DoF figure-eight flight of 120 s -- 20 Hz
frames, 200 Hz IMU, ~190 tracked features per frame -- is pushed through the reference's compile-time shape
(ALL_BUF_SIZE = 18, Vo_SIZE = 8, NUM_OF_F = 1000): 2383 solved frames, both slideWindow branches hundreds of times.

What can and cannot agree to 1e-6 m.  The estimator as the reference configures it is SENSITIVE to rounding on such a
stream: ~93 % of the solves stop at NUM_ITERATIONS = 10 with the cost still falling (every prior factor, the linear
speed/bias prior included, sits under CauchyLoss(1.0), src/estimator.cpp:1102-1117, so the accelerometer-bias valley of
a 0.9 s window flattens as the estimate moves along it), and a capped, unconverged solve hands its path-dependent
result to the next frame.  The CPU oracle run twice with bootstrap positions that differ by 1e-12 m drifts apart to
1e-7 m after ~100 frames and to centimetres after ~300 (the control measured below): two implementations that are not
bitwise identical -- the reference's Ceres build included -- cannot stay within 1e-6 m of each other over 2000 frames.
So parity is established the way that IS possible, and the free-running drift is reported beside the control:
  (1) EVERY one of the 2383 solves (+ the initFactorGraph solve and every triangulation), with the windows the
      restatement + oracle run produces, is ALSO solved on the MI355X from the same inputs: states within 1e-7,
      cost trace 1e-7 relative, iteration count / termination / accept pattern identical (flips counted, none allowed);
  (2) free-running, native window manager + MI355X against restatement + oracle: ATE <= 1e-6 m while rounding has not
      been amplified yet (the first 40 solved frames), identical keyframe decisions there; beyond that the divergence is
      printed next to the oracle-vs-perturbed-oracle control and must stay of the control's order;
  (3) the drift against the simulator's ground truth (the one anchor outside the restatement) stays below 1 m rmse.
tests/golden/euroc_standin_n18_seed0.npz holds the oracle-side trajectory (tests/golden/make_euroc_standin_golden.py);
the CPU test below keeps it honest on a prefix."""
import os
import time

import numpy as np
import pytest

from isvins_amd import abi
import sequence_harness as sh

N, NVO, N_FRAMES, SEED = 18, 8, 2400, 0
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "euroc_standin_n18_seed0.npz")


def record_stream(n_frames, seed=SEED):
    """the simulated camera-IMU stream, generated once: [(imu samples before the frame, stamp, {id: (x, y, 1)})]"""
    sim = sh.Simulator(seed, euroc_like=True)
    G = np.array([0, 0, 9.81007])
    out = []
    for i in range(n_frames):
        imu = sim.imu_between(i) if i > 0 else [(sim.frame_dt / sim.k, sim.traj.R(0).T @ (sim.traj.acc(0) + G) + sim.ba, sim.traj.gyro(0) + sim.bg)]
        t, image = sim.frame(i)
        out.append((imu, t, image))
    return sim, out


def bootstrap(sim, i, seed=SEED):
    P, R, V = sim.truth_window(i, N)
    nrng = np.random.default_rng(1000 + seed)
    return P + nrng.normal(0, 0.01, P.shape), R, V + nrng.normal(0, 0.02, V.shape)


def run_oracle_side(oracle, cfg, sim, stream):
    est = sh.Estimator(sh.OracleSolver(oracle, cfg), oracle, N, NVO)
    per_solve = []
    for i, (imu, t, image) in enumerate(stream):
        for (dt, a, g) in imu:
            est.process_imu(dt, a, g)
        boot = bootstrap(sim, i) if (est.solver_flag == "INITIAL" and est.frame_count == N - 1) else None
        n0 = len(est.summaries)
        est.process_image(image, t, bootstrap=boot)
        if len(est.summaries) > n0:
            s = est.summaries[-1]
            per_solve.append((s.iterations, s.termination, tuple(s.trace_accepted[1: s.iterations + 1]), bool(est.margin_history[-1])))
    traj = np.array([np.concatenate([[h], p, R.ravel()]) for (h, p, R) in est.trajectory])
    return est, traj, per_solve


def run_native_side(est, sim, stream):
    per_solve = []
    for i, (imu, t, image) in enumerate(stream):
        est.process_imu_n(0, [x[0] for x in imu], [x[1] for x in imu], [x[2] for x in imu])
        st = est.status(0)
        if st["solver_flag"] == 0 and st["frame_count"] == N - 1:
            est.set_bootstrap(0, *bootstrap(sim, i))
        ids = np.array(list(image.keys()), np.int32)
        pts = np.array([image[int(k)] for k in ids], float).reshape(-1, 3)
        est.push_image(0, t, ids, pts)
        if est.step() > 0:
            s = est.last_summary(0)
            per_solve.append((s.iterations, s.termination, tuple(s.trace_accepted[1: s.iterations + 1]), bool(est.status(0)["margin_old"])))
    return est.trajectory(0, 1), per_solve


def test_oracle_side_reproduces_the_committed_prefix(oracle):
    """the first 40 solved frames of the restatement + oracle run against the committed golden trajectory: bitwise
    (same library, same stream) -- the golden file is the oracle's output and nothing else"""
    g = np.load(GOLDEN)
    n = N - 1 + 40
    cfg = abi.make_config(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
    sim, stream = record_stream(n)
    _, traj, per = run_oracle_side(oracle, cfg, sim, stream)
    assert traj.shape == (40, 13)
    assert np.abs(traj - g["trajectory"][:40]).max() < 1e-9
    assert [p[0] for p in per] == list(g["iterations"][:40])


class TeacherForced(sh.OracleSolver):
    """the oracle solver of the restatement run; every call is repeated on the MI355X from the same inputs and compared"""

    def __init__(self, lib, cfg, be):
        super().__init__(lib, cfg)
        self.be = be
        self.n = 0
        self.flips = dict(iterations=0, termination=0, accept=0, solve_flag=0)
        self.worst = dict(state=0.0, cost=0.0, depth=0.0, prior=0.0, tri=0.0, marg=0.0)

    def triangulate(self, w):
        g = w.clone()
        super().triangulate(w)
        self.be.triangulate([g])
        if w.L:
            self.worst["tri"] = max(self.worst["tri"], float(np.abs(g.lm_depth[: w.L] / w.lm_depth[: w.L] - 1).max()))

    def init_factor_graph(self, w):
        g = w.clone()
        s = super().init_factor_graph(w)
        sg, _ = self.be.init_factor_graph(g)
        assert sg.iterations == s.iterations and sg.termination == s.termination
        assert np.abs(g.state_vector()[: -g.L] - w.state_vector()[: -w.L]).max() < 1e-6
        return s

    def optimize(self, w):
        g = w.clone()
        s, m = super().optimize(w)
        sg, mg = self.be.optimize(g)
        self.n += 1
        n = s.iterations
        self.flips["iterations"] += sg.iterations != n
        self.flips["termination"] += sg.termination != s.termination
        self.flips["accept"] += list(sg.trace_accepted[: n + 1]) != list(s.trace_accepted[: n + 1])
        self.flips["solve_flag"] += not np.array_equal(g.lm_solve_flag[: g.L], w.lm_solve_flag[: w.L])
        if sg.iterations == n:
            tc_o, tc_g = np.array(s.trace_cost[: n + 1]), np.array(sg.trace_cost[: n + 1])
            self.worst["cost"] = max(self.worst["cost"], float(np.abs(tc_g / tc_o - 1).max()))
            st = max(np.abs(getattr(g, k) - getattr(w, k)).max() for k in ("Ps", "Rs", "Vs", "Bas", "Bgs", "para_Pose", "para_SpeedBias"))
            self.worst["state"] = max(self.worst["state"], float(st))
            self.worst["prior"] = max(self.worst["prior"], float(np.abs(g.priors_vector() - w.priors_vector()).max()))
            if w.L:
                self.worst["depth"] = max(self.worst["depth"], float((np.abs(g.lm_depth[: w.L] - w.lm_depth[: w.L]) / np.maximum(1.0, np.abs(w.lm_depth[: w.L]))).max()))
            if w.margin_old:
                for name, k in (("forward_pose_prior", 6), ("backward_relpose", 6), ("backward_vb", 9), ("backward_rollpitch", 2)):
                    A = abi.arr(getattr(mg, name).sqrt_info, (k, k)); B = abi.arr(getattr(m, name).sqrt_info, (k, k))
                    self.worst["marg"] = max(self.worst["marg"], float(np.abs(A.T @ A - B.T @ B).max() / np.abs(B.T @ B).max()))
        return s, m


def run_oracle_perturbed(oracle, cfg, sim, stream, n, eps=1e-12):
    """control: the restatement + oracle on the first n frames with bootstrap positions moved by eps metres"""
    est = sh.Estimator(sh.OracleSolver(oracle, cfg), oracle, N, NVO)
    for i, (imu, t, image) in enumerate(stream[:n]):
        for (dt, a, g) in imu:
            est.process_imu(dt, a, g)
        boot = None
        if est.solver_flag == "INITIAL" and est.frame_count == N - 1:
            P, R, V = bootstrap(sim, i); boot = (P + eps, R, V)
        est.process_image(image, t, bootstrap=boot)
    return np.array([p for (_, p, _) in est.trajectory])


@pytest.mark.gpu
def test_euroc_standin_full_length_gpu_vs_oracle(oracle):
    from isvins_amd import backend, estimator as E
    cfg = abi.make_config(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
    t0 = time.time()
    sim, stream = record_stream(N_FRAMES)
    feats = [len(im) for (_, _, im) in stream]
    t1 = time.time()
    est = E.SequenceEstimator(sh.estimator_params(cfg), 1)
    rows, per_g = run_native_side(est, sim, stream)
    t2 = time.time()
    be = backend.Backend(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
    tf = TeacherForced(oracle, cfg, be)
    eo = sh.Estimator(tf, oracle, N, NVO)
    per_o = []
    for i, (imu, t, image) in enumerate(stream):
        for (dt, a, g) in imu:
            eo.process_imu(dt, a, g)
        boot = bootstrap(sim, i) if (eo.solver_flag == "INITIAL" and eo.frame_count == N - 1) else None
        n0 = len(eo.summaries)
        eo.process_image(image, t, bootstrap=boot)
        if len(eo.summaries) > n0:
            s = eo.summaries[-1]
            per_o.append((s.iterations, s.termination, tuple(s.trace_accepted[1: s.iterations + 1]), bool(eo.margin_history[-1])))
    traj_o = np.array([np.concatenate([[h], p, R.ravel()]) for (h, p, R) in eo.trajectory])
    t3 = time.time()
    n_ctl = 420
    ctl = run_oracle_perturbed(oracle, cfg, sim, stream, n_ctl)
    t4 = time.time()
    n_solved = N_FRAMES - (N - 1)
    assert len(rows) == len(traj_o) == n_solved >= 2000 and tf.n == n_solved
    assert est.failed_solves(0) == 0
    old = np.array([p[3] for p in per_o])
    truth = np.array([sim.traj.p(h) for h in traj_o[:, 0]])
    drift = np.linalg.norm(traj_o[:, 1:4] - truth, axis=1)
    dpos = np.linalg.norm(rows[:, 1:4] - traj_o[:, 1:4], axis=1)
    dctl = np.linalg.norm(ctl - traj_o[: len(ctl), 1:4], axis=1)
    marks = [m for m in (20, 40, 80, 160, 240, 320, 400) if m < len(ctl)]
    first_kf_flip = next((k for k, (a, b) in enumerate(zip(per_g, per_o)) if a[3] != b[3]), None)
    print(f"\nEuRoC stand-in, N={N} Vo={NVO}: {N_FRAMES} frames (120 s at 20 Hz), {n_solved} solved, features/frame mean {np.mean(feats):.0f} "
          f"[{min(feats)}, {max(feats)}], MARGIN_OLD {old.mean():.2f} / MARGIN_NEW {1 - old.mean():.2f}\n"
          f"  (1) every solve repeated on the MI355X from the oracle side's inputs ({tf.n} windows, landmarks {min(len(x[2]) for x in per_o)}..): "
          f"flips {tf.flips}; worst |dstate| {tf.worst['state']:.2e}, cost trace {tf.worst['cost']:.2e} rel, depth {tf.worst['depth']:.2e} rel, "
          f"priors {tf.worst['prior']:.2e}, marginalisation information {tf.worst['marg']:.2e} rel, triangulation {tf.worst['tri']:.2e} rel\n"
          f"  (2) free-running native + MI355X vs restatement + oracle, |dP| at solved frame {marks}: {[f'{dpos[m]:.1e}' for m in marks]}; "
          f"first keyframe-decision flip at solved frame {first_kf_flip}; full-length ATE {np.sqrt(np.mean(dpos ** 2)):.3f} m\n"
          f"      control, oracle vs oracle with a 1e-12 m bootstrap perturbation,  |dP| at the same frames: {[f'{dctl[m]:.1e}' for m in marks]}\n"
          f"  (3) drift of the oracle-side run against the simulator's ground truth: rmse {np.sqrt(np.mean(drift ** 2)):.3f} m, max {drift.max():.3f} m over {traj_o[-1, 0] - traj_o[0, 0]:.0f} s\n"
          f"  wall: stream {t1 - t0:.1f} s, native + GPU {t2 - t1:.1f} s ({1e3 * (t2 - t1) / N_FRAMES:.2f} ms / frame incl. the Python feed), "
          f"restatement + oracle + per-solve GPU repeats {t3 - t2:.1f} s, control {t4 - t3:.1f} s")
    # (1) per-solve parity over the whole run
    assert all(v == 0 for v in tf.flips.values()), tf.flips
    # (depths are 1 / lambda of far, weakly constrained points and the recovered information matrices go through an
    # eigen-decomposition and small inverses: 1e-4 relative over 2383 real windows; 1e-5 / 1e-6 on the synthetic ones)
    assert tf.worst["state"] < 1e-7 and tf.worst["prior"] < 1e-7 and tf.worst["cost"] < 1e-7 and tf.worst["depth"] < 1e-4
    assert tf.worst["marg"] < 1e-4 and tf.worst["tri"] < 1e-6
    # (2) free-running: exact while rounding has not been amplified, of the control's order afterwards
    assert np.sqrt(np.mean(dpos[:40] ** 2)) < 1e-6 and dpos[:40].max() < 1e-6
    assert all(a == b for a, b in zip(per_g[:40], per_o[:40]))
    assert first_kf_flip is None or first_kf_flip > 150
    assert dpos.max() < 3.0 and 0.05 < old.mean() < 0.95
    # (3) ground truth
    assert np.sqrt(np.mean(drift ** 2)) < 1.0
    # the committed golden is the oracle side's output (prefix: the full length is as chaotic as (2))
    g = np.load(GOLDEN)
    assert g["trajectory"].shape == traj_o.shape
    assert np.abs(traj_o[:40] - g["trajectory"][:40]).max() < 1e-6
    est.close(); be.close()
