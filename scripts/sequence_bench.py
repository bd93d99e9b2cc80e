"""Whole-sequence throughput of the native window manager (include/isvins_estimator.h) on the MI355X backend:
S simulated camera-IMU streams in lock step, one batched triangulate + backendOptimization per frame.
usage: python scripts/sequence_bench.py S [N Nvo n_frames]
Prints frames/s (sequences x frames / wall time of the feed + step calls once every window is full) and the per-step
breakdown the estimator records.  The streams are generated before the timed region (tests/sequence_harness.Simulator)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
import numpy as np
from isvins_amd import abi, estimator as E
import sequence_harness as sh

S = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 11; Nvo = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n_frames = int(sys.argv[4]) if len(sys.argv) > 4 else N + 20
n_streams = min(S, 8)                          # distinct simulated streams, reused round-robin (generation is Python-slow)
streams = []
for sd in range(n_streams):
    sim = sh.Simulator(sd)
    fr = []
    for i in range(n_frames):
        if i > 0:
            smp = sim.imu_between(i)
        else:
            smp = [(sim.frame_dt / sim.k, sim.traj.R(0).T @ (sim.traj.acc(0) + np.array([0, 0, 9.81007])) + sim.ba, sim.traj.gyro(0) + sim.bg)]
        t, image = sim.frame(i)
        ids = np.array(sorted(image), np.int32); pts = np.array([image[int(k)] for k in ids], float).reshape(-1, 3)
        fr.append((np.array([x[0] for x in smp]), np.array([x[1] for x in smp]), np.array([x[2] for x in smp]), t, ids, pts))
    boot = sim.truth_window(N - 1, N)
    streams.append((fr, boot))
cfg = abi.make_config(N, Nvo, max_landmarks=800, max_obs=800 * N, max_batch=S)
est = E.SequenceEstimator(sh.estimator_params(cfg), S)
timed, t_feed, t_step, parts = 0, 0.0, 0.0, []
for i in range(n_frames):
    t0 = time.perf_counter()
    for s in range(S):
        fr, boot = streams[s % n_streams]
        dts, accs, gyrs, t, ids, pts = fr[i]
        est.process_imu_n(s, dts, accs, gyrs)
        if i == N - 1:
            est.set_bootstrap(s, *boot)
        est.push_image(s, t, ids, pts)
    t1 = time.perf_counter()
    n = est.step()
    t2 = time.perf_counter()
    if i >= N + 2:                               # steady state: every sequence solves (the first solves include initFactorGraph)
        timed += n; t_feed += t1 - t0; t_step += t2 - t1; parts.append(est.last_step_ms())
    print(f"frame {i}: solved {n}, feed {1e3*(t1-t0):.1f} ms, step {1e3*(t2-t1):.1f} ms", flush=True)
st = est.status(0)
med = {k: float(np.median([p[k] for p in parts])) for k in parts[0]}
print(f"S={S} N={N} Nvo={Nvo}: {st['n_landmarks']} landmarks in the last window of sequence 0, {st['iterations']} iterations")
print(f"steady state: {timed / (t_feed + t_step):.0f} frames/s incl. the Python feed loop, {timed / t_step:.0f} frames/s for isv_estimator_step alone "
      f"({1e3 * t_step / len(parts):.2f} ms per lock-step frame of {S} sequences)")
print("median step breakdown (ms):", {k: round(v, 3) for k, v in med.items()})
est.close()
