"""Writes tests/golden/euroc_standin_n18_seed0.npz: the trajectory and per-solve control flow of the restated window
manager + CPU oracle on the simulated EuRoC stand-in stream (tests/test_sequence_long.py).  The fixture is the ORACLE's
output (the reference cannot be built here and ships no trajectories): it records the oracle, it does not pin it."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import isvins_loader; isvins_loader.load()  # noqa: E402
from isvins_amd import abi  # noqa: E402
import oracle_lib  # noqa: E402
import test_sequence_long as T  # noqa: E402

oracle = oracle_lib.load()
cfg = abi.make_config(T.N, T.NVO, max_landmarks=1000, max_obs=1000 * T.N, max_batch=1)
sim, stream = T.record_stream(T.N_FRAMES)
eo, traj, per = T.run_oracle_side(oracle, cfg, sim, stream)
truth = np.array([sim.traj.p(h) for h in traj[:, 0]])
np.savez_compressed(os.path.join(HERE, "euroc_standin_n18_seed0.npz"), trajectory=traj, iterations=np.array([p[0] for p in per], np.int32),
                    termination=np.array([p[1] for p in per], np.int32), margin_old=np.array([p[3] for p in per], np.int8), truth=truth)
print("solved frames", len(traj), "rmse vs truth", np.sqrt(np.mean(np.sum((traj[:, 1:4] - truth) ** 2, axis=1))))
