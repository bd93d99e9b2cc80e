"""CPU-only worker of tests/test_sequence_long.py: one restatement + oracle run of the EuRoC stand-in stream in its own process
(the GPU test starts several of these beside its own GPU work: the controls are independent and host-bound).

usage: python sequence_long_worker.py '<json: n_frames, num_iterations, no_update, eps>' out.npz
  eps          None: the unperturbed run; else the bootstrap positions are moved by eps metres (the control)
  no_update    the oracle's isvo_debug_no_update hook (the sensitivity study: NOT the reference's behaviour)
Test infrastructure: imports the oracle."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import isvins_loader  # noqa: E402
isvins_loader.load()
from isvins_amd import abi  # noqa: E402
import oracle_lib  # noqa: E402
import sequence_harness as sh  # noqa: E402
import test_sequence_long as T  # noqa: E402


def main():
    kw = json.loads(sys.argv[1]); out = sys.argv[2]
    oracle = oracle_lib.load()
    oracle.isvo_debug_no_update(1 if kw.get("no_update") else 0)
    cfg = abi.make_config(T.N, T.NVO, max_landmarks=1000, max_obs=1000 * T.N, max_batch=1, num_iterations=int(kw.get("num_iterations", 10)))
    sim, stream = T.record_stream(int(kw["n_frames"]))
    eps = kw.get("eps")
    est = sh.Estimator(sh.OracleSolver(oracle, cfg), oracle, T.N, T.NVO)
    its, term = [], []
    for i, (imu, t, image) in enumerate(stream):
        for (dt, a, g) in imu:
            est.process_imu(dt, a, g)
        boot = None
        if est.solver_flag == "INITIAL" and est.frame_count == T.N - 1:
            P, R, V = T.bootstrap(sim, i)
            boot = (P + (eps or 0.0), R, V)
        n0 = len(est.summaries)
        est.process_image(image, t, bootstrap=boot)
        if len(est.summaries) > n0:
            its.append(est.summaries[-1].iterations); term.append(est.summaries[-1].termination)
    traj = np.array([np.concatenate([[h], p, R.ravel()]) for (h, p, R) in est.trajectory])
    np.savez(out, trajectory=traj, iterations=np.array(its), termination=np.array(term))


if __name__ == "__main__":
    main()
