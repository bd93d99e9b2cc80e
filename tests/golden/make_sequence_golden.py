"""Generates tests/golden/sequence_*.npz: a simulated camera-IMU stream pushed through the restated window manager
(tests/sequence_harness.py) with the CPU oracle as the solver.  The stream is regenerated from the seeded simulator,
so a fixture stores a fingerprint of the inputs plus the expected outputs: the pose_output.txt rows
(src/System.cpp:401-410), the newest-frame trajectory, the final window and the bookkeeping counters.
Run:  python tests/golden/make_sequence_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
from isvins_amd import abi
import oracle_lib
import sequence_harness as sh

CASES = {"sequence_11kf_seed4": dict(N=11, Nvo=5, n_frames=24, seed=4)}


def run(case):
    kw = CASES[case]
    lib = oracle_lib.load()
    cfg = abi.make_config(kw["N"], kw["Nvo"], max_landmarks=800, max_obs=800 * kw["N"], max_batch=1)
    est, sim = sh.run_sequence(sh.OracleSolver(lib, cfg), lib, kw["N"], kw["Nvo"], kw["n_frames"], seed=kw["seed"])
    t_last, image = sh.Simulator(kw["seed"]).frame(0)
    fp = np.array([len(image), sum(image), sum(v[0] for v in image.values())])
    return est, dict(
        in_fingerprint=fp,
        pose_output=np.array([[t, *p, *sh._quat_from_R(R)] for (t, p, R) in est.pose_output]),
        newest=np.array([[t, *p, *R.ravel()] for (t, p, R) in est.trajectory]),
        Ps=est.Ps, Rs=est.Rs, Vs=est.Vs, Bas=est.Bas, Bgs=est.Bgs, Headers=est.Headers,
        counters=np.array([est.frame_count, len(est.tracks), len(est.rollpitch), int(est.margin_old), est.summaries[-1].iterations]),
        margin_history=np.array(est.margin_history, np.int32),
        iterations=np.array([s.iterations for s in est.summaries], np.int32))


if __name__ == "__main__":
    for case in CASES:
        _, d = run(case)
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **d)
        print(case, {k: v.shape for k, v in d.items()})
