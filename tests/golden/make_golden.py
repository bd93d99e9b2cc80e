"""Generates tests/golden/window_*.npz from the CPU oracle (the reference itself cannot run here:
its dependencies are absent -- see DESIGN.md).  Inputs are regenerated from `synth` (deterministic
SplitMix64 streams), so a fixture stores only a fingerprint of the inputs plus expected outputs.
Run:  python tests/golden/make_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
from isvins_amd import abi, synth
import oracle_lib

CASES = {"window_11kf_40lm": dict(window_id=0, n_frames=11, n_vo=5, n_landmarks=40),
         "window_18kf_60lm": dict(window_id=1, n_frames=18, n_vo=8, n_landmarks=60)}
dp = C.POINTER(C.c_double)


def run(case):
    kw = dict(CASES[case]); wid = kw.pop("window_id")
    w = synth.make_window(wid, **kw)
    lib = oracle_lib.load()
    cfg = abi.make_config(w.N, w.Nvo)
    F = w.n_factors
    ps = np.zeros((F, 28)); im = np.zeros((w.N - 1, 465)); cost = np.zeros(1)
    lib.isvo_linearize(C.byref(cfg), C.byref(w.c()), ps.ctypes.data_as(dp), im.ctypes.data_as(dp), None, cost.ctypes.data_as(dp))
    o = w.clone(); s = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
    lib.isvo_optimize(C.byref(cfg), C.byref(o.c()), C.byref(s), C.byref(mg))
    n = s.iterations
    return w, dict(
        in_obs_sum=np.array([w.obs_point.sum(), w.Ps.sum(), w.Rs.sum(), np.ctypeslib.as_array(w.imu[0].covariance).sum()]),
        proj_strips=ps, imu_strips=im, cost0=cost,
        iterations=np.array([n]), termination=np.array([s.termination]),
        trace_cost=np.array(s.trace_cost[: n + 1]), trace_accepted=np.array(s.trace_accepted[: n + 1]),
        Ps=o.Ps, Rs=o.Rs, Vs=o.Vs, Bas=o.Bas, Bgs=o.Bgs, depth=o.lm_depth[: o.L].copy(), priors=o.priors_vector(),
        fwd_info=np.array(mg.forward_pose_prior.sqrt_info).reshape(6, 6), bwd_rel_info=np.array(mg.backward_relpose.sqrt_info).reshape(6, 6),
        bwd_vb_info=np.array(mg.backward_vb.sqrt_info).reshape(9, 9), bwd_rp_info=np.array(mg.backward_rollpitch.sqrt_info).reshape(2, 2),
        backward_kld=np.array([mg.backward_kld]))


if __name__ == "__main__":
    for case in CASES:
        _, d = run(case)
        np.savez_compressed(os.path.join(HERE, case + ".npz"), **d)
        print(case, {k: v.shape for k, v in d.items()})
