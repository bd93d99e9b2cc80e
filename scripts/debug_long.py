"""lock-step native+GPU vs restatement+oracle on the EuRoC stand-in; stop at the first frame where they differ"""
import os, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
import ctypes as C
import numpy as np
from isvins_amd import abi, backend, estimator as E
import oracle_lib, sequence_harness as sh, test_sequence_long as T
oracle = oracle_lib.load()
N, NVO = T.N, T.NVO
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
cfg = abi.make_config(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
sim, stream = T.record_stream(n_frames)

class Tap(sh.OracleSolver):
    def optimize(self, w):
        self.pre = w.clone()
        return super().optimize(w)
tap = Tap(oracle, cfg)
eo = sh.Estimator(tap, oracle, N, NVO)
en = E.SequenceEstimator(sh.estimator_params(cfg), 1)
be = backend.Backend(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
worst = 0
for i, (imu, t, image) in enumerate(stream):
    for (dt, a, g) in imu:
        eo.process_imu(dt, a, g)
    en.process_imu_n(0, [x[0] for x in imu], [x[1] for x in imu], [x[2] for x in imu])
    boot = None
    if eo.solver_flag == "INITIAL" and eo.frame_count == N - 1:
        boot = T.bootstrap(sim, i)
        en.set_bootstrap(0, *boot)
    n0 = len(eo.summaries)
    eo.process_image(image, t, bootstrap=boot)
    ids = np.array(list(image.keys()), np.int32); pts = np.array([image[int(k)] for k in ids], float).reshape(-1, 3)
    en.push_image(0, t, ids, pts)
    solved = en.step() > 0
    if not solved:
        continue
    so = eo.summaries[-1]; sg = en.last_summary(0)
    wn = en.window(0)
    dP = np.abs(wn["Ps"] - eo.Ps).max(); dR = np.abs(wn["Rs"] - eo.Rs).max(); dV = np.abs(wn["Vs"] - eo.Vs).max()
    worst = max(worst, dP)
    same = (sg.iterations == so.iterations and sg.termination == so.termination and list(sg.trace_accepted[:so.iterations + 1]) == list(so.trace_accepted[:so.iterations + 1]))
    if i % 100 == 0:
        print(f"frame {i}: dP {dP:.2e} dR {dR:.2e} dV {dV:.2e} it {sg.iterations}/{so.iterations} cost {sg.final_cost:.6f}/{so.final_cost:.6f}", flush=True)
    if not same or dP > 1e-7:
        print(f"FIRST DIFFERENCE at frame {i} (t = {t:.2f}): dP {dP:.3e} dR {dR:.3e} dV {dV:.3e}")
        print("  gpu   : it", sg.iterations, "term", sg.termination, "acc", list(sg.trace_accepted[:sg.iterations + 1]), "L", en.status(0)["n_landmarks"])
        print("  oracle: it", so.iterations, "term", so.termination, "acc", list(so.trace_accepted[:so.iterations + 1]), "L", tap.pre.L)
        print("  gpu cost   ", [f"{x:.9f}" for x in sg.trace_cost[:sg.iterations + 1]])
        print("  oracle cost", [f"{x:.9f}" for x in so.trace_cost[:so.iterations + 1]])
        print("  gpu radius   ", [f"{x:.6g}" for x in sg.trace_radius[:sg.iterations + 1]])
        print("  oracle radius", [f"{x:.6g}" for x in so.trace_radius[:so.iterations + 1]])
        # replay the oracle side's pre-solve window on the bare backend
        w = tap.pre
        g = w.clone(); s2, m2 = be.optimize(g)
        o = w.clone(); s3 = abi.isv_summary_t(); m3 = abi.isv_marg_result_t(); oracle.isvo_optimize(C.byref(cfg), C.byref(o.c()), C.byref(s3), C.byref(m3))
        print("  same window, bare backend vs oracle: it", s2.iterations, s3.iterations, "term", s2.termination, s3.termination, "dP", np.abs(g.Ps - o.Ps).max())
        print("     gpu cost   ", [f"{x:.9f}" for x in s2.trace_cost[:s2.iterations + 1]])
        print("     oracle cost", [f"{x:.9f}" for x in s3.trace_cost[:s3.iterations + 1]])
        print("     acc", list(s2.trace_accepted[:s2.iterations + 1]), list(s3.trace_accepted[:s3.iterations + 1]))
        print("     step", [f"{x:.3e}" for x in s2.trace_step_norm[:s2.iterations + 1]], [f"{x:.3e}" for x in s3.trace_step_norm[:s3.iterations + 1]])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "first_diff_window.npz"), Ps=w.Ps, Rs=w.Rs, Vs=w.Vs, Bas=w.Bas, Bgs=w.Bgs, tic=w.tic, ric=w.ric,
                 lm_start_frame=w.lm_start_frame, lm_obs_ptr=w.lm_obs_ptr, obs_point=w.obs_point, lm_depth=w.lm_depth,
                 imu=np.frombuffer(bytes(w.imu), dtype=np.float64), pose_prior=np.frombuffer(bytes(w.pose_prior), dtype=np.uint8),
                 vb_prior=np.frombuffer(bytes(w.vb_prior), dtype=np.uint8), relpose=np.frombuffer(bytes(w.relpose), dtype=np.uint8),
                 rollpitch=np.frombuffer(bytes(w.rollpitch), dtype=np.uint8), n_rollpitch=w.n_rollpitch, margin_old=w.margin_old, header0=w.header0, L=w.L, n_obs=w.n_obs)
        break
print("worst dP before the difference", worst)
