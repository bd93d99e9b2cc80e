"""Generates tests/golden/marg_third_opinion.npz (GPU box).

The 2400-frame EuRoC stand-in (tests/test_sequence_long.py) is run teacher-forced: the restatement + oracle drive the sequence and
every MARGIN_OLD solve is repeated on the MI355X.  The K windows on which the recovered marginalisation information of the two
sides differs most are kept.  For each of them MargForward / MargBackward are then run on BOTH sides from ONE input -- the oracle's
solved window, handed to the oracle and to a HIP handle configured with NUM_ITERATIONS = 0 (no trust-region step: vector2double,
update() with old == new, double2vector, MargForward, MargBackward at exactly these states) -- and the inputs the two routines
read plus both sides' outputs go into the fixture.  tests/test_marg_third_opinion.py (CPU) recomputes the routines at 40 digits
in mpmath from the same inputs and says which side is closer.

  python scripts/marg_third_opinion_dump.py [n_frames] [K]
"""
import ctypes as C
import heapq
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
import numpy as np
from isvins_amd import abi, backend
import oracle_lib
import sequence_harness as sh
import test_sequence_long as tl

N, NVO = tl.N, tl.NVO
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else tl.N_FRAMES
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
FACTORS = (("forward_pose_prior", 6), ("backward_relpose", 6), ("backward_vb", 9), ("backward_rollpitch", 2))


def info_rel(ma, mb):
    worst = 0.0
    for name, k in FACTORS:
        A = abi.arr(getattr(ma, name).sqrt_info, (k, k)); B = abi.arr(getattr(mb, name).sqrt_info, (k, k))
        worst = max(worst, float(np.abs(A.T @ A - B.T @ B).max() / np.abs(B.T @ B).max()))
    return worst


class Collect(sh.OracleSolver):
    def __init__(self, lib, cfg, be):
        super().__init__(lib, cfg)
        self.be, self.n, self.heap = be, 0, []

    def triangulate(self, w):
        super().triangulate(w)

    def optimize(self, w):
        g = w.clone()
        s, m = super().optimize(w)
        sg, mg = self.be.optimize(g)
        self.n += 1
        if w.margin_old and sg.iterations == s.iterations:
            rel = info_rel(mg, m)
            item = (rel, self.n, w.clone())
            if len(self.heap) < K: heapq.heappush(self.heap, item)
            elif rel > self.heap[0][0]: heapq.heapreplace(self.heap, item)
        return s, m


def raw(struct):
    return np.frombuffer(bytes(struct), dtype=np.uint8).copy()


def main():
    oracle = oracle_lib.load()
    cfg = abi.make_config(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
    be = backend.Backend(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
    sim, stream = tl.record_stream(n_frames)
    col = Collect(oracle, cfg, be)
    est = sh.Estimator(col, oracle, N, NVO)
    for i, (imu, t, image) in enumerate(stream):
        for (dt, a, g) in imu:
            est.process_imu(dt, a, g)
        boot = tl.bootstrap(sim, i) if (est.solver_flag == "INITIAL" and est.frame_count == N - 1) else None
        est.process_image(image, t, bootstrap=boot)
    be.close()
    picks = sorted(col.heap, key=lambda x: -x[0])
    print(f"{col.n} solves; the {len(picks)} largest GPU / oracle differences of the recovered information (each side from its OWN solve): "
          + ", ".join(f"{r:.1e} (solve {k})" for r, k, _ in picks), flush=True)
    cfg0 = abi.make_config(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1, num_iterations=0)
    be0 = backend.Backend(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1, num_iterations=0)
    out = dict(K=len(picks), N=N, Nvo=NVO, alpha=cfg.alpha, gravity=np.array(cfg.gravity[:]), proj_sqrt_info=np.array(cfg.proj_sqrt_info[:]))
    for k, (rel, idx, w) in enumerate(picks):
        o = w.clone(); so = abi.isv_summary_t(); mo = abi.isv_marg_result_t()
        assert oracle.isvo_optimize(C.byref(cfg0), C.byref(o.c()), C.byref(so), C.byref(mo)) == 0 and so.iterations == 0
        g = w.clone()
        sg, mg = be0.optimize(g)
        assert sg.iterations == 0 and mg.valid == 1 and mo.valid == 1
        p = f"w{k}_"
        out[p + "solve"] = idx; out[p + "own_solve_rel"] = rel
        out[p + "same_input_rel"] = info_rel(mg, mo)
        out[p + "input_mismatch"] = max(np.abs(g.para_Pose - o.para_Pose).max(), np.abs(g.para_SpeedBias - o.para_SpeedBias).max())
        # what MargForward / MargBackward read (oracle side's values; src/estimator.cpp:1149-1539)
        v = NVO
        out[p + "pose"] = o.para_Pose[[0, 1, v - 1, v]].copy(); out[p + "sb"] = o.para_SpeedBias[[v - 1, v]].copy(); out[p + "ex"] = o.para_Ex_Pose.copy()
        sel = [l for l in range(o.L) if o.lm_start_frame[l] == 0 and o.lm_obs_ptr[l + 1] - o.lm_obs_ptr[l] >= 2]
        out[p + "lam"] = np.array([o.para_Feature[l] for l in sel])
        out[p + "pts"] = np.array([[o.obs_point[o.lm_obs_ptr[l]], o.obs_point[o.lm_obs_ptr[l] + 1]] for l in sel]).reshape(len(sel), 2, 3)
        out[p + "pose_prior"] = raw(o.pose_prior); out[p + "vb_prior"] = raw(o.vb_prior); out[p + "relpose0"] = raw(o.relpose[0])
        out[p + "imu"] = raw(o.imu[v - 1])
        for side, m in (("gpu", mg), ("oracle", mo)):
            for name, kk in FACTORS + (("combined_relpose", 6),):
                f = m.combined.relative_pose if name == "combined_relpose" else getattr(m, name)
                out[p + side + "_" + name] = abi.arr(f.sqrt_info, (kk, kk)).copy()
            out[p + side + "_covRel"] = abi.arr(m.combined.covRel, (6, 6)).copy()
        print(f"  window {k}: solve {idx}, own-solve difference {rel:.2e}, same-input difference {out[p + 'same_input_rel']:.2e}, input mismatch {out[p + 'input_mismatch']:.1e}, {len(sel)} marginalised landmarks", flush=True)
    be0.close()
    dst = os.path.join(ROOT, "gpurun_out", "marg_third_opinion.npz")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    np.savez_compressed(dst, **out)
    print("wrote", dst)


if __name__ == "__main__":
    main()
