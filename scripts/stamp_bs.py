import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 11
NVO = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ws = synth.make_windows(range(B), n_frames=N, n_vo=NVO)
be = backend.Backend(N, NVO, max_landmarks=300, max_obs=max(w.n_obs for w in ws), max_batch=B)
be.upload(ws)
be.run_optimize()
print(f"B = {B} N = {N} Nvo = {NVO}")
dbg = be.debug_read(21, B * 64).reshape(B, 64)
names = {0: "load Tvis + zero", 1: "imu gather", 2: "priors", 3: "scale+qT", 4: "sb chains", 5: "Y Y^T", 6: "pose cholesky", 7: "solves", 8: "outputs"}
tot = dbg[:, :9].sum(1)
if dbg[:, 32:38].sum() > 0:
    print("split solve, chain kernel (MODE 1), us per call:")
    for k, nm in {32: "load + zero", 33: "imu gather", 34: "priors", 35: "scale + t_Y + q_ss", 36: "sb chains", 37: "Y Y^T + hand-over"}.items():
        print(f"  {nm:30s} {np.median(dbg[:, k]) / 10 / 100:8.1f} us")
    print("split solve, pose kernel (MODE 2): slot 0 = loads + assembly + scaling, then pose cholesky / solves / outputs:")
print("per k_build_solve_sb call (us), median over windows; 10 calls/solve; wall_clock64 = 100 MHz")
for k, nm in names.items():
    print(f"  {nm:30s} {np.median(dbg[:, k]) / 10 / 100:8.1f} us")
print("  total           ", np.median(tot) / 10 / 100)
for k, nm in {48: "dogleg: backsub + first loops", 49: "dogleg: step, candidate", 50: "dogleg: norms + stage + sync", 51: "dogleg: w0 imu raw residual", 56: "dogleg: w1 priors", 62: "dogleg: w2 model pieces", 52: "dogleg: wait for slowest wave", 54: "k_dogleg: dogleg body (total)", 55: "k_dogleg: step control (total)", 53: "dogleg: weighted + sums", 20: "chol: panel (w0)", 21: "chol: barrier after panel", 22: "chol: w0 diag update + factor", 23: "chol: barrier after trailing", 40: "sweep: prologue (sched/off -> LDS)", 41: "sweep: wave 0 main loop", 42: "sweep: barrier wait", 43: "sweep: stage 2"}.items():
    print(f"  {nm:36s} {np.median(dbg[:, k]) / 10 / 100:8.1f} us")
for k, nm in {24: "lin_gram: staging (pose, sched, lam, pts) + sync", 25: "lin_gram: soff / wst + first stream load issued", 26: "lin_gram: factor evaluation + stores (wave 0, all chunks)",
              27: "lin_gram: LDS tile + MFMA rounds (wave 0) [rest: trailing sync]", 57: "lin_gram:   tile write + sync", 58: "lin_gram:   operand reads", 59: "lin_gram:   MFMA segments + group flushes", 28: "lin_gram: (loop exit)", 29: "lin_gram: wait for the slowest wavefront", 30: "lin_gram: fold + Tvis stores",
              10: "rank1: meta + landmark scalars prologue", 11: "rank1: barrier (pass consumed)", 12: "rank1: commit (wait gathers, LDS write) + barrier", 13: "rank1: issue next gathers",
              14: "rank1: MFMA loop", 15: "rank1: epilogue (Tvis read-modify-write)"}.items():
    print(f"  {nm:60s} {np.median(dbg[:, k]) / 10 / 100:8.1f} us")
