import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('TORCH'):
    import torch; torch.cuda.init(); torch.zeros(1, device='cuda')
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth
import numpy as np
B, N, Nvo, L = [int(x) for x in sys.argv[1:5]]
tf = int(sys.argv[5]) if len(sys.argv) > 5 and int(sys.argv[5]) > 0 else None
ws = synth.make_windows(range(B), n_frames=N, n_vo=Nvo, n_landmarks=L, target_factors=tf)
be = backend.Backend(N, Nvo, max_landmarks=L, max_obs=max(w.n_obs for w in ws), max_batch=B)
be.upload(ws)
for _ in range(2): be.run_optimize()
ts = []
for _ in range(int(os.environ.get('REPS', 5))):
    t = time.perf_counter(); be.run_optimize(); ts.append(time.perf_counter() - t)
if os.environ.get('ASYNC'):
    K = int(os.environ.get('REPS', 5))
    t = time.perf_counter()
    for _ in range(K): be.run_optimize(sync=False)
    be.sync(); ts = [(time.perf_counter() - t) / K]
F = sum(w.n_factors for w in ws)
print(f"B={B} N={N} Nvo={Nvo} L={L} F/window={F/B:.0f}: optimize {1e3*np.median(ts):.3f} ms/batch, {1e3*np.median(ts)/B:.4f} ms/window, {B/np.median(ts):.0f} windows/s; fused visual/control = {be.last_counts()[4:6]}", flush=True)
