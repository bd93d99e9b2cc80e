import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
t = time.time(); ws = synth.make_windows(range(B)); print("gen s", time.time() - t, flush=True)
Ftot = sum(w.n_factors for w in ws)
be = backend.Backend(11, 5, max_landmarks=300, max_obs=1600, max_batch=B)
t = time.time(); be.upload(ws); print("upload s", time.time() - t, flush=True)
for _ in range(3): be.run_linearize()
ts = []
for _ in range(20):
    be.run_linearize(); ts.append(be.last_timing().copy())
ts = np.array(ts)
print("Ftot", Ftot, "ms total/proj/imu+prior (median):", np.median(ts[:, 0]), np.median(ts[:, 1]), np.median(ts[:, 2]))
proj_ms = np.median(ts[:, 1])
print("proj GB/s (292 B/factor):", Ftot * 292 / proj_ms / 1e6)
