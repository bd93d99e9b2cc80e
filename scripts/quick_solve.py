import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ws = synth.make_windows(range(B))
be = backend.Backend(11, 5, max_landmarks=300, max_obs=1600, max_batch=B)
be.upload(ws)
for _ in range(2): be.run_optimize()
ts = []
for _ in range(reps):
    t = time.perf_counter(); be.run_optimize(); ts.append(time.perf_counter() - t)
print("B", B, "optimize ms (median wall):", 1e3 * np.median(ts), "ev total ms", be.last_timing()[0], "windows/s", B / np.median(ts), flush=True)
out = [w.clone() for w in ws[:4]]
