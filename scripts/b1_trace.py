import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
import torch; torch.cuda.set_device(0)
from isvins_amd import backend, synth
N, V = int(os.environ.get("N", 11)), int(os.environ.get("V", 5))
ws = synth.make_windows([0], n_frames=N, n_vo=V, n_landmarks=300)
b = backend.Backend(N, V, max_landmarks=300, max_obs=ws[0].n_obs, max_batch=1)
b.upload(ws)
import numpy as np
ts = []
for _ in range(12):
    b.run_optimize(sync=True); ts.append(b.last_timing()[0])
print("B=1 ms per optimize:", np.round(ts, 3))
