"""Would two half batches on two stream pairs beat one batch?  K handles of 1024 / K windows each, driven by K host threads
(ctypes releases the GIL inside the C calls), device-resident inputs, against one handle of 1024 windows."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
import numpy as np
from isvins_amd import backend, synth

W = int(os.environ.get("W", "1024")); STEPS = int(os.environ.get("STEPS", "60"))
windows = synth.make_windows(range(W), n_frames=11, n_vo=5, n_landmarks=300)
max_obs = max(w.n_obs for w in windows)
for K in (1, 2, 4):
    hs = []
    for k in range(K):
        ws = windows[k * W // K:(k + 1) * W // K]
        b = backend.Backend(11, 5, max_landmarks=300, max_obs=max_obs, max_batch=len(ws))
        b.upload(ws); b.run_optimize(sync=True)
        hs.append(b)
    def drive(b):
        for _ in range(STEPS):
            b.run_optimize(sync=True)
    th = [threading.Thread(target=drive, args=(b,)) for b in hs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"K={K}: {1e3 * dt / STEPS:.3f} ms per {W} windows = {W * STEPS / dt / 1e3:.1f} k windows/s  (solve_st per handle: {[int(b.last_counts()[6]) for b in hs]})", flush=True)
    for b in hs: b.close()
