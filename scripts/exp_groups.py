"""experiment: G handles x (1024 / G) windows enqueued concurrently vs one handle x 1024 (do kernels of different
iteration phases overlap on the CUs?)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[2] if len(sys.argv) > 2 else "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
import torch
torch.cuda.set_device(0)
from isvins_amd import backend, synth
N, V, L = int(os.environ.get("N", 11)), int(os.environ.get("V", 5)), 300
ws = synth.make_windows(range(1024), n_frames=N, n_vo=V, n_landmarks=L)
mo = max(w.n_obs for w in ws)
for G in [int(x) for x in sys.argv[1].split(",")]:
    per = 1024 // G
    bes = []
    for g in range(G):
        b = backend.Backend(N, V, max_landmarks=L, max_obs=mo, max_batch=per)
        b.upload(ws[g * per:(g + 1) * per]); bes.append(b)
    for _ in range(3):
        for b in bes: b.run_optimize(sync=False)
        for b in bes: b.sync()
    t0 = time.perf_counter()
    K = 30
    for _ in range(K):
        for b in bes: b.run_optimize(sync=False)
    for b in bes: b.sync()
    dt = (time.perf_counter() - t0) / K
    print(f"G={G}: {1e3 * dt:.3f} ms per 1024 windows = {1024 / dt:.0f} windows/s", flush=True)
    for b in bes: b.close()
