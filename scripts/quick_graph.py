"""single-window latency with and without the captured hipGraph (ISV_GRAPH=1, measurement hook of isv_batch_optimize):
usage: [ISV_GRAPH=1] python scripts/quick_graph.py N Nvo [L]   -> ms per optimize (HIP events), host ms of the first call"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth
import numpy as np
N, Nvo = int(sys.argv[1]), int(sys.argv[2]); L = int(sys.argv[3]) if len(sys.argv) > 3 else 300
w = synth.make_windows(range(1), n_frames=N, n_vo=Nvo, n_landmarks=L)
w2 = synth.make_windows(range(1, 2), n_frames=N, n_vo=Nvo, n_landmarks=L)
be = backend.Backend(N, Nvo) if N == 18 else backend.Backend(N, Nvo, max_landmarks=L, max_obs=max(w[0].n_obs, w2[0].n_obs), max_batch=1)
be.upload(w)
t = time.perf_counter(); be.run_optimize(sync=True); first = time.perf_counter() - t
ev, host = [], []
for _ in range(30):
    t = time.perf_counter(); be.run_optimize(sync=True); host.append(time.perf_counter() - t); ev.append(float(be.last_timing()[0]))
# a second batch of different counts: what every real frame does to the kernel arguments
be.upload(w2)
t = time.perf_counter(); be.run_optimize(sync=True); second = time.perf_counter() - t
print(f"ISV_GRAPH={'1' if os.environ.get('ISV_GRAPH') else '0'} N={N}: {np.median(ev[5:]):.3f} ms per optimize (HIP events), host wall {1e3 * np.median(host[5:]):.3f} ms; "
      f"first call {1e3 * first:.2f} ms, first call after a new upload {1e3 * second:.2f} ms", flush=True)
be.close()
