"""k_build_solve_st forced on small handles (ISV_SOLVE_ST=1) against k_build_solve_sb on the same windows: iterations and the
largest state difference per (N, Nvo, generic-N) case."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
import numpy as np
from isvins_amd import backend, synth
for N, Nvo, L, gen in [(11, 5, 120, 0), (11, 5, 120, 1), (10, 5, 100, 0), (9, 4, 80, 0), (8, 4, 80, 0), (7, 3, 60, 0), (6, 3, 60, 0), (12, 5, 100, 0), (13, 6, 100, 0)]:
    ws = synth.make_windows([31, 32, 33], n_frames=N, n_vo=Nvo, n_landmarks=L)
    kw = dict(max_landmarks=L, max_obs=max(w.n_obs for w in ws), max_batch=3)
    os.environ["ISV_SOLVE_ST"] = "1"
    if gen: os.environ["ISV_GENERIC_N"] = "1"
    be = backend.Backend(N, Nvo, **kw)
    os.environ["ISV_SOLVE_ST"] = "0"
    sb = backend.Backend(N, Nvo, **kw)
    os.environ.pop("ISV_SOLVE_ST"); os.environ.pop("ISV_GENERIC_N", None)
    gs = [w.clone() for w in ws]; sums, _ = be.optimize_batch(gs)
    hs = [w.clone() for w in ws]; sums1, _ = sb.optimize_batch(hs)
    print(f"N={N} Nvo={Nvo} generic={gen}: iterations st {[s.iterations for s in sums]} sb {[s.iterations for s in sums1]} "
          f"max|dstate| {max(np.abs(g.state_vector() - h.state_vector()).max() for g, h in zip(gs, hs)):.3e}", flush=True)
    be.close(); sb.close()
