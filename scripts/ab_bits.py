"""Bit-level A/B of two builds of the library: solves a fixed set of windows (several window lengths, batch sizes, both
marginalisation branches, a free extrinsic, the linearise API) with the library named by ISVINS_LIB (default: the in-tree one)
and prints one sha256 per case over the solved states, summaries and marginalisation records.  Run it once per library and diff."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
import numpy as np
from isvins_amd import backend, synth


def digest(ws, sums, margs):
    h = hashlib.sha256()
    for w in ws:
        for a in (w.Ps, w.Rs, w.Vs, w.Bas, w.Bgs, w.tic, w.ric, w.lm_depth, w.lm_solve_flag):
            h.update(np.ascontiguousarray(a).tobytes())
    for s in sums:
        h.update(bytes(s))
    for m in margs:
        h.update(bytes(m))
    return h.hexdigest()[:16]


cases = [("N11 B1", 11, 5, 300, 1, 1, 0), ("N11 B37", 11, 5, 120, 37, 64, 0), ("N11 B300 cap1024", 11, 5, 150, 300, 1024, 0),
         ("N18 B5", 18, 8, 200, 5, 8, 0), ("N18 B40 cap512", 18, 8, 100, 40, 512, 0), ("N18 B300 cap512", 18, 8, 100, 300, 512, 0), ("N11 B600 cap1024", 11, 5, 300, 600, 1024, 0), ("N11 ex B3", 11, 5, 120, 3, 4, 1), ("N24 B2", 24, 10, 80, 2, 2, 0)]
for name, N, Nvo, L, B, cap, ex in cases:
    ws = [synth.make_window(i, n_frames=N, n_vo=Nvo, n_landmarks=L, margin_old=i % 2) for i in range(B)]
    be = backend.Backend(N, Nvo, max_landmarks=L + 20, max_obs=max(w.n_obs for w in ws), max_batch=cap, estimate_extrinsic=ex)
    ps, im, cost = be.linearize(ws[0])
    hl = hashlib.sha256(np.ascontiguousarray(ps).tobytes() + np.ascontiguousarray(im).tobytes() + np.float64(cost).tobytes()).hexdigest()[:16]
    sums, margs = be.optimize_batch(ws)
    print(f"{name:22s} linearize {hl} solve {digest(ws, sums, margs)}", flush=True)
    be.close()
