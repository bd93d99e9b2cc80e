import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth
import numpy as np
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ws = synth.make_windows(range(B))
be = backend.Backend(11, 5, max_landmarks=300, max_obs=1600, max_batch=B)
be.upload(ws); be.run_optimize()
dbg = be.debug_read(21, B * 64).reshape(B, 64)[:, 32:42]
names = ["fwd landmark jac", "fwd 12x12 sums", "fwd priors", "pg edge (pinv, inv, chol)", "fwd prior (inv, kld, chol)", "bwd build+schur", "bwd jacobi 21", "bwd projections", "bwd kld"]
for k, nm in enumerate(names):
    print(f"  {nm:28s} {np.median(dbg[:, k]) / 100:9.1f} us")
print("  total", np.median(dbg.sum(1)) / 100)
j = dbg[:, 6] / 100
print("  bwd jacobi percentiles (us): 10/50/90/99/max", [round(float(np.percentile(j, p)), 1) for p in (10, 50, 90, 99, 100)])
b = dbg[:, 5:9].sum(1) / 100; f = dbg[:, 0:5].sum(1) / 100
print("  bwd sum percentiles (us) 10/50/90/max", [round(float(np.percentile(b, p)), 1) for p in (10, 50, 90, 100)])
print("  fwd sum percentiles (us) 10/50/90/max", [round(float(np.percentile(f, p)), 1) for p in (10, 50, 90, 100)])
for k in range(9): print("   comp", k, "p10/p50/p90/max", [round(float(np.percentile(dbg[:, k] / 100, p)), 1) for p in (10, 50, 90, 100)])
