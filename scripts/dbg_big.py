import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
from isvins_amd import backend, synth, abi
import oracle_lib, numpy as np
N, Nvo, L = [int(x) for x in sys.argv[1:4]]
lib = oracle_lib.load()
b = backend.Backend(N, Nvo, max_landmarks=L, max_obs=L * N, max_batch=2)
w = synth.make_window(50, n_frames=N, n_vo=Nvo, n_landmarks=L)
o = w.clone(); so = abi.isv_summary_t(); mo = abi.isv_marg_result_t()
lib.isvo_optimize(C.byref(b.cfg), C.byref(o.c()), C.byref(so), C.byref(mo))
g = w.clone(); sg, mg = b.optimize(g)
n = so.iterations
print("iters", so.iterations, sg.iterations, "term", so.termination, sg.termination)
print("acc o", list(so.trace_accepted[:n + 1])); print("acc g", list(sg.trace_accepted[:n + 1]))
print("cost o", np.array(so.trace_cost[:n + 1])); print("cost g", np.array(sg.trace_cost[:n + 1]))
print("rad o", np.array(so.trace_radius[:n + 1])); print("rad g", np.array(sg.trace_radius[:n + 1]))
print("step o", np.array(so.trace_step_norm[:n + 1])); print("step g", np.array(sg.trace_step_norm[:n + 1]))
