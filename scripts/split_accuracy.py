"""GPU vs oracle agreement of the one-launch reduced-system solve (ISV_CHAIN_SPLIT=0) and the split solve (default on small-batch
handles), same windows: worst relative difference of the cost trace / final cost and worst absolute state difference.  GPU box."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
import numpy as np
from isvins_amd import abi, backend, synth
import oracle_lib
oracle = oracle_lib.load()

def run(split, N, Nvo, ids, L, hook=None):
    os.environ["ISV_CHAIN_SPLIT"] = "1" if split else "0"
    if hook: os.environ["ISV_DEBUG_MIN_RADIUS"] = str(hook)
    ws = synth.make_windows(ids, n_frames=N, n_vo=Nvo, n_landmarks=L)
    b = backend.Backend(N, Nvo, max_landmarks=L, max_obs=max(w.n_obs for w in ws), max_batch=len(ws))
    os.environ.pop("ISV_DEBUG_MIN_RADIUS", None)
    if hook: oracle.isvo_debug_min_radius(float(hook))
    out = []
    try:
        gs = [w.clone() for w in ws]
        sums, _ = b.optimize_batch(gs)
        for w, g, sg in zip(ws, gs, sums):
            o = w.clone(); so = abi.isv_summary_t(); mg = abi.isv_marg_result_t()
            oracle.isvo_optimize(C.byref(b.cfg), C.byref(o.c()), C.byref(so), C.byref(mg))
            n = so.iterations
            tc_o, tc_g = np.array(so.trace_cost[: n + 1]), np.array(sg.trace_cost[: n + 1])
            same = sg.iterations == n and list(sg.trace_accepted[: n + 1]) == list(so.trace_accepted[: n + 1])
            out.append((same, np.abs(tc_g / tc_o - 1).max(), abs(sg.final_cost / so.final_cost - 1), max(np.abs(g.Ps - o.Ps).max(), np.abs(g.para_SpeedBias - o.para_SpeedBias).max())))
    finally:
        if hook: oracle.isvo_debug_min_radius(0.0)
        b.close()
    return out

for (N, Nvo, ids, L, hook) in [(11, 5, range(16), 300, None), (18, 8, range(16), 300, None), (11, 5, [5], 80, 8000.0), (11, 5, range(40, 56), 80, None), (11, 5, range(2000, 2016), 16, None), (11, 5, range(2300, 2316), 16, None)]:
    for split in (0, 1):
        r = run(split, N, Nvo, list(ids), L, hook)
        print(f"N={N} L={L} hook={hook} split={split}: pattern ok {all(x[0] for x in r)}; worst trace {max(x[1] for x in r):.2e}, final cost {max(x[2] for x in r):.2e}, state {max(x[3] for x in r):.2e}; median final cost {np.median([x[2] for x in r]):.2e}", flush=True)
