"""pose-graph optimisation on the MI355X: ms per PoseGraph::optimizeCS pass, one graph and a batch of graphs"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader; isvins_loader.load()
import ctypes as C
import numpy as np
import torch; torch.cuda.set_device(0)
from isvins_amd import posegraph as pg
import oracle_lib
oracle = oracle_lib.load()
from test_oracle_pgo import bind
bind(oracle)
CFGS = [tuple(int(x) for x in c.split(':')) for c in os.environ['PGO_CFGS'].split(',')] if os.environ.get('PGO_CFGS') else ((200, 5, 1), (1000, 10, 1), (200, 5, 256), (200, 5, 1024), (1000, 10, 256))
for K, loops, S in CFGS:
    graphs = [pg.make_pose_graph(100 + s, K, loops) for s in range(min(S, 8))]
    opt = pg.PoseGraphOptimizer(K, max_graphs=S, max_loop_blocks=8 * K)
    batch = [pg.clone_keyframes(graphs[s % len(graphs)][0]) for s in range(S)]
    firsts = [graphs[s % len(graphs)][2] for s in range(S)]; curs = [K - 1] * S
    opt.optimize_batch(batch, firsts, curs)
    ts = []
    for _ in range(3):
        batch = [pg.clone_keyframes(graphs[s % len(graphs)][0]) for s in range(S)]
        t0 = time.perf_counter(); res = opt.optimize_batch(batch, firsts, curs); ts.append(time.perf_counter() - t0)
    t = min(ts)
    its = np.mean([r.iterations for r in res])
    line = f"K={K} loops={loops} graphs={S}: {1e3 * t:.2f} ms per batch call (host prep + H2D + kernel + D2H + write-back), {1e3 * t / S:.3f} ms / graph, {its:.1f} LM iterations"
    if S == 1 and K <= 200 and not os.environ.get('PGO_NO_ORACLE'):      # (the oracle is dense: 6K x 6K, minutes beyond a few hundred keyframes)
        cfg = pg.make_config(K)
        o = pg.clone_keyframes(graphs[0][0]); r = pg.isv_pgo_result_t()
        t0 = time.perf_counter(); oracle.isvo_pgo_optimize(C.byref(cfg), K, o, firsts[0], K - 1, C.byref(r)); to = time.perf_counter() - t0
        line += f"; CPU oracle (dense normal equations, 1 thread): {1e3 * to:.1f} ms"
    print(line, flush=True)
    open(os.path.join(ROOT, 'gpurun_out', 'pgo_bench.log'), 'a').write(line + '\n')
    opt.close()
