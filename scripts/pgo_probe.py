"""PGO probe: the bench's pose-graph leg with the host-side breakdown (ISV_TRACE_HANDOVER=1 prints it to stderr)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
import numpy as np
from isvins_amd import posegraph as pgm
K, loops = 200, 5
graphs = [pgm.make_pose_graph(100 + s_, K, loops) for s_ in range(8)]
for n_graphs, reps in ((1, 3), (1024, 3)):
    opt = pgm.PoseGraphOptimizer(K, max_graphs=n_graphs, max_loop_blocks=8 * K)
    firsts = [graphs[s_ % 8][2] for s_ in range(n_graphs)]; curs = [K - 1] * n_graphs
    for rep in range(reps + 1):
        batch = [pgm.clone_keyframes(graphs[s_ % 8][0]) for s_ in range(n_graphs)]
        t1 = time.perf_counter(); res = opt.optimize_batch(batch, firsts, curs); dt = time.perf_counter() - t1
        print(f"graphs {n_graphs} rep {rep}: {1e3 * dt:.2f} ms, kernel {opt.last_kernel_ms()}, its {np.mean([r.iterations for r in res]):.2f}", flush=True)
    opt.close()
