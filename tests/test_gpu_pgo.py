"""GPU parity of the pose-graph optimisation (k_pgo: PoseGraph::optimizeCS, reference src/pose_graph/pose_graph.cpp:234-428)
against the CPU oracle, through the C ABI (include/isvins_posegraph.h).  The GPU factorises the normal equations block
sparse on the row envelope and gets the marginal covariances by a Takahashi selected inversion; the oracle forms them
densely and inverts.  Tolerances: iteration count, termination and accept pattern identical; cost trace 1e-8 relative;
optimised poses 1e-8; stored covariances 1e-6 relative to each block's largest entry; drift 1e-8."""
import ctypes as C

import numpy as np
import pytest

from isvins_amd import abi, posegraph as pg
from test_oracle_pgo import oracle_pgo

pytestmark = pytest.mark.gpu


def check_graph(o, ro, g, rg, first, cur_pos):
    assert rg.status == 0 and ro.status == 0
    assert (rg.iterations, rg.termination, rg.num_successful) == (ro.iterations, ro.termination, ro.num_successful)
    assert (rg.n_poses, rg.n_free, rg.n_loop_edges) == (ro.n_poses, ro.n_free, ro.n_loop_edges)
    n = ro.iterations
    assert list(rg.trace_accepted[: n + 1]) == list(ro.trace_accepted[: n + 1])
    tc_o, tc_g = np.array(ro.trace_cost[: n + 1]), np.array(rg.trace_cost[: n + 1])
    assert np.allclose(tc_g, tc_o, rtol=1e-8, atol=1e-14), np.abs(tc_g - tc_o).max()
    assert abs(rg.final_cost - ro.final_cost) <= 1e-8 * max(1e-12, ro.final_cost) + 1e-14
    for k in range(len(o)):
        assert np.abs(abi.arr(g[k].T_w_i) - abi.arr(o[k].T_w_i)).max() < 1e-8, k
        assert np.abs(abi.arr(g[k].R_w_i) - abi.arr(o[k].R_w_i)).max() < 1e-8, k
        assert g[k].cov_computed == o[k].cov_computed
        if o[k].cov_computed:
            co, cg = abi.arr(o[k].cov), abi.arr(g[k].cov)
            assert np.abs(cg - co).max() <= 1e-6 * max(np.abs(co).max(), 1e-300), (k, np.abs(cg - co).max() / max(np.abs(co).max(), 1e-300))
        for f in ("delta_t", "delta_R"):
            assert np.abs(abi.arr(getattr(g[k].relative_pose, f)) - abi.arr(getattr(o[k].relative_pose, f))).max() < 1e-8
    assert abs(rg.yaw_drift - ro.yaw_drift) < 1e-6
    assert np.abs(abi.arr(rg.r_drift) - abi.arr(ro.r_drift)).max() < 1e-8 and np.abs(abi.arr(rg.t_drift) - abi.arr(ro.t_drift)).max() < 1e-8


@pytest.fixture(scope="module")
def opt():
    from isvins_amd import backend
    backend.build()
    o = pg.PoseGraphOptimizer(1024, max_graphs=8)
    yield o
    o.close()


@pytest.mark.parametrize("seed,K,loops", [(0, 120, 3), (1, 200, 6), (2, 60, 0), (4, 400, 10), (5, 12, 1)])
def test_optimize_matches_oracle(oracle, opt, seed, K, loops):
    kf, P, first = pg.make_pose_graph(seed, K, loops)
    o, ro = oracle_pgo(oracle, kf, first, K - 1)
    g = pg.clone_keyframes(kf)
    rg = opt.optimize(g, first, K - 1)
    check_graph(o, ro, g, rg, first, K - 1)


def test_cur_in_the_middle_and_drift_correction(oracle, opt):
    """cur_index before the end of the list: the later keyframes are not optimised, they get the drift correction
    r_drift * P_vio + t_drift (pose_graph.cpp:400-407); the newest loop edge (cur's own) is NOT in the solve (:314)"""
    K = 150
    kf, P, first = pg.make_pose_graph(9, K, 4)
    loops = [k for k in range(K) if kf[k].has_loop]
    cur = loops[-1]
    o, ro = oracle_pgo(oracle, kf, first, cur)
    g = pg.clone_keyframes(kf)
    rg = opt.optimize(g, first, cur)
    check_graph(o, ro, g, rg, first, cur)
    assert ro.n_loop_edges == len([k for k in loops if first <= k < cur])
    assert np.abs(abi.arr(g[K - 1].T_w_i) - abi.arr(kf[K - 1].vio_T_w_i)).max() > 1e-6       # the tail moved with the drift correction


def test_constant_sequence_zero_and_huber_region(oracle, opt):
    """keyframes of sequence 0 are held constant (pose_graph.cpp:291): a graph whose first third belongs to sequence 0;
    and loop closures far outside the Huber radius (|r| >> 0.1: the linear branch of the corrector)"""
    K = 90
    kf, P, first = pg.make_pose_graph(13, K, 5, drift=0.01)
    for k in range(30):
        kf[k].sequence = 0
    for k in range(K):
        if kf[k].has_loop:
            kf[k].loop_weight = 5e3
    o, ro = oracle_pgo(oracle, kf, first, K - 1)
    g = pg.clone_keyframes(kf)
    rg = opt.optimize(g, first, K - 1)
    check_graph(o, ro, g, rg, first, K - 1)
    assert ro.n_free < ro.n_poses - 1
    for k in range(first, 30):
        assert list(g[k].T_w_i) == list(kf[k].vio_T_w_i)


def test_batch_of_graphs_is_bitwise_the_single_calls(oracle, opt):
    """several pose graphs (one per sequence) in one launch: every graph bitwise equal to its own single call"""
    specs = [(20, 100, 3), (21, 64, 2), (22, 250, 7), (23, 31, 0), (24, 180, 4)]
    graphs = [pg.make_pose_graph(*s) for s in specs]
    singles = []
    for (kf, P, first), s in zip(graphs, specs):
        g = pg.clone_keyframes(kf); r = opt.optimize(g, first, s[1] - 1); singles.append((g, r))
    batch = [pg.clone_keyframes(kf) for (kf, P, first) in graphs]
    res = opt.optimize_batch(batch, [first for (_, _, first) in graphs], [s[1] - 1 for s in specs])
    for (gs, rs), gb, rb in zip(singles, batch, res):
        assert bytes(gs) == bytes(gb)
        assert (rs.iterations, rs.termination, rs.final_cost) == (rb.iterations, rb.termination, rb.final_cost)
    o, ro = oracle_pgo(oracle, graphs[2][0], graphs[2][2], specs[2][1] - 1)
    check_graph(o, ro, batch[2], res[2], graphs[2][2], specs[2][1] - 1)


@pytest.mark.parametrize("chunks", ["1", "3", "4"])
def test_chunked_batch_pipeline_is_bitwise_the_single_chunk_call(monkeypatch, chunks):
    """round 4: a batch call goes to the device in up to four chunks of graphs on four streams (k_pgo launched per chunk with the
    index of its first graph; assembly, copies, kernels and write-back of different chunks overlap, one team of host threads).
    70 graphs of five different shapes (>= 64: the default takes four chunks) against the same call forced to 1 / 3 / 4 chunks
    (ISV_PGO_CHUNKS, read per call): every keyframe list and every result record bitwise equal, on a cold and on a cached call;
    the chunk boundaries fall inside runs of equal and of different graphs"""
    specs = [(30, 90, 3), (31, 40, 1), (32, 150, 5), (33, 25, 0), (34, 120, 2)]
    graphs = [pg.make_pose_graph(*s) for s in specs]
    n = 70
    firsts = [graphs[i % 5][2] for i in range(n)]; curs = [specs[i % 5][1] - 1 for i in range(n)]
    ref = pg.PoseGraphOptimizer(160, max_graphs=n, max_loop_blocks=8 * 160)
    tst = pg.PoseGraphOptimizer(160, max_graphs=n, max_loop_blocks=8 * 160)
    try:
        for rep in range(2):                         # (the second call hits the structure cache)
            a = [pg.clone_keyframes(graphs[i % 5][0]) for i in range(n)]
            b = [pg.clone_keyframes(graphs[i % 5][0]) for i in range(n)]
            ra = ref.optimize_batch(a, firsts, curs)
            monkeypatch.setenv("ISV_PGO_CHUNKS", chunks)
            rb = tst.optimize_batch(b, firsts, curs)
            monkeypatch.delenv("ISV_PGO_CHUNKS")
            for i in range(n):
                assert bytes(a[i]) == bytes(b[i]), (rep, i)
                assert bytes(ra[i]) == bytes(rb[i]), (rep, i)
                assert ra[i].status == 0 and ra[i].iterations > 0
        assert tst.structure_cache_hits() == n and tst.last_kernel_ms()[0] > 0
    finally:
        ref.close(); tst.close()


def test_global_index_path_is_bitwise_the_lds_one(opt, monkeypatch):
    """graphs whose envelope index arrays do not fit the kernel's 48 KB of LDS read them from global memory instead: forced on
    an ordinary graph (ISV_PGO_IDX_GLOBAL), the result has to be the same bits as with the LDS copies"""
    kf, P, first = pg.make_pose_graph(11, 300, 8)
    a = pg.clone_keyframes(kf)
    ra = opt.optimize(a, first, 299)
    monkeypatch.setenv("ISV_PGO_IDX_GLOBAL", "1")
    b = pg.clone_keyframes(kf)
    rb = opt.optimize(b, first, 299)
    assert ra.status == 0 and rb.status == 0
    assert (ra.iterations, ra.termination) == (rb.iterations, rb.termination)
    assert list(ra.trace_cost) == list(rb.trace_cost)
    for k in range(len(a)):
        assert np.array_equal(abi.arr(a[k].T_w_i), abi.arr(b[k].T_w_i)) and np.array_equal(abi.arr(a[k].R_w_i), abi.arr(b[k].R_w_i))
        assert np.array_equal(abi.arr(a[k].cov), abi.arr(b[k].cov))


@pytest.mark.parametrize("seed,K,loops", [(11, 300, 8), (3, 200, 5), (7, 40, 1), (9, 64, 0)])
def test_four_wavefronts_per_graph_are_bitwise_the_one_wavefront_form(opt, monkeypatch, seed, K, loops):
    """round 5: a call that leaves CUs idle runs k_pgo<4> (evaluation, assembly and the element-wise passes on four wavefronts, the
    recurrences on wavefront 0); a call that fills the GPU runs k_pgo<1>.  ISV_PGO_WAVES forces either form (read per call): every
    trace entry, pose, covariance and drift has to be the same bits."""
    kf, P, first = pg.make_pose_graph(seed, K, loops)
    out = []
    for waves in ("1", "4"):
        monkeypatch.setenv("ISV_PGO_WAVES", waves)
        g = pg.clone_keyframes(kf)
        out.append((g, opt.optimize(g, first, K - 1)))
    (a, ra), (b, rb) = out
    assert ra.status == 0 and rb.status == 0
    assert (ra.iterations, ra.termination, ra.num_successful) == (rb.iterations, rb.termination, rb.num_successful)
    assert list(ra.trace_cost) == list(rb.trace_cost) and list(ra.trace_accepted) == list(rb.trace_accepted)
    assert ra.final_cost == rb.final_cost and ra.yaw_drift == rb.yaw_drift
    assert np.array_equal(abi.arr(ra.r_drift), abi.arr(rb.r_drift)) and np.array_equal(abi.arr(ra.t_drift), abi.arr(rb.t_drift))
    for k in range(len(a)):
        assert np.array_equal(abi.arr(a[k].T_w_i), abi.arr(b[k].T_w_i)) and np.array_equal(abi.arr(a[k].R_w_i), abi.arr(b[k].R_w_i)), k
        assert np.array_equal(abi.arr(a[k].cov), abi.arr(b[k].cov)), k


def test_loop_pose_output_file(opt, tmp_path):
    """./loop_pose_output.txt (pose_graph.cpp:412-423): one `fixed` row per keyframe, stamp px py pz qw qx qy qz of getPose()"""
    kf, P, first = pg.make_pose_graph(30, 50, 2)
    opt.optimize(kf, first, 49)
    path = tmp_path / "loop_pose_output.txt"
    opt.write_loop_pose_output(path, kf)
    rows = np.loadtxt(path)
    assert rows.shape == (50, 8)
    from scipy.spatial.transform import Rotation as Rot
    for k in (0, 17, 49):
        assert abs(rows[k, 0] - kf[k].time_stamp) < 1e-6 and np.abs(rows[k, 1:4] - abi.arr(kf[k].T_w_i)).max() <= 5.0000001e-7
        q = rows[k, 4:]                                                     # w x y z
        R = Rot.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        assert np.abs(R - abi.arr(kf[k].R_w_i, (3, 3))).max() < 1e-5
    assert all(len(x.split(".")[1]) == 6 for x in open(path).readline().split())


def test_invalid_inputs_are_refused(opt):
    kf, P, first = pg.make_pose_graph(31, 20, 1)
    from isvins_amd import backend
    with pytest.raises(backend.BackendError):
        opt.optimize(kf, first, 999)                       # cur_index not in the list
    bad = pg.clone_keyframes(kf)
    j = [k for k in range(20) if kf[k].has_loop][0]
    bad[j].loop_index = 0
    if first > 0:
        with pytest.raises(backend.BackendError):
            opt.optimize(bad, first, 19)                   # loop partner before first_looped_index: the reference asserts
    bad2 = pg.clone_keyframes(kf); bad2[3].vio_R_w_i[0] = float("nan")
    with pytest.raises(backend.BackendError):
        opt.optimize(bad2, 0, 19)


def test_structure_cache_reuses_the_analysis_and_changes_nothing(oracle):
    """round 4: the structure analysis of a graph slot (parameter blocks, adjacency, skyline, column patterns) is kept with the handle
    and reused while the keyframe list's structure is unchanged -- PoseGraph::optimizeCS (src/pose_graph/pose_graph.cpp:234-428)
    re-optimises the same list with new numbers.  A second call on the same batch hits the cache for every graph and gives the
    bytes of the first; changed measurements on the same structure are picked up (against a fresh handle); a graph that grew by one
    keyframe misses and is analysed again (checked against the oracle)."""
    specs = [(40, 90, 3), (41, 64, 2), (42, 120, 4)]
    graphs = [pg.make_pose_graph(*s) for s in specs]
    firsts = [f for (_, _, f) in graphs]; curs = [s[1] - 1 for s in specs]
    opt = pg.PoseGraphOptimizer(200, max_graphs=4, max_loop_blocks=8 * 200)
    fresh = pg.PoseGraphOptimizer(200, max_graphs=4, max_loop_blocks=8 * 200)
    try:
        a = [pg.clone_keyframes(kf) for (kf, _, _) in graphs]; ra = opt.optimize_batch(a, firsts, curs)
        assert opt.structure_cache_hits() == 0
        b = [pg.clone_keyframes(kf) for (kf, _, _) in graphs]; rb = opt.optimize_batch(b, firsts, curs)
        assert opt.structure_cache_hits() == 3
        for x, y, r1, r2 in zip(a, b, ra, rb):
            assert bytes(x) == bytes(y) and (r1.iterations, r1.final_cost) == (r2.iterations, r2.final_cost)
        # same structure, other numbers: the cache must not serve stale measurements
        c = [pg.clone_keyframes(kf) for (kf, _, _) in graphs]
        for g in c:
            for k in range(5, 15):
                g[k].relative_pose.delta_t[0] += 0.01; g[k].vio_T_w_i[1] += 0.02
        c2 = [pg.clone_keyframes(g) for g in c]
        rc = opt.optimize_batch(c, firsts, curs); rf = fresh.optimize_batch(c2, firsts, curs)
        assert opt.structure_cache_hits() == 6 and fresh.structure_cache_hits() == 0
        for x, y, r1, r2 in zip(c, c2, rc, rf):
            assert bytes(x) == bytes(y) and (r1.iterations, r1.final_cost) == (r2.iterations, r2.final_cost)
        # the graph grows by one keyframe: a miss, analysed again, against the oracle
        kf, P, first = pg.make_pose_graph(40, 91, 3)
        g = pg.clone_keyframes(kf); r = opt.optimize(g, first, 90)
        assert opt.structure_cache_hits() == 6
        o, ro = oracle_pgo(oracle, kf, first, 90)
        check_graph(o, ro, g, r, first, 90)
    finally:
        opt.close(); fresh.close()
