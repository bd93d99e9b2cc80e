"""Golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the CPU oracle;
the reference cannot run in this image).  CPU: the oracle still reproduces them (pins the oracle
against accidental change, checks the input generator is deterministic).  GPU: the HIP path matches
them through the C ABI, including the reference's own window shape N = 18 / Nvo = 8 (generic,
global-scratch variant of k_build_solve)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402
from isvins_amd import abi, backend, synth  # noqa: E402

CASES = list(make_golden.CASES)


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_golden(case):
    g = np.load(os.path.join(HERE, "golden", case + ".npz"))
    w, d = make_golden.run(case)
    assert np.array_equal(d["in_obs_sum"], g["in_obs_sum"]), "synthetic generator is not deterministic"
    for k in ("proj_strips", "imu_strips", "cost0", "trace_cost", "Ps", "Rs", "Vs", "depth", "priors", "fwd_info", "bwd_vb_info"):
        assert np.allclose(d[k], g[k], rtol=1e-12, atol=1e-12 * max(1.0, np.abs(g[k]).max())), k
    assert int(d["iterations"][0]) == int(g["iterations"][0]) and int(d["termination"][0]) == int(g["termination"][0])
    assert np.array_equal(d["trace_accepted"], g["trace_accepted"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_matches_golden(case):
    g = np.load(os.path.join(HERE, "golden", case + ".npz"))
    kw = dict(make_golden.CASES[case]); wid = kw.pop("window_id")
    w = synth.make_window(wid, **kw)
    backend.build()
    be = backend.Backend(w.N, w.Nvo, max_landmarks=64, max_obs=64 * w.N, max_batch=2)
    ps, im, cost = be.linearize(w)
    sc = np.maximum(1.0, np.abs(g["proj_strips"]).max(axis=1, keepdims=True))
    assert (np.abs(ps - g["proj_strips"]) / sc).max() < 1e-9
    sc = np.maximum(1.0, np.abs(g["imu_strips"]).max(axis=1, keepdims=True))
    assert (np.abs(im - g["imu_strips"]) / sc).max() < 1e-9
    assert abs(cost - g["cost0"][0]) < 1e-11 * g["cost0"][0]
    o = w.clone()
    s, mg = be.optimize(o)
    n = int(g["iterations"][0])
    assert s.iterations == n and s.termination == int(g["termination"][0])
    assert list(s.trace_accepted[: n + 1]) == list(g["trace_accepted"])
    assert np.allclose(np.array(s.trace_cost[: n + 1]), g["trace_cost"], rtol=1e-7)
    for k in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
        assert np.abs(getattr(o, k) - g[k]).max() < 1e-7, k
    assert np.abs(o.priors_vector() - g["priors"]).max() < 1e-7
    assert np.abs(o.lm_depth[: o.L] - g["depth"]).max() < 1e-5 * max(1.0, np.abs(g["depth"]).max())
    for name, key, nn in (("forward_pose_prior", "fwd_info", 6), ("backward_relpose", "bwd_rel_info", 6), ("backward_vb", "bwd_vb_info", 9), ("backward_rollpitch", "bwd_rp_info", 2)):
        U = np.array(getattr(mg, name).sqrt_info).reshape(nn, nn); G = g[key]
        assert np.abs(U.T @ U - G.T @ G).max() < 1e-6 * np.abs(G.T @ G).max(), name
    be.close()


# ---- whole-sequence fixture (tests/golden/make_sequence_golden.py) ------------------------------------------------
import make_sequence_golden  # noqa: E402
import sequence_harness as sh  # noqa: E402

SEQ_CASES = list(make_sequence_golden.CASES)


@pytest.mark.parametrize("case", SEQ_CASES)
def test_restatement_reproduces_sequence_golden(case):
    g = np.load(os.path.join(HERE, "golden", case + ".npz"))
    _, d = make_sequence_golden.run(case)
    assert np.array_equal(d["in_fingerprint"], g["in_fingerprint"]), "the simulator is not deterministic"
    assert np.array_equal(d["margin_history"], g["margin_history"]) and np.array_equal(d["iterations"], g["iterations"])
    assert np.array_equal(d["counters"], g["counters"])
    for k in ("pose_output", "newest", "Ps", "Rs", "Vs", "Bas", "Bgs", "Headers"):
        assert np.allclose(d[k], g[k], rtol=0, atol=1e-10), k


def _native_vs_sequence_golden(case, solver, tol):
    from isvins_amd import estimator as E
    g = np.load(os.path.join(HERE, "golden", case + ".npz"))
    kw = make_sequence_golden.CASES[case]
    cfg = abi.make_config(kw["N"], kw["Nvo"], max_landmarks=800, max_obs=800 * kw["N"], max_batch=1)
    est = E.SequenceEstimator(sh.estimator_params(cfg), 1, solver=solver(cfg) if solver else None)
    sh.run_sequences_native(est, kw["N"], kw["n_frames"], (kw["seed"],))
    assert np.abs(est.trajectory(0, 0) - g["pose_output"]).max() < tol
    assert np.abs(est.trajectory(0, 1) - g["newest"]).max() < tol
    w = est.window(0)
    for k in ("Ps", "Rs", "Vs", "Bas", "Bgs"):
        assert np.abs(w[k] - g[k]).max() < tol, k
    assert np.array_equal(w["Headers"], g["Headers"])
    st = est.status(0)
    assert [st["frame_count"], st["n_tracks"], st["n_rollpitch"], st["margin_old"], st["iterations"]] == list(g["counters"])
    est.close()


@pytest.mark.parametrize("case", SEQ_CASES)
def test_native_window_manager_matches_sequence_golden(case):
    """include/isvins_estimator.h (C++) with the oracle injected as the solver, on the CPU"""
    import oracle_lib
    lib = oracle_lib.load()
    _native_vs_sequence_golden(case, lambda cfg: sh.oracle_vtbl(lib, cfg), 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SEQ_CASES)
def test_gpu_sequence_matches_golden(case):
    """the native window manager with every solve on the MI355X: north_star's 1e-6 m on the trajectory"""
    backend.build()
    _native_vs_sequence_golden(case, None, 1e-6)
