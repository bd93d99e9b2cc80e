"""GPU parity of the factor-linearisation kernels against the CPU oracle, through the C ABI.

Tolerance (float64): 1e-9 relative to each strip's largest entry.  The reprojection residual is a
460x-amplified difference of two nearly equal normalised coordinates, so evaluation-order noise
(rotation matrices + FMA contraction on the GPU vs Eigen-style quaternion products in the oracle)
is ~1e-12 absolute in r and ~1e-10 relative in the Cauchy-scaled Jacobian; measured max 9.5e-11.
The IMU sqrt_info mirrors the oracle's operation order and must be BITWISE equal."""
TOL = 1e-9
import ctypes as C

import numpy as np
import pytest

from isvins_amd import abi, backend, synth

pytestmark = pytest.mark.gpu
dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


def oracle_linearize(oracle, cfg, w, n_prior):
    F, N = w.n_factors, w.N
    ps = np.zeros((max(F, 1), 28)); im = np.zeros((N - 1, 465)); pr = np.zeros(n_prior); cost = np.zeros(1)
    assert oracle.isvo_linearize(C.byref(cfg), C.byref(w.c()), P(ps), P(im), P(pr), P(cost)) == 0
    return ps[:F], im, pr, cost[0]


@pytest.fixture(scope="module")
def be():
    backend.build()
    b = backend.Backend(11, 5, max_landmarks=400, max_obs=4400, max_batch=8)
    yield b
    b.close()


def rel_err(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("wid", [0, 7])
def test_linearize_matches_oracle(oracle, be, wid):
    w = synth.make_window(wid)
    n_prior = 6 + 9 + 6 * (w.Nvo - 1) + 2 * w.n_rollpitch
    ps_o, im_o, pr_o, cost_o = oracle_linearize(oracle, be.cfg, w, n_prior)
    ps, im, cost = be.linearize(w)
    # per-strip scale: Jacobian entries are O(460 * 1/depth); compare relative to each row's max
    sc = np.maximum(1.0, np.abs(ps_o).max(axis=1, keepdims=True))
    assert (np.abs(ps - ps_o) / sc).max() < TOL
    # IMU sqrt_info: mirrored operation order -> bitwise
    sq = be.debug_read(2, (w.N - 1) * 225).reshape(w.N - 1, 225)
    for i in range(w.N - 1):
        G = np.array([0, 0, 9.81007]); r = np.zeros(15); so = np.zeros(225)
        oracle.isvo_x_imu(C.byref(w.imu[i]), P(G), P(np.zeros(7) + [0, 0, 0, 0, 0, 0, 1.0]), P(np.zeros(9)), P(np.zeros(7) + [0, 0, 0, 0, 0, 0, 1.0]), P(np.zeros(9)), 1, P(r), None, None, None, None, P(so))
        assert np.array_equal(sq[i], so), f"imu sqrt_info {i} not bitwise equal (max diff {np.abs(sq[i]-so).max()})"
    sc = np.maximum(1.0, np.abs(im_o).max(axis=1, keepdims=True))
    assert (np.abs(im - im_o) / sc).max() < TOL
    assert abs(cost - cost_o) < 1e-11 * abs(cost_o)
    # prior residuals
    strip = be.debug_read(0, 132 + 78 * (w.Nvo - 1) + 14 * be.cfg.max_rollpitch)
    got = [strip[0:6], strip[42:51]]
    for k in range(w.Nvo - 1):
        got.append(strip[132 + 78 * k: 132 + 78 * k + 6])
    base = 132 + 78 * (w.Nvo - 1)
    for m in range(w.n_rollpitch):
        got.append(strip[base + 14 * m: base + 14 * m + 2])
    got = np.concatenate(got)
    assert np.abs(got - pr_o).max() < 1e-9 * max(1.0, np.abs(pr_o).max())


def test_linearize_at_perturbed_priors(oracle, be):
    """non-zero prior residuals: shift the state away from the prior measurements"""
    w = synth.make_window(11)
    rng = np.random.default_rng(0)
    w.Ps += 0.03 * rng.normal(size=w.Ps.shape)
    w.Vs += 0.03 * rng.normal(size=w.Vs.shape)
    from scipy.spatial.transform import Rotation as Rot
    for i in range(w.N):
        w.Rs[i] = w.Rs[i] @ Rot.from_rotvec(0.01 * rng.normal(size=3)).as_matrix()
    n_prior = 6 + 9 + 6 * (w.Nvo - 1) + 2 * w.n_rollpitch
    ps_o, im_o, pr_o, cost_o = oracle_linearize(oracle, be.cfg, w, n_prior)
    ps, im, cost = be.linearize(w)
    assert abs(cost - cost_o) < 1e-11 * abs(cost_o)
    strip = be.debug_read(0, 132 + 78 * (w.Nvo - 1) + 14 * be.cfg.max_rollpitch)
    got = np.concatenate([strip[0:6], strip[42:51]] + [strip[132 + 78 * k: 138 + 78 * k] for k in range(w.Nvo - 1)] +
                         [strip[132 + 78 * (w.Nvo - 1) + 14 * m: 134 + 78 * (w.Nvo - 1) + 14 * m] for m in range(w.n_rollpitch)])
    assert np.abs(pr_o).max() > 0.1
    assert np.abs(got - pr_o).max() < TOL * max(1.0, np.abs(pr_o).max())


def test_batch_linearize_equals_single(oracle, be):
    """ragged batch: strips of window b inside a batch are bitwise those of the single-window call"""
    ws = synth.make_windows([1, 2, 3], n_landmarks=120) + [synth.make_window(4, n_landmarks=37)]
    singles = [be.linearize(w) for w in ws]
    be.upload(ws); be.run_linearize()
    costs = be.debug_read(7, len(ws))
    for b, w in enumerate(ws):
        assert costs[b] == singles[b][2]
