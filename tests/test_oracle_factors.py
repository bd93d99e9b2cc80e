"""Pins the CPU oracle's cost functions (the reference ships no golden vectors for this path).

Finite-difference Jacobian checks that mirror the reference's own self-checks:
  ProjectionFactor::check     src/factor/projection_factor.cpp:197-299  (right perturbation Q*deltaQ)
  RelativePoseFactor::check   include/factor/relative_pose_factor.h:132-186 (central, eps 1e-8,
                              perturbation through Sophus exp on the right)
  SE3PriorFactor::check       include/factor/se3_prior_factor.h:83-133
  RollPitchFactor::check      include/factor/rollpitch_factor.h:84-131
and the same recipe applied to IMUFactor / YawFactor, which have no check() in the reference.
"""
import ctypes as C

import numpy as np
import pytest

from isvins_amd import abi, synth

dp = C.POINTER(C.c_double)


def P(a):
    return a.ctypes.data_as(dp)


def rand_pose(rng, scale=1.0):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if q[3] < 0:
        q = -q
    return np.concatenate([scale * rng.normal(size=3), q])


def plus(oracle, x, d):
    o = np.zeros(7)
    oracle.isvo_x_pose_plus(P(x), P(np.ascontiguousarray(d)), P(o))
    return o


def fd_pose(oracle, f, x, dim, eps=1e-6):
    """central difference wrt the 6-dof tangent with the reference's Plus"""
    J = np.zeros((dim, 6))
    for k in range(6):
        d = np.zeros(6); d[k] = eps
        J[:, k] = (f(plus(oracle, x, d)) - f(plus(oracle, x, -d))) / (2 * eps)
    return J


def test_pose_plus_is_right_multiplication(oracle):
    rng = np.random.default_rng(0)
    x = rand_pose(rng)
    d = 1e-3 * rng.normal(size=6)
    xp = plus(oracle, x, d)
    assert np.allclose(xp[:3], x[:3] + d[:3])
    q = x[3:]; dq = np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0])
    # Hamilton product q * dq in (x,y,z,w)
    w = q[3] * dq[3] - q[:3] @ dq[:3]
    v = q[3] * dq[:3] + dq[3] * q[:3] + np.cross(q[:3], dq[:3])
    e = np.concatenate([v, [w]]); e /= np.linalg.norm(e)
    assert np.allclose(xp[3:], e, atol=1e-15)
    assert abs(np.linalg.norm(xp[3:]) - 1) < 1e-15


@pytest.mark.parametrize("weighted", [0, 1])
def test_projection_factor_fd(oracle, weighted):
    rng = np.random.default_rng(1)
    sq = np.array([460.0, 3.0, -2.0, 455.0])
    for _ in range(5):
        pi, pj = rand_pose(rng, 0.3), rand_pose(rng, 0.3)
        pj[3:] = plus(oracle, pi, np.concatenate([np.zeros(3), 0.1 * rng.normal(size=3)]))[3:]
        ex = np.concatenate([synth.TIC, [0, 0, 0, 1.0]])
        oracle.isvo_x_R2q(P(np.ascontiguousarray(synth.RIC)), P(ex[3:]))
        ex = np.ascontiguousarray(ex)
        lam = 0.2 + 0.3 * rng.random()
        pts_i = np.array([0.3 * rng.normal(), 0.3 * rng.normal(), 1.0])
        pts_j = np.array([0.3 * rng.normal(), 0.3 * rng.normal(), 1.0])

        def ev(a, b, e, l):
            r = np.zeros(2)
            oracle.isvo_x_proj(P(a), P(b), P(e), C.c_double(l), P(pts_i), P(pts_j), P(sq), weighted, P(r), None, None, None, None)
            return r

        r = np.zeros(2); Ji = np.zeros((2, 7)); Jj = np.zeros((2, 7)); Jex = np.zeros((2, 7)); Jl = np.zeros(2)
        oracle.isvo_x_proj(P(pi), P(pj), P(ex), C.c_double(lam), P(pts_i), P(pts_j), P(sq), weighted, P(r), P(Ji), P(Jj), P(Jex), P(Jl))
        assert np.all(Ji[:, 6] == 0) and np.all(Jj[:, 6] == 0)
        sc = max(1.0, np.abs(Ji).max())
        assert np.allclose(fd_pose(oracle, lambda x: ev(x, pj, ex, lam), pi, 2), Ji[:, :6], atol=2e-6 * sc)
        assert np.allclose(fd_pose(oracle, lambda x: ev(pi, x, ex, lam), pj, 2), Jj[:, :6], atol=2e-6 * sc)
        assert np.allclose(fd_pose(oracle, lambda x: ev(pi, pj, x, lam), ex, 2), Jex[:, :6], atol=2e-6 * sc)
        e = 1e-7
        assert np.allclose((ev(pi, pj, ex, lam + e) - ev(pi, pj, ex, lam - e)) / (2 * e), Jl, rtol=1e-6, atol=1e-5 * sc)


def _imu(rng):
    acc = np.array([0.1, -0.2, 9.8]) + 0.5 * rng.normal(size=(1, 21, 3))
    gyr = 0.3 * rng.normal(size=(1, 21, 3))
    ba, bg = 0.02 * rng.normal(size=(1, 3)), 0.002 * rng.normal(size=(1, 3))
    pre = synth.preintegrate(0.005, acc, gyr, ba, bg)
    im = abi.isv_imu_t()
    im.delta_p[:] = pre["delta_p"][0]; im.delta_v[:] = pre["delta_v"][0]
    q = pre["delta_q"][0]; im.delta_q[:] = [q[1], q[2], q[3], q[0]]
    im.linearized_ba[:] = ba[0]; im.linearized_bg[:] = bg[0]; im.sum_dt = pre["sum_dt"]
    im.jacobian[:] = pre["jacobian"][0].ravel(); im.covariance[:] = pre["covariance"][0].ravel()
    return im, acc[0], gyr[0], ba[0], bg[0]


def test_preintegration_matches_numpy_producer(oracle):
    """oracle midpoint rule (integration_base.h:54-158) == the package's batched numpy producer"""
    rng = np.random.default_rng(2)
    im, acc, gyr, ba, bg = _imu(rng)
    o = abi.isv_imu_t()
    oracle.isvo_x_preint_init(C.byref(o), P(ba), P(bg))
    noise = np.array([synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W])
    for s in range(1, 21):
        oracle.isvo_x_preint_step(C.byref(o), C.c_double(0.005), P(np.ascontiguousarray(acc[s - 1])), P(np.ascontiguousarray(gyr[s - 1])),
                                  P(np.ascontiguousarray(acc[s])), P(np.ascontiguousarray(gyr[s])), P(noise))
    for f in ("delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
        a, b = abi.arr(getattr(o, f)), abi.arr(getattr(im, f))
        assert np.allclose(a, b, rtol=1e-12, atol=1e-18), f
    assert abs(o.sum_dt - im.sum_dt) < 1e-15


def test_imu_sqrt_info_is_cholesky_of_inverse_covariance(oracle):
    rng = np.random.default_rng(3)
    im, *_ = _imu(rng)
    sq = np.zeros((15, 15)); r = np.zeros(15)
    G = np.array([0, 0, 9.81007])
    pi, pj = rand_pose(rng), rand_pose(rng); si, sj = rng.normal(size=9), rng.normal(size=9)
    oracle.isvo_x_imu(C.byref(im), P(G), P(pi), P(si), P(pj), P(sj), 1, P(r), None, None, None, None, P(sq))
    cov = abi.arr(im.covariance, (15, 15))
    assert np.allclose(np.tril(sq, -1), 0)
    assert np.allclose(sq.T @ sq @ cov, np.eye(15), atol=1e-6)


@pytest.mark.parametrize("weighted", [0, 1])
def test_imu_factor_fd(oracle, weighted):
    rng = np.random.default_rng(4)
    G = np.array([0, 0, 9.81007])
    im, *_ = _imu(rng)
    pi = rand_pose(rng, 0.5)
    dq = abi.arr(im.delta_q)
    pj = pi.copy(); pj[:3] += 0.1 * rng.normal(size=3)
    pj = plus(oracle, pj, np.concatenate([np.zeros(3), 2 * dq[:3] + 0.02 * rng.normal(size=3)]))
    si = np.concatenate([0.5 * rng.normal(size=3), abi.arr(im.linearized_ba) + 0.01 * rng.normal(size=3), abi.arr(im.linearized_bg) + 0.001 * rng.normal(size=3)])
    sj = si + 0.01 * rng.normal(size=9)

    def ev(a, b, c, d):
        r = np.zeros(15)
        oracle.isvo_x_imu(C.byref(im), P(G), P(a), P(b), P(c), P(d), weighted, P(r), None, None, None, None, None)
        return r

    r = np.zeros(15); Jpi = np.zeros((15, 7)); Jsi = np.zeros((15, 9)); Jpj = np.zeros((15, 7)); Jsj = np.zeros((15, 9))
    oracle.isvo_x_imu(C.byref(im), P(G), P(pi), P(si), P(pj), P(sj), weighted, P(r), P(Jpi), P(Jsi), P(Jpj), P(Jsj), None)
    sc = max(1.0, np.abs(Jpi).max(), np.abs(Jsi).max())
    # the reference's analytic rotation blocks are first-order (VINS-Mono): tolerance 2e-3 relative
    tol = 3e-3 * sc
    assert np.allclose(fd_pose(oracle, lambda x: ev(x, si, pj, sj), pi, 15), Jpi[:, :6], atol=tol)
    assert np.allclose(fd_pose(oracle, lambda x: ev(pi, si, x, sj), pj, 15), Jpj[:, :6], atol=tol)
    e = 1e-6
    Jn = np.zeros((15, 9)); Jm = np.zeros((15, 9))
    for k in range(9):
        d = np.zeros(9); d[k] = e
        Jn[:, k] = (ev(pi, si + d, pj, sj) - ev(pi, si - d, pj, sj)) / (2 * e)
        Jm[:, k] = (ev(pi, si, pj, sj + d) - ev(pi, si, pj, sj - d)) / (2 * e)
    assert np.allclose(Jn, Jsi, atol=tol) and np.allclose(Jm, Jsj, atol=tol)


def _se3(rng):
    f = abi.isv_se3_prior_t()
    x = rand_pose(rng)
    R = np.zeros(9); t = x[:3] + 0.05 * rng.normal(size=3)
    from scipy.spatial.transform import Rotation
    Rm = Rotation.from_quat(x[3:]).as_matrix() @ Rotation.from_rotvec(0.2 * rng.normal(size=3)).as_matrix()
    f.t[:] = t; f.R[:] = Rm.ravel()
    f.sqrt_info[:] = (np.diag([100.] * 3 + [1000.] * 3) + np.triu(rng.normal(size=(6, 6)), 1)).ravel()
    return f, x


@pytest.mark.parametrize("weighted", [0, 1])
def test_se3_prior_fd(oracle, weighted):
    rng = np.random.default_rng(5)
    f, x = _se3(rng)

    def ev(a):
        r = np.zeros(6); oracle.isvo_x_se3prior(C.byref(f), weighted, P(a), P(r), None); return r
    r = np.zeros(6); J = np.zeros((6, 7))
    oracle.isvo_x_se3prior(C.byref(f), weighted, P(x), P(r), P(J))
    assert np.allclose(fd_pose(oracle, ev, x, 6, 1e-7), J[:, :6], rtol=1e-6, atol=1e-6 * max(1, np.abs(J).max()))


@pytest.mark.parametrize("weighted", [0, 1])
def test_relpose_fd(oracle, weighted):
    rng = np.random.default_rng(6)
    from scipy.spatial.transform import Rotation
    xi, xj = rand_pose(rng), rand_pose(rng)
    xj = plus(oracle, np.concatenate([xi[:3] + 0.3 * rng.normal(size=3), xi[3:]]), np.concatenate([np.zeros(3), 0.3 * rng.normal(size=3)]))
    Ri, Rj = Rotation.from_quat(xi[3:]).as_matrix(), Rotation.from_quat(xj[3:]).as_matrix()
    f = abi.isv_relpose_t()
    f.delta_t[:] = Ri.T @ (xj[:3] - xi[:3]) + 0.02 * rng.normal(size=3)
    f.delta_R[:] = (Ri.T @ Rj @ Rotation.from_rotvec(0.1 * rng.normal(size=3)).as_matrix()).ravel()
    f.sqrt_info[:] = (np.diag([100.] * 6) + np.triu(rng.normal(size=(6, 6)), 1)).ravel()

    def ev(a, b):
        r = np.zeros(6); oracle.isvo_x_relpose(C.byref(f), weighted, P(a), P(b), P(r), None, None); return r
    r = np.zeros(6); Ji = np.zeros((6, 7)); Jj = np.zeros((6, 7))
    oracle.isvo_x_relpose(C.byref(f), weighted, P(xi), P(xj), P(r), P(Ji), P(Jj))
    sc = max(1, np.abs(Ji).max())
    assert np.allclose(fd_pose(oracle, lambda x: ev(x, xj), xi, 6, 1e-7), Ji[:, :6], atol=2e-6 * sc)
    assert np.allclose(fd_pose(oracle, lambda x: ev(xi, x), xj, 6, 1e-7), Jj[:, :6], atol=2e-6 * sc)


@pytest.mark.parametrize("weighted", [0, 1])
def test_rollpitch_and_yaw_fd(oracle, weighted):
    rng = np.random.default_rng(7)
    from scipy.spatial.transform import Rotation
    x = rand_pose(rng)
    f = abi.isv_rollpitch_t()
    f.R[:] = (Rotation.from_quat(x[3:]).as_matrix() @ Rotation.from_rotvec(0.1 * rng.normal(size=3)).as_matrix()).ravel()
    f.sqrt_info[:] = [100., 2., 0., 90.]

    def ev(a):
        r = np.zeros(2); oracle.isvo_x_rollpitch(C.byref(f), weighted, P(a), P(r), None); return r
    r = np.zeros(2); J = np.zeros((2, 7))
    oracle.isvo_x_rollpitch(C.byref(f), weighted, P(x), P(r), P(J))
    assert np.allclose(fd_pose(oracle, ev, x, 2, 1e-7), J[:, :6], atol=2e-6 * max(1, np.abs(J).max()))
    # yaw factor (Jacobian only is used by MargBackward): residual = (R_i * R_z^-1 e_x).y
    J6 = np.zeros(6); r1 = np.zeros(1)
    oracle.isvo_x_yaw(P(x), P(r1), P(J6))
    Rz = Rotation.from_quat(x[3:]).as_matrix()
    ym = Rz.T @ np.array([1., 0, 0])

    def yaw_res(a):
        return np.array([(Rotation.from_quat(a[3:]).as_matrix() @ ym)[1]])
    assert np.allclose(fd_pose(oracle, yaw_res, x, 1, 1e-7)[0], J6, atol=1e-6)
    assert abs(r1[0]) < 1e-14     # measurement is the pose itself


def test_so3_log_exp_and_right_jacobian(oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(8)
    for scale in (1e-12, 1e-6, 0.3, 2.5):
        w = scale * rng.normal(size=3)
        R = np.zeros(9); oracle.isvo_x_so3_exp(P(w), P(R))
        assert np.allclose(R.reshape(3, 3), Rotation.from_rotvec(w).as_matrix(), atol=1e-14)
        w2 = np.zeros(3); oracle.isvo_x_so3_log(P(R), P(w2))
        assert np.allclose(w2, Rotation.from_matrix(R.reshape(3, 3)).as_rotvec(), atol=1e-12)
        # log(exp(w) exp(e)) ~ w + Jr^-1(w) e   (sophus_utils.hpp:194-236)
        J = np.zeros(9); oracle.isvo_x_rjacinv(P(w), P(J))
        e = 1e-7 * rng.normal(size=3)
        lhs = (Rotation.from_rotvec(w) * Rotation.from_rotvec(e)).as_rotvec()
        assert np.allclose(lhs, w + J.reshape(3, 3) @ e, atol=1e-12)


def test_update_pseudo_measurement_formulas(oracle):
    """update() (estimator.cpp:1133-1144) against an independent scipy restatement of the
    reference formulas: se3_prior_factor.h:73-81, rollpitch_factor.h:78-83,
    relative_pose_factor.h:103-117.  NOTE (reference quirk, reproduced): the rotation part is
    R <- R * exp(log(R_new^-1 R_old)), which does NOT keep the rotation residual unchanged the way
    t += P_new - P_old does for the translation."""
    rng = np.random.default_rng(9)
    from scipy.spatial.transform import Rotation as Rot
    x0 = rand_pose(rng); R0 = Rot.from_quat(x0[3:]).as_matrix()
    Rm = R0 @ Rot.from_rotvec(0.05 * rng.normal(size=3)).as_matrix()
    f = abi.isv_se3_prior_t(); f.t[:] = x0[:3] + 0.01; f.R[:] = Rm.ravel(); f.sqrt_info[:] = np.eye(6).ravel()
    x1 = plus(oracle, x0, 1e-2 * rng.normal(size=6)); R1 = Rot.from_quat(x1[3:]).as_matrix()
    oracle.isvo_x_update_se3(C.byref(f), P(np.ascontiguousarray(x0[:3])), P(np.ascontiguousarray(R0.ravel())), P(x1))
    assert np.allclose(abi.arr(f.t), x0[:3] + 0.01 + (x1[:3] - x0[:3]), atol=1e-15)
    assert np.allclose(abi.arr(f.R, (3, 3)), Rm @ R1.T @ R0, atol=1e-14)
    # translation residual is preserved
    r = np.zeros(6); oracle.isvo_x_se3prior(C.byref(f), 1, P(x1), P(r), None)
    assert np.allclose(r[:3], -0.01, atol=1e-14)
    g = abi.isv_rollpitch_t(); g.R[:] = Rm.ravel(); g.sqrt_info[:] = [1, 0, 0, 1.]
    oracle.isvo_x_update_rollpitch(C.byref(g), P(np.ascontiguousarray(R0.ravel())), P(x1))
    assert np.allclose(abi.arr(g.R, (3, 3)), Rm @ R1.T @ R0, atol=1e-14)
    # relative pose
    xj0 = rand_pose(rng); Rj0 = Rot.from_quat(xj0[3:]).as_matrix()
    h = abi.isv_relpose_t()
    dt0 = R0.T @ (xj0[:3] - x0[:3]); dR0 = R0.T @ Rj0
    h.delta_t[:] = dt0; h.delta_R[:] = dR0.ravel(); h.sqrt_info[:] = np.eye(6).ravel()
    xj1 = plus(oracle, xj0, 1e-3 * rng.normal(size=6)); xi1 = plus(oracle, x0, 1e-3 * rng.normal(size=6))
    Ri1, Rj1 = Rot.from_quat(xi1[3:]).as_matrix(), Rot.from_quat(xj1[3:]).as_matrix()
    oracle.isvo_x_update_relpose(C.byref(h), P(np.ascontiguousarray(x0[:3])), P(np.ascontiguousarray(R0.ravel())),
                                 P(np.ascontiguousarray(xj0[:3])), P(np.ascontiguousarray(Rj0.ravel())), P(xi1), P(xj1))
    d_tj, d_ti = xj1[:3] - xj0[:3], xi1[:3] - x0[:3]
    lgj = Rot.from_matrix(Rj1.T @ Rj0).as_rotvec(); lgi = Rot.from_matrix(Ri1.T @ R0).as_rotvec()
    sk = np.array([[0, -dt0[2], dt0[1]], [dt0[2], 0, -dt0[0]], [-dt0[1], dt0[0], 0]])
    assert np.allclose(abi.arr(h.delta_t), dt0 + R0.T @ d_tj - R0.T @ d_ti + sk @ lgi, atol=1e-14)
    Ji = -(Rj1.T @ Ri1)
    want = dR0 @ Rot.from_rotvec(Ji @ lgi).as_matrix() @ Rot.from_rotvec(lgj).as_matrix()
    assert np.allclose(abi.arr(h.delta_R, (3, 3)), want, atol=1e-14)
