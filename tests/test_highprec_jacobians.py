"""An INDEPENDENT derivation of every factor's Jacobian, at 40 digits.

The reference ships no golden vectors for this path ("parity unpinned"); tests/test_oracle_factors.py pins the
oracle's analytic Jacobians with double-precision finite differences (the reference's own check() recipe, 1e-6).  This file
tightens that pin by six orders of magnitude: each factor's RESIDUAL is written down once more, in mpmath, straight from the
reference's Evaluate() (file:line below), perturbed through the reference's Plus (PoseLocalParameterization: p + dp,
q * [1, dtheta / 2] normalised, src/factor/pose_local_parameterization.cpp:3-20), and differentiated numerically with a 1e-18
central step at 40 digits -- a derivative good to ~1e-30 that shares no code and no algebra with the analytic blocks the
oracle (and the HIP kernels, which are compared with the oracle) implement.

Two blocks of IMUFactor are NOT the exact derivative of its residual, and the test shows by how much and why (both are the
reference's formulas, reproduced on purpose): d r_q / d theta_i carries Qright(corrected delta_q) where
Utility::deltaQ does not normalise (include/utility/utility.h:11-24: off by |dq_dbg (bg - bg_lin)|^2 / 4, ~6e-9 here), and
d r_q / d bg_i uses the UNCORRECTED delta_q (include/factor/imu_factor.h:105: off in proportion to the bias offset times the
rotation residual).  With the gyro bias at its linearisation point both vanish and every block is exact to rounding.
"""
import ctypes as C

import mpmath as mp
import numpy as np
import pytest

from isvins_amd import abi, synth
from test_oracle_factors import P, _imu, _se3, plus, rand_pose

mp.mp.dps = 40
H = mp.mpf(10) ** -18


# ---- quaternion / rotation algebra in mpmath (quaternions as (w, x, y, z)) -------------------------------------------------
def qmul(a, b):
    aw, ax, ay, az = a; bw, bx, by, bz = b
    return (aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw)


def qconj(a):
    return (a[0], -a[1], -a[2], -a[3])


def qnorm(a):
    n = mp.sqrt(sum(c * c for c in a))
    return tuple(c / n for c in a)


def qrot(q, v):
    r = qmul(qmul(q, (mp.mpf(0),) + tuple(v)), qconj(q))
    return [r[1], r[2], r[3]]


def q_from_pose(x):          # pose block [px py pz qx qy qz qw] (src/estimator.cpp:481-485)
    return (x[6], x[3], x[4], x[5])


def q_from_R(R):             # R row-major 9, via the largest-trace branch in high precision
    m = [[R[3 * i + j] for j in range(3)] for i in range(3)]
    tr = m[0][0] + m[1][1] + m[2][2]
    cands = []
    w = mp.sqrt(max(mp.mpf(0), 1 + tr)) / 2
    x = mp.sqrt(max(mp.mpf(0), 1 + m[0][0] - m[1][1] - m[2][2])) / 2
    y = mp.sqrt(max(mp.mpf(0), 1 - m[0][0] + m[1][1] - m[2][2])) / 2
    z = mp.sqrt(max(mp.mpf(0), 1 - m[0][0] - m[1][1] + m[2][2])) / 2
    k = max(range(4), key=lambda i: [w, x, y, z][i])
    if k == 0:
        cands = (w, (m[2][1] - m[1][2]) / (4 * w), (m[0][2] - m[2][0]) / (4 * w), (m[1][0] - m[0][1]) / (4 * w))
    elif k == 1:
        cands = ((m[2][1] - m[1][2]) / (4 * x), x, (m[0][1] + m[1][0]) / (4 * x), (m[0][2] + m[2][0]) / (4 * x))
    elif k == 2:
        cands = ((m[0][2] - m[2][0]) / (4 * y), (m[0][1] + m[1][0]) / (4 * y), y, (m[1][2] + m[2][1]) / (4 * y))
    else:
        cands = ((m[1][0] - m[0][1]) / (4 * z), (m[0][2] + m[2][0]) / (4 * z), (m[1][2] + m[2][1]) / (4 * z), z)
    return qnorm(cands)


def so3_log(q):              # rotation vector of a unit quaternion (Sophus SO3::log)
    q = qnorm(q)
    if q[0] < 0:
        q = tuple(-c for c in q)
    n = mp.sqrt(q[1] ** 2 + q[2] ** 2 + q[3] ** 2)
    if n == 0:
        return [mp.mpf(0)] * 3
    ang = 2 * mp.atan2(n, q[0])
    return [ang * q[1] / n, ang * q[2] / n, ang * q[3] / n]


def plus_mp(x, d):           # PoseLocalParameterization::Plus
    q = qnorm(qmul(q_from_pose(x), (mp.mpf(1), d[3] / 2, d[4] / 2, d[5] / 2)))
    return [x[0] + d[0], x[1] + d[1], x[2] + d[2], q[1], q[2], q[3], q[0]]


def mpv(a):
    return [mp.mpf(float(v)) for v in np.asarray(a, dtype=np.float64).ravel()]


def jac_pose(f, x, dim):
    """d f / d (tangent of the pose block x), central difference at 40 digits"""
    J = np.zeros((dim, 6))
    for k in range(6):
        d = [mp.mpf(0)] * 6
        d[k] = H
        fp = f(plus_mp(x, d))
        d[k] = -H
        fm = f(plus_mp(x, d))
        J[:, k] = [float((a - b) / (2 * H)) for a, b in zip(fp, fm)]
    return J


def jac_vec(f, x, dim):
    J = np.zeros((dim, len(x)))
    for k in range(len(x)):
        xp = list(x); xp[k] = x[k] + H
        xm = list(x); xm[k] = x[k] - H
        J[:, k] = [float((a - b) / (2 * H)) for a, b in zip(f(xp), f(xm))]
    return J


def matvec(S, v, n):
    return [sum(S[r * n + c] * v[c] for c in range(n)) for r in range(len(S) // n)]


def close(a, b, tol=1e-10):
    sc = max(1.0, np.abs(b).max())
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= tol * sc


# ---- ProjectionFactor::Evaluate  src/factor/projection_factor.cpp:24-60 ----------------------------------------------------
def proj_res(xi, xj, ex, lam, pts_i, pts_j, sq):
    Qi, Qj, qic = q_from_pose(xi), q_from_pose(xj), q_from_pose(ex)
    pc = [c / lam for c in pts_i]
    pb = [a + b for a, b in zip(qrot(qic, pc), ex[:3])]
    pw = [a + b for a, b in zip(qrot(Qi, pb), xi[:3])]
    pbj = qrot(qconj(Qj), [a - b for a, b in zip(pw, xj[:3])])
    pcj = qrot(qconj(qic), [a - b for a, b in zip(pbj, ex[:3])])
    u = [pcj[0] / pcj[2] - pts_j[0], pcj[1] / pcj[2] - pts_j[1]]
    return [sq[0] * u[0] + sq[1] * u[1], sq[2] * u[0] + sq[3] * u[1]]


def test_projection_factor_against_40_digit_derivative(oracle):
    rng = np.random.default_rng(11)
    sq = np.array([460.0, 3.0, -2.0, 455.0])
    for _ in range(4):
        pi, pj = rand_pose(rng, 0.3), rand_pose(rng, 0.3)
        pj[3:] = plus(oracle, pi, np.concatenate([np.zeros(3), 0.1 * rng.normal(size=3)]))[3:]
        ex = np.concatenate([synth.TIC, [0, 0, 0, 1.0]])
        oracle.isvo_x_R2q(P(np.ascontiguousarray(synth.RIC)), P(ex[3:]))
        ex = np.ascontiguousarray(ex)
        lam = 0.2 + 0.3 * rng.random()
        pts_i = np.array([0.3 * rng.normal(), 0.3 * rng.normal(), 1.0])
        pts_j = np.array([0.3 * rng.normal(), 0.3 * rng.normal(), 1.0])
        r = np.zeros(2); Ji = np.zeros((2, 7)); Jj = np.zeros((2, 7)); Jex = np.zeros((2, 7)); Jl = np.zeros(2)
        oracle.isvo_x_proj(P(pi), P(pj), P(ex), C.c_double(lam), P(pts_i), P(pts_j), P(sq), 1, P(r), P(Ji), P(Jj), P(Jex), P(Jl))
        a, b, e, l, p1, p2, s = mpv(pi), mpv(pj), mpv(ex), mp.mpf(float(lam)), mpv(pts_i), mpv(pts_j), mpv(sq)
        assert close([float(v) for v in proj_res(a, b, e, l, p1, p2, s)], r, 1e-12)
        assert close(jac_pose(lambda x: proj_res(x, b, e, l, p1, p2, s), a, 2), Ji[:, :6])
        assert close(jac_pose(lambda x: proj_res(a, x, e, l, p1, p2, s), b, 2), Jj[:, :6])
        assert close(jac_pose(lambda x: proj_res(a, b, x, l, p1, p2, s), e, 2), Jex[:, :6])
        assert close(jac_vec(lambda v: proj_res(a, b, e, v[0], p1, p2, s), [l], 2)[:, 0], Jl)


# ---- SE3PriorFactor  se3_prior_factor.h:21-53;  RelativePoseFactor  relative_pose_factor.h:27-70;
#      RollPitchFactor  rollpitch_factor.h:26-57 -------------------------------------------------------------------------------
def se3_res(x, t, R, S):
    lg = so3_log(qmul(qconj(q_from_R(R)), qnorm(q_from_pose(x))))
    return matvec(S, [x[0] - t[0], x[1] - t[1], x[2] - t[2]] + lg, 6)


def relpose_res(xi, xj, dt, dR, S):
    Qi, Qj = q_from_pose(xi), q_from_pose(xj)
    qd = qrot(qconj(Qi), [a - b for a, b in zip(xj[:3], xi[:3])])
    lg = so3_log(qmul(qmul(q_from_R(dR), qconj(Qj)), Qi))        # Log(delta_R Rj^T Ri)
    return matvec(S, [dt[0] - qd[0], dt[1] - qd[1], dt[2] - qd[2]] + lg, 6)


def rollpitch_res(x, R, S):
    v = qrot(qmul(q_from_R(R), qconj(qnorm(q_from_pose(x)))), [mp.mpf(0), mp.mpf(0), mp.mpf(-1)])
    return matvec(S, v[:2], 2)


def test_prior_factors_against_40_digit_derivative(oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(12)
    f, x = _se3(rng)
    r = np.zeros(6); J = np.zeros((6, 7))
    oracle.isvo_x_se3prior(C.byref(f), 1, P(x), P(r), P(J))
    xm, t, R, S = mpv(x), mpv(abi.arr(f.t)), mpv(abi.arr(f.R)), mpv(abi.arr(f.sqrt_info))
    assert close([float(v) for v in se3_res(xm, t, R, S)], r, 1e-12)
    assert close(jac_pose(lambda q: se3_res(q, t, R, S), xm, 6), J[:, :6])

    xi = rand_pose(rng)
    xj = plus(oracle, np.concatenate([xi[:3] + 0.3 * rng.normal(size=3), xi[3:]]), np.concatenate([np.zeros(3), 0.3 * rng.normal(size=3)]))
    Ri, Rj = Rotation.from_quat(xi[3:]).as_matrix(), Rotation.from_quat(xj[3:]).as_matrix()
    g = abi.isv_relpose_t()
    g.delta_t[:] = Ri.T @ (xj[:3] - xi[:3]) + 0.02 * rng.normal(size=3)
    g.delta_R[:] = (Ri.T @ Rj @ Rotation.from_rotvec(0.1 * rng.normal(size=3)).as_matrix()).ravel()
    g.sqrt_info[:] = (np.diag([100.] * 6) + np.triu(rng.normal(size=(6, 6)), 1)).ravel()
    r = np.zeros(6); Ji = np.zeros((6, 7)); Jj = np.zeros((6, 7))
    oracle.isvo_x_relpose(C.byref(g), 1, P(xi), P(xj), P(r), P(Ji), P(Jj))
    a, b, dt, dR, S = mpv(xi), mpv(xj), mpv(abi.arr(g.delta_t)), mpv(abi.arr(g.delta_R)), mpv(abi.arr(g.sqrt_info))
    assert close([float(v) for v in relpose_res(a, b, dt, dR, S)], r, 1e-11)
    assert close(jac_pose(lambda q: relpose_res(q, b, dt, dR, S), a, 6), Ji[:, :6], 1e-9)
    assert close(jac_pose(lambda q: relpose_res(a, q, dt, dR, S), b, 6), Jj[:, :6], 1e-9)

    x = rand_pose(rng)
    p = abi.isv_rollpitch_t()
    p.R[:] = (Rotation.from_quat(x[3:]).as_matrix() @ Rotation.from_rotvec(0.1 * rng.normal(size=3)).as_matrix()).ravel()
    p.sqrt_info[:] = [100., 2., 0., 90.]
    r = np.zeros(2); J = np.zeros((2, 7))
    oracle.isvo_x_rollpitch(C.byref(p), 1, P(x), P(r), P(J))
    xm, R, S = mpv(x), mpv(abi.arr(p.R)), mpv(abi.arr(p.sqrt_info))
    assert close([float(v) for v in rollpitch_res(xm, R, S)], r, 1e-12)
    assert close(jac_pose(lambda q: rollpitch_res(q, R, S), xm, 2), J[:, :6], 1e-9)


# ---- IMUFactor / IntegrationBase::evaluate  integration_base.h:160-186, imu_factor.h:23-155 --------------------------------
def imu_res(im, G, xi, si, xj, sj, S=None):
    dp, dv = mpv(abi.arr(im.delta_p)), mpv(abi.arr(im.delta_v))
    dq = mpv(abi.arr(im.delta_q)); dq = (dq[3], dq[0], dq[1], dq[2])
    lba, lbg, dt = mpv(abi.arr(im.linearized_ba)), mpv(abi.arr(im.linearized_bg)), mp.mpf(float(im.sum_dt))
    Jm = mpv(abi.arr(im.jacobian))                                   # 15 x 15 row-major, order P R V BA BG
    blk = lambda r0, c0: [[Jm[(r0 + a) * 15 + c0 + b] for b in range(3)] for a in range(3)]
    mv = lambda M, v: [sum(M[a][b] * v[b] for b in range(3)) for a in range(3)]
    dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg = blk(0, 9), blk(0, 12), blk(3, 12), blk(6, 9), blk(6, 12)
    Qi, Qj = q_from_pose(xi), q_from_pose(xj)
    dba = [si[3 + k] - lba[k] for k in range(3)]
    dbg = [si[6 + k] - lbg[k] for k in range(3)]
    th = mv(dq_dbg, dbg)
    cdq = qmul(dq, (mp.mpf(1), th[0] / 2, th[1] / 2, th[2] / 2))      # Utility::deltaQ: NOT normalised (utility.h:11-24)
    cdv = [dv[k] + mv(dv_dba, dba)[k] + mv(dv_dbg, dbg)[k] for k in range(3)]
    cdp = [dp[k] + mv(dp_dba, dba)[k] + mv(dp_dbg, dbg)[k] for k in range(3)]
    u = [G[k] * dt * dt / 2 + xj[k] - xi[k] - si[k] * dt for k in range(3)]
    rp = [a - b for a, b in zip(qrot(qconj(Qi), u), cdp)]
    # Eigen's Quaternion::inverse() = conjugate / squared norm (cdq is not unit)
    n2 = sum(c * c for c in cdq)
    inv = tuple(c / n2 for c in qconj(cdq))
    qe = qmul(inv, qmul(qconj(Qi), Qj))
    rq = [2 * qe[1], 2 * qe[2], 2 * qe[3]]
    u = [G[k] * dt + sj[k] - si[k] for k in range(3)]
    rv = [a - b for a, b in zip(qrot(qconj(Qi), u), cdv)]
    r = rp + rq + rv + [sj[3 + k] - si[3 + k] for k in range(3)] + [sj[6 + k] - si[6 + k] for k in range(3)]
    return matvec(S, r, 15) if S is not None else r


@pytest.mark.parametrize("bg_at_linearisation_point", [True, False])
def test_imu_factor_against_40_digit_derivative(oracle, bg_at_linearisation_point):
    rng = np.random.default_rng(13)
    G = np.array([0, 0, 9.81007])
    im, *_ = _imu(rng)
    pi = rand_pose(rng, 0.5)
    si = np.concatenate([0.5 * rng.normal(size=3), abi.arr(im.linearized_ba) + 0.01 * rng.normal(size=3),
                         abi.arr(im.linearized_bg) + (0.0 if bg_at_linearisation_point else 0.001) * rng.normal(size=3)])
    sj = si + 0.01 * rng.normal(size=9)
    pj = pi.copy(); pj[:3] += 0.1 * rng.normal(size=3)
    dq = abi.arr(im.delta_q)                                              # x y z w
    pj = plus(oracle, pj, np.concatenate([np.zeros(3), 2 * dq[:3] + 0.1 * rng.normal(size=3)]))     # a rotation residual of ~0.2 rad
    r = np.zeros(15); Jpi = np.zeros((15, 7)); Jsi = np.zeros((15, 9)); Jpj = np.zeros((15, 7)); Jsj = np.zeros((15, 9))
    oracle.isvo_x_imu(C.byref(im), P(G), P(pi), P(si), P(pj), P(sj), 0, P(r), P(Jpi), P(Jsi), P(Jpj), P(Jsj), None)
    a, b, c, d, g = mpv(pi), mpv(si), mpv(pj), mpv(sj), mpv(G)
    assert close([float(v) for v in imu_res(im, g, a, b, c, d)], r, 1e-11)
    assert np.abs(r[3:6]).max() > 0.05
    Epi = jac_pose(lambda x: imu_res(im, g, x, b, c, d), a, 15)
    Epj = jac_pose(lambda x: imu_res(im, g, a, b, x, d), c, 15)
    Esi = jac_vec(lambda v: imu_res(im, g, a, v, c, d), b, 15)
    Esj = jac_vec(lambda v: imu_res(im, g, a, b, c, v), d, 15)
    rot = slice(3, 6)
    exact = np.ones((15, 6), bool); exact[rot, 3:6] = False              # every block but d r_q / d theta_i ...
    assert close(np.where(exact, Epi, 0), np.where(exact, Jpi[:, :6], 0), 1e-9)
    assert close(Epj, Jpj[:, :6], 1e-9)
    exact_s = np.ones((15, 9), bool); exact_s[rot, 6:9] = False          # ... and d r_q / d bg_i
    assert close(np.where(exact_s, Esi, 0), np.where(exact_s, Jsi, 0), 1e-9)
    assert close(Esj, Jsj, 1e-9)
    e_th = np.abs(Epi[rot, 3:6] - Jpi[rot, 3:6]).max()
    e_bg = np.abs(Esi[rot, 6:9] - Jsi[rot, 6:9]).max()
    if bg_at_linearisation_point:
        assert e_th < 1e-10 and e_bg < 1e-10      # nothing to correct: both blocks are the exact derivative
    else:
        th = np.asarray(abi.arr(im.jacobian)).reshape(15, 15)[3:6, 12:15] @ (si[6:9] - abi.arr(im.linearized_bg))
        assert 0.05 * (th @ th) < e_th < 2.0 * (th @ th)                  # the unnormalised deltaQ: |theta|^2 / 4 in the norm
        assert 1e-9 < e_bg < 2.0 * np.linalg.norm(th)                     # the uncorrected delta_q in Qleft(Qj^-1 Qi delta_q)
