"""MargForward / MargBackward (src/estimator.cpp:1149-1352 / :1354-1539) recomputed at 40 digits in mpmath -- a THIRD opinion beside
the CPU oracle (double, LU / cyclic Jacobi) and the HIP kernels (double, closed-form landmark elimination / Gauss-Jordan / parallel
Jacobi).  Test infrastructure: tests/test_marg_third_opinion.py feeds it the inputs both routines read, as
recorded in tests/golden/marg_third_opinion.npz by scripts/marg_third_opinion_dump.py.

Every residual is the one tests/test_highprec_jacobians.py writes down from the reference's Evaluate(); Jacobians of the factors whose
analytic blocks are exact derivatives (ProjectionFactor, SE3Prior, RelativePose, RollPitch, Yaw) are central differences through the
reference's Plus with a 1e-18 step at 40 digits (good to ~1e-30); IMUFactor's Jacobian is the reference's ANALYTIC formula
(include/factor/imu_factor.h:66-155, with its two inexact blocks: the unnormalised deltaQ and the uncorrected delta_q), restated
here in mpmath, because MargBackward linearises with that formula and not with the derivative.  The linear algebra (Schur
complements, pseudo-inverse, eigen-truncation at alpha, the projected covariances and their inverses) runs in mpmath."""
import ctypes as C

import mpmath as mp
import numpy as np

from isvins_amd import abi
from test_highprec_jacobians import (H, mpv, plus_mp, proj_res, q_from_pose, q_from_R, qconj, qmul, qnorm, qrot, relpose_res,
                                     rollpitch_res, se3_res)

mp.mp.dps = 40
ZERO, ONE = mp.mpf(0), mp.mpf(1)


def struct_of(cls, raw):
    s = cls()
    C.memmove(C.byref(s), raw.tobytes(), C.sizeof(cls))
    return s


def jac_pose_mp(f, x, dim):
    """d f / d (tangent of pose block x) as an mp.matrix (dim x 6)"""
    J = mp.zeros(dim, 6)
    for k in range(6):
        d = [ZERO] * 6
        d[k] = H
        fp = f(plus_mp(x, d))
        d[k] = -H
        fm = f(plus_mp(x, d))
        for r in range(dim):
            J[r, k] = (fp[r] - fm[r]) / (2 * H)
    return J


def jac_scalar_mp(f, x, dim):
    fp, fm = f(x + H), f(x - H)
    J = mp.zeros(dim, 1)
    for r in range(dim):
        J[r, 0] = (fp[r] - fm[r]) / (2 * H)
    return J


def q_to_R(q):                 # Eigen toRotationMatrix (oracle/isvo_math.h q_to_R), q = (w, x, y, z); row-major 3x3 as mp.matrix
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return mp.matrix([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def q_inv(q):
    n2 = sum(c * c for c in q)
    return (q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2)


def skew(v):
    return mp.matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def qleft33(q):
    return skew(q[1:]) + q[0] * mp.eye(3)


def qright33(q):
    return q[0] * mp.eye(3) - skew(q[1:])


def qleft44(q):
    M = mp.zeros(4, 4)
    B = qleft33(q)
    M[0, 0] = q[0]
    for k in range(3):
        M[0, k + 1] = -q[k + 1]; M[k + 1, 0] = q[k + 1]
        for j in range(3):
            M[k + 1, j + 1] = B[k, j]
    return M


def qright44(q):
    M = mp.zeros(4, 4)
    B = qright33(q)
    M[0, 0] = q[0]
    for k in range(3):
        M[0, k + 1] = -q[k + 1]; M[k + 1, 0] = q[k + 1]
        for j in range(3):
            M[k + 1, j + 1] = B[k, j]
    return M


def sym_info(S, n):
    """sqrt_info^T sqrt_info of a row-major n x n double array, as mp.matrix"""
    A = mp.matrix(n, n)
    v = mpv(S)
    for a in range(n):
        for b in range(n):
            A[a, b] = v[a * n + b]
    return A.T * A


def set_block(M, r0, c0, B):
    for a in range(B.rows):
        for b in range(B.cols):
            M[r0 + a, c0 + b] = B[a, b]


def add_block(M, r0, c0, B):
    for a in range(B.rows):
        for b in range(B.cols):
            M[r0 + a, c0 + b] += B[a, b]


def sub(M, r0, r1, c0, c1):
    B = mp.matrix(r1 - r0, c1 - c0)
    for a in range(r0, r1):
        for b in range(c0, c1):
            B[a - r0, b - c0] = M[a, b]
    return B


def eig_truncate(L, alpha):
    """eigenpairs of the symmetric L with eigenvalue > alpha (src/estimator.cpp:1311-1331 / :1479-1497)"""
    E, Q = mp.eigsy(L)
    keep = [i for i in range(L.rows) if E[i] > alpha]
    U = mp.matrix(L.rows, len(keep))
    for k, i in enumerate(keep):
        for r in range(L.rows):
            U[r, k] = Q[r, i]
    return U, [E[i] for i in keep], [E[i] for i in range(L.rows)]


def project_info(Jk, U, Dv):
    """((Jk U) D^-1 (Jk U)^T)^-1"""
    JU = Jk * U
    Dinv = mp.diag([1 / d for d in Dv])
    return mp.inverse(JU * Dinv * JU.T)


def full_pivot_rank(L, thr):
    """rank as Eigen's FullPivHouseholderQR with threshold thr sees it, restated (like the oracle) by full-pivot LU pivots"""
    A = L.copy(); n = A.rows; piv = []
    for k in range(n):
        best, pi, pj = ZERO, k, k
        for i in range(k, n):
            for j in range(k, n):
                if abs(A[i, j]) > best:
                    best, pi, pj = abs(A[i, j]), i, j
        piv.append(best)
        if best == 0:
            piv += [ZERO] * (n - k - 1)
            break
        for j in range(n):
            A[k, j], A[pi, j] = A[pi, j], A[k, j]
        for i in range(n):
            A[i, k], A[i, pj] = A[i, pj], A[i, k]
        for i in range(k + 1, n):
            f = A[i, k] / A[k, k]
            for j in range(k, n):
                A[i, j] -= f * A[k, j]
    mx = max(piv)
    return sum(1 for p in piv if p > thr * mx)


def marg_forward(z, p, alpha):
    """-> dict(forward_pose_prior = information of the new pose prior on T1, combined_relpose = information of the pose-graph edge,
    covRel) + conditioning figures.  Parameter order of Lam: T1 @0, T0 @6, landmarks @12.. (the oracle's, oracle/isv_oracle.c:632)."""
    pose0, pose1 = mpv(z[p + "pose"][0]), mpv(z[p + "pose"][1])
    ex, sq = mpv(z[p + "ex"]), mpv(z["proj_sqrt_info"])
    lam, pts = z[p + "lam"], z[p + "pts"]
    n0 = len(lam)
    S12 = mp.zeros(12, 12)                       # the two pose blocks with the landmarks eliminated (each landmark is a 1 x 1 block)
    L12 = mp.zeros(12, 12)                       # ... and without (Lam_rp, :1240: the reference takes the top-left corner of Lam as it is)
    for m in range(n0):
        l, pi, pj = mp.mpf(float(lam[m])), mpv(pts[m, 0]), mpv(pts[m, 1])
        Ji = jac_pose_mp(lambda x: proj_res(x, pose1, ex, l, pi, pj, sq), pose0, 2)
        Jj = jac_pose_mp(lambda x: proj_res(pose0, x, ex, l, pi, pj, sq), pose1, 2)
        Jl = jac_scalar_mp(lambda v: proj_res(pose0, pose1, ex, v, pi, pj, sq), l, 2)
        J = mp.zeros(2, 12)
        set_block(J, 0, 0, Jj); set_block(J, 0, 6, Ji)
        Hpp, c, dl = J.T * J, J.T * Jl, (Jl.T * Jl)[0, 0]
        L12 += Hpp
        S12 += Hpp - c * c.T / dl
    prior = struct_of(abi.isv_se3_prior_t, z[p + "pose_prior"])
    t, R, S = mpv(abi.arr(prior.t)), mpv(abi.arr(prior.R)), mpv(abi.arr(prior.sqrt_info))
    Jp = jac_pose_mp(lambda x: se3_res(x, t, R, S), pose0, 6)
    for M in (S12, L12):
        add_block(M, 6, 6, Jp.T * Jp)
    rel = struct_of(abi.isv_relpose_t, z[p + "relpose0"])
    dt, dR, S = mpv(abi.arr(rel.delta_t)), mpv(abi.arr(rel.delta_R)), mpv(abi.arr(rel.sqrt_info))
    Ji = jac_pose_mp(lambda x: relpose_res(x, pose1, dt, dR, S), pose0, 6)
    Jj = jac_pose_mp(lambda x: relpose_res(pose0, x, dt, dR, S), pose1, 6)
    J = mp.zeros(6, 12)
    set_block(J, 0, 0, Jj); set_block(J, 0, 6, Ji)
    S12 = S12 + J.T * J
    L12 = L12 + J.T * J
    # (i) the pose-graph edge (:1240-1283): measurement from the current poses, J = [J_i | J_j] against Lam_rp ordered [T1 | T0] (as the reference has it)
    Qi, Qj = q_from_pose(pose0), q_from_pose(pose1)
    d = [pose1[k] - pose0[k] for k in range(3)]
    pg_dt = qrot(q_inv(Qi), d)
    Rm = q_to_R(qmul(q_inv(Qi), Qj))
    pg_dR = [Rm[a, b] for a in range(3) for b in range(3)]
    I6 = [ONE if a == b else ZERO for a in range(6) for b in range(6)]
    Ji = jac_pose_mp(lambda x: relpose_res(x, pose1, pg_dt, pg_dR, I6), pose0, 6)
    Jj = jac_pose_mp(lambda x: relpose_res(pose0, x, pg_dt, pg_dR, I6), pose1, 6)
    Jg = mp.zeros(6, 12)
    set_block(Jg, 0, 0, Ji); set_block(Jg, 0, 6, Jj)
    Jpinv = Jg.T * mp.inverse(Jg * Jg.T)                      # full row rank: Utility::pseudoInverse keeps every singular value
    Om = Jpinv.T * L12 * Jpinv
    out = {"combined_relpose": Om, "covRel": mp.inverse(Om)}
    # (ii) the new pose prior on T1 (:1286-1351): Schur complement of everything but T1
    A, Bm, Dm = sub(S12, 0, 6, 0, 6), sub(S12, 0, 6, 6, 12), sub(S12, 6, 12, 6, 12)
    Lprior = A - Bm * mp.inverse(Dm) * Bm.T
    fp_t, fp_R = pose1[:3], [v for v in q_to_R(q_from_pose(pose1))]
    Jr = jac_pose_mp(lambda x: se3_res(x, fp_t, fp_R, I6), pose1, 6)
    rank = full_pivot_rank(Lprior, mp.mpf("1e-16"))
    if rank == 6:
        covi = Jr * mp.inverse(Lprior) * Jr.T
    else:
        U, Dv, _ = eig_truncate(Lprior, alpha)
        JU = Jr * U
        covi = JU * mp.diag([1 / dd for dd in Dv]) * JU.T
    out["forward_pose_prior"] = mp.inverse(covi)
    E = mp.eigsy(Lprior, eigvals_only=True)
    out["_fwd_rank"] = rank
    out["_fwd_cond"] = float(max(E) / min(E)) if min(E) > 0 else float("inf")
    out["_fwd_n_landmarks"] = n0
    return out


def imu_jacobians(im, G, xi, si, xj, sj):
    """IMUFactor's analytic raw Jacobians (include/factor/imu_factor.h:66-155 as oracle/isvo_factors.h:148-229 restates them), 6-column
    pose blocks: (Jpi 15x6, Jsi 15x9, Jpj 15x6, Jsj 15x9)"""
    Jm = mpv(abi.arr(im.jacobian))
    blk = lambda r0, c0: mp.matrix([[Jm[(r0 + a) * 15 + c0 + b] for b in range(3)] for a in range(3)])
    dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg = blk(0, 9), blk(0, 12), blk(3, 12), blk(6, 9), blk(6, 12)
    dqv = mpv(abi.arr(im.delta_q)); dq = (dqv[3], dqv[0], dqv[1], dqv[2])
    lbg, dt = mpv(abi.arr(im.linearized_bg)), mp.mpf(float(im.sum_dt))
    Qi, Qj = q_from_pose(xi), q_from_pose(xj)
    Qii = q_inv(Qi)
    RiT = q_to_R(Qii)
    dbg = mp.matrix([si[6 + k] - lbg[k] for k in range(3)])
    th = dq_dbg * dbg
    cdq = qmul(dq, (ONE, th[0] / 2, th[1] / 2, th[2] / 2))
    Jpi, Jsi, Jpj, Jsj = mp.zeros(15, 6), mp.zeros(15, 9), mp.zeros(15, 6), mp.zeros(15, 9)
    set_block(Jpi, 0, 0, -RiT)
    u = [G[k] * dt * dt / 2 + xj[k] - xi[k] - si[k] * dt for k in range(3)]
    set_block(Jpi, 0, 3, skew(qrot(Qii, u)))
    P4 = qleft44(qmul(q_inv(Qj), Qi)) * qright44(cdq)
    set_block(Jpi, 3, 3, -sub(P4, 1, 4, 1, 4))
    u = [G[k] * dt + sj[k] - si[k] for k in range(3)]
    set_block(Jpi, 6, 3, skew(qrot(Qii, u)))
    set_block(Jsi, 0, 0, -RiT * dt); set_block(Jsi, 0, 3, -dp_dba); set_block(Jsi, 0, 6, -dp_dbg)
    set_block(Jsi, 3, 6, -qleft33(qmul(qmul(q_inv(Qj), Qi), dq)) * dq_dbg)      # the uncorrected delta_q (:105)
    set_block(Jsi, 6, 0, -RiT); set_block(Jsi, 6, 3, -dv_dba); set_block(Jsi, 6, 6, -dv_dbg)
    set_block(Jsi, 9, 3, -mp.eye(3)); set_block(Jsi, 12, 6, -mp.eye(3))
    set_block(Jpj, 0, 0, RiT)
    set_block(Jpj, 3, 3, qleft33(qmul(qmul(q_inv(cdq), Qii), Qj)))
    set_block(Jsj, 6, 0, RiT); set_block(Jsj, 9, 3, mp.eye(3)); set_block(Jsj, 12, 6, mp.eye(3))
    return Jpi, Jsi, Jpj, Jsj


def yaw_res(x, m):                   # YawFactor (yaw_factor.h:15-19, 51-65): second component of R(x) m, m = Rz^-1 e_x fixed at construction
    return [qrot(qnorm(q_from_pose(x)), m)[1]]


def marg_backward(z, p, alpha):
    """-> dict(backward_relpose, backward_vb, backward_rollpitch = informations of the recovered factors) + conditioning figures.
    Parameter order: T1 = frame v @0, VB1 @6, T0 = frame v - 1 @15, VB0 @21 (oracle/isv_oracle.c:765)."""
    xi, xj = mpv(z[p + "pose"][2]), mpv(z[p + "pose"][3])
    si, sj = mpv(z[p + "sb"][0]), mpv(z[p + "sb"][1])
    G = mpv(z["gravity"])
    Lam = mp.zeros(30, 30)
    vb = struct_of(abi.isv_linear9_t, z[p + "vb_prior"])
    add_block(Lam, 21, 21, sym_info(abi.arr(vb.sqrt_info), 9))
    im = struct_of(abi.isv_imu_t, z[p + "imu"])
    cov = mp.matrix(15, 15)
    cv = mpv(abi.arr(im.covariance))
    for a in range(15):
        for b in range(15):
            cov[a, b] = cv[a * 15 + b]
    Om = mp.inverse(cov)                         # sqrt_info^T sqrt_info with sqrt_info = LLT(covariance^-1).L^T (imu_factor.h:36)
    Js = imu_jacobians(im, G, xi, si, xj, sj)
    ib = (15, 21, 0, 6)
    for a in range(4):
        for b in range(4):
            add_block(Lam, ib[a], ib[b], Js[a].T * Om * Js[b])
    A, Bm, Dm = sub(Lam, 0, 21, 0, 21), sub(Lam, 0, 21, 21, 30), sub(Lam, 21, 30, 21, 30)
    Lprior = A - Bm * mp.inverse(Dm) * Bm.T
    Qi, Qj = q_from_pose(xi), q_from_pose(xj)
    d = [xj[k] - xi[k] for k in range(3)]
    rp_dt = qrot(q_inv(Qi), d)
    Rm = q_to_R(qmul(q_inv(Qi), Qj))
    rp_dR = [Rm[a, b] for a in range(3) for b in range(3)]
    I6 = [ONE if a == b else ZERO for a in range(6) for b in range(6)]
    Jrp_i = jac_pose_mp(lambda x: relpose_res(x, xj, rp_dt, rp_dR, I6), xi, 6)
    Jrp_j = jac_pose_mp(lambda x: relpose_res(xi, x, rp_dt, rp_dR, I6), xj, 6)
    gR = [v for v in q_to_R(Qi)]
    Jg = jac_pose_mp(lambda x: rollpitch_res(x, gR, [ONE, ZERO, ZERO, ONE]), xi, 2)
    m_yaw = qrot(q_inv(Qi), [ONE, ZERO, ZERO])
    Jyaw = jac_pose_mp(lambda x: yaw_res(x, m_yaw), xi, 1)
    Jr = mp.zeros(21, 21)
    set_block(Jr, 0, 15, Jrp_i); set_block(Jr, 0, 0, Jrp_j)
    for a in range(9):
        Jr[6 + a, 6 + a] = ONE
    set_block(Jr, 15, 15, Jg)
    for a in range(3):
        Jr[17 + a, 15 + a] = ONE
    set_block(Jr, 20, 15, Jyaw)
    U, Dv, Eall = eig_truncate(Lprior, alpha)
    out = {"backward_relpose": project_info(sub(Jr, 0, 6, 0, 21), U, Dv), "backward_vb": project_info(sub(Jr, 6, 15, 0, 21), U, Dv),
           "backward_rollpitch": project_info(sub(Jr, 15, 17, 0, 21), U, Dv)}
    out["_bwd_rank"] = len(Dv)
    out["_bwd_eigs"] = [float(e) for e in Eall]
    out["_bwd_cond_kept"] = float(max(Dv) / min(Dv))
    return out


def to_np(M):
    return np.array([[float(M[a, b]) for b in range(M.cols)] for a in range(M.rows)])
