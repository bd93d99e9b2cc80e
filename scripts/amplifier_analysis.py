"""Which state direction does the estimator amplify from solve to solve, and at what gain per frame?  (VERDICT r3 task 8a)

CPU ONLY (the restated window manager tests/sequence_harness.py + the oracle; test infrastructure, not the product path).
On the EuRoC stand-in stream (tests/test_sequence_long.py: N = 18, Vo = 8, 20 Hz frames) the estimator is run to solved frame
K0, its complete state is cloned there (window states, pre-integrations, tracks with depths, every prior factor), and the
window-state part x = (P, theta, V, ba, bg) x 18 frames = 270 coordinates is perturbed by +-EPS along every coordinate.  Every
perturbed clone is pushed through the next n frames (processIMU ... processImage: propagate, triangulate,
backendOptimization, update() of the priors, slideWindow) and the difference of the window states it ends with, divided by
2 EPS, is one column of the finite-difference Jacobian J_n = d x(K0 + n) / d x(K0) of the solve -> solve map, for
n = 1, 2, 5, 10, 20 frames, once with the reference's update() of the prior pseudo-measurements (src/estimator.cpp:1133-1144)
and once with it switched off (the oracle's isvo_debug_no_update).  Reported: the leading singular values of J_n (the gain of
the most amplified direction after n frames), the per-frame gain sigma_1(J_n)^(1/n), and what the leading output direction
is made of: the share of its norm in P / theta / V / ba / bg, and its overlap with the unobservable-or-weak directions of a
monocular VIO window -- global translation, yaw about gravity, metric SCALE (positions and velocities scaled about the first
frame), accelerometer-bias and gyro-bias shifts common to all frames.

usage: python scripts/amplifier_analysis.py [--k0 200] [--eps 1e-8] [--workers 6] [--out profiles/r04_amplifier.json]"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import isvins_loader  # noqa: E402
isvins_loader.load()
from isvins_amd import abi  # noqa: E402
import oracle_lib  # noqa: E402
import sequence_harness as sh  # noqa: E402
import test_sequence_long as T  # noqa: E402

N, NVO = T.N, T.NVO
STEPS = (1, 2, 5, 10, 20, 40)


def _cp(src, Ty):
    o = Ty()
    C.memmove(C.byref(o), C.byref(src), C.sizeof(Ty))
    return o


def clone(est):
    """a deep copy of sequence_harness.Estimator (ctypes PODs by value; the solver / library handles are shared)"""
    e = sh.Estimator(est.solver, est.lib, est.N, est.Nvo)
    for k in ("g", "ric", "tic", "Ps", "Vs", "Rs", "Bas", "Bgs", "Headers", "acc_0", "gyr_0"):
        setattr(e, k, getattr(est, k).copy())
    e.frame_count, e.first_imu, e.solver_flag, e.margin_old = est.frame_count, est.first_imu, est.solver_flag, est.margin_old
    e.pre = []
    for p in est.pre:
        if p is None:
            e.pre.append(None); continue
        q = sh.PreInt.__new__(sh.PreInt)
        q.lib = p.lib; q.pod = _cp(p.pod, abi.isv_imu_t); q.acc_0 = p.acc_0.copy(); q.gyr_0 = p.gyr_0.copy(); q.noise = p.noise.copy()
        e.pre.append(q)
    e.bufs = [list(b) for b in est.bufs]
    e.tracks = []
    for t in est.tracks:
        u = sh.Track(t.id, t.start_frame); u.points = [p.copy() for p in t.points]; u.depth = t.depth; u.solve_flag = t.solve_flag
        e.tracks.append(u)
    e.pose_prior = _cp(est.pose_prior, abi.isv_se3_prior_t); e.vb_prior = _cp(est.vb_prior, abi.isv_linear9_t)
    e.relpose = [_cp(f, abi.isv_relpose_t) for f in est.relpose]
    e.rollpitch = [_cp(f, abi.isv_rollpitch_t) for f in est.rollpitch]
    e.to_add = None if est.to_add is None else (_cp(est.to_add[0], abi.isv_se3_prior_t), _cp(est.to_add[1], abi.isv_relpose_t), _cp(est.to_add[2], abi.isv_linear9_t))
    return e


def so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def so3_log(R):
    c = max(-1.0, min(1.0, (np.trace(R) - 1) / 2)); th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) / 2
    return v if th < 1e-9 else v * th / np.sin(th)


def get_x(est, ref=None):
    """window state coordinates: per frame P (3), theta (3: log(R_ref^T R), 0 when ref is None), V, ba, bg"""
    x = np.zeros((N, 15))
    x[:, 0:3] = est.Ps; x[:, 6:9] = est.Vs; x[:, 9:12] = est.Bas; x[:, 12:15] = est.Bgs
    if ref is not None:
        for j in range(N):
            x[j, 3:6] = so3_log(ref.Rs[j].T @ est.Rs[j])
    return x.ravel()


def perturb(est, k, eps):
    j, c = divmod(k, 15)
    if c < 3: est.Ps[j, c] += eps
    elif c < 6:
        w = np.zeros(3); w[c - 3] = eps; est.Rs[j] = est.Rs[j] @ so3_exp(w)
    elif c < 9: est.Vs[j, c - 6] += eps
    elif c < 12: est.Bas[j, c - 9] += eps
    else: est.Bgs[j, c - 12] += eps


def advance(est, stream, sim, i0, n, marks):
    """push frames i0 + 1 .. i0 + n through `est`; return {m: clone-free snapshot of (Ps, Rs, Vs, Bas, Bgs)} at the marks"""
    out = {}
    for s in range(1, n + 1):
        imu, t, image = stream[i0 + s]
        for (dt, a, g) in imu:
            est.process_imu(dt, a, g)
        est.process_image(image, t)
        if s in marks:
            snap = sh.Estimator.__new__(sh.Estimator)
            snap.Ps, snap.Rs, snap.Vs, snap.Bas, snap.Bgs = est.Ps.copy(), est.Rs.copy(), est.Vs.copy(), est.Bas.copy(), est.Bgs.copy()
            out[s] = snap
    return out


_G = {}


def _init(k0, no_update, nmax):
    oracle = oracle_lib.load()
    oracle.isvo_debug_no_update(1 if no_update else 0)
    cfg = abi.make_config(N, NVO, max_landmarks=1000, max_obs=1000 * N, max_batch=1)
    i0 = N - 1 + k0
    sim, stream = T.record_stream(i0 + nmax + 1)
    est = sh.Estimator(sh.OracleSolver(oracle, cfg), oracle, N, NVO)
    for i, (imu, t, image) in enumerate(stream[: i0 + 1]):
        for (dt, a, g) in imu:
            est.process_imu(dt, a, g)
        boot = T.bootstrap(sim, i) if (est.solver_flag == "INITIAL" and est.frame_count == N - 1) else None
        est.process_image(image, t, bootstrap=boot)
    base = advance(clone(est), stream, sim, i0, nmax, STEPS)
    _G.update(est=est, stream=stream, sim=sim, i0=i0, nmax=nmax, base=base)


def _column(args):
    k, eps = args
    res = {}
    outs = []
    for sgn in (+1, -1):
        e = clone(_G["est"]); perturb(e, k, sgn * eps)
        outs.append(advance(e, _G["stream"], _G["sim"], _G["i0"], _G["nmax"], STEPS))
    for m in STEPS:
        if m <= _G["nmax"]:
            res[m] = (get_x(outs[0][m], _G["base"][m]) - get_x(outs[1][m], _G["base"][m])) / (2 * eps)
    return k, res


def structured_directions(est):
    """unit vectors in x-coordinates: translation (3), yaw, scale, common accelerometer bias (3), common gyro bias (3)"""
    d = {}
    for c, nm in enumerate("xyz"):
        v = np.zeros((N, 15)); v[:, c] = 1; d["translation " + nm] = v.ravel()
    v = np.zeros((N, 15)); ez = np.array([0, 0, 1.0])
    for j in range(N):
        v[j, 0:3] = np.cross(ez, est.Ps[j] - est.Ps[0]); v[j, 3:6] = est.Rs[j].T @ ez; v[j, 6:9] = np.cross(ez, est.Vs[j])
    d["yaw about gravity"] = v.ravel()
    v = np.zeros((N, 15))
    for j in range(N):
        v[j, 0:3] = est.Ps[j] - est.Ps[0]; v[j, 6:9] = est.Vs[j]
    d["metric scale"] = v.ravel()
    for c, nm in enumerate("xyz"):
        v = np.zeros((N, 15)); v[:, 9 + c] = 1; d["accelerometer bias " + nm] = v.ravel()
        v = np.zeros((N, 15)); v[:, 12 + c] = 1; d["gyro bias " + nm] = v.ravel()
    return {k: v / np.linalg.norm(v) for k, v in d.items()}


GAUGE = ("translation x", "translation y", "translation z", "yaw about gravity")


def gauge_projector(est):
    """I - Q Q^T with Q the orthonormalised gauge directions (global translation, yaw about gravity) at the given window"""
    d = structured_directions(est)
    Q, _ = np.linalg.qr(np.stack([d[k] for k in GAUGE], 1))
    return np.eye(15 * N) - Q @ Q.T


def analyse(J, est_end, est_start=None):
    if est_start is not None:
        # the unobservable gauge of a VIO window (4 dof) is carried along unchanged by every solve: a perturbation along it is neither
        # damped nor amplified, and sigma = sqrt(18) for a translation is only the count of frames.  Project it out on both sides.
        J = gauge_projector(est_end) @ J @ gauge_projector(est_start)
    U, S, Vt = np.linalg.svd(J)
    u1 = U[:, 0].reshape(N, 15); v1 = Vt[0].reshape(N, 15)
    share = lambda a: {nm: float(np.sum(a[:, s] ** 2)) for nm, s in (("P", slice(0, 3)), ("theta", slice(3, 6)), ("V", slice(6, 9)), ("ba", slice(9, 12)), ("bg", slice(12, 15)))}
    dirs = structured_directions(est_end)
    ov = {k: float(abs(v @ U[:, 0])) for k, v in dirs.items()}
    # how much of the output direction the named directions span together (least squares on the orthonormalised set)
    Q, _ = np.linalg.qr(np.stack(list(dirs.values()), 1))
    span = float(np.linalg.norm(Q.T @ U[:, 0]))
    return dict(singular_values=[float(x) for x in S[:8]], out_share=share(u1), in_share=share(v1), out_overlap=ov, out_span_of_named=span)


def run(k0, eps, workers, no_update, nmax):
    t0 = time.time()
    with mp.Pool(workers, initializer=_init, initargs=(k0, no_update, nmax)) as pool:
        cols = pool.map(_column, [(k, eps) for k in range(15 * N)], chunksize=4)
    _init(k0, no_update, nmax)
    res = {}
    for m in STEPS:
        if m > nmax:
            continue
        J = np.zeros((15 * N, 15 * N))
        for k, r in cols:
            J[:, k] = r[m]
        end = _G["base"][m]
        a = analyse(J, end)
        a["gain_per_frame"] = float(a["singular_values"][0] ** (1.0 / m))
        a["gauge_free"] = analyse(J, end, _G["est"])
        a["gauge_free"]["gain_per_frame"] = float(a["gauge_free"]["singular_values"][0] ** (1.0 / m))
        res[str(m)] = a
    return res, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k0", type=int, default=200); ap.add_argument("--eps", type=float, default=1e-8)
    ap.add_argument("--workers", type=int, default=6); ap.add_argument("--nmax", type=int, default=20)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04_amplifier.json"))
    a = ap.parse_args()
    out = dict(stream="EuRoC stand-in (tests/test_sequence_long.py), N = 18, Vo = 8", k0=a.k0, eps=a.eps,
               coordinates="x = (P, theta, V, ba, bg) of the 18 window frames; J_n = d x(k0 + n) / d x(k0), central differences")
    for nm, nu in (("with_update", 0), ("no_update", 1)):
        res, dt = run(a.k0, a.eps, a.workers, nu, a.nmax)
        out[nm] = res
        print(f"{nm}: {dt:.0f} s")
        for m, r in res.items():
            for tag, rr in (("raw       ", r), ("gauge-free", r["gauge_free"])):
                top = sorted(rr["out_overlap"].items(), key=lambda kv: -kv[1])[:3]
                print(f"  n = {m:>2} {tag}: sigma = {['%.3g' % s for s in rr['singular_values'][:4]]} gain/frame {rr['gain_per_frame']:.3f}  out share {({k: round(v, 2) for k, v in rr['out_share'].items()})} "
                      f"in share {({k: round(v, 2) for k, v in rr['in_share'].items()})} top overlaps {[(k, round(v, 2)) for k, v in top]}")
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
