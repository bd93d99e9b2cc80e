"""Synthetic sliding-window generator (configs 2, 4, 5 of BASELINE.json).

Deterministic: every random draw comes from a SplitMix64 counter stream seeded with
0x15A51A5500000000 + window_id, so Python here and any other implementation of the same
counter rule produce identical windows.  IMU samples are pre-integrated with the midpoint rule of
IntegrationBase (reference include/factor/integration_base.h:54-158) in numpy -- this is the
producer side (Estimator::processIMU), which stays on the CPU in the reference too.
"""
import ctypes as C
import numpy as np
from . import abi

SEED0 = 0x15A51A5500000000
MASK = (1 << 64) - 1
# config/euroc_config.yaml:25-37,57-61
RIC = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422],
                [0.999557249008, 0.0149672133247, 0.025715529948],
                [-0.0257744366974, 0.00375618835797, 0.999660727178]])
TIC = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])
ACC_N, GYR_N, ACC_W, GYR_W = 0.22627, 0.003988, 0.001, 0.0001
G_NORM = 9.81007


class SplitMix64:
    """counter-based: the k-th 64-bit output is mix(seed + (k+1) * 0x9E3779B97F4A7C15)"""

    def __init__(self, seed):
        self.seed = seed & MASK
        self.k = 0

    def u64(self, n):
        ks = np.arange(self.k + 1, self.k + n + 1, dtype=np.uint64)
        self.k += n
        with np.errstate(over="ignore"):
            z = np.uint64(self.seed) + ks * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return z

    def uniform(self, n):
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def normal(self, n):
        u = self.uniform(2 * n)
        u1 = np.maximum(u[:n], 1e-300)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u[n:])


def _rot_zyx(y, p, r):
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def _exp_so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


class Trajectory:
    """body on a circle r=3 m at 0.5 m/s, yaw tangent, roll/pitch +-5 deg sinusoids"""

    def __init__(self, phase, pitch0=0.0, pitch_amp=None):
        self.r, self.v = 3.0, 0.5
        self.w = self.v / self.r
        self.ph = phase
        self.amp = np.deg2rad(5.0)
        # (test windows only: a constant pitch offset / another pitch amplitude, e.g. a body that looks straight up,
        # the double2vector branch of src/estimator.cpp:537-547)
        self.p0 = pitch0
        self.pamp = self.amp if pitch_amp is None else pitch_amp

    def ypr(self, t):
        y = self.w * t + self.ph + np.pi / 2
        p = self.p0 + self.pamp * np.sin(2 * np.pi * 0.5 * t)
        r = self.amp * np.sin(2 * np.pi * 0.3 * t + 1.0)
        return y, p, r

    def ypr_dot(self, t):
        return (self.w, self.pamp * 2 * np.pi * 0.5 * np.cos(2 * np.pi * 0.5 * t),
                self.amp * 2 * np.pi * 0.3 * np.cos(2 * np.pi * 0.3 * t + 1.0))

    def R(self, t):
        return _rot_zyx(*self.ypr(t))

    def p(self, t):
        a = self.w * t + self.ph
        return np.array([self.r * np.cos(a), self.r * np.sin(a), 0.0])

    def vel(self, t):
        a = self.w * t + self.ph
        return np.array([-self.r * self.w * np.sin(a), self.r * self.w * np.cos(a), 0.0])

    def acc(self, t):
        a = self.w * t + self.ph
        return np.array([-self.r * self.w ** 2 * np.cos(a), -self.r * self.w ** 2 * np.sin(a), 0.0])

    def gyro(self, t):
        y, p, r = self.ypr(t)
        yd, pd, rd = self.ypr_dot(t)
        return np.array([rd - yd * np.sin(p),
                         pd * np.cos(r) + yd * np.sin(r) * np.cos(p),
                         -pd * np.sin(r) + yd * np.cos(r) * np.cos(p)])


def _skew(v):
    z = np.zeros(v.shape[:-1])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def _qmul(a, b):  # (w,x,y,z)
    return np.stack([a[..., 0] * b[..., 0] - a[..., 1] * b[..., 1] - a[..., 2] * b[..., 2] - a[..., 3] * b[..., 3],
                     a[..., 0] * b[..., 1] + a[..., 1] * b[..., 0] + a[..., 2] * b[..., 3] - a[..., 3] * b[..., 2],
                     a[..., 0] * b[..., 2] + a[..., 2] * b[..., 0] + a[..., 3] * b[..., 1] - a[..., 1] * b[..., 3],
                     a[..., 0] * b[..., 3] + a[..., 3] * b[..., 0] + a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]], -1)


def _q2R(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def preintegrate(dt, acc, gyr, ba, bg):
    """Batched midpoint pre-integration (integration_base.h:54-158).
    acc, gyr: [B, S+1, 3] samples (sample 0 = acc_0/gyr_0), dt scalar, ba/bg: [B, 3] linearisation
    biases.  Returns dict of delta_p [B,3], delta_q [B,4] (w,x,y,z), delta_v, jacobian [B,15,15],
    covariance [B,15,15], sum_dt."""
    B, S1, _ = acc.shape
    dp = np.zeros((B, 3)); dv = np.zeros((B, 3)); dq = np.zeros((B, 4)); dq[:, 0] = 1
    Jm = np.tile(np.eye(15), (B, 1, 1)); Cv = np.zeros((B, 15, 15))
    nd = np.concatenate([[ACC_N ** 2] * 3, [GYR_N ** 2] * 3, [ACC_N ** 2] * 3, [GYR_N ** 2] * 3,
                         [ACC_W ** 2] * 3, [GYR_W ** 2] * 3])
    I3 = np.eye(3)
    for s in range(1, S1):
        a0 = acc[:, s - 1] - ba; a1 = acc[:, s] - ba
        w = 0.5 * (gyr[:, s - 1] + gyr[:, s]) - bg
        Rd = _q2R(dq)
        inc = np.concatenate([np.ones((B, 1)), w * dt / 2], -1)
        rdq = _qmul(dq, inc)
        Rr = _q2R(rdq)
        un0 = np.einsum("bij,bj->bi", Rd, a0); un1 = np.einsum("bij,bj->bi", Rr, a1)
        un = 0.5 * (un0 + un1)
        rdp = dp + dv * dt + 0.5 * un * dt * dt
        rdv = dv + un * dt
        Rw, Ra0, Ra1 = _skew(w), _skew(a0), _skew(a1)
        ImW = I3 - Rw * dt
        T1 = Rd @ Ra0; T2 = Rr @ Ra1; T3 = T2 @ ImW
        F = np.zeros((B, 15, 15)); V = np.zeros((B, 15, 18))
        F[:, 0:3, 0:3] = I3
        F[:, 0:3, 3:6] = -0.25 * T1 * dt * dt + -0.25 * T3 * dt * dt
        F[:, 0:3, 6:9] = I3 * dt
        F[:, 0:3, 9:12] = -0.25 * (Rd + Rr) * dt * dt
        F[:, 0:3, 12:15] = -0.25 * T2 * dt * dt * -dt
        F[:, 3:6, 3:6] = ImW
        F[:, 3:6, 12:15] = -1.0 * I3 * dt
        F[:, 6:9, 3:6] = -0.5 * T1 * dt + -0.5 * T3 * dt
        F[:, 6:9, 6:9] = I3
        F[:, 6:9, 9:12] = -0.5 * (Rd + Rr) * dt
        F[:, 6:9, 12:15] = -0.5 * T2 * dt * -dt
        F[:, 9:12, 9:12] = I3; F[:, 12:15, 12:15] = I3
        V[:, 0:3, 0:3] = 0.25 * Rd * dt * dt
        V[:, 0:3, 3:6] = 0.25 * -T2 * dt * dt * 0.5 * dt
        V[:, 0:3, 6:9] = 0.25 * Rr * dt * dt
        V[:, 0:3, 9:12] = V[:, 0:3, 3:6]
        V[:, 3:6, 3:6] = 0.5 * I3 * dt; V[:, 3:6, 9:12] = 0.5 * I3 * dt
        V[:, 6:9, 0:3] = 0.5 * Rd * dt
        V[:, 6:9, 3:6] = 0.5 * -T2 * dt * 0.5 * dt
        V[:, 6:9, 6:9] = 0.5 * Rr * dt
        V[:, 6:9, 9:12] = V[:, 6:9, 3:6]
        V[:, 9:12, 12:15] = I3 * dt; V[:, 12:15, 15:18] = I3 * dt
        Jm = F @ Jm
        Cv = F @ Cv @ F.transpose(0, 2, 1) + (V * nd) @ V.transpose(0, 2, 1)
        dq = rdq / np.linalg.norm(rdq, axis=-1, keepdims=True)
        dp, dv = rdp, rdv
    return dict(delta_p=dp, delta_q=dq, delta_v=dv, jacobian=Jm, covariance=Cv, sum_dt=dt * (S1 - 1))


def _track_lengths(rng, L, N, Nvo, target_F=None):
    host = np.minimum((rng.uniform(L) * Nvo).astype(np.int64), Nvo - 1)
    u = rng.uniform(L)
    geom = np.floor(np.log(np.maximum(u, 1e-300)) / np.log(1.0 - 0.25)).astype(np.int64)
    if target_F is not None:      # stress config: long tracks, then trimmed to hit F exactly
        geom = geom + (N - 2)
    k = np.minimum(2 + geom, N - host)
    if target_F is not None:
        F = int((k - 1).sum())
        i = 0
        while F != target_F:      # deterministic +-1 sweep over landmarks
            if F > target_F and k[i] > 2:
                k[i] -= 1; F -= 1
            elif F < target_F and k[i] < N - host[i]:
                k[i] += 1; F += 1
            i = (i + 1) % L
    return host.astype(np.int32), k.astype(np.int32)


def make_windows(window_ids, n_frames=11, n_vo=5, n_landmarks=300, target_factors=None, margin_old=1,
                 host_frames=None, max_track=None, pitch0_deg=0.0, pitch_amp_deg=None):
    """Build `abi.Window`s for the given ids (config 2: ids=[0]; config 4: range(1024);
    config 5: n_frames=20, n_vo=8, n_landmarks=2000, target_factors=30000).
    Test-only shaping (defaults leave the SURVEY 8d generator untouched): host_frames = (lo, hi) confines the
    landmarks' host frames to [lo, hi); max_track caps the track length; pitch0_deg / pitch_amp_deg shape the pitch."""
    N, Nvo, L = n_frames, n_vo, n_landmarks
    kf_dt, S, imu_dt = 0.1, 20, 0.005
    specs = []
    acc_all, gyr_all, ba_lin, bg_lin = [], [], [], []
    for wid in window_ids:
        rng = SplitMix64(SEED0 + int(wid))
        traj = Trajectory(phase=2 * np.pi * rng.uniform(1)[0], pitch0=np.deg2rad(pitch0_deg),
                          pitch_amp=None if pitch_amp_deg is None else np.deg2rad(pitch_amp_deg))
        t0 = 10.0 * rng.uniform(1)[0]
        tk = t0 + kf_dt * np.arange(N)
        P = np.stack([traj.p(t) for t in tk]); R = np.stack([traj.R(t) for t in tk]); Vv = np.stack([traj.vel(t) for t in tk])
        ba_true = 0.02 * rng.normal(3); bg_true = 0.002 * rng.normal(3)
        # IMU samples per interval: S+1 samples, white noise + bias random walk
        nA = rng.normal((N - 1) * (S + 1) * 3).reshape(N - 1, S + 1, 3)
        nG = rng.normal((N - 1) * (S + 1) * 3).reshape(N - 1, S + 1, 3)
        wA = rng.normal((N - 1) * 3).reshape(N - 1, 3); wG = rng.normal((N - 1) * 3).reshape(N - 1, 3)
        Ba = ba_true + np.concatenate([np.zeros((1, 3)), np.cumsum(ACC_W * np.sqrt(kf_dt) * wA, 0)])
        Bg = bg_true + np.concatenate([np.zeros((1, 3)), np.cumsum(GYR_W * np.sqrt(kf_dt) * wG, 0)])
        Gv = np.array([0, 0, G_NORM])
        acc = np.zeros((N - 1, S + 1, 3)); gyr = np.zeros((N - 1, S + 1, 3))
        for i in range(N - 1):
            for s in range(S + 1):
                t = tk[i] + s * imu_dt
                acc[i, s] = traj.R(t).T @ (traj.acc(t) + Gv) + Ba[i] + ACC_N * nA[i, s]
                gyr[i, s] = traj.gyro(t) + Bg[i] + GYR_N * nG[i, s]
        # landmarks
        host, k = _track_lengths(rng, L, N, Nvo, target_factors)
        if host_frames is not None:
            lo, hi = host_frames
            host = (lo + host % max(hi - lo, 1)).astype(np.int32)
            k = np.minimum(k, N - host).astype(np.int32)
        if max_track is not None:
            k = np.minimum(k, max_track).astype(np.int32)
        xyz = rng.uniform(3 * L).reshape(L, 3)
        pc = np.stack([-3 + 6 * xyz[:, 0], -3 + 6 * xyz[:, 1], 2 + 6 * xyz[:, 2]], -1)   # host camera frame
        n_obs = int(k.sum())
        obs_noise = rng.normal(2 * n_obs).reshape(n_obs, 2) / 460.0
        dep_noise = rng.normal(L)
        pert = rng.normal(N * 6).reshape(N, 6); vpert = rng.normal(N * 3).reshape(N, 3)
        bap = rng.normal(N * 3).reshape(N, 3); bgp = rng.normal(N * 3).reshape(N, 3)
        linp = rng.normal((N - 1) * 6).reshape(N - 1, 6)
        si_noise = rng.normal(36 + 81 + 36 * (Nvo - 1) + 4 * Nvo)
        specs.append(dict(wid=wid, P=P, R=R, V=Vv, Ba=Ba, Bg=Bg, host=host, k=k, pc=pc, n_obs=n_obs,
                          obs_noise=obs_noise, dep_noise=dep_noise, pert=pert, vpert=vpert, bap=bap,
                          bgp=bgp, si_noise=si_noise))
        acc_all.append(acc); gyr_all.append(gyr)
        ba_lin.append(Ba[:-1] + 0.005 * linp[:, :3]); bg_lin.append(Bg[:-1] + 0.0005 * linp[:, 3:])
    W = len(specs)
    pre = preintegrate(imu_dt, np.concatenate(acc_all), np.concatenate(gyr_all),
                       np.concatenate(ba_lin), np.concatenate(bg_lin))
    ba_lin = np.concatenate(ba_lin); bg_lin = np.concatenate(bg_lin)
    out = []
    for wi, sp in enumerate(specs):
        nrp = Nvo
        w = abi.Window(N, Nvo, L, sp["n_obs"], nrp)
        P, R = sp["P"], sp["R"]
        # observations (exact projection + pixel noise), landmark-major CSR
        w.lm_start_frame[:L] = sp["host"]
        w.lm_obs_ptr[1:] = np.cumsum(sp["k"])
        Rc = R @ RIC                                  # world <- camera
        tc = P + np.einsum("nij,j->ni", R, TIC)
        o = 0
        depth_true = np.zeros(L)
        for l in range(L):
            h, kk = int(sp["host"][l]), int(sp["k"][l])
            pw = Rc[h] @ sp["pc"][l] + tc[h]
            depth_true[l] = sp["pc"][l][2]
            for j in range(h, h + kk):
                pcj = Rc[j].T @ (pw - tc[j])
                w.obs_point[o, 0] = pcj[0] / pcj[2] + sp["obs_noise"][o, 0]
                w.obs_point[o, 1] = pcj[1] / pcj[2] + sp["obs_noise"][o, 1]
                w.obs_point[o, 2] = 1.0
                o += 1
        # the reference's inverse depth is w.r.t. the normalised observation of the host view
        w.lm_depth[:L] = 1.0 / ((1.0 / depth_true) * (1.0 + 0.1 * sp["dep_noise"]))
        # perturbed initial state
        for i in range(N):
            w.Ps[i] = P[i] + 0.02 * sp["pert"][i, :3]
            w.Rs[i] = R[i] @ _exp_so3(np.deg2rad(0.5) * sp["pert"][i, 3:])
            w.Vs[i] = sp["V"][i] + 0.05 * sp["vpert"][i]
            w.Bas[i] = sp["Ba"][i] + 0.02 * sp["bap"][i]
            w.Bgs[i] = sp["Bg"][i] + 0.002 * sp["bgp"][i]
        w.tic[:] = TIC; w.ric[:] = RIC
        for i in range(N - 1):
            b = wi * (N - 1) + i
            im = w.imu[i]
            im.delta_p[:] = pre["delta_p"][b]; im.delta_v[:] = pre["delta_v"][b]
            q = pre["delta_q"][b]; im.delta_q[:] = [q[1], q[2], q[3], q[0]]
            im.linearized_ba[:] = ba_lin[b]; im.linearized_bg[:] = bg_lin[b]
            im.sum_dt = pre["sum_dt"]
            im.jacobian[:] = pre["jacobian"][b].ravel(); im.covariance[:] = pre["covariance"][b].ravel()
        # priors: measurement = initial estimate (zero residual), upper-triangular sqrt_info
        z = sp["si_noise"]; zo = [0]

        def tri(n, d):
            e = z[zo[0]: zo[0] + n * n].reshape(n, n); zo[0] += n * n
            d = np.asarray(d, float)
            return (np.diag(d) + 0.05 * np.triu(e, 1) * d[:, None]).ravel()

        w.pose_prior.t[:] = w.Ps[0]; w.pose_prior.R[:] = w.Rs[0].ravel()
        w.pose_prior.sqrt_info[:] = tri(6, [1e2] * 3 + [1e3] * 3); w.pose_prior.index = 0
        w.vb_prior.VB[:] = np.concatenate([w.Vs[Nvo - 1], w.Bas[Nvo - 1], w.Bgs[Nvo - 1]])
        w.vb_prior.sqrt_info[:] = tri(9, [10.0] * 9); w.vb_prior.index = Nvo - 1
        for i in range(Nvo - 1):
            rp = w.relpose[i]
            rp.delta_t[:] = w.Rs[i].T @ (w.Ps[i + 1] - w.Ps[i])
            rp.delta_R[:] = (w.Rs[i].T @ w.Rs[i + 1]).ravel()
            rp.sqrt_info[:] = tri(6, [1e2] * 6); rp.imu_i, rp.imu_j = i, i + 1
        for i in range(nrp):
            g = w.rollpitch[i]
            g.R[:] = w.Rs[i].ravel(); g.sqrt_info[:] = tri(2, [1e2] * 2); g.index = i
        w.margin_old = margin_old
        w.header0 = float(sp["wid"])
        w.truth = dict(P=P, R=R, V=sp["V"], Ba=sp["Ba"], Bg=sp["Bg"], depth=depth_true)
        out.append(w)
    return out


def make_window(window_id=0, **kw):
    return make_windows([window_id], **kw)[0]
