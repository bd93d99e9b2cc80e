"""isv_batch_upload ships the caller's windows as raw CSR and k_upload_build derives the solver's view on the device (round 5:
landmark / factor records, the (host, observer) pair groups by a stable counting sort, the longest-first schedule, the factor
stream) -- what pack_window derives on the host (ISV_HOST_PACK=1: kept as the A/B and test path).  The two must be the SAME arrays:
every solve below is bitwise the host-packed one (a different order anywhere in the factor stream would change the last bits of
the Gram sums), and one of each shape is checked against the oracle."""
import numpy as np
import pytest

from isvins_amd import backend, synth
from test_gpu_solve import check_marg, check_window, oracle_run

pytestmark = pytest.mark.gpu


def _both(monkeypatch, N, Nvo, ws, **kw):
    cap = dict(max_landmarks=max(w.L for w in ws), max_obs=max(w.n_obs for w in ws), max_batch=len(ws), **kw)
    monkeypatch.setenv("ISV_HOST_PACK", "1")
    bh = backend.Backend(N, Nvo, **cap)
    monkeypatch.delenv("ISV_HOST_PACK")
    bd = backend.Backend(N, Nvo, **cap)
    try:
        gh = [w.clone() for w in ws]; gd = [w.clone() for w in ws]
        sh_, mh = bh.optimize_batch(gh)
        sd, md = bd.optimize_batch(gd)
        for a, b, sa, sb in zip(gh, gd, sh_, sd):
            assert sa.iterations == sb.iterations and sa.termination == sb.termination and sa.final_cost == sb.final_cost
            assert np.array_equal(a.state_vector(), b.state_vector()) and np.array_equal(a.priors_vector(), b.priors_vector())
        for a, b in zip(mh, md):
            assert bytes(a) == bytes(b)
        return bd.cfg, gd, sd, md
    finally:
        bh.close(); bd.close()


@pytest.mark.parametrize("N,Nvo,L,n", [(11, 5, 300, 24), (18, 8, 300, 6), (6, 3, 40, 5)])
def test_device_built_upload_is_bitwise_the_host_packed_one(oracle, monkeypatch, N, Nvo, L, n):
    ws = synth.make_windows(range(500, 500 + n), n_frames=N, n_vo=Nvo, n_landmarks=L)
    ws[1] = synth.make_window(777, n_frames=N, n_vo=Nvo, n_landmarks=max(4, L // 7))          # a ragged batch
    cfg, g, s, m = _both(monkeypatch, N, Nvo, ws)
    o, so, mo = oracle_run(oracle, cfg, ws[0])
    check_window(o, so, g[0], s[0]); check_marg(mo, m[0], Nvo)


def test_long_window_and_unfused_path(oracle, monkeypatch):
    """a 2000-landmark / 12 000-factor window beside a short one: the unfused k_proj_linearize<0> + k_sweep_mfma pair reads the
    device-built pair groups and the host-built tiles; every landmark hosted in the first frames gives > 64 groups per wavefront"""
    ws = [synth.make_window(900, n_frames=20, n_vo=8, n_landmarks=2000, target_factors=12000),
          synth.make_window(901, n_frames=20, n_vo=8, n_landmarks=150, host_frames=(0, 1), max_track=7)]
    _both(monkeypatch, 20, 8, ws)


def test_free_extrinsic(monkeypatch):
    ws = synth.make_windows([610, 611, 612], n_frames=11, n_vo=5, n_landmarks=120)
    _both(monkeypatch, 11, 5, ws, estimate_extrinsic=1)


def test_observation_offsets_that_do_not_start_at_zero(monkeypatch):
    """lm_obs_ptr is only required to delimit each landmark's observations: a table that starts at 3 (with three unused points in
    front) is the same window"""
    w = synth.make_window(620, n_landmarks=60)
    v = w.clone()
    pad = 3
    v.obs_point = np.concatenate([np.full((pad, 3), 7.0), w.obs_point[: w.n_obs]]).copy()
    v.lm_obs_ptr = (w.lm_obs_ptr + pad).astype(np.int32)
    v.n_obs = w.n_obs + pad
    bd = backend.Backend(11, 5, max_landmarks=60, max_obs=v.n_obs, max_batch=1)
    try:
        a = w.clone(); b = v.clone()
        sa, _ = bd.optimize(a); sb, _ = bd.optimize(b)
        assert sa.iterations == sb.iterations and np.array_equal(a.state_vector(), b.state_vector())
    finally:
        bd.close()


def test_two_part_upload_and_two_copy_download_change_nothing(monkeypatch):
    """round 5: a batch that uses most of a large handle goes up in two parts (the first while the host threads still pack the second)
    and comes back in two copies instead of 24; ISV_UPLOAD_ONE_COPY / ISV_DOWNLOAD_ARRAYS (read per call) keep the old forms.  Same
    handle, same windows: every state, prior, summary and marginalisation record has to be the same bits, with a ragged batch
    (the offsets of the second part depend on every window before it)."""
    n = 420
    ws = synth.make_windows(range(3000, 3000 + n), n_frames=11, n_vo=5, n_landmarks=150)
    ws[7] = synth.make_window(3999, n_frames=11, n_vo=5, n_landmarks=30)
    be = backend.Backend(11, 5, max_landmarks=160, max_obs=max(w.n_obs for w in ws) + 40, max_batch=512)
    try:
        a = [w.clone() for w in ws]; b = [w.clone() for w in ws]
        sa, ma = be.optimize_batch(a)
        monkeypatch.setenv("ISV_UPLOAD_ONE_COPY", "1"); monkeypatch.setenv("ISV_DOWNLOAD_ARRAYS", "1")
        sb, mb = be.optimize_batch(b)
        for x, y, p, q in zip(a, b, sa, sb):
            assert bytes(p) == bytes(q)
            assert np.array_equal(x.state_vector(), y.state_vector()) and np.array_equal(x.priors_vector(), y.priors_vector())
        for p, q in zip(ma, mb):
            assert bytes(p) == bytes(q)
        assert sa[0].iterations > 0 and sa[0].status == 0
    finally:
        be.close()
