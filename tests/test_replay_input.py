"""tools/isv_replay --euroc input path on the CPU (--dump-events: no GPU): the reference's imu0/data.csv reader
(test/run_euroc.cpp:26-50) and the IMU / image pairing of System::getMeasurements + ProcessBackEnd
(src/System.cpp:160-202, 262-296), against a Python restatement of those rules -- including the interpolation branch
(an IMU sample behind the image time) and the images the reference drops or never reaches."""
import os
import subprocess

import numpy as np

from isvins_amd import backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def expected_events(imu, frames):
    """imu: list of (t, gyr3, acc3); frames: list of (t, n).  -> list of ('imu', dt, acc, gyr) / ('frame', t, n)"""
    out, q, cur, last = [], 0, -1.0, None
    for (ft, n) in frames:
        if q >= len(imu) or not (imu[-1][0] > ft):
            break                                   # "wait for imu"
        if not (imu[q][0] < ft):
            continue                                # "throw img"
        take = []
        while imu[q][0] < ft:
            take.append(q); q += 1
        take.append(q)                              # the first sample at or after the image, not popped
        for k in take:
            t, g, a = imu[k]
            if t <= ft:
                if cur < 0:
                    cur = t
                dt = t - cur; cur = t
                last = (np.array(a, float), np.array(g, float))
            else:
                dt1, dt2 = ft - cur, t - ft
                cur = ft
                w1, w2 = dt2 / (dt1 + dt2), dt1 / (dt1 + dt2)
                last = (w1 * last[0] + w2 * np.array(a, float), w1 * last[1] + w2 * np.array(g, float))
                dt = dt1
            out.append(("imu", dt, last[0].copy(), last[1].copy()))
        out.append(("frame", ft, n))
    return out


def test_euroc_reader_pairs_imu_and_frames_like_the_reference(tmp_path):
    backend.build()
    tool = os.path.join(ROOT, "tools", "isv_replay")
    rng = np.random.default_rng(3)
    t0 = 2_000_000_000                                   # ns
    imu = [(t0 + 5_000_000 * k, rng.normal(0, 0.1, 3), rng.normal(0, 1, 3) + [0, 0, 9.8]) for k in range(70)]
    # frame stamps: one before any IMU sample (dropped), some on sample times, some between samples (interpolated),
    # the last one beyond the IMU data (never reached)
    frame_ns = [t0 - 1_000_000, t0 + 50_000_000, t0 + 100_000_000, t0 + 152_500_000, t0 + 201_000_000, t0 + 250_000_000, t0 + 400_000_000]
    os.makedirs(tmp_path / "mav0" / "imu0")
    with open(tmp_path / "mav0" / "imu0" / "data.csv", "w") as f:
        f.write("#timestamp [ns],w_RS_S_x [rad s^-1],w_y,w_z,a_x,a_y,a_z\n")
        for (t, g, a) in imu:
            f.write("%d,%r,%r,%r,%r,%r,%r\n" % (t, *map(float, g), *map(float, a)))
    with open(tmp_path / "tracks.csv", "w") as f:
        f.write("#timestamp [ns],id,x,y,z\n")
        for i, t in enumerate(frame_ns):
            for k in range(3 + i):
                f.write("%d,%d,%r,%r,1.0\n" % (t, 100 * i + k, 0.01 * k, -0.02 * k))
    with open(tmp_path / "config.txt", "w") as f:
        f.write("config 5 2 50 10 460.0 9.81007 0.1 5.0 0.2 0.004 0.001 0.0001 0.0217\nric 1 0 0 0 1 0 0 0 1\ntic 0 0 0\n")
    dump = tmp_path / "events.txt"
    subprocess.run([tool, "--euroc", str(tmp_path / "mav0"), "--tracks", str(tmp_path / "tracks.csv"), "--config", str(tmp_path / "config.txt"),
                    "--dump-events", str(dump)], check=True, capture_output=True, text=True, timeout=60)
    got = [l.split() for l in open(dump)]
    exp = expected_events([(t / 1e9, g, a) for (t, g, a) in imu], [(t / 1e9, 3 + i) for i, t in enumerate(frame_ns)])
    assert len(got) == len(exp)
    assert [e[0] for e in exp].count("frame") == 5       # the first image is dropped, the last one never reached
    n_interp = 0
    for g, e in zip(got, exp):
        assert g[0] == e[0]
        if e[0] == "imu":
            v = np.array(g[1:], float)
            assert abs(v[0] - e[1]) <= 1e-15 and np.abs(v[1:4] - e[2]).max() <= 1e-15 and np.abs(v[4:7] - e[3]).max() <= 1e-15
            n_interp += 0 < e[1] < 4.9e-3
        else:
            assert float(g[1]) == e[1] and int(g[2]) == e[2]
    assert n_interp >= 2                                 # both off-grid frames went through the interpolation branch
