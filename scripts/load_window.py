import ctypes as C, numpy as np
from isvins_amd import abi
def load(path, N=18, Nvo=8):
    z = np.load(path)
    L, n_obs, nrp = int(z["L"]), int(z["n_obs"]), int(z["n_rollpitch"])
    w = abi.Window(N, Nvo, L, n_obs, nrp)
    for k in ("Ps", "Rs", "Vs", "Bas", "Bgs", "tic", "ric", "lm_start_frame", "lm_obs_ptr", "obs_point", "lm_depth"):
        getattr(w, k)[...] = z[k].reshape(getattr(w, k).shape)
    C.memmove(w.imu, z["imu"].tobytes(), C.sizeof(w.imu))
    C.memmove(C.byref(w.pose_prior), z["pose_prior"].tobytes(), C.sizeof(w.pose_prior))
    C.memmove(C.byref(w.vb_prior), z["vb_prior"].tobytes(), C.sizeof(w.vb_prior))
    C.memmove(w.relpose, z["relpose"].tobytes(), C.sizeof(w.relpose))
    C.memmove(w.rollpitch, z["rollpitch"].tobytes(), min(C.sizeof(w.rollpitch), len(z["rollpitch"].tobytes())))
    w.margin_old = int(z["margin_old"]); w.header0 = float(z["header0"])
    return w
