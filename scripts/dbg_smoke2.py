import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
from isvins_amd import backend
mode = sys.argv[1]
if mode == "lib_first":
    lib = backend.load_library(); print("abi", lib.isv_abi_version())
import torch
print("avail", torch.cuda.is_available())
torch.zeros(1, device="cuda:0")
try:
    be = backend.Backend(11, 5, max_landmarks=64, max_obs=704, max_batch=1)
    print(mode, "create ok")
except Exception as e:
    print(mode, "create FAILED", e)
