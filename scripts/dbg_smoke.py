import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import isvins_loader; isvins_loader.load()
import torch
mode = sys.argv[1]
from isvins_amd import backend
print("avail", torch.cuda.is_available())
if mode == "zeros_first":
    torch.zeros(1, device="cuda:0")
try:
    be = backend.Backend(11, 5, max_landmarks=64, max_obs=704, max_batch=1)
    print(mode, "create ok")
except Exception as e:
    print(mode, "create FAILED", e)
if mode == "zeros_after":
    x = torch.zeros(1, device="cuda:0"); print("torch alloc after ok", x.device)
