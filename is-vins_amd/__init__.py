"""MI355X-native sliding-window VIO backend: drop-in for the IS-VINS per-frame solve path
(Estimator::solveOdometry -> backendOptimization, reference src/estimator.cpp:461-472,1541-1562).

The directory name carries a hyphen (`is-vins_amd`), so it is imported through
`isvins_loader.load()` (repo root), which registers it as the module `isvins_amd`.
  abi      ctypes mirror of include/isvins_backend.h
  synth    deterministic synthetic windows (BASELINE.json configs 2, 4, 5)
  backend  host-side mirror of the reference call surface over the HIP C-ABI library
  csrc/    hand-written gfx950 kernels + the C ABI (libisvins_hip.so)
"""
from . import abi, sharding, synth  # noqa: F401
