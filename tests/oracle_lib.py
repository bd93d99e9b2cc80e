"""ctypes binding of the CPU oracle (oracle/libisv_oracle.so).  TESTS ONLY: the product package
never imports this module."""
import ctypes as C
import os
import subprocess

from isvins_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_lib = None
dp = C.POINTER(C.c_double)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "libisv_oracle.so")
    src = [os.path.join(ORACLE_DIR, f) for f in ("isv_oracle.c", "isvo_factors.h", "isvo_math.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src if os.path.exists(s)):
        build()
    lib = C.CDLL(so)
    lib.isvo_optimize.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t),
                                  C.POINTER(abi.isv_summary_t), C.POINTER(abi.isv_marg_result_t)]
    lib.isvo_optimize.restype = C.c_int
    lib.isvo_linearize.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t), dp, dp, dp, dp]
    lib.isvo_linearize.restype = C.c_int
    lib.isvo_normal_equations.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t), dp, dp, C.POINTER(C.c_int)]
    lib.isvo_schur_solve.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t), dp, dp]
    lib.isvo_schur_solve.restype = C.c_int
    lib.isvo_cost.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t)]
    lib.isvo_cost.restype = C.c_double
    lib.isvo_init_factor_graph.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t), C.POINTER(abi.isv_summary_t), dp]
    lib.isvo_init_factor_graph.restype = C.c_int
    lib.isvo_triangulate.argtypes = [C.POINTER(abi.isv_config_t), C.POINTER(abi.isv_window_t)]
    lib.isvo_triangulate.restype = C.c_int
    lib.isvo_debug_force_retry.argtypes = [C.c_int]
    lib.isvo_debug_force_retry.restype = None
    lib.isvo_debug_force_invalid.argtypes = [C.c_int]
    lib.isvo_debug_force_invalid.restype = None
    lib.isvo_debug_min_radius.argtypes = [C.c_double]
    lib.isvo_debug_min_radius.restype = None
    for name in dir(abi):
        pass
    _lib = lib
    return lib
