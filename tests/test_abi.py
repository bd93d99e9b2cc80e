"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/isvins_backend.h declares, and the ctypes mirror has the header's struct sizes.
No compute call is made here (no GPU in this container)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from isvins_amd import abi, backend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "isvins_backend.h")
HEADER_EST = os.path.join(ROOT, "include", "isvins_estimator.h")


@pytest.fixture(scope="module")
def lib():
    backend.build()
    return backend.load_library()


def declared_functions(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(isv_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert "isv_backend_optimize" in names and "isv_batch_upload" in names and len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/isvins_backend.h but not exported"
    assert set(names) == set(backend.EXPORTS)
    assert lib.isv_abi_version() == 2


def test_every_estimator_symbol_is_exported(lib):
    from isvins_amd import estimator
    names = [n for n in declared_functions(HEADER_EST) if n.startswith("isv_estimator_")]
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/isvins_estimator.h but not exported"
    assert set(names) == set(estimator.EXPORTS)


def test_estimator_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from isvins_amd import estimator, synth
    p = estimator.make_params(abi.make_config(11, 5), synth.RIC, synth.TIC, 0.2, 0.004, 0.001, 0.0001, 10.0 / 460.0)
    with pytest.raises(backend.BackendError):
        estimator.SequenceEstimator(p, 2)


def test_struct_sizes_match_header(tmp_path):
    """compile a tiny C program against the header and compare sizeof with the ctypes mirror"""
    names = ["isv_config_t", "isv_imu_t", "isv_se3_prior_t", "isv_linear9_t", "isv_relpose_t", "isv_rollpitch_t",
             "isv_window_t", "isv_summary_t", "isv_combined_factors_t", "isv_marg_result_t"]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "isvins_backend.h"\nint main(){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in names) + "return 0;}")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    for n, s in zip(names, sizes):
        assert C.sizeof(getattr(abi, n)) == s, n


def test_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(backend.BackendError):
        backend.Backend(11, 5)


def test_create_rejects_bad_config(lib):
    cfg = abi.make_config(11, 5)
    h = C.c_void_p()
    cfg.n_frames = 2
    assert lib.isv_backend_create(C.byref(cfg), C.byref(h)) == -1
    cfg = abi.make_config(11, 5); cfg.estimate_extrinsic = 2        # (2 = online initial calibration: initial/, out of scope)
    assert lib.isv_backend_create(C.byref(cfg), C.byref(h)) == -5
    assert lib.isv_backend_create(None, C.byref(h)) == -1


def test_every_posegraph_symbol_is_exported_and_sized(lib, tmp_path):
    """include/isvins_posegraph.h: every declared entry point is exported, the ctypes mirror has the header's struct sizes,
    and the optimiser fails loudly without a GPU (no CPU path)"""
    from isvins_amd import posegraph as pg
    header = os.path.join(ROOT, "include", "isvins_posegraph.h")
    names = [n for n in declared_functions(header) if n.startswith("isv_pgo_") or n.startswith("isv_combined_")]
    assert set(names) == set(pg.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/isvins_posegraph.h but not exported"
    structs = ["isv_pg_keyframe_t", "isv_pgo_config_t", "isv_pgo_result_t"]
    src = tmp_path / "szp.c"
    src.write_text('#include <stdio.h>\n#include "isvins_posegraph.h"\nint main(){' +
                   "".join(f'printf("%zu\\n", sizeof({n}));' for n in structs) + "return 0;}")
    exe = tmp_path / "szp"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    for n, s in zip(structs, [int(x) for x in subprocess.check_output([str(exe)]).split()]):
        assert C.sizeof(getattr(pg, n)) == s, n
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(backend.BackendError):
            pg.PoseGraphOptimizer(64)
