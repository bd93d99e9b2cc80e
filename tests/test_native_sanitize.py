"""AddressSanitizer + UBSan run of the window manager's host C++ (is-vins_amd/csrc/isv_estimator.cpp) on the CPU, with a
stand-in solver behind the test seam (tests/native/estimator_sanitize.cpp): 19 sequences x 120 frames of tracks being
born, ageing and dying through both slideWindow branches.  (GPU sanitizers are not available on the pool; the host
logic is where the dynamic containers are.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_window_manager_under_asan_ubsan(tmp_path, monkeypatch):
    exe = tmp_path / "estimator_sanitize"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                           os.path.join(ROOT, "tests", "native", "estimator_sanitize.cpp"),
                           os.path.join(ROOT, "is-vins_amd", "csrc", "isv_estimator.cpp"), "-o", str(exe)])
    for threads in ("1", "4"):
        env = dict(os.environ, ISV_HOST_THREADS=threads, ASAN_OPTIONS="detect_leaks=1")
        env.pop("LD_PRELOAD", None)
        out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.startswith("ok:")
