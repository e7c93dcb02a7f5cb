import os
import sys

import pytest

try:  # torch (test plumbing only) must load its HIP runtime before libs360.so does, or torch.cuda finds no device
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RIG_JSON = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")

# Developer switch, tests only: S360_TEST_EMULATED_LIB=1 points the Python binding of THIS TEST PROCESS at
# tools/libs360_emu.so (the library's sources compiled for the CPU over tools/hip_wave_shim, see
# tests/test_cpu_library_emulation.py), so that `pytest -m gpu -k "not fullsize"` can be tried where no GPU is attached.
# The product never looks at this variable.
if os.environ.get("S360_TEST_EMULATED_LIB") == "1":
    from surround360_amd import _capi as _capi_for_emulation
    # (S360_TEST_EMULATED_LIB_PATH: another build of the emulated library, e.g. tools/fuzz/libs360_asan.so)
    _capi_for_emulation.LIB_PATH = os.environ.get("S360_TEST_EMULATED_LIB_PATH") or os.path.join(ROOT, "tools", "libs360_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fullsize: the subset of the gpu tests that runs at the BASELINE.json sizes "
                                       "(2048^2 pair, full 8K frame); minutes of oracle time on the host cores")


@pytest.fixture(scope="session")
def rig_json():
    return RIG_JSON


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def s360lib():
    """Loads libs360.so (building it if the sources are newer)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "surround360_amd", "csrc"), "-j8", "-s"])
    from surround360_amd import _capi
    return _capi.lib()


def _have_gpu():
    try:
        from surround360_amd import _capi
        return _capi.lib().s360_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_rig(s360lib, rig_json):
    from surround360_amd import render as R
    if s360lib.s360_device_count() <= 0:
        pytest.fail("no HIP device: -m gpu tests must run on the GPU box (there is no CPU fallback)")
    return R.RigDescription(rig_json)
