import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """CPU emulation of the HIP kernels (tests/hipemu) -- kernel-logic tests only."""
    import ctypes

    from faster_voxelpose_amd import _capi as capi
    os.environ.setdefault("FVP_WINO_GENERIC", "1")    # read once when the library loads: lets one test cover masked tiles
    here = os.path.join(ROOT, "tests", "hipemu")
    subprocess.run([os.path.join(here, "build_emu.sh")], check=True, capture_output=True)
    return capi.bind(ctypes.CDLL(os.path.join(here, "libfvp_emu.so")))
