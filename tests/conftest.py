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
    from faster_voxelpose_amd import netspec
    # the emulated library is a diagnostics build (-DFVP_DIAG=1): it honours the FVP_* switches.  Masked Winograd tiles
    # (rows that do not divide the workgroup tile) are switched on for the whole emulated session so that one test covers
    # them: read once when the library loads, mirrored on the host side by netspec.WINO_GENERIC (weight-blob layout).
    os.environ.setdefault("FVP_WINO_GENERIC", "1")
    # the register-direct 1x1 / transposed-conv kernel also takes the few planes of the emulated cases (the product keeps
    # k_conv_dma below 1024 (1x1) / 200 (transposed) tiles; the two kernels produce the same bits: test_register_direct_conv_equals_the_staged_kernel)
    os.environ.setdefault("FVP_CONV_REG_MIN_TILES", "1")
    netspec.WINO_GENERIC = True
    here = os.path.join(ROOT, "tests", "hipemu")
    subprocess.run([os.path.join(here, "build_emu.sh")], check=True, capture_output=True)
    return capi.bind(ctypes.CDLL(os.path.join(here, "libfvp_emu.so")))


def _diag_path():
    return os.path.join(ROOT, "tests", "diag", "libfvp_hip_diag.so")


@pytest.fixture(scope="session")
def diag_lib():
    """The diagnostics build of the HIP library (tests/diag/build_diag.sh, -DFVP_DIAG=1) for GPU tests that flip
    kernel-selection switches through the environment - the shipped library ignores the environment."""
    import ctypes

    import torch  # noqa: F401  (its HIP runtime first: see _capi.load)
    from faster_voxelpose_amd import _capi as capi
    marker = os.path.join(ROOT, "tests", "diag", "BUILD_FAILED")
    if os.path.isfile(marker):          # written by __graft_entry__.build() when the -DFVP_DIAG=1 build broke
        pytest.fail("the diagnostics library did not build in build():\n" + open(marker).read()[-2000:])
    if not os.path.isfile(_diag_path()):
        subprocess.run([os.path.join(ROOT, "tests", "diag", "build_diag.sh")], check=True, capture_output=True)
    lib = capi.bind(ctypes.CDLL(_diag_path()))
    assert lib.fvp_diag_build() == 1
    return lib


def pytest_sessionstart(session):
    # tools/gpu_switch_matrix.sh: the whole GPU suite on the diagnostics build, so that the FVP_* switches it sets act
    if os.environ.get("FVP_TEST_DIAG_LIB") == "1":
        from faster_voxelpose_amd import _capi as capi
        capi.LIB_PATH = _diag_path()
        if os.environ.get("FVP_WINO_GENERIC") == "1":    # host mirror of the library's switch (weight-blob layout)
            from faster_voxelpose_amd import netspec
            netspec.WINO_GENERIC = True
