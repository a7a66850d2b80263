"""Library selection shared by the diagnostics tools: FVP_LIB=<path> loads a variant built by tools/build_variant.sh;
otherwise, when any FVP_* kernel switch is set in the environment, the diagnostics build (tests/diag/libfvp_hip_diag.so)
is loaded - the shipped libfvp_hip.so ignores the environment."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_KNOBS = ("FVP_CONV_", "FVP_WINO_", "FVP_TRI", "FVP_C1D_", "FVP_BB_", "FVP_WHOLE_")


def select(capi):
    diag = os.path.join(ROOT, "tests", "diag", "libfvp_hip_diag.so")
    if os.environ.get("FVP_LIB"):
        capi.LIB_PATH = os.path.abspath(os.environ["FVP_LIB"])
    elif any(k.startswith(_KNOBS) for k in os.environ) and os.path.isfile(diag):
        capi.LIB_PATH = diag
    if os.environ.get("FVP_WINO_GENERIC") == "1":     # host mirror of the library's switch (weight-blob layout)
        from faster_voxelpose_amd import netspec
        netspec.WINO_GENERIC = True
    return capi.LIB_PATH
