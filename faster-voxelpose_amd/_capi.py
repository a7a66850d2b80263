"""ctypes binding of the C ABI in ``include/fvp.h`` (``libfvp_hip.so``).

The HIP library is the product: there is no CPU or PyTorch fallback.  ``load()`` raises if
the shared object has not been built (``python __graft_entry__.py build`` or
``faster-voxelpose_amd/csrc/build.sh``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libfvp_hip.so"
ABI_VERSION = 8            # include/fvp.h FVP_ABI_VERSION
LIB_PATH = os.path.join(_HERE, LIB_NAME)

FVP_CAM_FLOATS = 24
FVP_MAX_VIEWS = 8
FVP_MAX_JOINTS = 32

OP_CONV, OP_POOL2, OP_CONVT2 = 0, 1, 2
EPI_RELU, EPI_RES, EPI_RES_AFTER_RELU = 1, 2, 4
K_PROJECT_WHOLE, K_PROJECT_TRIPLANE, K_CONV, K_SOFTARGMAX, K_OTHER, K_CONV_WINO, K_BACKBONE, K_CONV_WINO_SMALL, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7, 8
BB_CONV, BB_MAXPOOL, BB_DECONV = 0, 1, 2
BB_OUT_HEAT = 8
BB_STEM = 16


class FvpGeom(C.Structure):
    _fields_ = [("clamp_max", C.c_float), ("rt", C.c_float * 6), ("hm_w", C.c_float), ("hm_h", C.c_float),
                ("img_w", C.c_float), ("img_h", C.c_float), ("W", C.c_int32), ("H", C.c_int32),
                ("V", C.c_int32), ("J", C.c_int32), ("JP", C.c_int32)]


class FvpConvOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32), ("res", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32), ("flags", C.c_int32), ("w_off", C.c_int32),
                ("e_off", C.c_int32), ("cinp", C.c_int32), ("coutp", C.c_int32), ("wino_off", C.c_int32), ("pair_off", C.c_int32)]


class FvpBbOp(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("kind", "src", "dst", "res", "cin", "cinp", "cout", "coutp", "cbuf", "kh", "kw",
                                         "stride", "pad", "h", "w", "oh", "ow", "flags", "w_off", "e_off")]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_G = C.POINTER(FvpGeom)

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/fvp.h 1:1
SIGNATURES = {
    "fvp_version": [],
    "fvp_diag_build": [],
    "fvp_sizeof": [_I],
    "fvp_error_string": [_I],
    "fvp_heatmaps_to_cl": [_P, _P, _I, _G, _P],
    "fvp_sample_grid": [_P, _P, _P, _I, _I, _I, _P, _G, _P, _P],
    "fvp_project_whole": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _G, _P, _P, _P],
    "fvp_project_columns": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _G, _P, _I, _P, _P],
    "fvp_zmax": [_P, _P, C.c_long, _I, _P],
    "fvp_person_boxes": [_P, _I, _P, _P, _P, _P, _P],
    "fvp_project_individual": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _G, _P, _P],
    "fvp_triplane_max": [_P, _P, _I, _I, _I, _P],
    "fvp_project_individual_triplane": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _G, _P, _I, _P, _P],
    "fvp_conv_stack_run": [C.POINTER(FvpConvOp), _I, _P, C.POINTER(_P), _I, _I, _P, _I, _P],
    "fvp_conv_stack_run_fused_1d": [C.POINTER(FvpConvOp), _I, _P, _P, _P, _I, _P],
    "fvp_pack_conv": [_P, _P, _P, _P, _P, _P, _F, _I, C.POINTER(FvpConvOp), _P, _P],
    "fvp_nms_topk": [_P, _I, _I, _I, _I, _P, _P, _P, _P],
    "fvp_gather_proposals": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "fvp_proposals": [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P],
    "fvp_proposal_layer": [_P, _P, _P, _P, _F, _I, _I, _P, _P],
    "fvp_softargmax_weightnet": [_P, _P, _P, _F, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "fvp_pack_weightnet": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _P, _P],
    "fvp_fuse_poses": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "fvp_rasterise_heatmaps": [_P, _P, _I, _I, _I, _I, _I, C.c_double, C.c_double, C.c_double, _P, _P, _I, _P],
    "fvp_bb_input": [_P, _P, _I, _I, _I, _I, _P],
    "fvp_bb_pack": [_P, _P, _P, _P, _P, _P, _F, C.POINTER(FvpBbOp), _P, _P, _P],
    "fvp_bb_run": [C.POINTER(FvpBbOp), _I, _P, _P, C.POINTER(_P), _I, _I, _P, _I, _P, _P],
    "fvp_bb_tune": [C.POINTER(FvpBbOp), _I, _P, _P, C.POINTER(_P), _I, _I, _P],
    "fvp_prof_enable": [_I],
    "fvp_prof_read": [_I, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)],
    "fvp_prof_reset": [],
}
_RESTYPES = {"fvp_error_string": C.c_char_p}


class FvpError(RuntimeError):
    pass


def bind(lib):
    """Attach argtypes/restypes for every symbol ``include/fvp.h`` declares."""
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)                 # AttributeError here = symbol missing from the .so
        fn.argtypes = args
        fn.restype = _RESTYPES.get(name, C.c_int)
    return lib


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise FvpError(
                f"{LIB_PATH} not found: the HIP extension is required (there is no CPU fallback). "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`faster-voxelpose_amd/csrc/build.sh`.")
        # PyTorch ships its own HIP runtime; it has to be the one already in the process when the library (which
        # needs libamdhip64 by soname) is loaded - with the system copy loaded first, the two runtimes both try to
        # own the device and every launch fails with "no ROCm-capable device"
        import torch  # noqa: F401
        _lib = bind(C.CDLL(LIB_PATH))
        if _lib.fvp_version() != ABI_VERSION:
            raise FvpError("libfvp_hip.so ABI version mismatch (rebuild with faster-voxelpose_amd/csrc/build.sh)")
        if _lib.fvp_sizeof(0) != C.sizeof(FvpGeom) or _lib.fvp_sizeof(1) != C.sizeof(FvpConvOp) \
                or _lib.fvp_sizeof(2) != C.sizeof(FvpBbOp):
            raise FvpError("libfvp_hip.so struct layout differs from the ctypes mirrors in _capi.py")
    return _lib


def check(lib, rc, what=""):
    if rc != 0:
        msg = lib.fvp_error_string(rc)
        raise FvpError(f"{what}: error {rc}: {msg.decode() if msg else '?'}")
