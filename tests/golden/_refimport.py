"""Import the read-only reference (``/root/reference/lib``) in the build container.

Only the golden-vector generator (``make_golden.py``) and ad-hoc pinning runs use this.
Nothing under ``tests/`` that runs on the GPU box may import it: ``/root/reference`` does
not exist there.  The reference's ``utils/transforms.py`` imports OpenCV, which this image
lacks; only ``cv2.getAffineTransform`` is reachable from the hot path, so a float64
three-point solve stands in for it.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "models"))


def _cv2_stub():
    m = types.ModuleType("cv2")

    def getAffineTransform(src, dst):
        a = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], axis=1)
        return np.linalg.solve(a, np.asarray(dst, np.float64)).T.copy()

    m.getAffineTransform = getAffineTransform
    return m


def import_reference():
    """Returns the reference's ``models`` package (faster_voxelpose, project_*, ...)."""
    if not available():
        raise RuntimeError("reference checkout not present; goldens can only be regenerated "
                           "in the build container")
    sys.modules.setdefault("cv2", _cv2_stub())
    lib = os.path.join(REF_ROOT, "lib")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k.split(".")[0] in ("models", "core", "utils")}
    sys.path.insert(0, lib)
    try:
        import models  # noqa: F401
        import models.faster_voxelpose  # noqa: F401
        import core.proposal  # noqa: F401
        ref = {k: v for k, v in sys.modules.items()
               if k.split(".")[0] in ("models", "core", "utils")}
    finally:
        sys.path.remove(lib)
        for k in list(sys.modules):
            if k.split(".")[0] in ("models", "core", "utils"):
                del sys.modules[k]
        sys.modules.update(saved)
    return types.SimpleNamespace(
        models=ref["models"],
        faster_voxelpose=ref["models.faster_voxelpose"],
        project_whole=ref["models.project_whole"],
        project_individual=ref["models.project_individual"],
        human_detection_net=ref["models.human_detection_net"],
        joint_localization_net=ref["models.joint_localization_net"],
        cnns_2d=ref["models.cnns_2d"],
        cnns_1d=ref["models.cnns_1d"],
        weight_net=ref["models.weight_net"],
        proposal=ref["core.proposal"],
        cameras=ref["utils.cameras"],
        transforms=ref["utils.transforms"],
    )


@contextlib.contextmanager
def quiet():
    """The reference prints on every grid build; keep generator logs readable."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield
