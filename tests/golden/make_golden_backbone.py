"""Golden vectors for the Pose-ResNet-50 backbone from the reference's own ``models/resnet.py``
(build container only): seeded weights (``synthetic.fill_backbone_state_dict``), a seeded
[2,3,96,128] image batch, the reference's fp32 heatmaps [2,15,24,32]."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]
import _refimport as R  # noqa: E402
import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd.core import config as CFG  # noqa: E402

SHAPE, WSEED, XSEED = (2, 3, 96, 128), 3, 5


def inputs():
    return torch.from_numpy(np.random.default_rng(XSEED).random(SHAPE, dtype=np.float32))


def main():
    R.import_reference()
    spec = importlib.util.spec_from_file_location("ref_resnet", os.path.join(R.REF_ROOT, "lib", "models", "resnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = CFG.default_config()
    m = mod.get(cfg).eval()
    sd = S.fill_backbone_state_dict(m.state_dict(), seed=WSEED)
    m.load_state_dict(sd)
    with torch.no_grad():
        y = m(inputs())
    np.savez_compressed(os.path.join(HERE, "backbone_r50.npz"), heatmaps=y.numpy(),
                        keys=np.array(list(sd)), nparams=np.int64(sum(v.numel() for v in sd.values())))
    print(y.shape, float(y.abs().max()), float(y.std()), len(sd))


if __name__ == "__main__":
    main()
