"""Golden vectors for the input-heatmap rasteriser, from the reference itself (build container).

    python tests/golden/make_golden_heatmaps.py

Calls ``JointsDataset.generate_input_heatmap`` / ``utils.transforms.affine_transform`` of
/root/reference on an instance created without ``__init__`` (only the attributes those two
methods read are set) and stores the resulting [V,J,H,W] float32 heatmaps (compressed; they are
mostly zeros).  Inputs are regenerated from the recipe in ``heatmap_cases.py``."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]

import _refimport as R  # noqa: E402
from heatmap_cases import HEATMAP_CASES, make_pred2d  # noqa: E402


def main():
    R.import_reference()                       # installs the cv2 stand-in
    sys.path.insert(0, os.path.join(R.REF_ROOT, "lib"))
    # load the module file directly: the package's __init__ pulls in dataset readers with
    # dependencies (json_tricks) this image does not have
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_joints_dataset", os.path.join(R.REF_ROOT, "lib", "dataset", "JointsDataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    JointsDataset = mod.JointsDataset
    from utils.transforms import affine_transform
    for case in HEATMAP_CASES:
        cfg, all_preds, rt, sigma = make_pred2d(case)
        ds = object.__new__(JointsDataset)
        ds.image_size = np.array(cfg.DATASET.IMAGE_SIZE)
        ds.heatmap_size = np.array(cfg.DATASET.HEATMAP_SIZE)
        ds.sigma = sigma
        ds.data_augmentation = False
        out = []
        for preds in all_preds:                # JointsDataset.__getitem__ :144-154
            preds = [p.copy() for p in preds]
            for n in range(len(preds)):
                for i in range(len(preds[n])):
                    preds[n][i, :2] = affine_transform(preds[n][i, :2], rt)
            if len(preds) == 0:
                out.append(np.zeros((cfg.DATASET.NUM_JOINTS, ds.heatmap_size[1], ds.heatmap_size[0]), np.float32))
            else:
                out.append(ds.generate_input_heatmap(preds))
        hm = np.stack(out).astype(np.float32)
        path = os.path.join(HERE, case + ".npz")
        np.savez_compressed(path, heatmaps=hm, nonzero=np.int64((hm > 0).sum()), total=np.float64(hm.astype(np.float64).sum()))
        print(f"{case}: {hm.shape} nonzero {(hm > 0).sum()} max {hm.max():.4f} -> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
