"""Golden vectors for the image-geometry helpers (build container only): random centre / scale /
rotation / shift / inverse cases through the reference's ``get_affine_transform`` and ``get_scale``
(``lib/utils/transforms.py:15,81``; OpenCV replaced by the float64 three-point solve of
``_refimport``).  Stores inputs and the reference's outputs -> ``transforms.npz``."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport as R  # noqa: E402


def main():
    ref = R.import_reference()
    rng = np.random.default_rng(5)
    n = 64
    center = rng.uniform(0, 1000, (n, 2))
    scale = rng.uniform(0.5, 10, (n, 2)).astype(np.float32)
    rot = rng.uniform(-180, 180, n)
    rot[:8] = 0.0
    out = rng.integers(32, 1000, (n, 2))
    shift = rng.uniform(-0.3, 0.3, (n, 2)).astype(np.float32)
    shift[:16] = 0.0
    inv = rng.integers(0, 2, n)
    affine = np.stack([ref.transforms.get_affine_transform(center[i], scale[i], rot[i], out[i], shift=shift[i],
                                                           inv=int(inv[i])) for i in range(n)])
    sizes = np.array([[1920, 1080, 960, 512], [1032, 776, 800, 608], [360, 288, 800, 640], [100, 300, 64, 64],
                      [300, 100, 64, 64]], np.int64)
    scales = np.stack([ref.transforms.get_scale((a, b), (c, d)) for a, b, c, d in sizes])
    np.savez_compressed(os.path.join(HERE, "transforms.npz"), center=center, scale=scale, rot=rot, out=out,
                        shift=shift, inv=inv, affine=affine, sizes=sizes, scales=scales)
    print("wrote transforms.npz", affine.shape)


if __name__ == "__main__":
    main()
