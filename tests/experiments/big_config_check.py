"""BASELINE.json configs[3] shape (Panoptic 5-view, 128x128x32 voxels, jln128) on a GPU box:
HIP path vs CPU oracle on the same seeded inputs.  Diagnostics; the committed regression test is
tests/test_gpu_parity.py::test_jln128_config_vs_oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch  # noqa: E402

import fvp_oracle as O  # noqa: E402
import fvp_synthetic as S  # noqa: E402
from faster_voxelpose_amd.models import faster_voxelpose as FV  # noqa: E402

vox, cube = [128, 128, 32], [128, 128, 128]
people = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = S.make_cfg("panoptic", device="cuda:0", min_score=0.3, voxels=vox, cube=cube, max_people=people)
cams, seq = S.load_cameras("panoptic")
rt = S.resize_transform(cfg)
heat = S.heatmaps_blobs(cfg, cams, seq, 1, people=people, seed=11)
meta = {"seq": [seq]}
model = FV.get(cfg).to("cuda:0")
sd = S.fill_state_dict(model.state_dict(), seed=7)
model.load_state_dict(sd)
with torch.no_grad():
    fused, planes, centers, _, _ = model(meta=meta, input_heatmaps=heat.cuda(), cameras=cams, resize_transform=rt.cuda())
torch.cuda.synchronize()
print("gpu ok", fused.shape, "valid", int((centers[..., 3] >= 0).sum()))
t0 = time.time()
cfg_cpu = S.make_cfg("panoptic", device="cpu", min_score=0.3, voxels=vox, cube=cube, max_people=people)
of, op, oc = O.Oracle(cfg_cpu, sd).forward(heat, meta, cams, rt)
print("oracle s", time.time() - t0)
c = centers.cpu()
print("centres equal", torch.equal(c[..., :3], oc[..., :3]), "valid equal", torch.equal(c[..., 3], oc[..., 3]))
v = oc[..., 3] >= 0
err = (fused.cpu()[..., :3] - of[..., :3]).norm(dim=-1)[v]
print("max joint err mm", err.max().item() if err.numel() else None)
