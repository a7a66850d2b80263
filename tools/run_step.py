"""Three serial forward passes of the B = 8 Panoptic-shape step (target of the rocprofv3 wrappers in tools/)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fvp_synthetic as S
from faster_voxelpose_amd.models import faster_voxelpose as FV
dev="cuda:0"
B=int(os.environ.get("B","8"))
cfg = S.make_cfg("panoptic", device=dev, min_score=-1.0)
cams, seq = S.load_cameras("panoptic"); rt = S.resize_transform(cfg).to(dev)
heat = S.heatmaps_blobs(cfg, cams, seq, B, people=4, seed=100).to(dev)
meta={"seq":[seq]*B}
model = FV.get(cfg).to(dev); model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
with torch.no_grad():
    for _ in range(3): out = model(meta=meta, input_heatmaps=heat, cameras=cams, resize_transform=rt)
torch.cuda.synchronize()
