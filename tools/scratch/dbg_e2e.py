import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fvp_synthetic as S
from faster_voxelpose_amd import _capi as _c
if os.environ.get("FVP_LIB"): _c.LIB_PATH = os.path.abspath(os.environ["FVP_LIB"])
from faster_voxelpose_amd.core import config as CFG
from faster_voxelpose_amd.models import faster_voxelpose as FV, resnet as RN
cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
cams, seq = S.load_cameras("panoptic")
rt = S.resize_transform(cfg).cuda()
model = FV.get(cfg).to("cuda:0")
model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
bb = RN.get(CFG.default_config()).to("cuda:0")
bb.load_state_dict(S.fill_backbone_state_dict(bb.state_dict(), seed=3))
W, H = cfg.DATASET.IMAGE_SIZE
views = torch.rand(1, 5, 3, H, W, device="cuda")
meta = {"seq": [seq]}
with torch.no_grad():
    fused, planes, centers, heat, _ = model(backbone=bb, views=views, meta=meta, cameras=cams, resize_transform=rt)
    tri0 = model.engine.last_jln["planes"].clone(); feat0 = model.engine.last_jln["feat"].clone(); cubes0 = model.engine.last["cubes"].clone()
    torch.cuda.synchronize()
    for trial in range(3):
        pipe = FV.PipelinedForward(model, depth=3)
        outs = [pipe.submit(backbone=bb, views=views, meta=meta, cameras=cams, resize_transform=rt) for _ in range(4)]
        pipe.synchronize()
        for i, ((pf, _, pc, ph, _), _) in enumerate(outs):
            m = pipe.models[i % 3]
            print(trial, i, "fused", torch.equal(pf, fused), float((pf-fused).abs().max()), "centers", torch.equal(pc, centers), "heat", torch.equal(ph, heat),
                  "tri", torch.equal(m.engine.last_jln["planes"], tri0), "cubes", torch.equal(m.engine.last["cubes"], cubes0), "feat", torch.equal(m.engine.last_jln["feat"], feat0))
