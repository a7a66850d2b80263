import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fvp_synthetic as S
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
def snap(m):
    return {k: v.clone() for k, v in m.engine._scratch.items() if isinstance(k, tuple) and (".buf" in k[0])}
with torch.no_grad():
    fused, planes, centers, heat, _ = model(backbone=bb, views=views, meta=meta, cameras=cams, resize_transform=rt)
    torch.cuda.synchronize()
    ref = snap(model)
    for trial in range(4):
        pipe = FV.PipelinedForward(model, depth=3)
        outs = [pipe.submit(backbone=bb, views=views, meta=meta, cameras=cams, resize_transform=rt) for _ in range(3)]
        pipe.synchronize()
        for i, ((pf, _, pc, ph, _), _) in enumerate(outs):
            if torch.equal(pf, fused): continue
            cur = snap(pipe.models[i])
            bad = [(k[0], tuple(k[1]), float((cur[k]-ref[k]).abs().max()), int((cur[k]!=ref[k]).sum())) for k in sorted(ref, key=lambda k: (k[0].split(".buf")[0], int(k[0].split(".buf")[1]))) if k in cur and not torch.equal(cur[k], ref[k])]
            print("trial", trial, "replica", i, "differing bufs:", bad[:4])
            k = [kk for kk in ref if kk[0]==bad[0][0]][0]
            d = (cur[k]!=ref[k]).nonzero()
            print("   first diffs idx", d[:6].tolist(), "last", d[-3:].tolist())
