import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import fvp_synthetic as S
from faster_voxelpose_amd.models import faster_voxelpose as FV
B=int(os.environ.get("B","1"))
cfg = S.make_cfg("panoptic", device="cuda:0", min_score=-1.0)
cams, seq = S.load_cameras("panoptic")
rt = S.resize_transform(cfg).cuda()
model = FV.get(cfg).to("cuda:0")
model.load_state_dict(S.fill_state_dict(model.state_dict(), seed=7))
heats = [S.heatmaps_blobs(cfg, cams, seq, B, people=3, seed=40 + i).cuda() for i in range(3)]
meta = {"seq": [seq]*B}
bad=0
with torch.no_grad():
    want=[]
    for h in heats:
        f,p,c,_,_ = model(meta=meta, input_heatmaps=h, cameras=cams, resize_transform=rt); want.append((f.clone(), c.clone()))
    torch.cuda.synchronize()
    pipe = FV.PipelinedForward(model, depth=3)
    for trial in range(6):
        outs = [pipe.submit(meta=meta, input_heatmaps=heats[i%3], cameras=cams, resize_transform=rt) for i in range(6)]
        pipe.synchronize()
        for i, ((pf, _, pc, _, _), _) in enumerate(outs):
            if not (torch.equal(pf, want[i%3][0]) and torch.equal(pc, want[i%3][1])): bad+=1
print("B", B, "mismatches", bad, "of 36")
