#!/usr/bin/env python
"""Pipelined images -> joints vs the plain forward: which outputs differ? (GPU debugging aid; FVP_LIB selects a variant)"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import fvp_synthetic as S
from faster_voxelpose_amd import _capi as capi
import _lib; _lib.select(capi)
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
    torch.cuda.synchronize()
    nbad = 0
    for rep in range(int(os.environ.get('REPS', '10'))):
        pipe = FV.PipelinedForward(model, depth=3)
        outs = [pipe.submit(backbone=bb, views=views, meta=meta, cameras=cams, resize_transform=rt) for _ in range(4)]
        pipe.synchronize(); torch.cuda.synchronize()
        for i, ((pf, pp, pc, ph, _), _) in enumerate(outs):
            nbad += not torch.equal(pf, fused)
            if not torch.equal(pf, fused) or os.environ.get("VERBOSE"): print(rep, i, "fused", torch.equal(pf, fused), float((pf - fused).abs().max()), "planes", torch.equal(pp, planes),
                  "centers", torch.equal(pc, centers), float((pc - centers).abs().max()), "heat", torch.equal(ph, heat), float((ph - heat).abs().max()))
print("mismatching batches:", nbad)
