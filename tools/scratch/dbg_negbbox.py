import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"oracle")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import torch, numpy as np
import fvp_oracle as O, fvp_synthetic as S
from faster_voxelpose_amd.models import faster_voxelpose as FV
dev="cuda:0"
for shape in ("tiny",):
    for variant in ("fresh", "after_forward", "pc_offcentre"):
        cfg=S.make_cfg(shape, device=dev, min_score=-1.0); cams,seq=S.load_cameras(shape); rt=S.resize_transform(cfg)
        m=FV.get(cfg).to(dev); m.load_state_dict(S.fill_state_dict(m.state_dict(), seed=7))
        heat = S.heatmaps_uniform(cfg, 1, seed=3)
        cen = torch.tensor(cfg.CAPTURE_SPEC.SPACE_CENTER)
        pc = torch.zeros(3, 7); pc[:, :3] = cen
        if variant=="pc_offcentre": pc[:, 0] += 333.0; pc[:,1] -= 217.0
        pc[:, 5:7] = torch.tensor([[0.6, 0.6], [-0.5, 0.7], [0.7, -1.5]])
        meta={"seq":[seq]}
        hd, rd = heat.to(dev), rt.to(dev)
        with torch.no_grad():
            if variant=="after_forward":
                m(meta=meta, input_heatmaps=hd, cameras=cams, resize_transform=rd)
            cubes, offset = m.joint_net.project_layer(hd, 0, meta, pc.to(dev), cams, rd)
        spec=O.IndividualSpec(S.make_cfg(shape, min_score=-1.0))
        oc, ooff, (tl,start,end) = O.project_individual(spec, S.make_cfg(shape), heat[0], pc, [cams[seq][i] for i in range(len(cams[seq]))], rt)
        d=(cubes.cpu()[0]-oc[0]).abs()
        print(shape, variant, "max", float(d.max()), "per-channel max", d.amax(dim=(1,2,3)).numpy())
        idx = (d>2.5e-7).nonzero()
        print("  n>2.5e-7", len(idx), "first", idx[:5].tolist())
        # grid check at those voxels
        e=m.engine
        V=len(cams[seq])
        fine = e.sample_grid(e.fine_axes, seq, rd, V).cpu().view(V,*e.fine,2)
        pts=spec.fine_points(torch.zeros(3,dtype=torch.int64), torch.tensor(e.fine))
        go=O.build_sample_grids(pts,[cams[seq][i] for i in range(V)],S.make_cfg(shape),rt).view(V,*e.fine,2)
        print("  full fine grid equal oracle:", torch.equal(fine, go), float((fine-go).abs().max()))
