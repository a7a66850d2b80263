import sys; sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import fvp_synthetic as S
from faster_voxelpose_amd.models import faster_voxelpose as FV
import fvp_oracle as O
cfg = S.make_cfg("panoptic", device="cuda:0")
model = FV.get(cfg).to("cuda:0")
sd = S.fill_state_dict(model.state_dict(), seed=7); model.load_state_dict(sd)
z = torch.from_numpy(np.random.default_rng(12).random((80, 15, 20), dtype=np.float32)).cuda()
model.engine.fused_c2c = True; f = model.pose_net.c2c_net(z)
model.engine.fused_c2c = False; g = model.pose_net.c2c_net(z)
o = O.c2c_net(sd, "pose_net.c2c_net", z.cpu())
print('fused vs generic max', (f-g).abs().max().item(), 'neq count', (f!=g).sum().item(), 'of', f.numel())
print('fused vs oracle', (f.cpu()-o).abs().max().item(), 'generic vs oracle', (g.cpu()-o).abs().max().item())
