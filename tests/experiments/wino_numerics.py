"""Experiment (test infrastructure, not collected by pytest): does a Winograd F(2x2,3x3) fp32 evaluation of the 3x3 convs stay inside the
parity criteria of tests/common.py?  Patches the ORACLE's conv (test infrastructure) only."""
import sys, os
import numpy as np, torch
import torch.nn.functional as F
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "oracle"), os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import fvp_oracle as O
from cases import CASES, make_inputs
from common import load_golden, joint_errors
import fvp_synthetic as S

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)

def wino3x3(x, w, b):
    N, C, H, W = x.shape
    K = w.shape[0]
    U = torch.einsum("ai,kcij,bj->abkc", G, w, G)                 # [4,4,K,C]
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                          # [N,C,th,tw,4,4]
    V = torch.einsum("ai,nchwij,bj->abnchw", Bt, t, Bt)
    M = torch.einsum("abkc,abnchw->abnkhw", U, V)
    Y = torch.einsum("ia,abnkhw,jb->nkhiwj", At, M, At)
    y = Y.reshape(N, K, Hp, Wp)[:, :, :H, :W]
    return y + b.view(1, -1, 1, 1)

orig = O._convnd
def patched(x, w, b, dim, pad):
    if dim == 2 and w.shape[-1] == 3 and x.dtype == torch.float32:
        return wino3x3(x, w, b)
    return orig(x, w, b, dim, pad)

for case in sys.argv[1:] or ["panoptic_g_b1_all", "tiny_g_b2_all", "shelf_g_b1_all"]:
    cfg, cams, seq, rt, heat, meta, wseed = make_inputs(case)
    g = load_golden(case)
    sd = S.fill_state_dict(O.reference_state_dict_shapes(cfg), seed=wseed)
    res = {}
    for name, fn in (("direct", orig), ("winograd", patched)):
        O._convnd = fn
        orc = O.Oracle(cfg, sd)
        fused, planes, centers = orc.forward(heat, meta, cams, rt)
        e32, e64, floor = joint_errors(fused, g)
        v = g["valid"]
        fe = 0.0
        for f in range(v.shape[0]):
            if f"jl{f}_feat" in g:
                feat = orc.trace["jln"][f].get("feat")
        res[name] = (e32.max(), e64.max(), floor)
        same_topk = np.array_equal((orc.trace["topk_index"].numpy()[..., 0] * cfg.CAPTURE_SPEC.VOXELS_PER_AXIS[0] + orc.trace["topk_index"].numpy()[..., 1]), g["topk_flat"])
        print(case, name, "e32 %.2e e64 %.2e floor %.2e topk_same %s hm2d_err %.2e" % (e32.max(), e64.max(), floor, same_topk, np.abs(orc.trace["hm2d"][:, 0].numpy() - g["hm2d"]).max()))
