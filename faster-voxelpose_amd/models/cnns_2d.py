"""2-D conv stacks -- drop-ins for ``CenterNet`` / ``P2PNet`` of the reference's
``lib/models/cnns_2d.py`` (:147-178, :115-135).

The modules hold parameters only (identical ``state_dict`` keys and shapes: front_layers,
encoder_decoder.*, output_*); ``forward`` runs the whole stack through the HIP interpreter
``fvp_conv_stack_run`` (fp32 MFMA implicit GEMMs with fused BN / bias / ReLU / residual).
"""
import ctypes as C

import torch

from ..engine import HotPath, _ptr
from ._netmodule import PackedNet


class CenterNet(PackedNet):
    def __init__(self, input_channels, output_channels, head_conv=32, _engine=None):
        super().__init__()
        assert _engine is not None, "CenterNet is built by HumanDetectionNet (needs the volume size)"
        assert output_channels == 1 and head_conv == 32
        self.output_channels = output_channels
        _engine.specs["center_net"].build_tree(self)
        self._init_packing(_engine, "center_net")

    def forward(self, x):
        """x: cubes [B,J,X,Y,Z] -> (hm [B,1,X,Y], size [B,2,X,Y]) (cnns_2d.py:173-178)."""
        e = self.engine
        self.ensure_packed()
        e._check_tensor(x, "cubes")
        x = x.contiguous()
        B, J, X, Y, Z = x.shape
        zmax = e.scratch("zmax", (B, J, X, Y))
        e._call("fvp_zmax", _ptr(x), _ptr(zmax), C.c_long(B * J * X * Y), Z, e.stream())
        return self.forward_zmax(zmax)

    def forward_zmax(self, zmax):
        self.ensure_packed()
        heads = self.engine.run_stack("center_net", zmax, zmax.shape[0])
        return heads["output_hm"].clone(), heads["output_size"].clone()


class P2PNet(PackedNet):
    def __init__(self, input_channels, output_channels, _engine=None):
        super().__init__()
        assert _engine is not None, "P2PNet is built by JointLocalizationNet (needs the cube size)"
        self.output_channels = output_channels
        _engine.specs["conv_net"].build_tree(self)
        self._init_packing(_engine, "conv_net")

    def forward(self, x, plane_valid=None, valid_div=1, _clone=True):
        """x: [n,J,C,C] -> [n,J,C,C] (cnns_2d.py:131-135)."""
        e = self.engine
        self.ensure_packed()
        e._check_tensor(x, "planes")
        out = e.run_stack("conv_net", x.contiguous(), x.shape[0], plane_valid, valid_div)["out"]
        return out.clone() if _clone else out
