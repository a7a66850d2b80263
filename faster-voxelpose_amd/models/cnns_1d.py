"""1-D conv stack -- drop-in for ``C2CNet`` of the reference's ``lib/models/cnns_1d.py``
(:112-132): the CenterNet trunk along z, executed by the same HIP conv interpreter with
H = 1 (Conv1d k7/k3/k1, max_pool1d, ConvTranspose1d k2 s2)."""
from ._netmodule import PackedNet


class C2CNet(PackedNet):
    def __init__(self, input_channels, output_channels, head_conv=32, _engine=None):
        super().__init__()
        assert _engine is not None, "C2CNet is built by HumanDetectionNet (needs the column length)"
        assert output_channels == 1
        self.output_channels = output_channels
        _engine.specs["c2c_net"].build_tree(self)
        self._init_packing(_engine, "c2c_net")

    def forward(self, x, _clone=True):
        """x: [n,J,Z] -> [n,1,Z] (cnns_1d.py:128-132)."""
        e = self.engine
        self.ensure_packed()
        e._check_tensor(x, "columns")
        n, J, Z = x.shape
        xin = x.contiguous().view(n, J, 1, Z)
        if e.fused_c2c and Z <= 24:
            out = e.run_stack_fused_1d("c2c_net", xin, n)["out"].view(n, 1, Z)
        else:
            out = e.run_stack("c2c_net", xin, n)["out"].view(n, 1, Z)
        return out.clone() if _clone else out
