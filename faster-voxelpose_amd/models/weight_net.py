"""WeightNet -- parameter holder with the reference's ``state_dict`` keys
(``lib/models/weight_net.py:48-67``).  Its arithmetic (conv 1->F k3, BN, max-pool, ReLU,
global average, MLP, sigmoid; :69-80) runs fused with the soft-argmax in the HIP kernel
``fvp_softargmax_weightnet``, launched by ``JointLocalizationNet.forward`` (or standalone by
``WeightNet.forward``)."""
import torch

from ._netmodule import PackedNet


class WeightNet(PackedNet):
    def __init__(self, cfg, _engine=None):
        super().__init__()
        assert _engine is not None, "WeightNet is built by JointLocalizationNet"
        self.voxels_per_axis = cfg.INDIVIDUAL_SPEC.VOXELS_PER_AXIS
        self.num_joints = cfg.DATASET.NUM_JOINTS
        self.num_channel_joint_feat = F = cfg.NETWORK.NUM_CHANNEL_JOINT_FEAT
        self.num_channel_joint_hidden = Hd = cfg.NETWORK.NUM_CHANNEL_JOINT_HIDDEN
        self.add("heatmap_feature_net.0.weight", torch.zeros(F, 1, 3, 3))
        self.add("heatmap_feature_net.0.bias", torch.zeros(F))
        self.add("heatmap_feature_net.1.weight", torch.ones(F))
        self.add("heatmap_feature_net.1.bias", torch.zeros(F))
        self.add("heatmap_feature_net.1.running_mean", torch.zeros(F), buffer=True)
        self.add("heatmap_feature_net.1.running_var", torch.ones(F), buffer=True)
        self.add("heatmap_feature_net.1.num_batches_tracked", torch.zeros((), dtype=torch.long), buffer=True)
        self.add("output.0.weight", torch.zeros(Hd, F))
        self.add("output.0.bias", torch.zeros(Hd))
        self.add("output.2.weight", torch.zeros(1, Hd))
        self.add("output.2.bias", torch.zeros(1))
        self._init_packing(_engine, "weight_net")

    def _pack(self):
        self.engine.pack_weightnet(self)

    def forward(self, x):
        """x = joint features [3, P, J, C, C] -> fusion weights [3P, J, 1] (weight_net.py:69-80).
        Standalone launch of the same fused kernel JointLocalizationNet.forward uses."""
        self.ensure_packed()
        return self.engine.softargmax_weightnet(x)[2]
