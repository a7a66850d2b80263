"""Experiment configuration: the default tree and the YAML overlay of the reference
(``lib/core/config.py``: defaults :11-147, ``update_config`` / ``_update_dict`` :150-185).

Same rules as the reference: a YAML file may only set keys that exist in the default tree -- an
unknown key inside a known section raises ``ValueError("SECTION.KEY not exist in config.py")`` --
unknown top-level sections are rejected the same way, size-like entries are held as numpy arrays.
The tree is an attribute-style mapping, so ``cfg.DATASET.NUM_JOINTS`` and
``cfg['DATASET']['NUM_JOINTS']`` both work (the reference uses EasyDict; not available here).
"""
import copy

import numpy as np
import yaml


class Node(dict):
    """dict with attribute access (nested dicts are converted on assignment)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Node(v) if isinstance(v, dict) and not isinstance(v, Node) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__

    def __deepcopy__(self, memo):
        return Node({k: copy.deepcopy(v, memo) for k, v in self.items()})


_ARRAYS = {("DATASET", "ORI_IMAGE_SIZE"), ("DATASET", "IMAGE_SIZE"), ("DATASET", "HEATMAP_SIZE"),
           ("CAPTURE_SPEC", "SPACE_SIZE"), ("CAPTURE_SPEC", "SPACE_CENTER"), ("CAPTURE_SPEC", "VOXELS_PER_AXIS"),
           ("INDIVIDUAL_SPEC", "SPACE_SIZE"), ("INDIVIDUAL_SPEC", "VOXELS_PER_AXIS")}


def _hrnet_stage(modules, branches, blocks, channels):
    return dict(NUM_MODULES=modules, NUM_BRANCHES=branches, BLOCK="BASIC", NUM_BLOCKS=blocks, NUM_CHANNELS=channels,
                FUSE_METHOD="SUM")


def default_config():
    """A fresh copy of the reference's defaults (config.py:11-147)."""
    vis = ["2d_planes", "image_with_poses", "heatmaps"]
    return Node(
        CUDNN=dict(BENCHMARK=True, DETERMINISTIC=False, ENABLED=True),
        BACKBONE="resnet", DEVICE="cuda:0", WORKERS=8, PRINT_FREQ=100, OUTPUT_DIR="output", LOG_DIR="log",
        MODEL="voxelpose",
        DATASET=dict(DATADIR="", COLOR_RGB=False, DATA_AUGMENTATION=False, TRAIN_DATASET="panoptic",
                     TRAIN_HEATMAP_SRC="image", TEST_DATASET="panoptic", TEST_HEATMAP_SRC="image", CAMERA_NUM=5,
                     ORI_IMAGE_SIZE=np.array([1920, 1080]), IMAGE_SIZE=np.array([960, 512]),
                     HEATMAP_SIZE=np.array([240, 128]), NUM_JOINTS=15, ROOT_JOINT_ID=2),
        SYNTHETIC=dict(CAMERA_FILE="", POSE_FILE="", MAX_PEOPLE=10, NUM_DATA=10000, DATA_AUGMENTATION=True),
        NETWORK=dict(PRETRAINED_BACKBONE="", NUM_CHANNEL_JOINT_FEAT=32, NUM_CHANNEL_JOINT_HIDDEN=64, SIGMA=3, BETA=100),
        HIGHER_HRNET=dict(PRETRAINED_LAYERS=["*"], FINAL_CONV_KERNEL=1, STEM_INPLANES=64,
                          STAGE2=_hrnet_stage(1, 2, [4, 4], [48, 96]),
                          STAGE3=_hrnet_stage(4, 3, [4, 4, 4], [48, 96, 192]),
                          STAGE4=_hrnet_stage(3, 4, [4, 4, 4, 4], [48, 96, 192, 384]),
                          DECONV=dict(NUM_DECONVS=1, NUM_CHANNELS=32, KERNEL_SIZE=4, NUM_BASIC_BLOCKS=4,
                                      CAT_OUTPUT=True)),
        RESNET=dict(NUM_LAYERS=50, DECONV_WITH_BIAS=False, NUM_DECONV_LAYERS=3, NUM_DECONV_FILTERS=[256, 256, 256],
                    NUM_DECONV_KERNELS=[4, 4, 4], FINAL_CONV_KERNEL=1),
        TRAIN=dict(BATCH_SIZE=8, SHUFFLE=True, BEGIN_EPOCH=0, END_EPOCH=10, RESUME=False, OPTIMIZER="adam", LR=1e-4,
                   LAMBDA_LOSS_2D=1.0, LAMBDA_LOSS_1D=1.0, LAMBDA_LOSS_BBOX=0.1, LAMBDA_LOSS_FUSED=5.0,
                   VISUALIZATION=True, VIS_TYPE=list(vis)),
        TEST=dict(BATCH_SIZE=8, MODEL_FILE="", VISUALIZATION=True, VIS_TYPE=list(vis)),
        CAPTURE_SPEC=dict(SPACE_SIZE=np.array([4000.0, 5200.0, 2400.0]), SPACE_CENTER=np.array([300.0, 300.0, 300.0]),
                          VOXELS_PER_AXIS=np.array([24, 32, 16]), MAX_PEOPLE=10, MIN_SCORE=0.1),
        INDIVIDUAL_SPEC=dict(SPACE_SIZE=np.array([2000.0, 2000.0, 2000.0]), VOXELS_PER_AXIS=np.array([64, 64, 64])),
    )


def _coerce(section, key, value):
    if section == "DATASET" and key in ("MEAN", "STD") and value:
        return np.array([eval(x) if isinstance(x, str) else x for x in value])      # as the reference (:152-156)
    if section == "NETWORK" and key in ("HEATMAP_SIZE", "IMAGE_SIZE"):
        return np.array([value, value]) if isinstance(value, int) else np.array(value)
    if (section, key) in _ARRAYS:
        return np.array(value)
    return value


def merge(cfg, overlay):
    """Apply a parsed YAML mapping with the reference's unknown-key rule."""
    for k, v in overlay.items():
        if k not in cfg:
            raise ValueError("{} not exist in config.py".format(k))
        if isinstance(v, dict):
            if not isinstance(cfg[k], dict):
                raise ValueError("{} is not a section in config.py".format(k))
            for vk, vv in v.items():
                if vk not in cfg[k]:
                    raise ValueError("{}.{} not exist in config.py".format(k, vk))
                if isinstance(vv, dict) and isinstance(cfg[k][vk], dict):          # HIGHER_HRNET.STAGEn
                    for k3, v3 in vv.items():
                        if k3 not in cfg[k][vk]:
                            raise ValueError("{}.{}.{} not exist in config.py".format(k, vk, k3))
                        cfg[k][vk][k3] = v3
                else:
                    cfg[k][vk] = _coerce(k, vk, vv)
        elif k == "SCALES":
            cfg[k][0] = tuple(v)
        else:
            cfg[k] = v
    return cfg


def update_config(config_file, cfg=None):
    """YAML file -> config tree (a fresh default tree unless ``cfg`` is given)."""
    cfg = default_config() if cfg is None else cfg
    with open(config_file) as f:
        overlay = yaml.safe_load(f) or {}
    return merge(cfg, overlay)
