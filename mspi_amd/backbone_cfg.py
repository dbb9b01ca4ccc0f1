"""Backbone hyper-parameter files.

The reference builds each Kinetics backbone from a PySlowFast YAML merged over
SlowFast/slowfast/config/defaults.py (parser.py:67-94).  Only the architecture keys matter on
the inference path; their defaults are restated here (values as in defaults.py) and a YAML in
the same schema -- ours under mspi_amd/configs/, or the reference's own file -- is merged on top.
"""
import os

from .attrdict import AttrDict

_HERE = os.path.dirname(os.path.abspath(__file__))


def _defaults():
    return AttrDict({
        "BN": {"NORM_TYPE": "batchnorm"},
        "DATA": {"INPUT_CHANNEL_NUM": [3, 3], "NUM_FRAMES": 8, "TRAIN_CROP_SIZE": 224, "TEST_CROP_SIZE": 256},
        "MODEL": {"ARCH": "slowfast", "MODEL_NAME": "SlowFast", "DROPCONNECT_RATE": 0.0, "NUM_CLASSES": 400},
        "RESNET": {"TRANS_FUNC": "bottleneck_transform", "NUM_GROUPS": 1, "WIDTH_PER_GROUP": 64, "STRIDE_1X1": False,
                   "DEPTH": 50, "NUM_BLOCK_TEMP_KERNEL": [[3], [4], [6], [3]],
                   "SPATIAL_STRIDES": [[1], [2], [2], [2]], "SPATIAL_DILATIONS": [[1], [1], [1], [1]]},
        "X3D": {"WIDTH_FACTOR": 1.0, "DEPTH_FACTOR": 1.0, "BOTTLENECK_FACTOR": 1.0, "DIM_C5": 2048, "DIM_C1": 12,
                "SCALE_RES2": False, "BN_LIN5": False, "CHANNELWISE_3x3x3": True},
        "NONLOCAL": {"LOCATION": [[[]], [[]], [[]], [[]]], "GROUP": [[1], [1], [1], [1]]},
        "SLOWFAST": {"BETA_INV": 8, "ALPHA": 8, "FUSION_CONV_CHANNEL_RATIO": 2, "FUSION_KERNEL_SZ": 5},
        "MVIT": {"MODE": "conv", "POOL_FIRST": False, "CLS_EMBED_ON": True, "PATCH_KERNEL": [3, 7, 7],
                 "PATCH_STRIDE": [2, 4, 4], "PATCH_PADDING": [2, 4, 4], "PATCH_2D": False, "EMBED_DIM": 96,
                 "NUM_HEADS": 1, "MLP_RATIO": 4.0, "QKV_BIAS": True, "DROPPATH_RATE": 0.1,
                 "LAYER_SCALE_INIT_VALUE": 0.0, "DEPTH": 16, "NORM": "layernorm", "DIM_MUL": [], "HEAD_MUL": [],
                 "POOL_KV_STRIDE": [], "POOL_KV_STRIDE_ADAPTIVE": None, "POOL_Q_STRIDE": [],
                 "POOL_KVQ_KERNEL": None, "NORM_STEM": False, "SEP_POS_EMBED": False, "DROPOUT_RATE": 0.0,
                 "USE_ABS_POS": True, "REL_POS_SPATIAL": False, "REL_POS_TEMPORAL": False,
                 "REL_POS_ZERO_INIT": False, "RESIDUAL_POOLING": False, "DIM_MUL_IN_ATT": False,
                 "SEPARATE_QKV": False, "USE_MEAN_POOLING": False, "USE_FIXED_SINCOS_POS": False,
                 "REV": {"ENABLE": False}},
        # backbones/Uniformer/defaults.py:404-446 (+ MODEL.NUM_CLASSES :274)
        "UNIFORMER": {"EMBED_DIM": [64, 128, 320, 512], "DEPTH": [3, 4, 8, 3], "HEAD_DIM": 64, "MLP_RATIO": 4,
                      "QKV_BIAS": True, "QKV_SCALE": None, "REPRESENTATION_SIZE": None, "SPLIT": False, "STD": False},
    })


def resolve(path):
    """`./configs/X3D_L.yaml` as written in config.py resolves against the cwd first (the
    reference's behaviour) and then against the YAMLs shipped in mspi_amd/configs/."""
    if os.path.exists(path):
        return path
    alt = os.path.join(_HERE, "configs", os.path.basename(path))
    if os.path.exists(alt):
        return alt
    raise FileNotFoundError(path)


def load_backbone_cfg(path):
    cfg = _defaults()
    cfg.merge_from_file(resolve(path))
    assert cfg.RESNET.NUM_GROUPS > 0 and cfg.RESNET.WIDTH_PER_GROUP > 0
    return cfg
