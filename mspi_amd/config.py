"""`cfg` -- same fields, names and defaults as the reference's config.py:1-101.

Selection of the motion encoder: the reference edits the module constant `_model_name`
(config.py:59); here `select_model(name)` re-derives the dependent fields, and the
environment variable MSPI_MOTION_ENCODER picks the import-time default (reference default:
'mvitv2s').  Additions (do not exist upstream, defaults reproduce upstream behaviour):
  MODEL.NUM_AUD_TOKENS  -- SyncBlock audio positional table rows; 36 = 9x4 for a 257x111
                           spectrogram (model/model_utils.py:224); 90 for 257x300.
  MODEL.SWIN.DEPTHS     -- [2,2,18,2] = the Swin-S that `SwinTransformer3D()` defaults to.
"""
import os

from .attrdict import AttrDict

cfg = AttrDict()

cfg.RECORD = AttrDict()
cfg.RECORD.LOG = "./experiments"

cfg.DATA = AttrDict()
cfg.DATA.ROOT = "./AuViDataset"
cfg.DATA.NUM_FRAMES = 16
cfg.DATA.USE_SOUND = True
cfg.DATA.RESOLUTION = (224, 384)

cfg.TRAIN = AttrDict()
cfg.TRAIN.BATCH_SIZE = 2

cfg.SOLVER = AttrDict()
cfg.SOLVER.LR = 1e-4
cfg.SOLVER.MIN_LR = 1e-5
cfg.SOLVER.MAX_EPOCH = 120
cfg.SOLVER.OPTIMIZING_METHOD = "adamw"
cfg.SOLVER.MONITORED_EPOCHES = list(range(60, 121, 20))

_MOTION_ENCODERS = ("mvitv2s", "s3d", "slowfast4x16", "morphmlps", "uniformerb", "videoswins", "x3dl")
_MOTION_WEIGHTS = {
    "mvitv2s": "./weights/MViTv2_S_16x4_k400_f302660347.pyth",
    "s3d": "./weights/S3D_kinetics400_rm_fc.pt",
    "slowfast4x16": "./weights/SLOWFAST_4x16_R50.pkl",
    "morphmlps": "./weights/mlp_s16x4_k400.pth",
    "uniformerb": "./weights/uniformer_base_k400_16x4.pth",
    "videoswins": "./weights/swin_small_patch244_window877_kinetics400_1k.pth",
    "x3dl": "./weights/x3d_l.pyth",
}
_LATERAL_BOOL = {
    "mvitv2s": [True, True, True, True],
    "s3d": [True, True, False, False],
    "slowfast4x16": [False, False, False, False],
    "morphmlps": [True, True, True, True],
    "uniformerb": [True, True, True, True],
    "videoswins": [True, True, True, True],
    "x3dl": [True, True, True, True],
}
_NUM_VIS_TOKENS = {
    "mvitv2s": 8 * 7 * 12,
    "s3d": 4 * 7 * 7,
    "slowfast4x16": 4 * 7 * 7,
    "morphmlps": 8 * 7 * 7,
    "uniformerb": 8 * 7 * 7,
    "videoswins": 8 * 7 * 7,
    "x3dl": 16 * 7 * 7,
}

_model_name = os.environ.get("MSPI_MOTION_ENCODER", _MOTION_ENCODERS[0])

cfg.MODEL = AttrDict()
cfg.MODEL.MOTION_ENCODER_EMBEDS = {
    "mvitv2s": (96, 192, 384, 768),
    "s3d": (192, 480, 832, 1024),
    "slowfast4x16": (320, 640, 1280, 2048),
    "morphmlps": (112, 224, 392, 784),
    "uniformerb": (64, 128, 320, 512),
    "videoswins": (96, 192, 384, 768),
    "x3dl": (24, 48, 96, 192),
}
cfg.MODEL.NUM_VIS_TOKENS = dict(_NUM_VIS_TOKENS)
cfg.MODEL.NUM_AUD_TOKENS = 36
cfg.MODEL.IMAGE_SALIENCY_ENCODER_WEIGHT = "./weights/image_saliency_encoder_convnext_tiny.pt"
cfg.MODEL.AUDIO_ENCODER_WEIGHT = "./weights/resnet18_vggsound.pt"

cfg.MODEL.S3D = AttrDict()
cfg.MODEL.S3D.POOL_STRIDE = 1
cfg.MODEL.MVIT2 = AttrDict()
cfg.MODEL.MVIT2.PATH_CFG = ["./configs/MVITv2_S_16x4.yaml"]
cfg.MODEL.SLOWFAST = AttrDict()
cfg.MODEL.SLOWFAST.PATH_CFG = ["./configs/SLOWFAST_4x16_R50.yaml"]
cfg.MODEL.MORPH = AttrDict()
cfg.MODEL.MORPH.PATH_CFG = "./configs/K400_MLP_S16x4.yaml"
cfg.MODEL.X3D = AttrDict()
cfg.MODEL.X3D.PATH_CFG = ["./configs/X3D_L.yaml"]
cfg.MODEL.UNIFORMER = AttrDict()
cfg.MODEL.UNIFORMER.PATH_CFG = "./configs/uniformer_b16x4_k400.yaml"
cfg.MODEL.SWIN = AttrDict()
cfg.MODEL.SWIN.DEPTHS = [2, 2, 18, 2]


def select_model(name, target=None):
    """Point cfg (or a clone passed as `target`) at motion encoder `name` (config.py:59-65)."""
    c = cfg if target is None else target
    if name not in _MOTION_ENCODERS:
        raise Exception("Invalid Motion Encoder!")
    c.MODEL.LATERAL_BOOL = list(_LATERAL_BOOL[name])
    c.MODEL.LATERAL_STRIDE = [4, 4, 4, 4] if name == "x3dl" else [2, 2, 2, 2]
    c.MODEL.MOTION_ENCODER = name
    c.MODEL.MOTION_ENCODER_WEIGHT = _MOTION_WEIGHTS[name]
    return c


select_model(_model_name)
