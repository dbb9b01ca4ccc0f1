"""Backbone factory -- same contract as the reference's model/get_video_backbones.py:11-31:
`video_motion_extractor(cfg) -> nn.Module` whose forward returns four NCDHW fp32 feature maps
and which has `load_weight(path)`; raises Exception("Invalid Motion Encoder!") otherwise.
Only the selected backbone is imported (the reference imports all seven eagerly)."""

_MOTION_ENCODERS = ("mvitv2s", "s3d", "slowfast4x16", "morphmlps", "uniformerb", "videoswins", "x3dl")


def video_motion_extractor(cfg):
    name = cfg.MODEL.MOTION_ENCODER
    motion_encoder = None
    if name == "x3dl":
        from ..backbones.X3D import X3D
        motion_encoder = X3D(path_to_config=cfg.MODEL.X3D.PATH_CFG)
    elif name == "slowfast4x16":
        from ..backbones.sf import SlowFast
        motion_encoder = SlowFast(path_to_config=cfg.MODEL.SLOWFAST.PATH_CFG)
    elif name == "mvitv2s":
        from ..backbones.MViT import MViT
        motion_encoder = MViT(path_to_configs=cfg.MODEL.MVIT2.PATH_CFG)
    elif name == "videoswins":
        from ..backbones.video_swin_transformer import SwinTransformer3D
        depths = cfg.MODEL.get("SWIN", {}).get("DEPTHS", [2, 2, 18, 2])
        motion_encoder = SwinTransformer3D(depths=list(depths))
    elif name == "s3d":
        from ..backbones.s3d import S3D_features_only
        motion_encoder = S3D_features_only(pool=cfg.MODEL.S3D.POOL_STRIDE)
    elif name == "uniformerb":
        from ..backbones.uniformer import Uniformer
        motion_encoder = Uniformer(yaml_path=cfg.MODEL.UNIFORMER.PATH_CFG)
    elif name == "morphmlps":
        from ..backbones.MorphMLP import MorphMLP_32_features_only
        motion_encoder = MorphMLP_32_features_only(path_to_config=cfg.MODEL.MORPH.PATH_CFG)
    if motion_encoder is None:
        raise Exception("Invalid Motion Encoder!")
    return motion_encoder
