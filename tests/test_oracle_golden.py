"""CPU: the oracle restatement (oracle/restate.py) against the golden vectors that
oracle/gen_golden.py captured from the reference import.  Weights and inputs are rebuilt from
seeds through the product's own module tree (mspi_amd.testing), so these tests also pin the
state-dict layout: a changed key/shape/init changes the checksum stored in the fixture."""
import os

import numpy as np
import pytest
import torch

from mspi_amd import testing as T
from oracle import restate as R


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _err(a, b):
    return float(np.abs(a.detach().numpy() - b).max())


def test_sinusoid_table(golden_dir):
    g = _g(golden_dir, "sinusoid_90x512")
    assert _err(R.sinusoid_table(90, 512), g["table"]) == 0.0
    from mspi_amd.model.model_utils import get_sinusoid_encoding_table
    assert _err(get_sinusoid_encoding_table(90, 512)[0], g["table"]) == 0.0


def test_x3dl_backbone(golden_dir):
    from mspi_amd.backbones.X3D import X3D
    from mspi_amd.config import cfg
    g = _g(golden_dir, "x3dl_backbone_64")
    m = T.seeded(lambda: X3D(cfg.MODEL.X3D.PATH_CFG), int(g["seed"]))
    sd = m.state_dict()
    assert len(sd) == 1137 and T.sd_checksum(sd) == int(g["sd_crc"])
    clips, _ = T.synth_inputs(int(g["batch"]), 16, int(g["size"]), int(g["size"]), seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.x3d_forward(sd, clips)
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


def test_s3d_backbone(golden_dir):
    """S3D_features_only; the fixture is the reference's own forward on the product's state dict."""
    from mspi_amd.backbones.s3d import S3D_features_only
    g = _g(golden_dir, "s3d_backbone_64")
    m = T.seeded(lambda: S3D_features_only(), int(g["seed"]))
    T.randomize_(m, int(g["seed"]) + 1)
    sd = m.state_dict()
    assert len(sd) == 462 and T.sd_checksum(sd) == int(g["sd_crc"])
    clips, _ = T.synth_inputs(int(g["batch"]), 16, int(g["size"]), int(g["size"]), seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.s3d_forward(sd, clips)
    assert [f.shape[1] for f in feats] == [192, 480, 832, 1024]
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


@pytest.mark.parametrize("case", ["uniformer_backbone_64", "uniformer_backbone_224"])
def test_uniformer_backbone(golden_dir, case):
    """Uniformer (UniFormer-B); the fixtures are the reference's own forward on the product's state dict
    (64x64: 128 / 32 attention tokens, 224x224: 1568 / 392)."""
    from mspi_amd.backbones.uniformer import Uniformer
    from mspi_amd.config import cfg
    g = _g(golden_dir, case)
    m = T.condition_(T.seeded(lambda: Uniformer(cfg.MODEL.UNIFORMER.PATH_CFG), int(g["seed"])), "uniformerb")
    sd = m.state_dict()
    assert T.sd_checksum(sd) == int(g["sd_crc"])
    clips, _ = T.synth_inputs(int(g["batch"]), 16, int(g["size"]), int(g["size"]), seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.uniformer_forward(sd, clips)
    assert [f.shape[1] for f in feats] == [64, 128, 320, 512]
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


def test_morphmlp_backbone(golden_dir):
    """MorphMLP_32_features_only (MorphMLP-S); fixture = the reference's own forward at 224x224 (the only
    extent upstream's reshapes accept).  The reference's permuted views take other matmul paths than the restatement's
    contiguous ones: equal to 1e-6 of the feature scale, not bit for bit."""
    from mspi_amd.backbones.MorphMLP import MorphMLP_32_features_only
    from mspi_amd.config import cfg
    g = _g(golden_dir, "morphmlp_backbone_224")
    m = T.seeded(lambda: MorphMLP_32_features_only(cfg.MODEL.MORPH.PATH_CFG), int(g["seed"]))
    sd = m.state_dict()
    assert T.sd_checksum(sd) == int(g["sd_crc"])
    clips, _ = T.synth_inputs(int(g["batch"]), 16, int(g["size"]), int(g["size"]), seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.morphmlp_forward(sd, clips)
    assert [f.shape[1] for f in feats] == [112, 224, 392, 784]
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 5e-6


def test_slowfast_backbone(golden_dir):
    from mspi_amd.backbones.sf import SlowFast
    from mspi_amd.config import cfg
    g = _g(golden_dir, "slowfast_backbone_64")
    m = T.seeded(lambda: SlowFast(cfg.MODEL.SLOWFAST.PATH_CFG), int(g["seed"]))
    sd = m.state_dict()
    assert len(sd) == 660 and T.sd_checksum(sd) == int(g["sd_crc"])
    clips, _ = T.synth_inputs(int(g["batch"]), 16, int(g["size"]), int(g["size"]), seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.slowfast_forward(sd, R.pack_clips("slowfast4x16", clips))
    assert [f.shape[1] for f in feats] == [320, 640, 1280, 2048]
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


def test_mvit_backbone(golden_dir):
    from mspi_amd.backbones.MViT import MViT
    from mspi_amd.config import cfg
    g = _g(golden_dir, "mvit_backbone_224")
    m = T.seeded(lambda: MViT(cfg.MODEL.MVIT2.PATH_CFG), int(g["seed"]))
    sd = m.state_dict()
    assert len(sd) == 394 and T.sd_checksum(sd) == int(g["sd_crc"])
    # the product's resolved per-block geometry equals the table the oracle restates from the reference ctor
    arch = [(b.attn.num_heads, b.attn.stride_q, b.attn.stride_kv) for b in m.blocks]
    assert arch == R.MVIT_S_ARCH["blocks"]
    clips, _ = T.synth_inputs(1, 16, 224, 224, seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.mvit_forward(sd, clips, R.MVIT_S_ARCH)
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


def test_mvit_backbone_224x384(golden_dir):
    """The reference's default frame size (config.py:14): rel_pos_w tables are interpolated (MViT.py:207-220)."""
    from mspi_amd.backbones.MViT import MViT
    from mspi_amd.config import cfg
    g = _g(golden_dir, "mvit_backbone_224x384")
    sd = T.seeded(lambda: MViT(cfg.MODEL.MVIT2.PATH_CFG), int(g["seed"])).state_dict()
    assert T.sd_checksum(sd) == int(g["sd_crc"])
    clips, _ = T.synth_inputs(1, 16, int(g["H"]), int(g["W"]), seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.mvit_forward(sd, clips, R.MVIT_S_ARCH)
    assert [tuple(f.shape[2:]) for f in feats] == [(8, 56, 96), (8, 28, 48), (8, 14, 24), (8, 7, 12)]
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


def test_swin_backbone(golden_dir):
    from mspi_amd.backbones.video_swin_transformer import SwinTransformer3D
    g = _g(golden_dir, "swin_t_backbone_224")
    m = T.seeded(lambda: SwinTransformer3D(depths=[2, 2, 6, 2]), int(g["seed"]))
    sd = m.state_dict()
    assert T.sd_checksum(sd) == int(g["sd_crc"])
    assert len(SwinTransformer3D().state_dict()) == 349                      # default = Swin-S
    clips, _ = T.synth_inputs(1, 16, 224, 224, seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.swin_forward(sd, clips)
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


def test_swin_backbone_padded_windows(golden_dir):
    """128x192 frames: token grids that are not multiples of the 7x7 window (the reference's zero-padding path)."""
    from mspi_amd.backbones.video_swin_transformer import SwinTransformer3D
    g = _g(golden_dir, "swin_t_backbone_128x192")
    m = T.seeded(lambda: SwinTransformer3D(depths=[2, 2, 6, 2]), int(g["seed"]))
    sd = m.state_dict()
    assert T.sd_checksum(sd) == int(g["sd_crc"])
    clips, _ = T.synth_inputs(int(g["batch"]), 16, int(g["H"]), int(g["W"]), seed=int(g["seed"]))
    with torch.no_grad():
        feats = R.swin_forward(sd, clips)
    for i, f in enumerate(feats):
        assert T.feature_error(f, g, "v%d" % (i + 1)) <= 1e-6


def test_swin_index_and_mask_helpers():
    """The product's window index / mask helpers against roll + window_partition done with torch."""
    from mspi_amd.backbones import video_swin_transformer as S
    D, H, W, ws, ss = 4, 14, 14, (4, 7, 7), (0, 3, 3)
    x = torch.arange(D * H * W, dtype=torch.float32).view(1, D, H, W, 1)
    ref = R._swin_partition(torch.roll(x, (-ss[0], -ss[1], -ss[2]), (1, 2, 3)), ws).squeeze(-1).to(torch.int32)
    assert torch.equal(S.window_token_index(D, H, W, ws, ss), ref)
    assert torch.equal(S.compute_mask(D, H, W, ws, ss), R._swin_mask(D, H, W, ws, ss))
    assert S.get_window_size((8, 56, 56), (8, 7, 7), (4, 3, 3)) == ((8, 7, 7), (0, 3, 3))
    assert S.get_window_size((8, 7, 7), (8, 7, 7), (4, 3, 3)) == ((8, 7, 7), (0, 0, 0))
    # padded grid: positions outside D x H x W map to the extra row D*H*W, the rest to what roll + partition of the
    # zero-padded tensor gives
    D, H, W, ws, ss = 4, 9, 12, (4, 7, 7), (0, 3, 3)
    Dp, Hp, Wp = 4, 14, 14
    xp = torch.full((1, Dp, Hp, Wp, 1), float(D * H * W))
    xp[:, :D, :H, :W, 0] = torch.arange(D * H * W, dtype=torch.float32).view(D, H, W)
    ref = R._swin_partition(torch.roll(xp, (-ss[0], -ss[1], -ss[2]), (1, 2, 3)), ws).squeeze(-1).to(torch.int32)
    assert torch.equal(S.window_token_index(D, H, W, ws, ss, (Dp, Hp, Wp)), ref)


@pytest.mark.parametrize("wa", [111, 300])
def test_resnet18_audio(golden_dir, wa):
    from mspi_amd.backbones.resnet import ResNet
    g = _g(golden_dir, "resnet18_audio_%d" % wa)
    m = T.seeded(ResNet, int(g["seed"]))
    sd = m.state_dict()
    assert T.sd_checksum(sd) == int(g["sd_crc"])
    _, audio = T.synth_inputs(int(g["batch"]), Wa=wa, H=8, W=8, seed=int(g["seed"]))
    with torch.no_grad():
        out = R.resnet18_forward(sd, audio)
    assert tuple(out.shape) == (2, 512, 9, (wa + 31) // 32)
    assert _err(out, g["out"]) <= 1e-5 * max(1.0, float(np.abs(g["out"]).max()))


def _model(g, name, cls):
    from mspi_amd.model import model_utils as pm
    cfg = T.golden_cfg(g, name)
    m = T.condition_(T.seeded(lambda: getattr(pm, cls)(cfg), int(g["seed"])), name)
    sd = m.state_dict()
    assert T.sd_checksum(sd) == int(g["sd_crc"]), "seeded weights drifted from the ones the golden was made with"
    H, W = T.golden_hw(g)
    clips, audio = T.synth_inputs(int(g["batch"]), 16, H, W, Wa=int(g["wa"]), seed=int(g["seed"]))
    return cfg, sd, clips, audio


@pytest.mark.parametrize("case,name", [("av_x3dl_64", "x3dl"), ("av_x3dl_224", "x3dl"), ("av_slowfast_64", "slowfast4x16"),
                                       ("av_mvit_224", "mvitv2s"), ("av_swin_s_224", "videoswins"), ("av_s3d_64", "s3d"),
                                       ("av_uniformer_64", "uniformerb"), ("av_morphmlp_224", "morphmlps"),
                                       ("av_slowfast_224", "slowfast4x16"), ("av_uniformer_224", "uniformerb"), ("av_s3d_224", "s3d"),
                                       ("av_swin_t_224", "videoswins"), ("av_mvit_224_wa300", "mvitv2s"),
                                       ("av_mvit_224x384", "mvitv2s")])
def test_audio_visual_model(golden_dir, case, name):
    g = _g(golden_dir, case)
    cfg, sd, clips, audio = _model(g, name, "AudioVisualSaliencyModel")
    with torch.no_grad():
        out, loss = R.audio_visual_forward(sd, clips, audio, name, cfg.MODEL.LATERAL_BOOL, cfg.MODEL.LATERAL_STRIDE)
    assert tuple(out.shape) == g["out"].shape
    assert _err(out, g["out"]) <= 5e-5 and abs(float(loss) - float(g["loss"])) <= 1e-5
    assert abs(float(torch.logsumexp(out[0], (0, 1)))) < 1e-4   # F1: a log-probability map


def test_visual_model(golden_dir):
    g = _g(golden_dir, "vis_x3dl_64")
    cfg, sd, clips, _ = _model(g, "x3dl", "VisualSaliencyModel")
    with torch.no_grad():
        out, zero = R.visual_forward(sd, clips, "x3dl", cfg.MODEL.LATERAL_BOOL, cfg.MODEL.LATERAL_STRIDE)
    assert zero == 0 and _err(out, g["out"]) <= 5e-5


def test_unit_level_outputs_against_reference_hooks(golden_dir):
    """the oracle's sub-module outputs (image encoder, Adapter, audio ResNet, SyncBlock, lateral layers, SA
    gates, readout) against what forward hooks on the REFERENCE's own sub-modules recorded during the same forward."""
    g = _g(golden_dir, "av_x3dl_64_units")
    cfg, sd, clips, audio = _model(g, "x3dl", "AudioVisualSaliencyModel")
    tr = {}
    with torch.no_grad():
        out, _ = R.audio_visual_forward(sd, clips, audio, "x3dl", cfg.MODEL.LATERAL_BOOL, cfg.MODEL.LATERAL_STRIDE, trace=tr)
    keys = [str(k) for k in g["keys"]]
    assert len(keys) == 13 and set(keys) == set(tr)
    for i, k in enumerate(keys):
        assert T.feature_error(tr[k], g, "v%d" % (i + 1)) <= 2e-6, k
    assert _err(out, g["out"]) <= 5e-5


def _unit_check(g, tag, trace, rename=None, tol=2e-6):
    keys = [str(k) for k in g[tag + "_keys"]]
    sub = {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "_v")}
    for i, k in enumerate(keys):
        t = trace[rename(k) if rename else k]
        shape = tuple(int(v) for v in sub["v%d_shape" % (i + 1)])
        assert T.feature_error(t.reshape(shape), sub, "v%d" % (i + 1)) <= tol, "%s %s" % (tag, k)
    return len(keys)


def test_backbone_unit_level_outputs_against_reference_hooks(golden_dir):
    """sub-module outputs inside the four BASELINE backbones against forward hooks on the reference's own
    modules -- X3D stem, SlowFast stems / pathways / first fusion, all 16 MViTv2-S blocks, all Swin-T blocks + PatchMerging."""
    from mspi_amd.backbones.MViT import MViT
    from mspi_amd.backbones.X3D import X3D
    from mspi_amd.backbones.sf import SlowFast
    from mspi_amd.backbones.video_swin_transformer import SwinTransformer3D
    from mspi_amd.config import cfg
    g = _g(golden_dir, "backbone_units")
    seed = int(g["seed"])
    clips, _ = T.synth_inputs(2, 16, 64, 64, seed=seed)
    clips224, _ = T.synth_inputs(1, 16, 224, 224, seed=seed)
    sd = T.seeded(lambda: X3D(cfg.MODEL.X3D.PATH_CFG), seed).state_dict()
    assert T.sd_checksum(sd) == int(g["x3dl_crc"])
    tr = {}
    with torch.no_grad():
        R.x3d_forward(sd, clips, trace=tr)
    assert _unit_check(g, "x3dl", tr) == 1
    sd = T.seeded(lambda: SlowFast(cfg.MODEL.SLOWFAST.PATH_CFG), seed).state_dict()
    assert T.sd_checksum(sd) == int(g["slowfast_crc"])
    tr = {}
    with torch.no_grad():
        R.slowfast_forward(sd, R.pack_clips("slowfast4x16", clips), trace=tr)
    tr["s1_fuse.1"] = tr["s1.1"]          # FuseFastToSlow returns [fused slow, untouched fast]
    assert _unit_check(g, "slowfast", tr) == 10
    sd = T.seeded(lambda: MViT(cfg.MODEL.MVIT2.PATH_CFG), seed).state_dict()
    assert T.sd_checksum(sd) == int(g["mvit_crc"])
    tr = {}
    with torch.no_grad():
        R.mvit_forward(sd, clips224, R.MVIT_S_ARCH, trace=tr)
    assert _unit_check(g, "mvit", tr) == 16
    sd = T.seeded(lambda: SwinTransformer3D(depths=[2, 2, 6, 2]), seed).state_dict()
    assert T.sd_checksum(sd) == int(g["swin_crc"])
    tr = {}
    with torch.no_grad():
        R.swin_forward(sd, clips224, trace=tr)
    assert _unit_check(g, "swin", tr) == 15
