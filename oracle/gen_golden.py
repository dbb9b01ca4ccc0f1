"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

For each case: build the *product* module tree (mspi_amd, parameter holders with the
reference's key names) under a seed, load its state dict with strict=True into the
*reference* module imported from /root/reference (which also proves key/shape parity), run
the reference CPU fp32 forward on seeded inputs and store the outputs under tests/golden/.
Weights and inputs are reproducible from seeds (mspi_amd.testing), so only outputs, seeds and
a state-dict checksum are committed.

ConvNeXt-Tiny: timm is absent, so the reference's `timm.create_model` call is routed to a
module that runs oracle.restate.convnext_tiny_features (PARITY UNPINNED for that sub-network;
everything downstream of it in the reference's forward runs unmodified).

Usage: python oracle/gen_golden.py [case ...]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from mspi_amd import testing as T  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402
from oracle import restate as R  # noqa: E402

torch.set_num_threads(8)


class _RefConvNeXt(nn.Module):
    """Stand-in for timm's FeatureListNet inside the reference model: parameters named like timm's,
    arithmetic = the oracle's restatement."""

    def __init__(self):
        super().__init__()
        from mspi_amd.backbones.convnext import ConvNeXtTinyFeatures
        holder = ConvNeXtTinyFeatures()
        for n, c in holder.named_children():
            self.add_module(n, c)

    def forward(self, x):
        return R.convnext_tiny_features(self.state_dict(), x)


class _NoWeights(dict):
    def __getitem__(self, k):
        return _NoWeights()


def build_reference_model(name, cls_name, num_vis_tokens=None, swin_depths=None):
    """Construct the reference AudioVisualSaliencyModel / VisualSaliencyModel offline:
    create_model -> stand-in, the three torch.load of absent weight files -> no-ops.  `swin_depths`: the reference's
    factory calls SwinTransformer3D() (= Swin-S, model/get_video_backbones.py:25); BASELINE configs[4] names Swin-T,
    which is the SAME reference class with depths=[2,2,6,2] -- the factory's name is bound to a partial of it."""
    cfg = rh.with_config(name)
    if num_vis_tokens is not None:
        cfg.MODEL.NUM_VIS_TOKENS[name] = num_vis_tokens
    rh.set_create_model(lambda *a, **k: _RefConvNeXt())
    import model.model_utils as mu
    import backbones.sf as ref_sf
    import backbones.s3d as ref_s3d
    import backbones.MorphMLP as ref_morph
    real_load, real_lsd, real_sf_lw = torch.load, nn.Module.load_state_dict, ref_sf.SlowFast.load_weight
    real_s3d_lw = ref_s3d.S3D_features_only.load_weight
    real_morph_lw = ref_morph.MorphMLP_32_features_only.load_weight
    import functools
    import model.get_video_backbones as ref_factory
    real_swin = ref_factory.SwinTransformer3D
    if swin_depths is not None:
        ref_factory.SwinTransformer3D = functools.partial(real_swin, depths=list(swin_depths))
    ref_morph.MorphMLP_32_features_only.load_weight = lambda self, path: None   # deletes head.* from the (absent) checkpoint
    ref_sf.SlowFast.load_weight = lambda self, path: None    # caffe2 .pkl loader opens the (absent) file itself
    ref_s3d.S3D_features_only.load_weight = lambda self, path: None   # raises on the absent file before any torch.load
    torch.load = lambda *a, **k: _NoWeights()
    nn.Module.load_state_dict = lambda self, sd, *a, **k: None if isinstance(sd, _NoWeights) else real_lsd(self, sd, *a, **k)
    try:
        m = getattr(mu, cls_name)(cfg)
    finally:
        torch.load, nn.Module.load_state_dict, ref_sf.SlowFast.load_weight = real_load, real_lsd, real_sf_lw
        ref_s3d.S3D_features_only.load_weight = real_s3d_lw
        ref_morph.MorphMLP_32_features_only.load_weight = real_morph_lw
        ref_factory.SwinTransformer3D = real_swin
    return m.eval()


def _save(case, **arrs):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, case + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("  wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def _feat_fixture(feats, max_elems=1 << 15):
    """Feature maps are too big to commit whole: keep every stride-th element of the flattened tensor
    (stride prime to the innermost dims so samples wander over all axes) plus mean / abs-max of the whole."""
    out = {}
    for i, f in enumerate(feats):
        flat = f.detach().reshape(-1)
        stride = max(1, flat.numel() // max_elems)
        stride += 1 - stride % 2          # odd
        while stride > 1 and any(d % stride == 0 for d in f.shape if d > 1):
            stride += 2
        k = "v%d" % (i + 1)
        out[k + "_shape"] = np.array(f.shape)
        out[k + "_stride"] = stride
        out[k + "_sample"] = flat[::stride].clone()
        out[k + "_mean"] = flat.double().mean().item()
        out[k + "_absmax"] = flat.abs().max().item()
    return out


def _check_restatement(name, ref_out, ora_out, tol=2e-5):
    err = max((a - b).abs().max().item() for a, b in zip(ref_out, ora_out))
    print("  %s: max |reference - oracle| = %.3e" % (name, err))
    assert err < tol, "oracle restatement drifted from the reference"


# ------------------------------------------------------------------------------- cases
def case_sinusoid():
    rh.enter_reference()
    import model.model_utils as mu
    ref = mu.get_sinusoid_encoding_table(90, 512)[0]
    _check_restatement("sinusoid", [ref], [R.sinusoid_table(90, 512)], 1e-7)
    _save("sinusoid_90x512", table=ref)


def case_x3dl_backbone(size=64, seed=0):
    from mspi_amd.backbones.X3D import X3D
    from mspi_amd.config import cfg as pcfg
    prod = T.seeded(lambda: X3D(pcfg.MODEL.X3D.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("x3dl")
    from backbones.X3D import X3D as RefX3D
    ref = RefX3D(path_to_config=rcfg.MODEL.X3D.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(2, 16, size, size, seed=seed)
    with torch.no_grad():
        feats = ref([clips])
        ora = R.x3d_forward(sd, clips)
    _check_restatement("x3dl backbone", feats, ora)
    _save("x3dl_backbone_%d" % size, seed=seed, size=size, batch=2, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_s3d_backbone(size=64, seed=0):
    """S3D_features_only (backbones/s3d.py:379-421) with the product's seeded + randomised state dict, strict load."""
    from mspi_amd.backbones.s3d import S3D_features_only
    prod = T.seeded(lambda: S3D_features_only(), seed)
    T.randomize_(prod, seed + 1)
    sd = prod.state_dict()
    rh.with_config("x3dl")                         # enters the reference tree (stubs, sys.path); the config itself is unused
    from backbones.s3d import S3D_features_only as RefS3D
    ref = RefS3D(pool=1).eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(2, 16, size, size, seed=seed)
    with torch.no_grad():
        feats = ref(clips)
        ora = R.s3d_forward(sd, clips)
    _check_restatement("s3d backbone", feats, ora)
    _save("s3d_backbone_%d" % size, seed=seed, size=size, batch=2, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_slowfast_backbone(size=64, seed=0):
    from mspi_amd.backbones.sf import SlowFast
    from mspi_amd.config import cfg as pcfg
    prod = T.seeded(lambda: SlowFast(pcfg.MODEL.SLOWFAST.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("slowfast4x16")
    from backbones.sf import SlowFast as RefSF
    ref = RefSF(path_to_config=rcfg.MODEL.SLOWFAST.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(2, 16, size, size, seed=seed)
    x = R.pack_clips("slowfast4x16", clips)
    with torch.no_grad():
        feats = ref(x)
        ora = R.slowfast_forward(sd, x)
    _check_restatement("slowfast backbone", feats, ora)
    _save("slowfast_backbone_%d" % size, seed=seed, size=size, batch=2, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_mvit_backbone(size=224, seed=0):
    """MViT's feature taps hard-code h in [56,28,14,7] (backbones/MViT.py:2063,2073): 224x224 only."""
    from mspi_amd.backbones.MViT import MViT
    from mspi_amd.config import cfg as pcfg
    prod = T.seeded(lambda: MViT(pcfg.MODEL.MVIT2.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("mvitv2s")
    from backbones.MViT import MViT as RefMViT
    ref = RefMViT(path_to_configs=rcfg.MODEL.MVIT2.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(1, 16, size, size, seed=seed)
    with torch.no_grad():
        feats = ref([clips])
        ora = R.mvit_forward(sd, clips, R.MVIT_S_ARCH)
    _check_restatement("mvit backbone", feats, ora, 5e-5)
    _save("mvit_backbone_%d" % size, seed=seed, size=size, batch=1, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_mvit_backbone_224x384(seed=0):
    """MViTv2-S backbone on the reference's default 224x384 frames (rel-pos table interpolation along W)."""
    from mspi_amd.backbones.MViT import MViT
    from mspi_amd.config import cfg as pcfg
    prod = T.seeded(lambda: MViT(pcfg.MODEL.MVIT2.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("mvitv2s")
    from backbones.MViT import MViT as RefMViT
    ref = RefMViT(path_to_configs=rcfg.MODEL.MVIT2.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(1, 16, 224, 384, seed=seed)
    with torch.no_grad():
        feats = ref([clips])
        ora = R.mvit_forward(sd, clips, R.MVIT_S_ARCH)
    _check_restatement("mvit backbone 224x384", feats, ora, 5e-5)
    _save("mvit_backbone_224x384", seed=seed, H=224, W=384, batch=1, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_swin_backbone(seed=0):
    """Swin-T depths through the reference class keeps the CPU forward short; Swin-S differs only in
    the number of stage-2 blocks.  224x224 (no padding anywhere)."""
    from mspi_amd.backbones.video_swin_transformer import SwinTransformer3D
    prod = T.seeded(lambda: SwinTransformer3D(depths=[2, 2, 6, 2]), seed)
    sd = prod.state_dict()
    rh.with_config("videoswins")
    from backbones.video_swin_transformer import SwinTransformer3D as RefSwin
    ref = RefSwin(depths=[2, 2, 6, 2])
    ref.eval()                       # the reference's train() override returns None
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(1, 16, 224, 224, seed=seed)
    with torch.no_grad():
        feats = ref(clips)
        ora = R.swin_forward(sd, clips)
    _check_restatement("swin-T backbone", feats, ora, 5e-5)
    _save("swin_t_backbone_224", seed=seed, size=224, batch=1, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_swin_backbone_padded(seed=0):
    """Swin-T depths on 128x192 frames (the aspect of the reference's default 224x384): token grids 32x48 and 16x24 are
    not multiples of the 7x7 window, so stages 1-2 run the reference's zero-padding path (:240-246, :419-423), with the
    shifted-window mask on the padded grid; stages 3-4 (8x12, 4x6) clamp the window instead."""
    from mspi_amd.backbones.video_swin_transformer import SwinTransformer3D
    prod = T.seeded(lambda: SwinTransformer3D(depths=[2, 2, 6, 2]), seed)
    sd = prod.state_dict()
    rh.with_config("videoswins")
    from backbones.video_swin_transformer import SwinTransformer3D as RefSwin
    ref = RefSwin(depths=[2, 2, 6, 2])
    ref.eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(2, 16, 128, 192, seed=seed)
    with torch.no_grad():
        feats = ref(clips)
        ora = R.swin_forward(sd, clips)
    _check_restatement("swin-T backbone, padded windows", feats, ora, 5e-5)
    _save("swin_t_backbone_128x192", seed=seed, H=128, W=192, batch=2, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_av_swin_224():
    """Full AV model with the reference's default SwinTransformer3D() = Swin-S."""
    _model_case("videoswins", "AudioVisualSaliencyModel", 224, 1, 111, 0, "av_swin_s_224")


def case_av_mvit_224():
    _model_case("mvitv2s", "AudioVisualSaliencyModel", 224, 1, 111, 0, "av_mvit_224")


def case_av_swin_t_224():
    """BASELINE configs[4]: VideoSwin-T (reference class, depths=[2,2,6,2]) + audio at Wa=300, whole model."""
    _model_case("videoswins", "AudioVisualSaliencyModel", 224, 1, 300, 0, "av_swin_t_224", swin_depths=[2, 2, 6, 2])


def case_av_mvit_224_wa300():
    """BASELINE configs[3]: MViTv2-S + audio with the 257x300 spectrogram (90 audio tokens)."""
    _model_case("mvitv2s", "AudioVisualSaliencyModel", 224, 1, 300, 0, "av_mvit_224_wa300")


def case_av_mvit_224x384():
    """The reference's DEFAULT deployment shape (config.py:14,49,59): mvitv2s on 224x384 frames, Wa=111, 8*7*12 visual
    tokens.  W=384 gives 96/48/24/12-wide token grids against rel_pos_w tables sized for 56/28/14/7: get_rel_pos's linear
    interpolation (backbones/MViT.py:207-220) runs in every block."""
    _model_case("mvitv2s", "AudioVisualSaliencyModel", (224, 384), 1, 111, 0, "av_mvit_224x384")


def case_av_slowfast_64():
    _model_case("slowfast4x16", "AudioVisualSaliencyModel", 64, 2, 111, 0, "av_slowfast_64")


def case_resnet18_audio(seed=0):
    from mspi_amd.backbones.resnet import ResNet
    prod = T.seeded(ResNet, seed)
    sd = prod.state_dict()
    rh.enter_reference()
    from backbones.resnet import get_resnet18
    ref = get_resnet18(pretrained=False).eval()
    ref.load_state_dict(sd, strict=True)
    for wa in (111, 300):
        _, audio = T.synth_inputs(2, Wa=wa, H=8, W=8, seed=seed)
        with torch.no_grad():
            out = ref(audio)
            ora = R.resnet18_forward(sd, audio)
        _check_restatement("resnet18 audio Wa=%d" % wa, [out], [ora])
        _save("resnet18_audio_%d" % wa, seed=seed, batch=2, sd_crc=T.sd_checksum(sd), out=out)


def _model_case(name, cls_name, size, B, wa, seed, tag, swin_depths=None):
    """`size`: frame edge, or (H, W) for rectangular frames (the reference default is 224x384, config.py:14)."""
    from mspi_amd.model import model_utils as pm
    H, W = (size, size) if isinstance(size, int) else size
    t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
    nvt = t_tok * (H // 32) * (W // 32)
    aud_tok = 9 * ((wa + 31) // 32)
    pcfg = T.make_cfg(name, num_aud_tokens=aud_tok, num_vis_tokens=nvt, swin_depths=swin_depths)
    prod = T.condition_(T.seeded(lambda: getattr(pm, cls_name)(pcfg), seed), name)
    sd = prod.state_dict()
    ref = build_reference_model(name, cls_name, num_vis_tokens=nvt, swin_depths=swin_depths)
    missing = set(ref.state_dict()) ^ set(sd)
    assert not missing, "state-dict keys differ: %s" % sorted(missing)[:8]
    ref.load_state_dict(sd, strict=True)
    if cls_name == "AudioVisualSaliencyModel" and aud_tok != 36:   # F3: rebuild the plain-tensor table
        import model.model_utils as mu
        ref.aud_vis_sync_block.aud_pos_embed = mu.get_sinusoid_encoding_table(aud_tok, 512)
    clips, audio = T.synth_inputs(B, 16, H, W, Wa=wa, seed=seed)
    t0 = time.time()
    with torch.no_grad():
        if cls_name == "AudioVisualSaliencyModel":
            out, loss = ref(clips, audio)
            o2, l2 = R.audio_visual_forward(sd, clips, audio, name, pcfg.MODEL.LATERAL_BOOL, pcfg.MODEL.LATERAL_STRIDE)
            _check_restatement("%s loss" % tag, [loss], [l2], 1e-5)
        else:
            out, loss = ref(clips)
            o2, _ = R.visual_forward(sd, clips, name, pcfg.MODEL.LATERAL_BOOL, pcfg.MODEL.LATERAL_STRIDE)
            loss = torch.zeros(())
    print("  reference forward %.1fs; logsumexp=%.2e" % (time.time() - t0, torch.logsumexp(out[0], (0, 1)).item()))
    _check_restatement(tag, [out], [o2], 5e-5)
    _save(tag, seed=seed, size=H, H=H, W=W, batch=B, wa=wa, num_vis_tokens=nvt, num_aud_tokens=aud_tok,
          swin_depths=np.array(swin_depths if swin_depths is not None else [], dtype=np.int64),
          sd_crc=T.sd_checksum(sd), out=out, loss=loss)


def case_av_x3dl_64():
    _model_case("x3dl", "AudioVisualSaliencyModel", 64, 2, 111, 0, "av_x3dl_64")


def case_av_x3dl_224():
    _model_case("x3dl", "AudioVisualSaliencyModel", 224, 1, 300, 0, "av_x3dl_224")


def case_av_s3d_64():
    """t tokens = 4 (16 frames / 2 in the stem / 2 in maxpooling3), so 4 * (64/32)^2 visual tokens."""
    _model_case("s3d", "AudioVisualSaliencyModel", 64, 2, 111, 0, "av_s3d_64")


def case_vis_x3dl_64():
    _model_case("x3dl", "VisualSaliencyModel", 64, 2, 111, 0, "vis_x3dl_64")


def case_uniformer_backbone(seed=0):
    """Uniformer (backbones/uniformer.py:280-492, UniFormer-B per configs/uniformer_b16x4_k400.yaml) with the product's
    seeded + randomised state dict, strict load.  64x64 batch 2 (128 / 32 tokens in the attention stages) and 224x224
    batch 1 (1568 / 392 tokens)."""
    from mspi_amd.backbones.uniformer import Uniformer
    from mspi_amd.config import cfg as pcfg
    prod = T.condition_(T.seeded(lambda: Uniformer(pcfg.MODEL.UNIFORMER.PATH_CFG), seed), "uniformerb")
    sd = prod.state_dict()
    rcfg = rh.with_config("uniformerb")
    from backbones.uniformer import Uniformer as RefUni
    ref = RefUni(yaml_path=rcfg.MODEL.UNIFORMER.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    for size, B in ((64, 2), (224, 1)):
        clips, _ = T.synth_inputs(B, 16, size, size, seed=seed)
        with torch.no_grad():
            feats = ref([clips])
            ora = R.uniformer_forward(sd, clips)
        scale = max(f.abs().max().item() for f in feats)
        _check_restatement("uniformer-B backbone %d (abs-max %.1f)" % (size, scale), feats, ora, 2e-5 * max(scale, 1.0))
        _save("uniformer_backbone_%d" % size, seed=seed, size=size, batch=B, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_av_slowfast_224():
    """BASELINE configs[2]'s model at full frame size (one clip; Wa = 300)."""
    _model_case("slowfast4x16", "AudioVisualSaliencyModel", 224, 1, 300, 0, "av_slowfast_224")


def case_av_uniformer_224():
    _model_case("uniformerb", "AudioVisualSaliencyModel", 224, 1, 111, 0, "av_uniformer_224")


def case_av_s3d_224():
    _model_case("s3d", "AudioVisualSaliencyModel", 224, 1, 300, 0, "av_s3d_224")


def case_av_uniformer_64():
    _model_case("uniformerb", "AudioVisualSaliencyModel", 64, 2, 111, 0, "av_uniformer_64")


def case_morphmlp_backbone(seed=0):
    """MorphMLP_32_features_only (backbones/MorphMLP.py:371-519, MorphMLP-S per configs/K400_MLP_S16x4.yaml) with the
    product's seeded state dict, strict load.  224x224 only: upstream's reshapes need H*W of every stage to be a
    multiple of its segment_dim (14, 28, 28, 49)."""
    from mspi_amd.backbones.MorphMLP import MorphMLP_32_features_only
    from mspi_amd.config import cfg as pcfg
    prod = T.seeded(lambda: MorphMLP_32_features_only(pcfg.MODEL.MORPH.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("morphmlps")
    from backbones.MorphMLP import MorphMLP_32_features_only as RefMorph
    ref = RefMorph(path_to_config=rcfg.MODEL.MORPH.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(1, 16, 224, 224, seed=seed)
    with torch.no_grad():
        feats = ref(clips)
        ora = R.morphmlp_forward(sd, clips)
    scale = max(f.abs().max().item() for f in feats)
    _check_restatement("morphmlp-S backbone 224 (abs-max %.1f)" % scale, feats, ora, 2e-5 * max(scale, 1.0))
    _save("morphmlp_backbone_224", seed=seed, size=224, batch=1, sd_crc=T.sd_checksum(sd), **_feat_fixture(feats))


def case_av_morphmlp_224():
    _model_case("morphmlps", "AudioVisualSaliencyModel", 224, 1, 111, 0, "av_morphmlp_224")


def case_av_x3dl_64_units():
    """Unit-level outputs of the reference model: forward hooks on the reference's own sub-modules during the
    av_x3dl_64 forward -- image encoder (o1, o0), Adapter masks, audio ResNet-18, SyncBlock tokens, the four lateral layers
    (after SA gating for 0-2 the hook sits on sa_k) and the readout stack -- stored as strided samples (_feat_fixture)."""
    from mspi_amd.model import model_utils as pm
    name, size, B, wa, seed = "x3dl", 64, 2, 111, 0
    nvt, aud_tok = 16 * (size // 32) ** 2, 36
    pcfg = T.make_cfg(name, num_aud_tokens=aud_tok, num_vis_tokens=nvt)
    prod = T.seeded(lambda: pm.AudioVisualSaliencyModel(pcfg), seed)
    sd = prod.state_dict()
    ref = build_reference_model(name, "AudioVisualSaliencyModel", num_vis_tokens=nvt)
    ref.load_state_dict(sd, strict=True)
    got = {}

    def hook(key):
        def fn(mod, inp, out):
            got[key] = [o.detach().clone() for o in out] if isinstance(out, (tuple, list)) else [out.detach().clone()]
        return fn

    names = ["image_encoder", "adapter", "audnet", "aud_vis_sync_block", "latlayer_0", "latlayer_1", "latlayer_2", "latlayer_3",
             "sa_0", "sa_1", "sa_2", "readout"]
    hs = [getattr(ref, n).register_forward_hook(hook(n)) for n in names]
    clips, audio = T.synth_inputs(B, 16, size, size, Wa=wa, seed=seed)
    with torch.no_grad():
        out, loss = ref(clips, audio)
    for h in hs:
        h.remove()
    tensors, keys = [], []
    for n in names:
        for i, t in enumerate(got[n]):
            tensors.append(t.float())
            keys.append("%s.%d" % (n, i))
    fx = _feat_fixture(tensors, max_elems=1 << 12)
    _save("av_x3dl_64_units", seed=seed, size=size, batch=B, wa=wa, num_vis_tokens=nvt, num_aud_tokens=aud_tok,
          sd_crc=T.sd_checksum(sd), keys=np.array(keys), out=out, **fx)
    print("  hooked:", ", ".join("%s%s" % (k, tuple(t.shape)) for k, t in zip(keys, tensors)))


def _hook_outputs(ref, names):
    got = {}

    def hook(key):
        def fn(mod, inp, out):
            outs = out if isinstance(out, (tuple, list)) else [out]
            for i, o in enumerate(outs):
                if torch.is_tensor(o):
                    got["%s.%d" % (key, i) if len(outs) > 1 else key] = o.detach().clone().float()
        return fn

    mods = dict(ref.named_modules())
    hs = [mods[n].register_forward_hook(hook(n)) for n in names]
    return got, hs


def case_backbone_units(seed=0):
    """Unit-level outputs inside the four BASELINE backbones, recorded by forward hooks on the reference's
    own sub-modules: X3D stem; SlowFast stems, both pathways of s2-s4 before each fusion, the fused slow pathway after s1;
    every MViTv2-S block (the seven distinct (stride_q, stride_kv, width) cases among them); every Video-Swin-T block
    (plain and shifted windows) and PatchMerging output."""
    from mspi_amd.config import cfg as pcfg
    from mspi_amd.backbones.X3D import X3D
    from mspi_amd.backbones.sf import SlowFast
    from mspi_amd.backbones.MViT import MViT
    from mspi_amd.backbones.video_swin_transformer import SwinTransformer3D
    out = {}

    def record(tag, ref, names, run, ora_trace):
        got, hs = _hook_outputs(ref, names)
        with torch.no_grad():
            run()
        for h in hs:
            h.remove()
        keys = sorted(got)
        assert set(keys) == set(ora_trace), (sorted(set(keys) ^ set(ora_trace)))
        _check_restatement("%s units" % tag, [got[k] for k in keys], [ora_trace[k].reshape(got[k].shape) for k in keys], 5e-5)
        fx = _feat_fixture([got[k] for k in keys], max_elems=1 << 11)
        out.update({"%s_%s" % (tag, k): v for k, v in fx.items()})
        out[tag + "_keys"] = np.array(keys)

    # X3D-L, 64x64
    prod = T.seeded(lambda: X3D(pcfg.MODEL.X3D.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("x3dl")
    from backbones.X3D import X3D as RefX3D
    ref = RefX3D(path_to_config=rcfg.MODEL.X3D.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    clips, _ = T.synth_inputs(2, 16, 64, 64, seed=seed)
    tr = {}
    with torch.no_grad():
        R.x3d_forward(sd, clips, trace=tr)
    record("x3dl", ref, ["s1"], lambda: ref([clips]), {"s1": tr["s1"]})
    out["x3dl_crc"] = T.sd_checksum(sd)
    # SlowFast, 64x64
    prod = T.seeded(lambda: SlowFast(pcfg.MODEL.SLOWFAST.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("slowfast4x16")
    from backbones.sf import SlowFast as RefSF
    ref = RefSF(path_to_config=rcfg.MODEL.SLOWFAST.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    packed = R.pack_clips("slowfast4x16", clips)
    tr = {}
    with torch.no_grad():
        R.slowfast_forward(sd, packed, trace=tr)
    tr = {k: v for k, v in tr.items()}
    tr["s1_fuse.0"] = tr["s1_fuse.0"]
    record("slowfast", ref, ["s1", "s1_fuse", "s2", "s3", "s4"], lambda: ref(packed),
           {"s1.0": tr["s1.0"], "s1.1": tr["s1.1"], "s1_fuse.0": tr["s1_fuse.0"], "s1_fuse.1": tr["s1.1"],
            "s2.0": tr["s2.0"], "s2.1": tr["s2.1"], "s3.0": tr["s3.0"], "s3.1": tr["s3.1"], "s4.0": tr["s4.0"], "s4.1": tr["s4.1"]})
    out["slowfast_crc"] = T.sd_checksum(sd)
    # MViTv2-S, 224x224 (its relative-position tables are sized for 224)
    prod = T.seeded(lambda: MViT(pcfg.MODEL.MVIT2.PATH_CFG), seed)
    sd = prod.state_dict()
    rcfg = rh.with_config("mvitv2s")
    from backbones.MViT import MViT as RefMViT
    ref = RefMViT(path_to_configs=rcfg.MODEL.MVIT2.PATH_CFG).eval()
    ref.load_state_dict(sd, strict=True)
    clips224, _ = T.synth_inputs(1, 16, 224, 224, seed=seed)
    tr = {}
    with torch.no_grad():
        R.mvit_forward(sd, clips224, R.MVIT_S_ARCH, trace=tr)
    got, hs = _hook_outputs(ref, ["blocks.%d" % i for i in range(16)])
    with torch.no_grad():
        ref([clips224])
    for h in hs:
        h.remove()
    got = {".".join(k.split(".")[:2]): v for k, v in got.items()}       # a block returns (x, thw): "blocks.N.0" -> "blocks.N"
    keys = sorted(got, key=lambda k: int(k.split(".")[1]))
    _check_restatement("mvit units", [got[k] for k in keys], [tr[k] for k in keys], 5e-5)
    fx = _feat_fixture([got[k] for k in keys], max_elems=1 << 11)
    out.update({"mvit_%s" % k: v for k, v in fx.items()})
    out["mvit_keys"] = np.array(keys)
    out["mvit_crc"] = T.sd_checksum(sd)
    # Video-Swin-T, 224x224
    prod = T.seeded(lambda: SwinTransformer3D(depths=[2, 2, 6, 2]), seed)
    sd = prod.state_dict()
    rh.with_config("videoswins")
    from backbones.video_swin_transformer import SwinTransformer3D as RefSwin
    ref = RefSwin(depths=[2, 2, 6, 2])
    ref.eval()
    ref.load_state_dict(sd, strict=True)
    tr = {}
    with torch.no_grad():
        R.swin_forward(sd, clips224, trace=tr)
    names = [k for k in tr]
    record("swin", ref, names, lambda: ref(clips224), tr)
    out["swin_crc"] = T.sd_checksum(sd)
    _save("backbone_units", seed=seed, **out)


def c2_name_corpus():
    """caffe2 blob names of the ResNet / SlowFast / X3D / non-local model-zoo families (R50 depths), plus optimizer
    blobs and a few names that match no rule -- the input side of tests/golden/c2_names.json."""
    names = ["lr", "model_iter", "pred_w", "pred_b", "conv1_xy_w", "conv1_xy_w_momentum", "conv_5_w", "conv_5_bn_s",
             "lin_5_w", "lin_5_b", "res2_0_branch2b_bn_fc1_w", "res2_0_branch2b_bn_fc2_b", "unrelated_blob", "w", "foo_bar_w"]
    bn = ("s", "b", "rm", "riv")
    for pre in ("", "t_"):
        names += [pre + "conv1_w", pre + "conv1_w_momentum", pre + "res_conv1_w"] + [pre + "res_conv1_bn_" + f for f in bn]
        for stage, depth in ((2, 3), (3, 4), (4, 6), (5, 3)):
            for blk in range(depth):
                for letter in "abc":
                    base = "%sres%d_%d_branch2%s" % (pre, stage, blk, letter)
                    names += [base + "_w", base + "_w_momentum"] + [base + "_bn_" + f for f in bn]
                if blk == 0:
                    base = "%sres%d_0_branch1" % (pre, stage)
                    names += [base + "_w"] + [base + "_bn_" + f for f in bn]
    names += ["t_pool1_subsample_w"] + ["t_pool1_subsample_bn_" + f for f in bn]
    for stage, last in ((2, 2), (3, 3), (4, 5)):
        base = "t_res%d_%d_branch2c_bn_subsample" % (stage, last)
        names += [base + "_w"] + [base + "_bn_" + f for f in bn]
    for stage, blk in ((3, 1), (3, 3), (4, 1), (4, 5)):
        for part in ("theta", "g", "phi", "out"):
            names += ["nonlocal_conv%d_%d_%s_w" % (stage, blk, part), "nonlocal_conv%d_%d_%s_b" % (stage, blk, part)]
        names += ["nonlocal_conv%d_%d_bn_%s" % (stage, blk, f) for f in bn]
    return names


def case_c2_names():
    """Blob-name translation of the reference (SlowFast/slowfast/utils/c2_model_loading.py:9-120) on the corpus above."""
    import json
    cwd = os.getcwd()
    rh.enter_reference()
    from SlowFast.slowfast.utils.c2_model_loading import get_name_convert_func
    os.chdir(cwd)
    conv = get_name_convert_func()
    names = c2_name_corpus()
    out = {n: conv(n) for n in names}
    path = os.path.join(GOLD, "c2_names.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("  wrote %s (%d names)" % (path, len(out)))


def case_saliency_metrics(seed=0):
    """utils/compute_saliency_metrics.py kldiv / cc / similarity / nss (batch means) and utils/loss.py's combination on
    seeded maps; `cv2` (imported at the top of that file, used only by auc_judd's resize) is an inert stub here."""
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    cwd = os.getcwd()
    rh.enter_reference()
    from utils import compute_saliency_metrics as M
    os.chdir(cwd)
    g = torch.Generator().manual_seed(seed)
    B, H, W = 3, 24, 40
    logits = torch.rand(B, H, W, generator=g) * 6
    pred = torch.softmax(logits.flatten(1), 1).view(B, H, W)               # a probability map, like exp(model output)
    gt = torch.rand(B, H, W, generator=g) ** 4                             # peaky density
    fix = (torch.rand(B, H, W, generator=g) > 0.97).float()
    ref = torch.stack([M.kldiv(pred, gt), M.cc(pred, gt), M.similarity(pred, gt), M.nss(pred, fix)])
    ora = R.saliency_metrics(pred, gt, fix)
    _check_restatement("saliency_metrics", [ref], [ora.mean(0)], tol=1e-6)
    _save("saliency_metrics", pred=pred, gt=gt, fix=fix, ref_means=ref, per_sample=ora)


CASES = {k[5:]: v for k, v in list(globals().items()) if k.startswith("case_")}

if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        print("[golden] %s" % n)
        CASES[n]()
