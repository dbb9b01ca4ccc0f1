"""CPU: the C-ABI library loads and exports every symbol include/mspi_hip.h declares; host-side
logic (config surface, factory errors, weight packing, algebraic fusions) without any GPU call."""
import ctypes
import math
import os
import re

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from mspi_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "mspi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mspi_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None
    assert lib.mspi_version() == 2
    assert lib.mspi_last_error() is not None


def test_graft_entry_build_runs_without_gpu():
    """The driver's build check: make + load + ABI version against the header (the assert once went stale on a version bump)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    importlib.import_module("__graft_entry__").build()


def test_descriptor_validation_without_gpu():
    """Bad descriptors are rejected on the host before any launch (no GPU needed)."""
    from mspi_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    rc = lib.mspi_conv_fwd(ctypes.byref(d), None, None, None, None, None, None, None)
    assert rc == -1 and b"null" in lib.mspi_last_error()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    d.N = d.T = d.H = d.W = 1
    d.C = 4
    d.kT = d.kH = d.kW = d.strT = d.strH = d.strW = 1
    d.To, d.Ho, d.Wo, d.Cout = 2, 1, 1, 4       # wrong output extent
    d.ldy = d.ldw = 4
    assert lib.mspi_conv_fwd(ctypes.byref(d), p, p, None, None, None, p, None) == -1
    assert b"does not match" in lib.mspi_last_error()
    a = _lib.AttnDesc()
    a.B = a.Hh = a.Nq = a.Nk = 1
    a.D = a.Dv = 48
    assert lib.mspi_attn_fwd(ctypes.byref(a), p, p, p, None, None, None, None, p, None) == -1
    assert b"not in" in lib.mspi_last_error()
    a.D = a.Dv = 64
    a.prec = 7                                   # neither MSPI_PREC_F32 nor MSPI_PREC_F16X3
    assert lib.mspi_attn_fwd(ctypes.byref(a), p, p, p, None, None, None, None, p, None) == -1
    assert b"prec" in lib.mspi_last_error()
    # fused MLP: only C in {96, 192}, GELU between the layers, LayerNorm needs its parameters
    m = _lib.MlpDesc()
    m.M, m.C, m.hidden, m.ldx, m.ldy, m.act, m.w1_scale, m.w2_scale = 10, 128, 512, 128, 128, 2, 1.0, 1.0
    assert lib.mspi_mlp_fwd(ctypes.byref(m), p, None, None, p, p, p, None, p, None) == -1
    assert b"not supported" in lib.mspi_last_error()
    m.C, m.hidden, m.ldx, m.ldy, m.ln = 96, 384, 96, 96, 1
    assert lib.mspi_mlp_fwd(ctypes.byref(m), p, None, None, p, p, p, None, p, None) == -1
    assert b"gamma" in lib.mspi_last_error()
    m.ln, m.act = 0, 1
    assert lib.mspi_mlp_fwd(ctypes.byref(m), p, None, None, p, p, p, None, p, None) == -1
    assert b"GELU" in lib.mspi_last_error()
    assert lib.mspi_mlp_packed_bytes(96, 384) == 12 * (6 * 2048 + 2 * 3 * 2048)
    # thin GEMM: K <= 224 storage columns, multiples of 4
    assert lib.mspi_rowgemm_supported(216, 96) == 1 and lib.mspi_rowgemm_supported(416, 96) == 0
    assert lib.mspi_rowgemm_packed_bytes(96, 216) == 7 * 8 * 2048
    r = _lib.RowGemmDesc()
    r.M, r.K, r.N, r.ldx, r.ldy, r.w_scale = 5, 416, 96, 416, 96, 1.0
    assert lib.mspi_rowgemm_fwd(ctypes.byref(r), p, p, None, None, None, p, None) == -1
    assert b"outside" in lib.mspi_last_error()
    r.K, r.ldx = 54, 56
    assert lib.mspi_rowgemm_fwd(ctypes.byref(r), p, p, None, None, None, p, None) == -1      # K not a multiple of 4
    assert lib.mspi_saliency_metrics(p, p, None, p, 0, 100, 0, None) == -1                     # empty batch


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from mspi_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MspiError):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    from mspi_amd import engine as E
    from mspi_amd._lib import MspiError
    pk = E.pack_conv(torch.randn(8, 4, 1, 1, 1))
    with pytest.raises(MspiError):
        E.conv(torch.zeros(1, 4, 1, 2, 2), pk)


def test_config_surface_matches_reference_fields():
    from mspi_amd import config as C
    cfg = C.cfg
    assert cfg.DATA.NUM_FRAMES == 16 and cfg.DATA.USE_SOUND is True and tuple(cfg.DATA.RESOLUTION) == (224, 384)
    assert C._MOTION_ENCODERS == ("mvitv2s", "s3d", "slowfast4x16", "morphmlps", "uniformerb", "videoswins", "x3dl")
    assert cfg.MODEL.MOTION_ENCODER_EMBEDS["x3dl"] == (24, 48, 96, 192)
    assert cfg.MODEL.NUM_VIS_TOKENS["mvitv2s"] == 8 * 7 * 12 and cfg.MODEL.NUM_VIS_TOKENS["x3dl"] == 16 * 49
    c = cfg.clone()
    C.select_model("x3dl", c)
    assert c.MODEL.LATERAL_STRIDE == [4, 4, 4, 4] and c.MODEL.LATERAL_BOOL == [True] * 4
    C.select_model("slowfast4x16", c)
    assert c.MODEL.LATERAL_STRIDE == [2, 2, 2, 2] and c.MODEL.LATERAL_BOOL == [False] * 4
    assert cfg.MODEL.MOTION_ENCODER != "slowfast4x16" or os.environ.get("MSPI_MOTION_ENCODER") == "slowfast4x16"
    with pytest.raises(Exception, match="Invalid Motion Encoder!"):
        C.select_model("resnet3d", c)


def test_factory_contract():
    from mspi_amd import testing as T
    from mspi_amd.model.get_video_backbones import video_motion_extractor
    m = video_motion_extractor(T.make_cfg("x3dl"))
    assert hasattr(m, "load_weight") and len(m.state_dict()) == 1137
    c = T.make_cfg("x3dl")
    c.MODEL.MOTION_ENCODER = "bogus"
    with pytest.raises(Exception, match="Invalid Motion Encoder!"):
        video_motion_extractor(c)


def test_train_mode_is_refused():
    from mspi_amd.backbones.resnet import ResNet
    from mspi_amd._lib import MspiError
    with pytest.raises(MspiError, match="eval"):
        ResNet().train().forward_cl(torch.zeros(1, 1, 64, 64))


def test_pack_conv_folds_bn_and_orders_taps():
    from mspi_amd import engine as E
    torch.manual_seed(0)
    conv = nn.Conv3d(6, 10, (3, 1, 2), bias=True)
    bn = nn.BatchNorm3d(10, eps=1e-3).eval()
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2)
        bn.weight.normal_()
        bn.bias.normal_()
    pk = E.pack_conv(conv.weight, conv.bias, bn, cin_stored=8, prec=E.PREC_F32)
    assert pk.w.shape == (12, 3 * 1 * 2 * 8) and pk.cout_s == 12 and pk.ldw % 4 == 0
    ph = E.pack_conv(conv.weight, conv.bias, bn, cin_stored=8, prec=E.PREC_F16X3)
    assert ph.w.dtype == torch.float16 and ph.w.shape == (2, 12, 64) and math.log2(ph.w_scale) % 1 == 0
    rec = (ph.w[0].double() + ph.w[1].double()) / ph.w_scale          # hi + lo reproduces the fp32 weights to 2^-21
    assert (rec[:, :48] - pk.w.double()).abs().max() <= 2.0 ** -20 * pk.w.abs().max()
    x = torch.randn(2, 6, 5, 4, 4)
    ref = bn(conv(x))
    # emulate the kernel's A row layout: (tap, ci) with ci fastest, channels padded to 8
    xp = F.pad(x, (0, 0, 0, 0, 0, 0, 0, 2))
    cols = torch.stack([xp[:, :, t:t + 3, h, w:w + 2].permute(0, 2, 3, 1).reshape(2, -1)
                        for t in range(3) for h in range(4) for w in range(3)], 1)   # [N, To*Ho*Wo, K]
    out = cols @ pk.w.t() + pk.bias
    got = out[:, :, :10].reshape(2, 3, 4, 3, 10).permute(0, 4, 1, 2, 3)
    assert (got - ref).abs().max() < 1e-5
    assert (out[:, :, 10:] == 0).all()


def test_lateral_composition_is_exact():
    from mspi_amd.model.model_utils import _compose_lateral
    torch.manual_seed(1)
    c0 = nn.Conv3d(24, 192, 1)
    c1 = nn.Conv3d(192, 192, (4, 1, 1), stride=(4, 1, 1), bias=False)
    x = torch.randn(1, 24, 16, 5, 5)
    w, b = _compose_lateral(c0, c1)
    with torch.no_grad():
        assert (F.conv3d(x, w, b, (4, 1, 1)) - c1(c0(x))).abs().max() < 1e-5


def test_readout_reorder_is_exact():
    """upsample x4 then (4,1,1)/4 conv + ReLU == conv then upsample then ReLU (both linear; bias commutes)."""
    torch.manual_seed(2)
    c = nn.Conv3d(64, 32, (4, 1, 1), stride=(4, 1, 1))
    x = torch.randn(1, 64, 4, 7, 7)
    up = lambda t: F.interpolate(t, scale_factor=(1, 4, 4), mode="trilinear", align_corners=False)
    with torch.no_grad():
        assert (F.relu(c(up(x))) - F.relu(up(c(x)))).abs().max() < 1e-5


def test_state_dict_roots_match_reference_names():
    from mspi_amd import testing as T
    from mspi_amd.model.model_utils import AudioVisualSaliencyModel
    m = AudioVisualSaliencyModel(T.make_cfg("x3dl"))
    roots = {k.split(".")[0] for k in m.state_dict()}
    assert roots == {"audnet", "image_encoder", "visnet", "aud_vis_sync_block", "vis_projector", "mlp_vis",
                     "aud_projector", "mlp_aud", "latlayer_0", "latlayer_1", "latlayer_2", "latlayer_3", "readout",
                     "adapter", "sa_0", "sa_1", "sa_2"}
    sd = m.state_dict()
    for k in ("visnet.s2.pathway0_res0.branch2.se.fc1.weight", "latlayer_0.2.norm.norm.weight", "readout.12.bias",
              "sa_0.conv_mask.0.bn.running_var", "adapter.conv.branch1.1.conv_t.weight",
              "aud_vis_sync_block.blocks.2.attn.qkv.weight", "image_encoder.smooth_1.1.running_mean",
              "image_encoder.encoder.stages_2.blocks.8.mlp.fc2.bias", "audnet.layer4.0.downsample.0.weight"):
        assert k in sd, k
    assert "aud_vis_sync_block.blocks.0.attn.qkv.bias" not in sd and "aud_vis_sync_block.vis_pos_embed" not in sd


def test_autotune_cache_roundtrip(tmp_path):
    """engine.save_autotune / load_autotune: the shape keys (nested tuples) survive the JSON file."""
    from mspi_amd import engine as E
    saved = dict(E.AUTOTUNE["cache"])
    try:
        E.AUTOTUNE["cache"].clear()
        key = (25088, 768, 192, (1, 1, 1), (1, 1, 1), E.PREC_F16X3, True, False, False)
        E.AUTOTUNE["cache"][key] = 7
        E.save_autotune(str(tmp_path / "t.json"))
        E.AUTOTUNE["cache"].clear()
        assert E.load_autotune(str(tmp_path / "t.json")) == 1
        assert E.AUTOTUNE["cache"] == {key: 7}
    finally:
        E.AUTOTUNE["cache"].clear()
        E.AUTOTUNE["cache"].update(saved)


def test_uniformer_folds_are_exact():
    """UniFormer's two graph-level folds on the host: BatchNorm in front of a 1x1x1 conv folded into the conv's input
    side, and `x + dwconv(x)` as one depthwise conv with the identity in the centre tap."""
    from mspi_amd import engine as E
    from mspi_amd import testing as T
    from mspi_amd.backbones import uniformer as U
    blk = T.seeded(lambda: U.CBlock(8, 4), 3)
    x = torch.rand(2, 8, 3, 5, 5, generator=torch.Generator().manual_seed(1)) - 0.5
    with torch.no_grad():
        ref = blk.conv1(blk.norm1(x))
        pk = U._pack_prenorm_conv(blk.norm1, blk.conv1)
        w = (pk.w[0].double() + pk.w[1].double())[:8, :8] / pk.w_scale if pk.w.dtype == torch.float16 else pk.w[:8, :8].double()
        got = torch.einsum("oc,ncthw->nothw", w, x.double()) + pk.bias[:8].double().view(1, -1, 1, 1, 1)
        assert (got - ref.double()).abs().max() < 1e-5
        ref = x + blk.pos_embed(x)
        pd = U._pack_pos_embed(blk.pos_embed)
        wd = pd.w[:, :8].t().reshape(8, 1, 3, 3, 3)
        got = F.conv3d(x, wd, pd.bias[:8], 1, 1, 1, 8)
        assert (got - ref).abs().max() < 1e-6


def test_uniformer_factory_and_keys():
    from mspi_amd import testing as T
    from mspi_amd.model.get_video_backbones import video_motion_extractor
    m = video_motion_extractor(T.make_cfg("uniformerb"))
    keys = set(m.state_dict())
    assert {"patch_embed1.proj.weight", "patch_embed1.norm.bias", "blocks1.4.attn.weight", "blocks2.7.mlp.fc2.bias",
            "blocks3.19.attn.qkv.bias", "blocks4.6.mlp.fc1.weight", "norm.running_var", "head.weight"} <= keys
    assert m.blocks1[0].attn.weight.shape == (64, 1, 5, 5, 5) and m.blocks3[0].attn.qkv.weight.shape == (960, 320)


def _strided(src, dims, strides):
    return src.reshape(-1).as_strided(tuple(dims), tuple(strides))


@pytest.mark.parametrize("H,W,C,sd", [(56, 56, 112, 14), (28, 28, 224, 28), (14, 14, 392, 28), (28, 14, 56, 14), (7, 14, 56, 14)])
def test_morph_regroupings_match_reference_expressions(H, W, C, sd):
    """The (dims, strides) gathers of mspi_amd.backbones.MorphMLP against the reference's reshape/permute chains
    (backbones/MorphMLP.py:84-102 for MorphFC_S, :134-137 for MorphFC_T), evaluated with as_strided on the host."""
    from mspi_amd.backbones import MorphMLP as M
    B, T, S = 2, 8, C // sd
    x = torch.arange(B * T * H * W * C, dtype=torch.float32).reshape(B, T, H, W, C)
    n = H * W // sd
    # W branch
    w = x.reshape(B, T, n, sd, sd, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, n, sd, sd * S)
    assert torch.equal(_strided(x, *M.w_gather(B * T, H * W, C, sd)).reshape(w.shape), w)
    back = w.reshape(B, T, n, sd, sd, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, H, W, C)
    assert torch.equal(_strided(w.contiguous(), *M.w_scatter(B * T, H * W, C, sd)).reshape(x.shape), back) and torch.equal(back, x)
    # H branch (transposed grid)
    h = x.transpose(3, 2).reshape(B, T, n, sd, sd, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, n, sd, sd * S)
    assert torch.equal(_strided(x, *M.h_gather(B * T, H, W, C, sd)).reshape(h.shape), h)
    back = h.reshape(B, T, n, sd, sd, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, W, H, C).transpose(3, 2)
    assert torch.equal(_strided(h.contiguous(), *M.h_scatter(B * T, H, W, C, sd)).reshape(x.shape), back) and torch.equal(back, x)
    # T branch
    St = C // 8
    t = x.reshape(B, T, H, W, 8, St).permute(0, 4, 2, 3, 1, 5).reshape(B, 8, H, W, T * St)
    assert torch.equal(_strided(x, *M.t_gather(B, T, H * W, C)).reshape(t.shape), t)
    back = t.reshape(B, 8, H, W, T, St).permute(0, 4, 2, 3, 1, 5).reshape(B, T, H, W, C)
    assert torch.equal(_strided(t.contiguous(), *M.t_scatter(B, T, H * W, C)).reshape(x.shape), back)


@pytest.mark.parametrize("H,W,C,sd", [(7, 7, 784, 49), (14, 7, 98, 49)])
def test_morph_s2_regrouping(H, W, C, sd):
    """MorphFC_S2 (backbones/MorphMLP.py:49-58)."""
    from mspi_amd.backbones import MorphMLP as M
    B, T, S, n = 1, 8, C // sd, H * W // sd
    x = torch.arange(B * T * H * W * C, dtype=torch.float32).reshape(B, T, H, W, C)
    h = x.reshape(B, T, sd, n, sd, S).permute(0, 1, 4, 3, 2, 5).reshape(B, T, sd, n, sd * S)
    assert torch.equal(_strided(x, *M.s2_gather(B * T, H * W, C, sd)).reshape(h.shape), h)
    back = h.reshape(B, T, sd, n, sd, S).permute(0, 1, 4, 3, 2, 5).reshape(B, T, H, W, C)
    assert torch.equal(_strided(h.contiguous(), *M.s2_scatter(B * T, H * W, C, sd)).reshape(x.shape), back)


def test_morphmlp_factory_and_keys():
    from mspi_amd import testing as T
    from mspi_amd.model.get_video_backbones import video_motion_extractor
    m = video_motion_extractor(T.make_cfg("morphmlps"))
    keys = set(m.state_dict())
    assert {"patch_embed1.proj1.weight", "patch_embed1.norm2.running_var", "patch_embed4.norm.bias", "blocks1.2.t_fc.mlp_t.bias",
            "blocks3.8.fc.mlp_w.weight", "blocks3.0.fc.reweight.fc2.bias", "blocks4.2.fc.mlp_h.weight", "blocks2.3.mlp.fc1.weight"} <= keys
    assert "blocks4.0.fc.mlp_w.weight" not in keys and m.blocks4[0].fc.reweight.fc2.out_features == 2 * 784


def test_library_has_no_cross_half_packed_fp32():
    """The built gfx950 code must not contain packed-fp32 instructions whose low result reads the HIGH half of a source
    (`v_pk_*_f32 ... op_sel:[..1..]`): on MI355X that form returns wrong low halves ~1e-7 of the time while another
    stream's MFMA kernel shares the CU (tools/pk_overlap_probe.hip).  hipcc's SLP vectoriser emits
    it for broadcast coefficients, which is why csrc/Makefile builds with -fno-slp-vectorize."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_no_packed_f32", os.path.join(root, "tools", "check_no_packed_f32.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    if not os.path.exists(chk.OBJDUMP):
        pytest.skip("llvm-objdump not found")
    n_obj, found = chk.packed_f32(os.path.join(root, "mspi_amd", "csrc", "libmspi_hip.so"))
    assert n_obj >= 1
    bad = [f for f in found if chk.cross_half(f[1])]
    assert not bad, bad[:5]
    assert chk.cross_half("v_pk_mul_f32 v[0:1], v[2:3], v[0:1] op_sel:[0,1] op_sel_hi:[0,1]")
    assert not chk.cross_half("v_pk_add_f32 v[16:17], v[16:17], v[2:3] neg_lo:[0,1] neg_hi:[0,1]")
