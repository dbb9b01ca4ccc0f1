"""Pretrained-weight loading: the caffe2 blob-name parser against the reference's converter, and a
round trip through a synthesised caffe2-format pickle of the SlowFast backbone.  CPU only."""
import contextlib
import io
import json
import os
import pickle

import numpy as np
import pytest
import torch

from mspi_amd import testing as T
from mspi_amd import weights as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_caffe2_names_match_reference_converter():
    """713 blob names (ResNet/SlowFast/X3D/non-local families, optimizer blobs, junk) -> the module paths the
    reference's c2_model_loading.get_name_convert_func() gives (fixture written by oracle/gen_golden.py c2_names)."""
    gold = json.load(open(os.path.join(GOLD, "c2_names.json")))
    assert len(gold) > 700
    bad = {k: (v, W.caffe2_to_pytorch_name(k)) for k, v in gold.items() if W.caffe2_to_pytorch_name(k) != v}
    assert not bad, "first mismatches: %s" % list(bad.items())[:5]


def _slowfast():
    from mspi_amd.model.get_video_backbones import video_motion_extractor
    with contextlib.redirect_stdout(io.StringIO()):
        return video_motion_extractor(T.make_cfg("slowfast4x16"))


def test_slowfast_caffe2_pickle_round_trip(tmp_path):
    """Every tensor of the SlowFast 4x16 R50 backbone survives: state dict -> caffe2 blob names (+ caffe2 shape quirks:
    optimizer blobs, a junk blob) -> pickle -> SlowFast.load_weight('.pkl') on a differently initialised model."""
    src = T.seeded(_slowfast, 3)
    T.randomize_(src, 5)
    sd = src.state_dict()
    blobs = {}
    for k, v in sd.items():
        name = W.pytorch_to_caffe2_name(k, fuse_block={2: 2, 3: 3, 4: 5})
        if name is None:
            assert k.endswith("num_batches_tracked"), k
            continue
        assert W.caffe2_to_pytorch_name(name) == k
        blobs[name] = v.numpy().copy()
    blobs["conv1_w_momentum"] = np.zeros(3, np.float32)
    blobs["lr"] = np.float32(0.1)
    blobs["pred_w"] = np.zeros((400, 2304), np.float32)      # the classification head has no counterpart here
    path = str(tmp_path / "SLOWFAST_4x16_R50.pkl")
    with open(path, "wb") as f:
        pickle.dump({"blobs": blobs}, f, protocol=2)
    dst = T.seeded(_slowfast, 9)
    with contextlib.redirect_stdout(io.StringIO()):
        rep = dst.load_weight(path)
    assert rep["missing"] == [] and rep["shape_mismatch"] == []
    assert rep["unmatched"] == ["pred_w"]
    out = dst.state_dict()
    for k, v in sd.items():
        if not k.endswith("num_batches_tracked"):
            assert torch.equal(out[k], v), k


def test_caffe2_shape_rules():
    """Trailing singleton dims are appended (Linear blob -> 1x1x1 conv weight), BN vectors tile into Sub-BN, a wrong
    shape is reported and skipped (SlowFast/slowfast/utils/checkpoint.py:235-262)."""
    model_state = {"s1.pathway0_stem.conv.weight": torch.zeros(4, 3, 1, 1, 1), "s1.pathway0_stem.bn.weight": torch.zeros(8),
                   "s1.pathway0_stem.bn.split_bn.running_mean": torch.zeros(8), "s2.pathway0_res0.branch1.weight": torch.zeros(2, 2)}
    blobs = {"conv1_w": np.ones((4, 3), np.float32), "res_conv1_bn_s": np.arange(4, dtype=np.float32),
             "res_conv1_bn_rm": np.arange(8, dtype=np.float32), "res2_0_branch1_w": np.ones((3, 3), np.float32)}
    sd, rep = W.convert_caffe2_blobs(blobs, model_state)
    assert tuple(sd["s1.pathway0_stem.conv.weight"].shape) == (4, 3, 1, 1, 1)
    assert sd["s1.pathway0_stem.bn.weight"].tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    assert "s1.pathway0_stem.bn.split_bn.running_mean" in sd
    assert [m[0] for m in rep["shape_mismatch"]] == ["res2_0_branch1_w"]


def test_caffe2_pickle_with_code_is_refused(tmp_path):
    """A checkpoint is a downloaded file: a pickle that names any callable outside numpy's array helpers must not load."""
    class Evil:
        def __reduce__(self):
            import os as _os
            return (_os.system, ("echo pwned > %s" % (tmp_path / "pwned"),))

    for i, payload in enumerate(({"blobs": {"conv1_w": Evil()}}, Evil())):
        path = str(tmp_path / ("evil%d.pkl" % i))
        with open(path, "wb") as f:
            pickle.dump(payload, f, protocol=2)
        with pytest.raises(pickle.UnpicklingError, match="refusing"):
            W.load_caffe2_pkl(path, _slowfast())
    assert not (tmp_path / "pwned").exists()


def _round_trip(make, wrap, path, strict_equal=True, extra=None):
    """state dict of a seeded model -> `wrap`ped into the checkpoint layout upstream's loader expects -> torch.save ->
    load_weight() on a differently seeded model -> tensors equal."""
    src = T.seeded(make, 3)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    ck = wrap(dict(sd))
    if extra:
        ck_inner = ck
        while not any(torch.is_tensor(v) for v in ck_inner.values()):
            ck_inner = next(v for v in ck_inner.values() if isinstance(v, dict))
        ck_inner.update(extra)
    torch.save(ck, path)
    dst = T.seeded(make, 9)
    assert any(not torch.equal(dst.state_dict()[k], v) for k, v in sd.items())
    with contextlib.redirect_stdout(io.StringIO()):
        dst.load_weight(path)
    out = dst.state_dict()
    for k, v in sd.items():
        assert torch.equal(out[k], v), k


def test_x3d_and_mvit_pyth_loaders(tmp_path):
    """backbones/X3D.py:248-250 and MViT.py:2078-2081: the PySlowFast '.pyth' layout {'model_state': state_dict}, loaded
    with strict=False (the released files also carry the classification head, which the feature extractor lacks)."""
    from mspi_amd.backbones.MViT import MViT
    from mspi_amd.backbones.X3D import X3D
    from mspi_amd.config import cfg
    head = {"head.projection.weight": torch.zeros(400, 2048), "head.projection.bias": torch.zeros(400)}
    _round_trip(lambda: X3D(cfg.MODEL.X3D.PATH_CFG), lambda sd: {"model_state": sd, "epoch": 300}, str(tmp_path / "x3d_l.pyth"),
                extra=head)
    _round_trip(lambda: MViT(cfg.MODEL.MVIT2.PATH_CFG), lambda sd: {"model_state": sd}, str(tmp_path / "mvit.pyth"), extra=head)


def test_swin_state_dict_loader(tmp_path):
    """backbones/video_swin_transformer.py:593-605: mmaction layout {'state_dict': {'backbone.<key>': ...,
    'cls_head.<key>': ...}}; only 'backbone.' entries are taken, prefix stripped."""
    from mspi_amd.backbones.video_swin_transformer import SwinTransformer3D

    def wrap(sd):
        ck = {"backbone." + k: v for k, v in sd.items()}
        ck["cls_head.fc_cls.weight"] = torch.zeros(400, 768)
        return {"state_dict": ck, "meta": {"epoch": 30}}

    _round_trip(lambda: SwinTransformer3D(depths=[2, 2, 6, 2]), wrap, str(tmp_path / "swin_tiny.pth"))


def test_plain_state_dict_loaders(tmp_path):
    """s3d.py:420-425 (bare state dict; a missing file raises), uniformer.py:497-500, MorphMLP.py:510-519 (head.* dropped)."""
    from mspi_amd.backbones.MorphMLP import MorphMLP_32_features_only
    from mspi_amd.backbones.s3d import S3D_features_only
    from mspi_amd.backbones.uniformer import Uniformer
    from mspi_amd.config import cfg
    _round_trip(lambda: S3D_features_only(), lambda sd: sd, str(tmp_path / "s3d.pt"))
    with pytest.raises(FileNotFoundError):
        S3D_features_only().load_weight(str(tmp_path / "absent.pt"))
    _round_trip(lambda: Uniformer(cfg.MODEL.UNIFORMER.PATH_CFG), lambda sd: sd, str(tmp_path / "uniformer.pth"))
    _round_trip(lambda: MorphMLP_32_features_only(cfg.MODEL.MORPH.PATH_CFG), lambda sd: sd, str(tmp_path / "morph.pth"),
                extra={"head.weight": torch.zeros(400, 784), "head.bias": torch.zeros(400)})


def test_slowfast_pytorch_checkpoint_loader(tmp_path):
    """mspi_amd/backbones/sf.py: PyTorch-format SlowFast checkpoints ({'model_state': ...}) load without the caffe2 path."""
    _round_trip(_slowfast, lambda sd: {"model_state": sd}, str(tmp_path / "slowfast.pyth"))
