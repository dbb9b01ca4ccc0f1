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
