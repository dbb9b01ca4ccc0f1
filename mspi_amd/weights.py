"""Pretrained-weight loading.

The reference loads four checkpoint formats:
  * PySlowFast `.pyth` (`{'model_state': ...}`)          -- X3D (backbones/X3D.py:248-250), MViTv2 (MViT.py:2078-2081)
  * mmaction `.pth` (`{'state_dict': ...}`, `backbone.`)  -- Video-Swin (video_swin_transformer.py:593-605)
  * plain state dicts                                     -- MSPI release checkpoints (inference.py:186), VGGSound ResNet-18
  * a caffe2 pickle (`{'blobs': {name: ndarray}}`)        -- SLOWFAST_4x16_R50.pkl (backbones/sf.py:387-388 ->
    SlowFast/slowfast/utils/checkpoint.py:226-292 with the blob-name translation of utils/c2_model_loading.py:9-120)

The first three are `torch.load` + `load_state_dict` in the backbone modules.  This file is the fourth: a blob-name
PARSER (not a rewrite-rule list) that produces the PySlowFast module path of every caffe2 blob, and the loader around
it.  The parser is pinned against the reference's converter on a corpus of blob names (tests/golden/c2_names.json,
written by oracle/gen_golden.py).
"""
import pickle
import re
from collections import OrderedDict

import numpy as np
import torch

_BN_FIELD = {"s": "weight", "b": "bias", "rm": "running_mean", "riv": "running_var"}
_NONLOCAL_CONV = ("theta", "g", "phi", "out")


def _param_suffix(tail, after_bn):
    """Last name component: caffe2 `_w/_b` (and, behind a BN, `_s/_b/_rm/_riv`) -> torch parameter names.  Anything
    else (optimizer blobs such as `w_momentum`) passes through unchanged, as upstream leaves it."""
    if after_bn and tail in _BN_FIELD:
        return _BN_FIELD[tail]
    if tail == "w":
        return "weight"
    if tail == "b":
        return "bias"
    return tail


def caffe2_to_pytorch_name(name):
    """PySlowFast parameter path of caffe2 blob `name` (ResNet / SlowFast / X3D / non-local families).

    Grammar (pathway prefix `t_` = fast pathway -> pathway1, none -> pathway0):
      conv1_<f> | res_conv1_<f> | res_conv1_bn_<f>            stem            s1.pathwayP_stem.{conv,bn}.<f>
      conv1_xy<...>                                            X3D stem        s1.pathway0_stem.conv_xy<...>
      res<S>_<B>_branch<N><l>_<f> | ..._bn_<f>                 block conv/BN   s<S>.pathwayP_res<B>.branch<N>.<l>[_bn].<f>
      res<S>_<B>_branch<N>_<f> | ..._bn_<f>                    shortcut        s<S>.pathwayP_res<B>.branch<N>[_bn].<f>
      t_pool1_subsample[_bn]_<f>                               fusion s1       s1_fuse.{conv_f2s,bn}.<f>
      t_res<S>_<B>_branch2c_bn_subsample[_bn]_<f>              fusion s<S>     s<S>_fuse.{conv_f2s,bn}.<f>
      nonlocal_conv<S>_<B>_<theta|g|phi|out|bn>_<f>            non-local       s<S>.pathway0_nonlocal<B>.{conv_*,bn}.<f>
      pred_<f>, conv_5<...>, lin_5<...>                        heads           head.{projection,conv_5,lin_5}...
    """
    n = name
    m = re.match(r"^nonlocal_conv(\d+)_(\d+)_(.*)$", n)
    if m:
        s, b, rest = m.groups()
        path = "s%s.pathway0_nonlocal%s" % (s, b)
        for c in _NONLOCAL_CONV:
            if rest.startswith(c + "_") or rest == c:
                return _tail_fix("%s.conv_%s%s" % (path, c, rest[len(c):]))
        if rest.startswith("bn_"):
            return "%s.bn.%s" % (path, _param_suffix(rest[3:], True))
        return _tail_fix("%s_%s" % (path, rest))
    m = re.match(r"^t_pool1_subsample_(bn_)?(.*)$", n)
    if m:
        return "s1_fuse.%s.%s" % ("bn" if m.group(1) else "conv_f2s", _param_suffix(m.group(2), bool(m.group(1))))
    m = re.match(r"^t_res(\d+)_(\d+)_branch2c_bn_subsample_(bn_)?(.*)$", n)
    if m:
        return "s%s_fuse.%s.%s" % (m.group(1), "bn" if m.group(3) else "conv_f2s", _param_suffix(m.group(4), bool(m.group(3))))
    fast = n.startswith("t_")
    body = n[2:] if fast else n
    pw = "pathway1" if fast else "pathway0"
    m = re.match(r"^res(\d+)_(\d+)_branch(\d+)([a-z])_(.*)$", body)
    if m:
        s, b, br, letter, rest = m.groups()
        return _tail_fix("s%s.%s_res%s.branch%s.%s_%s" % (s, pw, b, br, letter, rest))
    m = re.match(r"^res(\d+)_(\d+)_branch(\d+)_(.*)$", body)
    if m:
        s, b, br, rest = m.groups()
        return _tail_fix("s%s.%s_res%s.branch%s_%s" % (s, pw, b, br, rest))
    m = re.match(r"^res_conv1_bn_(.*)$", body)
    if m:
        return "s1.%s_stem.bn.%s" % (pw, _param_suffix(m.group(1), True))
    if not fast and body.startswith("conv1_xy"):
        return _tail_fix("s1.pathway0_stem.conv_xy" + body[len("conv1_xy"):])
    m = re.match(r"^(?:res_)?conv1_(.*)$", body)
    if m:
        return _tail_fix("s1.%s_stem.conv.%s" % (pw, m.group(1)))
    return _tail_fix(n)


def _tail_fix(n):
    """Head / squeeze-excite renames and the trailing parameter field, for names that keep caffe2's `_`-joined tail."""
    n = re.sub(r"pred_(.*)", r"head.projection.\1", n)
    n = re.sub(r"(.*)b_bn_fc(.*)", r"\1se.fc\2", n)
    n = re.sub(r"conv_5(.*)", r"head.conv_5\1", n)
    n = re.sub(r"lin_5(.*)", r"head.lin_5\1", n)
    m = re.match(r"^(.*)bn[._](s|b|rm|riv)$", n)
    if m:
        return "%sbn.%s" % (m.group(1), _BN_FIELD[m.group(2)])
    m = re.match(r"^(.*)[._](w|b)$", n)
    if m:
        return "%s.%s" % (m.group(1), "weight" if m.group(2) == "w" else "bias")
    return n


def convert_caffe2_blobs(blobs, model_state):
    """{caffe2 blob name: ndarray} -> (state dict restricted to `model_state`'s keys, report).

    Shape rules of SlowFast/slowfast/utils/checkpoint.py:235-262: trailing singleton dims are appended (Linear -> 1x1x1
    conv), a BN vector that is a whole fraction of the model's is tiled (Sub-BN), everything else must match exactly.
    report = {'loaded': [...], 'shape_mismatch': [(blob, shape, key, shape)], 'unmatched': [blob, ...], 'missing': [key, ...]}
    """
    out = OrderedDict()
    rep = {"loaded": [], "shape_mismatch": [], "unmatched": [], "missing": []}
    for blob, arr in blobs.items():
        key = caffe2_to_pytorch_name(blob)
        if "bn.running_" in key and key not in model_state:
            alt = key.replace("bn.running_", "bn.split_bn.running_")
            key = alt if alt in model_state else key
        if key not in model_state:
            if not any(t in blob for t in ("momentum", "lr", "model_iter")):
                rep["unmatched"].append(blob)
            continue
        want = tuple(model_state[key].shape)
        a = np.asarray(arr)
        if a.ndim < len(want):
            a = a.reshape(a.shape + (1,) * (len(want) - a.ndim))
        if a.ndim == 1 and len(want) == 1 and want[0] > a.shape[0] and want[0] % a.shape[0] == 0:
            a = np.concatenate([a] * (want[0] // a.shape[0]))
        if tuple(a.shape) != want:
            rep["shape_mismatch"].append((blob, tuple(a.shape), key, want))
            continue
        out[key] = torch.tensor(a).clone()
        rep["loaded"].append(key)
    rep["missing"] = sorted(k for k in model_state if k not in out and "num_batches_tracked" not in k)
    return out, rep


class _ArrayOnlyUnpickler(pickle.Unpickler):
    """A model-zoo pickle is a downloaded file: plain `pickle.load` would run whatever callable it names.  A caffe2
    checkpoint is a dict of str/bytes -> numpy arrays (plus scalars), so only numpy's array reconstruction helpers and
    inert builtin containers may be resolved; any other global (os.system, builtins.eval, ...) is refused."""
    _ALLOWED = {
        ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy", "float32"), ("numpy", "float64"), ("numpy", "int32"), ("numpy", "int64"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("collections", "OrderedDict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "str"),
        ("builtins", "bytes"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"),
        ("__builtin__", "dict"), ("__builtin__", "list"), ("__builtin__", "tuple"), ("__builtin__", "str"),
        ("__builtin__", "int"), ("__builtin__", "float"), ("__builtin__", "bool"),
        ("_codecs", "encode"),             # protocol-2 pickles written by python 2 carry array bytes through it
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError("refusing to unpickle %s.%s: a caffe2 checkpoint holds numpy arrays only" % (module, name))


def load_caffe2_pkl(path, model, strict_report=False):
    """Load a caffe2 model-zoo pickle (e.g. SLOWFAST_4x16_R50.pkl) into `model` (load_state_dict(strict=False), like
    upstream).  Returns the report of convert_caffe2_blobs; strict_report=True raises when a model parameter was not
    covered or a blob had the wrong shape."""
    with open(path, "rb") as f:
        ck = _ArrayOnlyUnpickler(f, encoding="latin1").load()
    blobs = ck["blobs"] if isinstance(ck, dict) and "blobs" in ck else ck
    sd, rep = convert_caffe2_blobs(blobs, model.state_dict())
    if strict_report and (rep["missing"] or rep["shape_mismatch"]):
        raise RuntimeError("caffe2 checkpoint does not cover the model: missing %s, shape mismatches %s" % (
            rep["missing"][:5], rep["shape_mismatch"][:3]))
    model.load_state_dict(sd, strict=False)
    return rep


def pytorch_to_caffe2_name(key, fuse_block=None):
    """Inverse of caffe2_to_pytorch_name on the SlowFast/ResNet backbone keys (used to synthesise caffe2-style
    checkpoints for tests and to export weights); returns None for keys caffe2 has no blob for.
    fuse_block: {stage: index of the fast pathway's last block in that stage} (R50: {2: 2, 3: 3, 4: 5}) -- caffe2 names
    the fast-to-slow fusion of stage S after that block."""
    if key.endswith("num_batches_tracked"):
        return None
    field_bn = {v: k for k, v in _BN_FIELD.items()}
    m = re.match(r"^s(\d+)_fuse\.(conv_f2s|bn)\.(\w+)$", key)
    if m:
        s, mod, f = m.groups()
        if s == "1":
            base = "t_pool1_subsample"
        elif fuse_block and int(s) in fuse_block:
            base = "t_res%s_%d_branch2c_bn_subsample" % (s, fuse_block[int(s)])
        else:
            return None
        return "%s_%s" % (base, "bn_" + field_bn[f] if mod == "bn" else {"weight": "w", "bias": "b"}[f])
    m = re.match(r"^s1\.pathway(\d)_stem\.(conv|bn)\.(\w+)$", key)
    if m:
        pw, mod, f = m.groups()
        pre = "t_" if pw == "1" else ""
        return pre + ("res_conv1_bn_" + field_bn[f] if mod == "bn" else "conv1_" + {"weight": "w", "bias": "b"}[f])
    m = re.match(r"^s(\d+)\.pathway(\d)_res(\d+)\.branch(\d)(?:\.([a-z]))?(_bn)?\.(\w+)$", key)
    if m:
        s, pw, b, br, letter, bn, f = m.groups()
        pre = "t_" if pw == "1" else ""
        tail = ("bn_" + field_bn[f]) if bn else {"weight": "w", "bias": "b"}[f]
        return "%sres%s_%s_branch%s%s_%s" % (pre, s, b, br, letter or "", tail)
    return None
