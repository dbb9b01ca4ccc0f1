"""Container-only tooling: import the reference (/root/reference) on CPU.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `oracle/gen_golden.py` (which writes the
fixtures under tests/golden/) uses this module; nothing here ships to the GPU box and
nothing under mspi_amd/ may import it.

The reference imports a handful of third-party packages that are absent in this image
(SURVEY.md section 8c).  None of them carries arithmetic that is on the hot path except
timm's ConvNeXt-T, which cannot be had offline at all ("parity unpinned").
We install inert in-process stand-ins for the *symbols the reference touches at import
time* and leave every line of the reference's own arithmetic untouched.
"""
import ast
import copy
import os
import sys
import types

import torch
import torch.nn as nn
import yaml

REF = os.environ.get("MSPI_REFERENCE", "/root/reference")


class CfgNode(dict):
    """Attr-dict with the slice of fvcore/yacs CfgNode the reference calls
    (SlowFast/slowfast/config/defaults.py, SlowFast/slowfast/utils/parser.py:67-94)."""

    def __init__(self, init=None, **kw):
        super().__init__()
        if init:
            for k, v in init.items():
                self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    @staticmethod
    def _coerce(v):
        # yacs turns "(3, 7, 7)"-style strings into python literals
        if isinstance(v, str):
            try:
                lit = ast.literal_eval(v)
                if isinstance(lit, (tuple, list)):
                    return list(lit)
            except (ValueError, SyntaxError):
                pass
        if isinstance(v, tuple):
            return list(v)
        return v

    def _merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict):
                if k not in self or not isinstance(self[k], CfgNode):
                    self[k] = CfgNode()
                self[k]._merge(v)
            else:
                self[k] = self._coerce(v)

    def merge_from_file(self, path):
        with open(path) as f:
            self._merge(yaml.safe_load(f))

    def merge_from_list(self, lst):
        pass

    def freeze(self):
        pass

    def defrost(self):
        pass


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _DropPath(nn.Module):
    """timm DropPath: identity in eval mode (the only mode the oracle runs)."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not (self.training and self.drop_prob > 0.0), "oracle harness is eval-only"
        return x


def _to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class _PathMgr:
    def exists(self, *_a, **_k):
        return True

    def mkdirs(self, *_a, **_k):
        return None

    def ls(self, *_a, **_k):
        return []

    def open(self, *a, **k):
        return open(*a, **k)


class _PathManagerFactory:
    @staticmethod
    def get(key=None):
        return _PathMgr()


_CREATE_MODEL = {"fn": None}


def set_create_model(fn):
    """Route timm.models.create_model(name, ...) (model/model_utils.py:361) to `fn`."""
    _CREATE_MODEL["fn"] = fn


def _create_model(name, *a, **k):
    if _CREATE_MODEL["fn"] is None:
        raise RuntimeError("timm is absent offline: call set_create_model() first")
    return _CREATE_MODEL["fn"](name, *a, **k)


def install_stubs():
    if "fvcore" in sys.modules and getattr(sys.modules["fvcore"], "_mspi_stub", False):
        return
    _mod("fvcore", _mspi_stub=True)
    _mod("fvcore.common")
    _mod("fvcore.common.config", CfgNode=CfgNode)
    _mod("fvcore.nn", FlopCountAnalysis=None, flop_count_table=None)
    _mod("iopath")
    _mod("iopath.common")
    _mod("iopath.common.file_io", PathManagerFactory=_PathManagerFactory)
    _mod("simplejson", dumps=lambda *a, **k: "")
    _mod("pytorchvideo")
    _mod("pytorchvideo.layers")
    _mod(
        "pytorchvideo.layers.distributed",
        cat_all_gather=None,
        get_local_process_group=None,
        get_local_rank=lambda: 0,
        get_local_size=lambda: 1,
        get_world_size=lambda: 1,
        init_distributed_training=None,
    )

    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    _mod("easydict", EasyDict=EasyDict)
    timm = _mod("timm")
    models = _mod("timm.models", create_model=_create_model)
    timm.models = models
    timm.create_model = _create_model
    _mod("timm.models.layers", to_2tuple=_to_2tuple, DropPath=_DropPath,
         trunc_normal_=torch.nn.init.trunc_normal_)
    _mod("timm.layers", to_2tuple=_to_2tuple, DropPath=_DropPath,
         trunc_normal_=torch.nn.init.trunc_normal_)
    _mod("timm.models.vision_transformer", VisionTransformer=object, _cfg=lambda **k: dict(k))
    _mod("timm.utils")
    _mod("timm.data")
    _mod("timm.data.constants", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406),
         IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    _mod("mmcv")
    _mod("mmcv.utils", get_logger=lambda *a, **k: None)
    _mod("mmcv.runner", load_checkpoint=None)


def enter_reference():
    """sys.path + cwd the way the reference expects (YAML paths are relative, config.py:85-101)."""
    install_stubs()
    sys.dont_write_bytecode = True  # the reference tree is read-only
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)


def with_config(model_name):
    """Import the reference `config` module and re-derive the fields that hang off the
    module-level constant `_model_name` (F12, config.py:59-65) for `model_name`."""
    enter_reference()
    import config as ref_config

    cfg = ref_config.cfg
    assert model_name in ref_config._MOTION_ENCODERS
    cfg.MODEL.MOTION_ENCODER = model_name
    cfg.MODEL.LATERAL_BOOL = ref_config._LATERAL_BOOL[model_name]
    cfg.MODEL.LATERAL_STRIDE = [4, 4, 4, 4] if model_name == "x3dl" else [2, 2, 2, 2]
    cfg.MODEL.MOTION_ENCODER_WEIGHT = ref_config._MOTION_WEIGHTS[model_name]
    return cfg
