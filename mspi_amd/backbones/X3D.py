"""X3D backbone (features only), HIP-backed.  Mirrors backbones/X3D.py:111-250 of the
reference: same constructor argument, same sub-module names (s1..s5), same outputs
(the last four stage outputs as NCDHW fp32 tensors), `load_weight(path)`."""
import math

import torch

from .. import engine as E
from ..backbone_cfg import load_backbone_cfg
from ..module import HipModule
from . import blocks3d as B

_MODEL_STAGE_DEPTH = {18: (2, 2, 2, 2), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}
_X3D_TEMPORAL_KERNELS = [[[5]], [[3]], [[3]], [[3]], [[3]]]  # conv1, res2..res5


def round_width(width, multiplier, min_width=1, divisor=1):
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


class X3D(HipModule):
    def __init__(self, path_to_config, features_only=True):
        super().__init__()
        cfg = load_backbone_cfg(path_to_config[0])
        assert cfg.MODEL.ARCH == "x3d" and cfg.BN.NORM_TYPE == "batchnorm"
        assert features_only, "only the feature-pyramid form is on MSPI's path"
        self.features_only = features_only
        self.num_pathways = 1
        x = cfg.X3D
        self.dim_c1 = x.DIM_C1
        dim_res2 = round_width(self.dim_c1, 2.0, divisor=8) if x.SCALE_RES2 else self.dim_c1
        dim_res3 = round_width(dim_res2, 2.0, divisor=8)
        dim_res4 = round_width(dim_res3, 2.0, divisor=8)
        dim_res5 = round_width(dim_res4, 2.0, divisor=8)
        block_basis = [[1, dim_res2, 2], [2, dim_res3, 2], [5, dim_res4, 2], [3, dim_res5, 2]]
        w_mul, d_mul = x.WIDTH_FACTOR, x.DEPTH_FACTOR
        dim_res1 = round_width(self.dim_c1, w_mul)
        tk = _X3D_TEMPORAL_KERNELS
        self.s1 = B.VideoModelStem(dim_in=cfg.DATA.INPUT_CHANNEL_NUM, dim_out=[dim_res1], kernel=[tk[0][0] + [3, 3]],
                                   stride=[[1, 2, 2]], padding=[[tk[0][0][0] // 2, 1, 1]], stem_func_name="x3d_stem")
        dim_in = dim_res1
        for stage, (reps, width, stride) in enumerate(block_basis):
            dim_out = round_width(width, w_mul)
            dim_inner = int(x.BOTTLENECK_FACTOR * dim_out)
            n_rep = int(math.ceil(d_mul * reps)) if d_mul else reps
            s = B.ResStage(dim_in=[dim_in], dim_out=[dim_out], dim_inner=[dim_inner], temp_kernel_sizes=tk[1],
                           stride=[stride], num_blocks=[n_rep],
                           num_groups=[dim_inner] if x.CHANNELWISE_3x3x3 else [cfg.RESNET.NUM_GROUPS],
                           num_block_temp_kernel=[n_rep], nonlocal_inds=cfg.NONLOCAL.LOCATION[0],
                           trans_func_name=cfg.RESNET.TRANS_FUNC, stride_1x1=cfg.RESNET.STRIDE_1X1,
                           dilation=cfg.RESNET.SPATIAL_DILATIONS[stage])
            dim_in = dim_out
            self.add_module("s{}".format(stage + 2), s)

    def _stages(self):
        return [self.s2, self.s3, self.s4, self.s5]

    @torch.no_grad()
    def forward_cl(self, x):
        """x: [clips] with clips [N,3,T,H,W] fp32 on the GPU (any strides).  Returns 4 CL features."""
        self._check_eval()
        clips = x[0] if isinstance(x, (list, tuple)) else x
        y = self.s1.run([clips])
        feats = []
        for s in self._stages():
            y = s.run(y)
            feats.append(y[0])
        return feats

    def forward(self, x):
        return [f.as_ncdhw() for f in self.forward_cl(x)]

    def load_weight(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu")["model_state"], strict=False)
        print("LOAD!!!")
