"""S3D motion encoder (features only), HIP-backed -- SURVEY.md section 8f, rank 4.

Same module tree and state-dict keys as the reference's `backbones/s3d.py:379-421` (`S3D_features_only`: `base1..base4`
+ the three max-pools, `Mixed_3b .. Mixed_5c` with `branch0..3`), so `S3D_kinetics400_rm_fc.pt` loads unchanged.  Every
conv is the implicit-GEMM kernel with eval-BN (eps 1e-3) + ReLU folded in; the four branches of a Mixed block write
straight into the channel slices of its output (the `torch.cat` of :140 disappears); max-pools are the dw kernel.
"""
import os

import torch
import torch.nn as nn

from .. import engine as E
from ..module import HipModule
from ..model.model_utils import BasicConv3d, SepConv3d

# name: (in, branch0, (branch1 reduce, out), (branch2 reduce, out), branch3)   backbones/s3d.py:118-370
_MIXED = {
    "3b": (192, 64, (96, 128), (16, 32), 32), "3c": (256, 128, (128, 192), (32, 96), 64),
    "4b": (480, 192, (96, 208), (16, 48), 64), "4c": (512, 160, (112, 224), (24, 64), 64),
    "4d": (512, 128, (128, 256), (24, 64), 64), "4e": (512, 112, (144, 288), (32, 64), 64),
    "4f": (528, 256, (160, 320), (32, 128), 128), "5b": (832, 256, (160, 320), (32, 128), 128),
    "5c": (832, 384, (192, 384), (48, 128), 128),
}


def _run_sep(x, pk, out=None):
    return E.conv(E.conv(x, pk[0]), pk[1], out=out)


class Mixed(HipModule):
    def __init__(self, name):
        super().__init__()
        cin, b0, (r1, o1), (r2, o2), b3 = _MIXED[name]
        self.branch0 = nn.Sequential(BasicConv3d(cin, b0, kernel_size=1, stride=1))
        self.branch1 = nn.Sequential(BasicConv3d(cin, r1, kernel_size=1, stride=1), SepConv3d(r1, o1, kernel_size=3, stride=1, padding=1))
        self.branch2 = nn.Sequential(BasicConv3d(cin, r2, kernel_size=1, stride=1), SepConv3d(r2, o2, kernel_size=3, stride=1, padding=1))
        self.branch3 = nn.Sequential(nn.MaxPool3d(kernel_size=(3, 3, 3), stride=1, padding=1), BasicConv3d(cin, b3, kernel_size=1, stride=1))
        self.widths = (b0, o1, o2, b3)

    def _pack(self):
        return {"b0": self.branch0[0].packed(), "b1": (self.branch1[0].packed(), self.branch1[1].packed()),
                "b2": (self.branch2[0].packed(), self.branch2[1].packed()), "b3": self.branch3[1].packed()}

    def run(self, x):
        pk = self.pk
        b0, o1, o2, b3 = self.widths
        out = E.alloc(x.N, x.T, x.H, x.W, b0 + o1 + o2 + b3, x.buf.device)
        E.conv(x, pk["b0"], out=out.slice(0, b0))
        _run_sep(E.conv(x, pk["b1"][0]), pk["b1"][1], out=out.slice(b0, o1))
        _run_sep(E.conv(x, pk["b2"][0]), pk["b2"][1], out=out.slice(b0 + o1, o2))
        E.conv(E.maxpool(x, (3, 3, 3), (1, 1, 1), (1, 1, 1)), pk["b3"], out=out.slice(b0 + o1 + o2, b3))
        return out


class S3D_features_only(HipModule):
    def __init__(self, pool=1):
        super().__init__()
        self.pool = pool
        self.base1 = nn.Sequential(
            SepConv3d(3, 64, kernel_size=7, stride=2, padding=3),
            nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1)),
            BasicConv3d(64, 64, kernel_size=1, stride=1),
            SepConv3d(64, 192, kernel_size=3, stride=1, padding=1))
        self.maxpooling2 = nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.base2 = nn.Sequential(Mixed("3b"), Mixed("3c"))
        self.maxpooling3 = nn.MaxPool3d(kernel_size=(3, 3, 3), stride=(2, 2, 2), padding=(1, 1, 1))
        self.base3 = nn.Sequential(Mixed("4b"), Mixed("4c"), Mixed("4d"), Mixed("4e"), Mixed("4f"))
        self.maxpooling4 = nn.MaxPool3d(kernel_size=(pool, 2, 2), stride=(pool, 2, 2), padding=(0, 0, 0))
        self.base4 = nn.Sequential(Mixed("5b"), Mixed("5c"))

    def _pack(self):
        return {"s0": self.base1[0].packed(), "c2": self.base1[2].packed(), "s3": self.base1[3].packed()}

    @torch.no_grad()
    def forward_cl(self, x):
        """x: [N,3,T,H,W] clips (or the one-element list the factory contract allows) -> 4 CL feature maps."""
        self._check_eval()
        if isinstance(x, (list, tuple)):
            x = x[0]
        pk = self.pk
        y = _run_sep(x.float(), pk["s0"])                              # (1,7,7)/(1,2,2) on the NCDHW input, then (7,1,1)/(2,1,1)
        y = E.maxpool(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        base1 = _run_sep(E.conv(y, pk["c2"]), pk["s3"])
        y = E.maxpool(base1, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        for blk in self.base2:
            y = blk.run(y)
        base2 = y
        y = E.maxpool(base2, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        for blk in self.base3:
            y = blk.run(y)
        base3 = y
        y = E.maxpool(base3, (self.pool, 2, 2), (self.pool, 2, 2), (0, 0, 0))
        for blk in self.base4:
            y = blk.run(y)
        return [base1, base2, base3, y]

    def forward(self, x):
        return [f.as_ncdhw() for f in self.forward_cl(x)]

    def load_weight(self, weight_path):
        if os.path.exists(weight_path):
            self.load_state_dict(torch.load(weight_path, map_location="cpu"))
            print("S3D Weight Loaded!")
        else:
            raise FileNotFoundError("S3D pretrained weight file ?")
