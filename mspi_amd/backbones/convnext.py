"""ConvNeXt-Tiny image encoder (features_only), HIP-backed.

The reference obtains this network from timm==0.6.12 (`timm.create_model("convnext_tiny",
pretrained=True, features_only=True)`, model/model_utils.py:361); timm is not part of the
reference tree and is unavailable offline, so this file restates the published architecture
with timm's FeatureListNet key names (`stem_0, stem_1, stages_{i}.downsample.{0,1},
stages_{i}.blocks.{j}.{conv_dw,norm,mlp.fc1,mlp.fc2,gamma}`) -- PARITY UNPINNED.

Per block: dw7x7 (dwconv kernel) -> LayerNorm 1e-6 -> fc1+GELU (GEMM epilogue) -> fc2 with the
layer-scale gamma folded into its weights and the shortcut add in its epilogue: 4 launches for
dim >= 384; for dim 96 / 192 everything after the dwconv is ONE launch (mspi_mlp_fwd: the 4C-wide
hidden activation never leaves the CU).
"""
import torch
import torch.nn as nn

from .. import engine as E
from ..module import HipModule

DEPTHS = (3, 3, 9, 3)
DIMS = (96, 192, 384, 768)


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(4 * dim, dim)


class ConvNeXtBlock(HipModule):
    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, 7, 1, 3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))

    def _pack(self):
        return {"dw": E.pack_dwconv(self.conv_dw.weight, self.conv_dw.bias, None, (1, 1, 1), (0, 3, 3)),
                "ln": (self.norm.weight.detach().float().contiguous(), self.norm.bias.detach().float().contiguous()),
                "mlp": E.pack_mlp_tail(self.mlp.fc1, self.mlp.fc2, out_scale=self.gamma)}   # stages 1-2: ONE launch

    def run(self, x):
        pk = self.pk
        return E.mlp_tail(E.dwconv(x, pk["dw"]), pk["mlp"], pk["ln"], 1e-6, res=x)


class ConvNeXtStage(HipModule):
    def __init__(self, dim_in, dim_out, depth, downsample):
        super().__init__()
        if downsample:
            self.downsample = nn.Sequential(nn.LayerNorm(dim_in, eps=1e-6), nn.Conv2d(dim_in, dim_out, 2, 2))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[ConvNeXtBlock(dim_out) for _ in range(depth)])

    def _pack(self):
        if isinstance(self.downsample, nn.Identity):
            return None
        ln, cv = self.downsample[0], self.downsample[1]
        return (ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(),
                E.pack_conv(cv.weight, cv.bias, None, (1, 2, 2), (0, 0, 0)))

    def run(self, x):
        if self.pk is not None:
            g, b, cv = self.pk
            x = E.conv(E.layernorm(x, g, b, 1e-6), cv)
        for blk in self.blocks:
            x = blk.run(x)
        return x


class ConvNeXtTinyFeatures(HipModule):
    """Returns the four stage outputs (strides 4, 8, 16, 32) like timm's FeatureListNet."""

    def __init__(self):
        super().__init__()
        self.stem_0 = nn.Conv2d(3, DIMS[0], 4, 4)
        self.stem_1 = nn.LayerNorm(DIMS[0], eps=1e-6)
        prev = DIMS[0]
        for i, (d, n) in enumerate(zip(DIMS, DEPTHS)):
            self.add_module("stages_%d" % i, ConvNeXtStage(prev, d, n, downsample=i > 0))
            prev = d
        for m in self.modules():  # timm's default init: trunc_normal(.02) weights, zero bias
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def _pack(self):
        return (E.pack_conv(self.stem_0.weight, self.stem_0.bias, None, (1, 4, 4), (0, 0, 0)),
                self.stem_1.weight.detach().float().contiguous(), self.stem_1.bias.detach().float().contiguous())

    @torch.no_grad()
    def forward_cl(self, clips):
        """clips: [B,3,T,H,W] (every frame is an image: the (b t) fold of model/model_utils.py:557
        is the row order of the channels-last output) or [N,3,H,W].  Returns 4 CLs with N*T images."""
        self._check_eval()
        stem, g, b = self.pk
        x = E.conv(clips, stem)
        x = x.reshape(x.N * x.T, 1, x.H, x.W)
        x = E.layernorm(x, g, b, 1e-6)
        outs = []
        for i in range(4):
            x = getattr(self, "stages_%d" % i).run(x)
            outs.append(x)
        return outs

    def forward(self, x):
        return [o.as_ncdhw().squeeze(2) for o in self.forward_cl(x)]
