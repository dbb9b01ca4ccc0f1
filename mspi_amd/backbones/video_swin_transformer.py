"""Video Swin Transformer backbone (features only), HIP-backed.  Mirrors the reference's
backbones/video_swin_transformer.py (SwinTransformer3D :476-713, BasicLayer :349-431,
SwinTransformerBlock3D :193-293, WindowAttention3D :108-190, PatchMerging :296-329, compute_mask :333-346):
same constructor defaults (= Swin-S; pass depths=[2,2,6,2] for Swin-T), same parameter / buffer names
(`patch_embed.proj`, `layers.L.blocks.B.{norm1,attn.{qkv,proj,relative_position_bias_table,relative_position_index},
norm2,mlp.fc{1,2}}`, `layers.L.downsample.{norm,reduction}`, `norm`), same outputs: each stage's pre-merge output.

On channels-last token rows a block is: LN -> qkv GEMM (token-wise, so computed on the UNshifted layout) ->
fused window attention that looks its tokens up through an index table (cyclic shift + window partition and
their inverses are index arithmetic inside the kernel), with the learned relative-position bias and the
shifted-window mask (-100) added key-major -> proj GEMM (+shortcut) -> LN -> fc1+GELU -> fc2 (+residual).
Inputs whose token grid is not a multiple of the (clamped) window, or odd grids at a PatchMerging, would need
the reference's zero padding: not built (224x224 clips never pad)."""
from functools import reduce
from operator import mul

import numpy as np
import torch
import torch.nn as nn

from .. import engine as E
from .._lib import MspiError
from ..module import HipModule


def _f(t):
    return t.detach().float().contiguous()


def get_window_size(x_size, window_size, shift_size=None):
    """Clamp window (and zero the shift) along dims not larger than the window (reference :92-105)."""
    ws = list(window_size)
    ss = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            ws[i] = x_size[i]
            if ss is not None:
                ss[i] = 0
    return tuple(ws) if ss is None else (tuple(ws), tuple(ss))


def compute_mask(D, H, W, window_size, shift_size):
    """[nW, N, N] additive mask (0 / -100) of the shifted-window partition (reference :333-346), on CPU."""
    img = torch.zeros(D, H, W)
    cnt = 0
    for d in (slice(-window_size[0]), slice(-window_size[0], -shift_size[0]), slice(-shift_size[0], None)):
        for h in (slice(-window_size[1]), slice(-window_size[1], -shift_size[1]), slice(-shift_size[1], None)):
            for w in (slice(-window_size[2]), slice(-window_size[2], -shift_size[2]), slice(-shift_size[2], None)):
                img[d, h, w] = cnt
                cnt += 1
    wd, wh, ww = window_size
    mw = img.view(D // wd, wd, H // wh, wh, W // ww, ww).permute(0, 2, 4, 1, 3, 5).reshape(-1, wd * wh * ww)
    diff = mw[:, None, :] - mw[:, :, None]
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0))


def window_token_index(D, H, W, window_size, shift_size, padded=None):
    """int32 [nW, N]: row (within a sample's D*H*W token rows) of token t of window w after the cyclic shift
    roll(x, -shift) and window_partition (reference :61-70,255-262).  padded = (Dp, Hp, Wp): the grid is zero-padded to
    multiples of the window first (reference :240-246); padding positions map to the extra row D*H*W."""
    wd, wh, ww = window_size
    Dp, Hp, Wp = padded if padded is not None else (D, H, W)
    d = (torch.arange(Dp) + shift_size[0]) % Dp
    h = (torch.arange(Hp) + shift_size[1]) % Hp
    w = (torch.arange(Wp) + shift_size[2]) % Wp
    rows = (d[:, None, None] * H + h[None, :, None]) * W + w[None, None, :]          # shifted position -> source row
    inside = (d[:, None, None] < D) & (h[None, :, None] < H) & (w[None, None, :] < W)
    rows = torch.where(inside, rows, torch.full_like(rows, D * H * W))
    rows = rows.view(Dp // wd, wd, Hp // wh, wh, Wp // ww, ww).permute(0, 2, 4, 1, 3, 5).reshape(-1, wd * wh * ww)
    return rows.to(torch.int32).contiguous()


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class WindowAttention3D(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.scale = (dim // num_heads) ** -0.5
        wd, wh, ww = window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += wd - 1
        rel[:, :, 1] += wh - 1
        rel[:, :, 2] += ww - 1
        rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
        rel[:, :, 1] *= 2 * ww - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class SwinTransformerBlock3D(HipModule):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio=4.0, qkv_bias=True):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention3D(dim, window_size, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _pack(self):
        a, m = self.attn, self.mlp
        return {"n1": (_f(self.norm1.weight), _f(self.norm1.bias)), "n2": (_f(self.norm2.weight), _f(self.norm2.bias)),
                "qkv": E.pack_conv(a.qkv.weight, a.qkv.bias), "proj": E.pack_conv(a.proj.weight, a.proj.bias),
                "mlp": E.pack_mlp_tail(m.fc1, m.fc2), "geo": {}}

    def _geometry(self, pk, D, H, W):
        """Per input grid: clamped window, token index table, key-major bias and mask (device tensors)."""
        key = (D, H, W)
        if key not in pk["geo"]:
            ws, ss = get_window_size((D, H, W), self.window_size, self.shift_size)
            Dp, Hp, Wp = (-(-D // ws[0]) * ws[0], -(-H // ws[1]) * ws[1], -(-W // ws[2]) * ws[2])   # reference :240-246
            a = self.attn
            dev = a.qkv.weight.device
            N = reduce(mul, ws)
            idx = a.relative_position_index[:N, :N].reshape(-1).to(a.relative_position_bias_table.device)
            bias = a.relative_position_bias_table.detach().float()[idx].reshape(N, N, -1)      # [q, k, head]
            biasT = bias.permute(2, 1, 0).contiguous().to(dev)                                  # [head, k, q]
            maskT = None
            if any(s > 0 for s in ss):
                maskT = compute_mask(Dp, Hp, Wp, ws, ss).transpose(1, 2).contiguous().to(dev)   # [nW, k, q]
            padded = (Dp, Hp, Wp) != (D, H, W)
            pk["geo"][key] = (N, window_token_index(D, H, W, ws, ss, (Dp, Hp, Wp)).to(dev), biasT, maskT, padded)
        return pk["geo"][key]

    def run(self, x):
        pk, a = self.pk, self.attn
        N, tok_idx, biasT, maskT, padded = self._geometry(pk, x.T, x.H, x.W)
        nwin = tok_idx.shape[0]
        if not padded:
            xn = E.layernorm_for_gemm(x, *pk["n1"], 1e-5, pk["qkv"])
            qkv = E.conv(xn, pk["qkv"])
            o = E.attention(qkv, x.N * nwin, N, a.num_heads, self.dim // a.num_heads, a.scale, biasT=biasT, maskT=maskT,
                            tok_idx=tok_idx)
        else:
            # The reference zero-pads norm1(x) up to a multiple of the window (:240-246), so a padding token enters the
            # attention as qkv(0) = the qkv bias.  All padding tokens share ONE extra row per sample: row D*H*W of the
            # normed buffer is zero, the qkv GEMM turns it into the bias, tok_idx points every padding position at it,
            # and what the attention writes back for those positions lands in that row of `o` and is never read.
            R = x.T * x.H * x.W
            xn = E.alloc(x.N, R + 1, 1, 1, x.C, x.buf.device)
            xn.buf.view(x.N, R + 1, xn.ld)[:, R].zero_()
            E.layernorm(x, *pk["n1"], 1e-5, out=xn.tokens(0, x.T, x.H, x.W))
            qkv = E.conv(xn, pk["qkv"])
            o = E.attention(qkv, x.N * nwin, N, a.num_heads, self.dim // a.num_heads, a.scale, biasT=biasT, maskT=maskT,
                            tok_idx=tok_idx, rows_per_sample=R + 1).tokens(0, x.T, x.H, x.W)
        x = E.conv(o, pk["proj"], res=x)
        return E.mlp_tail(x, pk["mlp"], pk["n2"], 1e-5, res=x)   # dim 96 / 192: one fused launch (mspi_mlp_fwd)


class PatchMerging(HipModule):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def _pack(self):
        return _f(self.norm.weight), _f(self.norm.bias), E.pack_conv(self.reduction.weight, None)

    def run(self, x):
        g, b, red = self.pk
        if x.H % 2 or x.W % 2:
            raise MspiError("Video-Swin PatchMerging on an odd grid %s needs the reference's zero padding: not built" % ((x.H, x.W),))
        return E.conv(E.layernorm(E.space_to_depth(x), g, b, 1e-5), red)


class BasicLayer(HipModule):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, qkv_bias, downsample):
        super().__init__()
        self.window_size = window_size
        self.shift_size = tuple(i // 2 for i in window_size)
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim, num_heads, window_size, (0, 0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio, qkv_bias)
            for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None

    def run(self, x):
        for blk in self.blocks:
            x = blk.run(x)
        return (self.downsample.run(x) if self.downsample is not None else x), x


class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = None


class SwinTransformer3D(HipModule):
    def __init__(self, pretrained=None, pretrained2d=False, patch_size=(2, 4, 4), in_chans=3, embed_dim=96,
                 depths=[2, 2, 18, 2], num_heads=[3, 6, 12, 24], window_size=(8, 7, 7), mlp_ratio=4.0, qkv_bias=True,
                 patch_norm=False):
        super().__init__()
        assert not patch_norm, "MSPI builds SwinTransformer3D() with patch_norm=False"
        self.num_layers = len(depths)
        self.embed_dim, self.window_size, self.patch_size = embed_dim, window_size, patch_size
        self.patch_embed = PatchEmbed3D(patch_size, in_chans, embed_dim)
        self.layers = nn.ModuleList([
            BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias,
                       downsample=i < self.num_layers - 1) for i in range(self.num_layers)])
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.norm = nn.LayerNorm(self.num_features)   # in the state dict; unused by the feature outputs

    def _pack(self):
        c = self.patch_embed.proj
        return E.pack_conv(c.weight, c.bias, None, c.stride, (0, 0, 0))

    @torch.no_grad()
    def forward_cl(self, x):
        """x: clips [N,3,T,H,W] (a bare tensor, as upstream).  T, H, W must be multiples of the patch (2,4,4)."""
        self._check_eval()
        clips = x[0] if isinstance(x, (list, tuple)) else x
        ps = self.patch_size
        if clips.shape[2] % ps[0] or clips.shape[3] % ps[1] or clips.shape[4] % ps[2]:
            raise MspiError("Video-Swin: clip extent must be a multiple of the patch size %s" % (ps,))
        y = E.conv(clips, self.pk)
        feats = []
        for layer in self.layers:
            y, tap = layer.run(y)
            feats.append(tap)
        return feats

    def forward(self, x):
        return [f.as_ncdhw() for f in self.forward_cl(x)]

    def load_weight(self, path=None):
        if path is not None:
            ck = torch.load(path, map_location="cpu")
            self.load_state_dict({k[9:]: v for k, v in ck["state_dict"].items() if "backbone" in k}, strict=False)
