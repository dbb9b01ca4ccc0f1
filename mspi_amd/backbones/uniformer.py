"""UniFormer-B motion encoder (features only), HIP-backed -- SURVEY.md section 8f, rank 4.

Mirrors the reference's backbones/uniformer.py (`Uniformer` :280-492, `CBlock` :117-137, `SABlock` :140-163,
`SpeicalPatchEmbed` :205-231, `PatchEmbed` :234-264, `Attention` :71-96, `Mlp`/`CMlp` :50-68,99-114): same constructor
argument (`yaml_path`), same parameter names (`patch_embed{1..4}.{proj,norm}`, `blocks{1,2}.N.{pos_embed,norm1,conv1,
conv2,attn,norm2,mlp.fc1,mlp.fc2}`, `blocks{3,4}.N.{pos_embed,norm1,attn.{qkv,proj},norm2,mlp.fc1,mlp.fc2}`, `norm`,
`head`), same outputs: the four stage outputs as NCDHW fp32 tensors, `forward(x)` takes the one-element list `[clips]`.

On channels-last rows:
  * `x + pos_embed(x)` is ONE depthwise 3x3x3 launch: the identity is folded into the centre tap;
  * the BatchNorms of a CBlock sit BEFORE a 1x1x1 conv, so they are folded into that conv's input side
    (W' = W diag(s), b' = b + W t) -- no normalisation pass at all in stages 1-2;
  * conv1 -> depthwise 5x5x5 -> conv2 (+x) and fc1 -> GELU -> fc2 (+x) are GEMMs with fused epilogues;
  * stages 3-4 are global self-attention over all T*H*W tokens (1568 / 392 at 224^2) through the fused attention
    kernel (head_dim 64), LN -> qkv, proj (+x), LN -> fc1 + GELU -> fc2 (+x).
`SplitSABlock` (UNIFORMER.SPLIT) is not on MSPI's configuration (`configs/uniformer_b16x4_k400.yaml`: SPLIT False).
"""
import torch
import torch.nn as nn

from .. import engine as E
from .._lib import MspiError
from ..backbone_cfg import load_backbone_cfg
from ..module import HipModule


def _f(t):
    return t.detach().float().contiguous()


def _bn_affine(bn):
    """Eval BatchNorm as y = s*x + t."""
    s = _f(bn.weight) / torch.sqrt(_f(bn.running_var) + bn.eps)
    return s, _f(bn.bias) - _f(bn.running_mean) * s


def _pack_prenorm_conv(bn, conv, act=E.ACT_NONE):
    """conv1x1x1(BN(x)) as one GEMM: the affine is folded into the input side of the weights."""
    s, t = _bn_affine(bn)
    w = _f(conv.weight).reshape(conv.out_channels, conv.in_channels)
    b = _f(conv.bias) + w @ t
    return E.pack_conv(w * s[None, :], b, act=act)


def _pack_pos_embed(conv):
    """x + dwconv3x3x3(x): identity folded into the centre tap."""
    w = _f(conv.weight).clone()
    w[:, 0, 1, 1, 1] += 1.0
    return E.pack_dwconv(w, _f(conv.bias), None, (1, 1, 1), (1, 1, 1))


class CMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Conv3d(dim, hidden, 1)
        self.act = nn.GELU()
        self.fc2 = nn.Conv3d(hidden, dim, 1)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias, qk_scale):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class CBlock(HipModule):
    """Local "attention" block: dw3x3x3 positional conv, BN -> 1x1x1 -> dw5x5x5 -> 1x1x1, BN -> conv MLP."""

    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.pos_embed = nn.Conv3d(dim, dim, 3, 1, 1, groups=dim)
        self.norm1 = nn.BatchNorm3d(dim)
        self.conv1 = nn.Conv3d(dim, dim, 1)
        self.conv2 = nn.Conv3d(dim, dim, 1)
        self.attn = nn.Conv3d(dim, dim, 5, 1, 2, groups=dim)
        self.norm2 = nn.BatchNorm3d(dim)
        self.mlp = CMlp(dim, int(dim * mlp_ratio))

    def _pack(self):
        return {"pos": _pack_pos_embed(self.pos_embed),
                "c1": _pack_prenorm_conv(self.norm1, self.conv1),
                "dw5": E.pack_dwconv(self.attn.weight, self.attn.bias, None, (1, 1, 1), (2, 2, 2)),
                "c2": E.pack_conv(self.conv2.weight, self.conv2.bias),
                "fc1": _pack_prenorm_conv(self.norm2, self.mlp.fc1, act=E.ACT_GELU),
                "fc2": E.pack_conv(self.mlp.fc2.weight, self.mlp.fc2.bias)}

    def run(self, x):
        pk = self.pk
        x = E.dwconv(x, pk["pos"])
        x = E.conv(E.dwconv(E.conv(x, pk["c1"]), pk["dw5"]), pk["c2"], res=x)
        return E.conv(E.conv(x, pk["fc1"]), pk["fc2"], res=x)


class SABlock(HipModule):
    """Global block: dw3x3x3 positional conv, pre-LN multi-head self-attention over all T*H*W tokens, pre-LN MLP."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, qk_scale):
        super().__init__()
        self.dim = dim
        self.pos_embed = nn.Conv3d(dim, dim, 3, 1, 1, groups=dim)
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads, qkv_bias, qk_scale)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _pack(self):
        a, m = self.attn, self.mlp
        return {"pos": _pack_pos_embed(self.pos_embed),
                "n1": (_f(self.norm1.weight), _f(self.norm1.bias)), "n2": (_f(self.norm2.weight), _f(self.norm2.bias)),
                "qkv": E.pack_conv(a.qkv.weight, a.qkv.bias), "proj": E.pack_conv(a.proj.weight, a.proj.bias),
                "mlp": E.pack_mlp_tail(m.fc1, m.fc2)}

    def run(self, x):
        pk, a = self.pk, self.attn
        x = E.dwconv(x, pk["pos"])
        qkv = E.conv(E.layernorm_for_gemm(x, *pk["n1"], self.norm1.eps, pk["qkv"]), pk["qkv"])
        o = E.attention(qkv, x.N, x.T * x.H * x.W, a.num_heads, self.dim // a.num_heads, a.scale)
        x = E.conv(o, pk["proj"], res=x)
        return E.mlp_tail(x, pk["mlp"], pk["n2"], self.norm2.eps, res=x)


class PatchEmbed(HipModule):
    """Strided conv + LayerNorm (eps 1e-5: the reference builds these norms with nn.LayerNorm's default)."""

    def __init__(self, in_chans, embed_dim, kernel, stride, pad):
        super().__init__()
        self.norm = nn.LayerNorm(embed_dim)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel, stride, pad)

    def _pack(self):
        c = self.proj
        return E.pack_conv(c.weight, c.bias, None, c.stride, c.padding), _f(self.norm.weight), _f(self.norm.bias)

    def run(self, x):
        pc, g, b = self.pk
        return E.layernorm(E.conv(x, pc), g, b, self.norm.eps)


class Uniformer(HipModule):
    def __init__(self, yaml_path):
        super().__init__()
        cfg = load_backbone_cfg(yaml_path)
        u = cfg.UNIFORMER
        if u.SPLIT or u.STD or u.REPRESENTATION_SIZE:
            raise MspiError("Uniformer: SPLIT / STD / REPRESENTATION_SIZE variants are not on MSPI's path")
        depth, dims = list(u.DEPTH), list(u.EMBED_DIM)
        heads = [d // u.HEAD_DIM for d in dims]
        in_chans = cfg.DATA.INPUT_CHANNEL_NUM[0]
        self.num_classes = cfg.MODEL.NUM_CLASSES
        self.embed_dim = dims
        self.patch_embed1 = PatchEmbed(in_chans, dims[0], (3, 4, 4), (2, 4, 4), (1, 0, 0))
        self.patch_embed2 = PatchEmbed(dims[0], dims[1], (1, 2, 2), (1, 2, 2), (0, 0, 0))
        self.patch_embed3 = PatchEmbed(dims[1], dims[2], (1, 2, 2), (1, 2, 2), (0, 0, 0))
        self.patch_embed4 = PatchEmbed(dims[2], dims[3], (1, 2, 2), (1, 2, 2), (0, 0, 0))
        self.blocks1 = nn.ModuleList([CBlock(dims[0], u.MLP_RATIO) for _ in range(depth[0])])
        self.blocks2 = nn.ModuleList([CBlock(dims[1], u.MLP_RATIO) for _ in range(depth[1])])
        self.blocks3 = nn.ModuleList([SABlock(dims[2], heads[2], u.MLP_RATIO, u.QKV_BIAS, u.QKV_SCALE) for _ in range(depth[2])])
        self.blocks4 = nn.ModuleList([SABlock(dims[3], heads[3], u.MLP_RATIO, u.QKV_BIAS, u.QKV_SCALE) for _ in range(depth[3])])
        self.norm = nn.BatchNorm3d(dims[-1])                      # in the state dict; the feature outputs never use them
        self.head = nn.Linear(dims[-1], self.num_classes) if self.num_classes > 0 else nn.Identity()

    @torch.no_grad()
    def forward_cl(self, x):
        """x: `[clips]` (reference :478 takes x[0]) or clips [N,3,T,H,W]; T even... H, W multiples of 32."""
        self._check_eval()
        clips = x[0] if isinstance(x, (list, tuple)) else x
        if clips.shape[3] % 32 or clips.shape[4] % 32:
            # the stride-2 patch convs silently drop odd rows/columns upstream; keep the contract explicit here
            raise MspiError("Uniformer: H and W must be multiples of 32, got %s" % (tuple(clips.shape[3:]),))
        feats = []
        y = clips.float()
        for pe, blocks in ((self.patch_embed1, self.blocks1), (self.patch_embed2, self.blocks2),
                           (self.patch_embed3, self.blocks3), (self.patch_embed4, self.blocks4)):
            y = pe.run(y)
            for blk in blocks:
                y = blk.run(y)
            feats.append(y)
        return feats

    def forward(self, x):
        return [f.as_ncdhw() for f in self.forward_cl(x)]

    def load_weight(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu"))
        print("Uniformer Loaded!")
