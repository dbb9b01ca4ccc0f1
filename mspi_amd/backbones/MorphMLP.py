"""MorphMLP-S motion encoder (features only), HIP-backed -- SURVEY.md section 8f, rank 4.

Mirrors the reference's backbones/MorphMLP.py (`MorphMLP_32_features_only` :371-519, `PermutatorBlock` :161-187,
`MorphFC_S` :71-113, `MorphFC_S2` :38-68, `MorphFC_T` :116-158, `PatchEmbed` :190-208, `Downsample` :211-225): same
constructor argument (`path_to_config`), same parameter names (`patch_embed1.{proj1,norm1,proj2,norm2}`,
`patch_embed{2,3,4}.{proj,norm}`, `blocks{1..4}.N.{norm1,t_norm1,t_fc.{mlp_t,proj},fc.{mlp_h,mlp_w,mlp_c,reweight.fc1,
reweight.fc2,proj},norm2,mlp.fc1,mlp.fc2}`), same outputs: the four stage outputs as NCDHW fp32 tensors; `forward(x)`
takes the bare clip tensor (model/model_utils.py:527-528).

The model is channels-last upstream ([B,T,H,W,C]) -- exactly this engine's activation layout.  A MorphFC layer is a
Linear over tokens regrouped by `reshape -> permute -> reshape` (chunks of `segment_dim` neighbouring positions along
W, along H, or the T frames, times a 1/segment_dim slice of the channels).  Each regrouping and its inverse is ONE
strided-gather launch (`mspi_permute_fwd`; the index maps are the `*_gather` / `*_scatter` functions below, checked against the
reference's tensor expressions on the host), every Linear is the GEMM kernel, the branch re-weighting
softmax(reweight(mean(h+w+c))) . (h, w, c) is one element-wise launch (`mspi_gated_sum_fwd`), and
norm2 -> fc1 -> GELU -> fc2 -> +x is the usual MLP tail.

Like upstream, the reshapes only work when H*W of every stage is a multiple of its segment_dim (14, 28, 28, 49): 224x224
clips (or multiples); anything else raises here where upstream dies in `reshape`.
"""
import torch
import torch.nn as nn
import yaml

from .. import engine as E
from .._lib import MspiError
from ..backbone_cfg import resolve
from ..module import HipModule


def _f(t):
    return t.detach().float().contiguous()


# ----------------------------------------------------------------------------- regroupings as (dims, strides)
# Each function returns (dims, strides) for engine.permute: out[i...] = src.flat[sum i_k * strides[k]], `out` dense.
# Token tensors are [BT, H*W, C] (BT = B*T frames), C = seg * S with the segment index major (c = g*S + s).
def w_gather(BT, HW, C, sd):
    """MorphFC_S `w` (:94-98): rows (bt, chunk, g), columns (p, s); chunk*sd + p = flattened (h, w) position."""
    S = C // sd
    return (BT, HW // sd, sd, sd, S), (HW * C, sd * C, S, C, 1)


def w_scatter(BT, HW, C, sd):
    """Inverse of w_gather (:99-102): out [bt, chunk, p, g, s] from rows (bt, chunk, g) x columns (p, s)."""
    S = C // sd
    return (BT, HW // sd, sd, sd, S), (HW * C, sd * C, S, C, 1)


def h_gather(BT, H, W, C, sd):
    """MorphFC_S `h` (:84-88): the same grouping on the TRANSPOSED grid, position index q = w*H + h = chunk*sd + p."""
    S = C // sd
    if H % sd == 0:      # a chunk is a run of sd rows inside one column: q -> (w, hq, p), h = hq*sd + p
        return (BT, W, H // sd, sd, sd, S), (H * W * C, C, sd * W * C, S, W * C, 1)
    if sd % H == 0:      # a chunk covers sd/H whole columns: p -> (pw, h), w = chunk*(sd/H) + pw
        pw = sd // H
        return (BT, W // pw, sd, pw, H, S), (H * W * C, pw * C, S, C, W * C, 1)
    raise MspiError("MorphFC_S: segment_dim %d and H %d must divide one another" % (sd, H))


def h_scatter(BT, H, W, C, sd):
    """Inverse of h_gather (:89-92): out [bt, h, w, g, s] from rows (bt, chunk, g) x columns (p, s)."""
    S = C // sd
    if H % sd == 0:      # out dims (bt, hq, p, w, g, s); row = ((bt*W + w)*(H/sd) + hq)*sd + g
        hq = H // sd
        return (BT, hq, sd, W, sd, S), (W * hq * sd * C, sd * C, S, hq * sd * C, C, 1)
    if sd % H == 0:      # out dims (bt, h, chunk, pw, g, s); row = (bt*(W/pw) + chunk)*sd + g, column (pw*H + h)*S + s
        pw = sd // H
        return (BT, H, W // pw, pw, sd, S), ((W // pw) * sd * C, S, sd * C, H * S, C, 1)
    raise MspiError("MorphFC_S: segment_dim %d and H %d must divide one another" % (sd, H))


def s2_gather(BT, HW, C, sd):
    """MorphFC_S2 `h` (:49-54): rows (bt, g, chunk'), columns (a, s); position = a*(HW/sd) + chunk'."""
    S, n = C // sd, HW // sd
    return (BT, sd, n, sd, S), (HW * C, S, C, n * C, 1)


def s2_scatter(BT, HW, C, sd):
    """Inverse of s2_gather (:55-58): out [bt, a, chunk', g, s] from rows (bt, g, chunk') x columns (a, s)."""
    S, n = C // sd, HW // sd
    return (BT, sd, n, sd, S), (sd * n * C, S, C, n * C, 1)


def t_gather(B, T, HW, C, seg=8):
    """MorphFC_T (:134-136): rows (b, g, hw), columns (t, s) with S = C/8."""
    S = C // seg
    return (B, seg, HW, T, S), (T * HW * C, S, C, HW * C, 1)


def t_scatter(B, T, HW, C, seg=8):
    """Inverse of t_gather (:137): out [b, t, hw, g, s] from rows (b, g, hw) x columns (t, s)."""
    S = C // seg
    return (B, T, HW, seg, S), (seg * HW * T * S, S, T * S, HW * T * S, 1)


# ----------------------------------------------------------------------------- parameter holders
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class MorphFC_S(nn.Module):
    branches = 3

    def __init__(self, dim, segment_dim, qkv_bias):
        super().__init__()
        self.segment_dim = segment_dim
        self.mlp_h = nn.Linear(dim, dim, bias=qkv_bias)
        self.mlp_w = nn.Linear(dim, dim, bias=qkv_bias)
        self.mlp_c = nn.Linear(dim, dim, bias=qkv_bias)
        self.reweight = Mlp(dim, dim // 4, dim * 3)
        self.proj = nn.Linear(dim, dim)


class MorphFC_S2(nn.Module):
    branches = 2

    def __init__(self, dim, segment_dim, qkv_bias):
        super().__init__()
        self.segment_dim = segment_dim
        self.mlp_c = nn.Linear(dim, dim, bias=qkv_bias)
        self.mlp_h = nn.Linear(dim, dim, bias=qkv_bias)
        self.reweight = Mlp(dim, dim // 4, dim * 2)
        self.proj = nn.Linear(dim, dim)


class MorphFC_T(nn.Module):
    def __init__(self, dim, qkv_bias):
        super().__init__()
        self.segment_dim = 8                                  # hard-coded upstream (:120)
        self.mlp_t = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


def _rows(flat, M, C):
    """A dense [M, C] device buffer as a one-sample CL of M rows."""
    return E.CL(flat, 0, 1, 1, 1, M, C, C)


class PermutatorBlock(HipModule):
    def __init__(self, dim, segment_dim, mlp_ratio, qkv_bias, mlp_fn):
        super().__init__()
        self.dim = dim
        self.norm1 = nn.LayerNorm(dim)
        self.t_norm1 = nn.LayerNorm(dim)
        self.t_fc = MorphFC_T(dim, qkv_bias)
        self.fc = mlp_fn(dim, segment_dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.skip_lam = 1.0

    def _pack(self):
        lin = lambda m: E.pack_conv(m.weight, m.bias)                 # noqa: E731
        ln = lambda m: (_f(m.weight), _f(m.bias))                     # noqa: E731
        fc, pk = self.fc, {}
        pk["tn"], pk["n1"], pk["n2"] = ln(self.t_norm1), ln(self.norm1), ln(self.norm2)
        pk["mlp_t"], pk["t_proj"] = lin(self.t_fc.mlp_t), lin(self.t_fc.proj)
        pk["mlp_h"], pk["mlp_c"], pk["proj"] = lin(fc.mlp_h), lin(fc.mlp_c), lin(fc.proj)
        if fc.branches == 3:
            pk["mlp_w"] = lin(fc.mlp_w)
        pk["rw1"] = E.pack_conv(fc.reweight.fc1.weight, fc.reweight.fc1.bias, act=E.ACT_GELU)
        pk["rw2"] = E.pack_conv(fc.reweight.fc2.weight, fc.reweight.fc2.bias, cin_stored=pk["rw1"].cout_s)
        pk["mlp"] = E.pack_mlp_tail(self.mlp.fc1, self.mlp.fc2)
        return pk

    def _regroup(self, src, gather, lin, scatter, like):
        """scatter(Linear(gather(src))) as a CL shaped like `like`."""
        Cc = self.dim
        g = E.permute(src, *gather)
        o = E.conv(_rows(g, g.numel() // Cc, Cc), lin)
        return E.CL(E.permute(o, *scatter), 0, like.N, like.T, like.H, like.W, Cc, Cc)


    def run(self, x):
        pk, fc, Cc = self.pk, self.fc, self.dim
        B, T, H, W = x.N, x.T, x.H, x.W
        HW, BT, sd = H * W, B * T, fc.segment_dim
        if T * (Cc // 8) != Cc or Cc % 8 or Cc % sd or HW % sd:
            raise MspiError("MorphMLP block (dim %d, segment_dim %d) cannot regroup a %dx%dx%d grid: upstream needs 8 "
                            "frames after the stem and H*W a multiple of segment_dim (224x224 clips)" % (Cc, sd, T, H, W))
        dev = x.buf.device
        # xt = x + t_fc(t_norm1(x))                                                        (:184)
        tn = E.layernorm(x, *pk["tn"], 1e-5)
        tt = self._regroup(tn, t_gather(B, T, HW, Cc), pk["mlp_t"], t_scatter(B, T, HW, Cc), x)
        xt = E.conv(tt, pk["t_proj"], res=x)
        # x = x + fc(norm1(xt))                                                            (:185; the shortcut is x, not xt)
        n1 = E.layernorm(xt, *pk["n1"], 1e-5)
        c = E.conv(n1, pk["mlp_c"])
        if fc.branches == 3:
            h = self._regroup(n1, h_gather(BT, H, W, Cc, sd), pk["mlp_h"], h_scatter(BT, H, W, Cc, sd), x)
            w = self._regroup(n1, w_gather(BT, HW, Cc, sd), pk["mlp_w"], w_scatter(BT, HW, Cc, sd), x)
            srcs = (h, w, c)
        else:
            h = self._regroup(n1, s2_gather(BT, HW, Cc, sd), pk["mlp_h"], s2_scatter(BT, HW, Cc, sd), x)
            srcs = (h, c)
        means = torch.empty(len(srcs), B, Cc, dtype=torch.float32, device=dev)
        for j, s_ in enumerate(srcs):
            E.mean_rows(s_, B, T * HW, means[j])
        for j in range(1, len(srcs)):
            E.add(means[0], means[j], means[0])
        logit = E.conv(E.conv(E.CL(means[0].reshape(-1), 0, B, 1, 1, 1, Cc, Cc), pk["rw1"]), pk["rw2"])
        mix = E.gated_sum(srcs, logit.buf.view(B, -1))
        x = E.conv(mix, pk["proj"], res=x)
        # x = x + mlp(norm2(x))                                                            (:186)
        return E.mlp_tail(x, pk["mlp"], pk["n2"], 1e-5, res=x)


class PatchEmbed(HipModule):
    """(3,3,3)/(2,2,2) conv + BN + GELU, (1,3,3)/(1,2,2) conv + BN (:190-208)."""

    def __init__(self, in_chans, embed_dim):
        super().__init__()
        self.proj1 = nn.Conv3d(in_chans, embed_dim // 2, (3, 3, 3), (2, 2, 2), (1, 1, 1))
        self.norm1 = nn.BatchNorm3d(embed_dim // 2)
        self.act = nn.GELU()
        self.proj2 = nn.Conv3d(embed_dim // 2, embed_dim, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.norm2 = nn.BatchNorm3d(embed_dim)

    def _pack(self):
        a, b = self.proj1, self.proj2
        p1 = E.pack_conv(a.weight, a.bias, self.norm1, a.stride, a.padding, act=E.ACT_GELU)
        return p1, E.pack_conv(b.weight, b.bias, self.norm2, b.stride, b.padding, cin_stored=p1.cout_s)

    def run(self, clips):
        p1, p2 = self.pk
        return E.conv(E.conv(clips, p1), p2)


class Downsample(HipModule):
    """(1,3,3)/(1,2,2) conv + LayerNorm (:211-225)."""

    def __init__(self, in_embed_dim, out_embed_dim):
        super().__init__()
        self.proj = nn.Conv3d(in_embed_dim, out_embed_dim, (1, 3, 3), (1, 2, 2), (0, 1, 1))
        self.norm = nn.LayerNorm(out_embed_dim)

    def _pack(self):
        c = self.proj
        return E.pack_conv(c.weight, c.bias, None, c.stride, c.padding), _f(self.norm.weight), _f(self.norm.bias)

    def run(self, x):
        pc, g, b = self.pk
        return E.layernorm(E.conv(x, pc), g, b, 1e-5)


class MorphMLP_32_features_only(HipModule):
    def __init__(self, path_to_config):
        super().__init__()
        with open(resolve(path_to_config)) as f:
            cfg = yaml.safe_load(f)
        m = cfg["MORPH"]
        layers, dims, seg, ratios = m["LAYERS"], m["EMBED_DIMS"], m["SEGMENT_DIM"], m["MLP_RATIOS"]
        self.num_classes = cfg["MODEL"]["NUM_CLASSES"]
        self.embed_dims = list(dims)
        if not all(m["TRANSITIONS"]):
            raise MspiError("MorphMLP: only the 4-stage pyramid with a down-sampling transition per stage is built")
        in_chans = cfg["DATA"]["INPUT_CHANNEL_NUM"][0]
        fns = [MorphFC_S] * 3 + [MorphFC_S2]
        self.patch_embed1 = PatchEmbed(in_chans, dims[0])
        for i in range(4):
            if i:
                setattr(self, "patch_embed%d" % (i + 1), Downsample(dims[i - 1], dims[i]))
            setattr(self, "blocks%d" % (i + 1), nn.ModuleList(
                [PermutatorBlock(dims[i], seg[i], ratios[i], m["QKV_BIAS"], fns[i]) for _ in range(layers[i])]))

    @torch.no_grad()
    def forward_cl(self, x):
        """x: clips [N,3,T,H,W] (a bare tensor, as upstream); 16 frames, H*W per stage a multiple of its segment_dim."""
        self._check_eval()
        clips = x[0] if isinstance(x, (list, tuple)) else x
        y = self.patch_embed1.run(clips.float())
        feats = []
        for i in range(1, 5):
            if i > 1:
                y = getattr(self, "patch_embed%d" % i).run(y)
            for blk in getattr(self, "blocks%d" % i):
                y = blk.run(y)
            feats.append(y)
        return feats

    def forward(self, x):
        return [f.as_ncdhw() for f in self.forward_cl(x)]

    def load_weight(self, path):
        ck = torch.load(path, map_location="cpu")
        if self.num_classes != 1000:                      # :505-507
            ck.pop("head.weight", None)
            ck.pop("head.bias", None)
        self.load_state_dict(ck, strict=False)
        print("LOAD")
