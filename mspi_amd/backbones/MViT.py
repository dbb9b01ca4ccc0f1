"""MViTv2-S backbone (features only), HIP-backed.  Mirrors the live part of backbones/MViT.py of the
reference (MViT :1658-2081, MultiScaleBlock :1311-1434, MultiScaleAttention :1016-1308, attention_pool
:170-204, cal_rel_pos_* :905-997): same constructor argument, same parameter names
(`patch_embed.proj`, `blocks.N.{norm1,attn.{qkv,proj,pool_{q,k,v},norm_{q,k,v},rel_pos_{h,w,t}},norm2,mlp.fc{1,2},proj}`,
`norm`), same outputs: the token maps after blocks {0,2,13,15} as [B,C,T,H,W].

Per block on channels-last token rows (the '(t h w) c' order IS the CL layout, so no rearrange exists):
  LN -> qkv GEMM -> 3 depthwise 3x3x3 pooling convs over all heads at once (the per-head conv weights are
  shared, so they are tiled over the heads) -> per-head LN(96) -> q/k augmentation with the decomposed
  relative-position terms -> one fused attention (+ residual pooling) -> proj GEMM (+ pooled skip) ->
  LN -> fc1+GELU -> fc2 (+ residual).
Reversible-MViT / cls-token / absolute-position variants are dead for MSPI's config and not built."""
import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import engine as E
from ..backbone_cfg import load_backbone_cfg
from ..module import HipModule


def round_width(width, multiplier, min_width=1, divisor=1):
    if not multiplier:
        return width
    width *= multiplier
    min_width = min_width or divisor
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


def _f(t):
    return t.detach().float().contiguous()


class PatchEmbed(nn.Module):
    def __init__(self, dim_in, dim_out, kernel, stride, padding):
        super().__init__()
        self.proj = nn.Conv3d(dim_in, dim_out, kernel_size=kernel, stride=stride, padding=padding)


REL_GEMM = os.environ.get("MSPI_MVIT_REL_GEMM", "1") != "0"   # A/B switch: relative-position dot products as a GEMM


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features, out_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, out_features)


def _rel_rows(rel_pos, q_size, k_size):
    """(length-matched table [2 max(q,k) - 1, head_dim], index dist[q][k] into it): rel_pos'[dist(q,k)] of
    cal_rel_pos_spatial/temporal (backbones/MViT.py:905-990), with get_rel_pos's linear interpolation (:207-220) when the
    stored table has a different length."""
    d = int(2 * max(q_size, k_size) - 1)
    tab = rel_pos.detach().float()
    if tab.shape[0] != d:
        tab = F.interpolate(tab.reshape(1, tab.shape[0], -1).permute(0, 2, 1), size=d, mode="linear")
        tab = tab.reshape(-1, d).permute(1, 0)
    q_ratio = max(k_size / q_size, 1.0)
    k_ratio = max(q_size / k_size, 1.0)
    dist = torch.arange(q_size)[:, None] * q_ratio - torch.arange(k_size)[None, :] * k_ratio
    dist += (k_size - 1) * k_ratio
    return tab, dist.long()


def _rel_tables(rel_pos, q_size, k_size):
    """Gathered table R[q][k][c] = rel_pos'[dist(q,k)]."""
    tab, dist = _rel_rows(rel_pos, q_size, k_size)
    return tab[dist.to(tab.device)].contiguous()      # [q_size, k_size, head_dim]


class MultiScaleAttention(nn.Module):
    """Parameter holder; arithmetic in MultiScaleBlock.run."""

    def __init__(self, dim, dim_out, input_size, num_heads, qkv_bias, kernel_q, kernel_kv, stride_q, stride_kv):
        super().__init__()
        self.num_heads, self.dim_out = num_heads, dim_out
        hd = dim_out // num_heads
        self.scale = hd ** -0.5
        self.kernel_q, self.kernel_kv = tuple(kernel_q), tuple(kernel_kv)
        self.stride_q, self.stride_kv = tuple(stride_q), tuple(stride_kv)
        self.qkv = nn.Linear(dim, dim_out * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim_out, dim_out)

        def pool(k, s):
            return nn.Conv3d(hd, hd, k, stride=s, padding=[int(v // 2) for v in k], groups=hd, bias=False)

        self.pool_q, self.norm_q = pool(kernel_q, stride_q), nn.LayerNorm(hd, eps=1e-6)
        self.pool_k, self.norm_k = pool(kernel_kv, stride_kv), nn.LayerNorm(hd, eps=1e-6)
        self.pool_v, self.norm_v = pool(kernel_kv, stride_kv), nn.LayerNorm(hd, eps=1e-6)
        assert input_size[1] == input_size[2]
        size = input_size[1]
        rel_sp_dim = 2 * max(size // stride_q[1], size // stride_kv[1]) - 1
        self.rel_pos_h = nn.Parameter(torch.zeros(rel_sp_dim, hd))
        self.rel_pos_w = nn.Parameter(torch.zeros(rel_sp_dim, hd))
        self.rel_pos_t = nn.Parameter(torch.zeros(2 * 8 - 1, hd))
        for p in (self.rel_pos_h, self.rel_pos_w, self.rel_pos_t):
            nn.init.trunc_normal_(p, std=0.02)


class MultiScaleBlock(HipModule):
    def __init__(self, dim, dim_out, num_heads, input_size, mlp_ratio, qkv_bias, kernel_q, kernel_kv, stride_q, stride_kv):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        att_dim = dim_out                       # DIM_MUL_IN_ATT
        self.attn = MultiScaleAttention(dim, att_dim, input_size, num_heads, qkv_bias, kernel_q, kernel_kv, stride_q, stride_kv)
        self.norm2 = nn.LayerNorm(att_dim, eps=1e-6)
        self.mlp = Mlp(att_dim, int(att_dim * mlp_ratio), dim_out)
        if dim != dim_out:
            self.proj = nn.Linear(dim, dim_out)
        self.stride_skip = tuple(stride_q)
        self.kernel_skip = tuple(s + 1 if s > 1 else s for s in stride_q)
        self.pool_skip = (nn.MaxPool3d(self.kernel_skip, self.stride_skip, [int(k // 2) for k in self.kernel_skip])
                          if math.prod(stride_q) > 1 else None)

    def _pack(self):
        a = self.attn
        h = a.num_heads

        def dw(conv):   # the per-head depthwise weights are shared by all heads: tile them
            return E.pack_dwconv(conv.weight.detach().repeat(h, 1, 1, 1, 1), None, None, conv.stride, conv.padding)

        pk = {"n1": (_f(self.norm1.weight), _f(self.norm1.bias)), "n2": (_f(self.norm2.weight), _f(self.norm2.bias)),
              "qkv": E.pack_conv(a.qkv.weight, a.qkv.bias), "proj": E.pack_conv(a.proj.weight, a.proj.bias),
              "pq": dw(a.pool_q), "pk": dw(a.pool_k), "pv": dw(a.pool_v),
              "nq": (_f(a.norm_q.weight), _f(a.norm_q.bias)), "nk": (_f(a.norm_k.weight), _f(a.norm_k.bias)),
              "nv": (_f(a.norm_v.weight), _f(a.norm_v.bias)),
              "mlp": E.pack_mlp_tail(self.mlp.fc1, self.mlp.fc2), "rel": {}}
        if hasattr(self, "proj"):
            pk["skip"] = E.pack_conv(self.proj.weight, self.proj.bias)
        return pk

    def _rel(self, pk, q_thw, k_thw):
        key = (tuple(q_thw), tuple(k_thw))
        if key not in pk["rel"]:
            a = self.attn
            gathered = (_rel_tables(a.rel_pos_h, q_thw[1], k_thw[1]), _rel_tables(a.rel_pos_w, q_thw[2], k_thw[2]),
                        _rel_tables(a.rel_pos_t, q_thw[0], k_thw[0]))
            # GEMM form: all distinct table rows stacked [Dh + Dw + Dt, head_dim] as a Linear layer, plus, per axis, the
            # column of that stack which holds the row for (own position, key position)
            (th, dh), (tw, dw), (tt, dt) = (_rel_rows(a.rel_pos_h, q_thw[1], k_thw[1]), _rel_rows(a.rel_pos_w, q_thw[2], k_thw[2]),
                                            _rel_rows(a.rel_pos_t, q_thw[0], k_thw[0]))
            dev = th.device
            stack = torch.cat([th, tw, tt], 0)
            idx = [d_.to(torch.int32).contiguous().to(dev) + off for d_, off in ((dh, 0), (dw, th.shape[0]), (dt, th.shape[0] + tw.shape[0]))]
            pk["rel"][key] = gathered + ((E.pack_conv(stack, None), idx[0], idx[1], idx[2]),)
        return pk["rel"][key]

    def run(self, x):
        """x: CL [B,T,H,W,dim] (token rows) -> CL [B,T',H',W',dim_out]."""
        pk, a = self.pk, self.attn
        heads, att = a.num_heads, a.dim_out
        hd = att // heads
        B = x.N
        xn = E.layernorm_for_gemm(x, *pk["n1"], 1e-6, *([pk["qkv"], pk["skip"]] if "skip" in pk else [pk["qkv"]]))
        qkv = E.conv(xn, pk["qkv"])                                   # [B,T,H,W, 3*att]: [q | k | v], heads inside
        q = E.dwconv(qkv.slice(0, att), pk["pq"])
        k = E.dwconv(qkv.slice(att, att), pk["pk"])
        v = E.dwconv(qkv.slice(2 * att, att), pk["pv"])
        q_thw, k_thw = (q.T, q.H, q.W), (k.T, k.H, k.W)

        def head_ln(t, gb):   # LayerNorm over each head's 96 channels: rows = (token, head)
            rows = E.CL(t.buf, t.off, t.M * heads, 1, 1, 1, hd, hd)
            E.layernorm(rows, gb[0], gb[1], 1e-6, out=rows)
            return t

        q, k, v = head_ln(q, pk["nq"]), head_ln(k, pk["nk"]), head_ln(v, pk["nv"])
        Rh, Rw, Rt, rel_gemm = self._rel(pk, q_thw, k_thw)
        o = E.mvit_attention(q, k, v, B, heads, hd, a.scale, q_thw, k_thw, Rh, Rw, Rt,
                             rel_gemm=rel_gemm if REL_GEMM and E.DEFAULT_PREC == E.PREC_F16X3 else None)
        skip = E.conv(xn, pk["skip"]) if "skip" in pk else x          # DIM_MUL_IN_ATT: proj(norm1(x)) (MViT.py:1414-1415)
        if self.pool_skip is not None:
            skip = E.maxpool(skip, self.kernel_skip, self.stride_skip, tuple(int(kk // 2) for kk in self.kernel_skip))
        x = E.conv(o, pk["proj"], res=skip)
        return E.mlp_tail(x, pk["mlp"], pk["n2"], 1e-6, res=x)   # dim 96 / 192: one fused launch (mspi_mlp_fwd)


class MViT(HipModule):
    def __init__(self, path_to_configs):
        super().__init__()
        cfg = load_backbone_cfg(path_to_configs[0])
        self.cfg = cfg
        m = cfg.MVIT
        assert m.MODE == "conv" and not m.POOL_FIRST and not m.CLS_EMBED_ON and not m.USE_ABS_POS and m.DIM_MUL_IN_ATT \
            and m.RESIDUAL_POOLING and m.REL_POS_SPATIAL and m.REL_POS_TEMPORAL and not m.SEPARATE_QKV \
            and not m.REV.ENABLE and not m.PATCH_2D and not m.NORM_STEM and m.NORM == "layernorm", \
            "only the MViTv2 (conv pooling, decomposed rel-pos, residual pooling) variant MSPI ships is built"
        self.patch_stride = m.PATCH_STRIDE
        self.patch_embed = PatchEmbed(cfg.DATA.INPUT_CHANNEL_NUM[0], m.EMBED_DIM, m.PATCH_KERNEL, m.PATCH_STRIDE, m.PATCH_PADDING)
        input_size = [cfg.DATA.NUM_FRAMES // m.PATCH_STRIDE[0], cfg.DATA.TRAIN_CROP_SIZE // m.PATCH_STRIDE[1],
                      cfg.DATA.TRAIN_CROP_SIZE // m.PATCH_STRIDE[2]]
        depth = m.DEPTH
        dim_mul, head_mul = [1.0] * (depth + 1), [1.0] * (depth + 1)
        for i, v in m.DIM_MUL:
            dim_mul[i] = v
        for i, v in m.HEAD_MUL:
            head_mul[i] = v
        pool_q = [[] for _ in range(depth)]
        pool_kv = [[] for _ in range(depth)]
        stride_q = [[] for _ in range(depth)]
        stride_kv = [[] for _ in range(depth)]
        for e in m.POOL_Q_STRIDE:
            stride_q[e[0]] = e[1:]
            pool_q[e[0]] = m.POOL_KVQ_KERNEL if m.POOL_KVQ_KERNEL is not None else [s + 1 if s > 1 else s for s in e[1:]]
        kv = list(m.POOL_KV_STRIDE)
        if m.POOL_KV_STRIDE_ADAPTIVE is not None:   # backbones/MViT.py:1801-1812
            _s = list(m.POOL_KV_STRIDE_ADAPTIVE)
            kv = []
            for i in range(depth):
                if len(stride_q[i]) > 0:
                    _s = [max(_s[d] // stride_q[i][d], 1) for d in range(len(_s))]
                kv.append([i] + _s)
        for e in kv:
            stride_kv[e[0]] = e[1:]
            pool_kv[e[0]] = m.POOL_KVQ_KERNEL if m.POOL_KVQ_KERNEL is not None else [s + 1 if s > 1 else s for s in e[1:]]
        embed_dim, num_heads = m.EMBED_DIM, m.NUM_HEADS
        self.blocks = nn.ModuleList()
        for i in range(depth):
            num_heads = round_width(num_heads, head_mul[i])
            dim_out = round_width(embed_dim, dim_mul[i], divisor=round_width(num_heads, head_mul[i]))
            assert len(pool_q[i]) == 3 and len(pool_kv[i]) == 3, "every MViTv2-S block pools q and kv with a 3x3x3 conv"
            self.blocks.append(MultiScaleBlock(embed_dim, dim_out, num_heads, input_size, m.MLP_RATIO, m.QKV_BIAS,
                                               pool_q[i], pool_kv[i], stride_q[i], stride_kv[i]))
            input_size = [s // st for s, st in zip(input_size, stride_q[i])]
            embed_dim = dim_out
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)      # in the state dict, unused by the feature taps
        for mod in self.modules():                            # backbones/MViT.py:1956-1963
            if isinstance(mod, (nn.Linear, nn.Conv3d)):
                nn.init.trunc_normal_(mod.weight, std=0.02)
                if isinstance(mod, nn.Linear) and mod.bias is not None:
                    nn.init.constant_(mod.bias, 0.02)
            elif isinstance(mod, nn.LayerNorm):
                nn.init.constant_(mod.bias, 0.02)
                nn.init.constant_(mod.weight, 1.0)

    def _pack(self):
        c = self.patch_embed.proj
        return E.pack_conv(c.weight, c.bias, None, c.stride, c.padding)

    @torch.no_grad()
    def forward_cl(self, x):
        self._check_eval()
        clips = x[0] if isinstance(x, (list, tuple)) else x
        y = E.conv(clips, self.pk)
        feats = []
        for i, blk in enumerate(self.blocks):
            y = blk.run(y)
            if i in (0, 2, 13, 15):
                feats.append(y)
        return feats

    def forward(self, x, bboxes=None, return_attn=False):
        return [f.as_ncdhw() for f in self.forward_cl(x)]

    def load_weight(self, path):
        weight = torch.load(path, map_location="cpu")["model_state"]
        self.load_state_dict(weight, strict=False)
        print("MViTv2 Weight Loaded!")
