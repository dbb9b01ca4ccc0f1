"""ORACLE -- CPU restatement of MSPI's saliency-inference path in plain torch (fp32, functional).

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; mspi_amd/ never does (the product path has no CPU
fallback).  Every function takes a state dict `sd` (reference key names) and a key prefix and
cites the reference lines it restates.

Pinning: the reference ships no tests or golden vectors, so this
restatement is pinned against the reference itself, imported on CPU in the build container by
oracle/gen_golden.py; the outputs are committed under tests/golden/ and
tests/test_oracle_golden.py checks every function here against them.  Exception: ConvNeXt-Tiny
(`convnext_tiny_features`) restates timm==0.6.12's published architecture, which is not in the
reference tree and not importable offline -> PARITY UNPINNED for that one function.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------- helpers
def _bn(sd, p, x, eps):
    """Eval-mode BatchNorm{2,3}d: running stats + affine."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def _conv3(sd, p, x, stride=1, padding=0, groups=1):
    return F.conv3d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, 1, groups)


def _conv2(sd, p, x, stride=1, padding=0, groups=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, 1, groups)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd, p, x, eps=1e-5):
    w = sd[p + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[p + ".bias"], eps)


def _swish(x):
    return x * torch.sigmoid(x)  # SlowFast/resnet_helper.py:76-103


# ------------------------------------------------------------------------------- X3D
def x3d_stem(sd, p, x):
    """SlowFast/stem_helper.py:207-290: conv_xy -> depthwise temporal conv -> BN(1e-5) -> ReLU."""
    k = sd[p + ".conv_xy.weight"].shape
    x = _conv3(sd, p + ".conv_xy", x, (1, 2, 2), (0, k[3] // 2, k[4] // 2))
    kt = sd[p + ".conv.weight"].shape[2]
    x = _conv3(sd, p + ".conv", x, 1, (kt // 2, 0, 0), groups=x.shape[1])
    return F.relu(_bn(sd, p + ".bn", x, 1e-5))


def se_block(sd, p, x):
    """SlowFast/resnet_helper.py:27-73."""
    s = x.mean((2, 3, 4), keepdim=True)
    s = F.relu(_conv3(sd, p + ".fc1", s))
    s = torch.sigmoid(_conv3(sd, p + ".fc2", s))
    return x * s


def x3d_transform(sd, p, x, stride):
    """SlowFast/resnet_helper.py:213-351; child order a,a_bn,a_relu,b,b_bn,[se],b_relu(Swish),c,c_bn."""
    x = F.relu(_bn(sd, p + ".a_bn", _conv3(sd, p + ".a", x), 1e-5))
    kt = sd[p + ".b.weight"].shape[2]
    x = _bn(sd, p + ".b_bn", _conv3(sd, p + ".b", x, (1, stride, stride), (kt // 2, 1, 1), groups=x.shape[1]), 1e-5)
    if p + ".se.fc1.weight" in sd:
        x = se_block(sd, p + ".se", x)
    x = _swish(x)
    return _bn(sd, p + ".c_bn", _conv3(sd, p + ".c", x), 1e-5)


def bottleneck_transform(sd, p, x, stride):
    """SlowFast/resnet_helper.py:354-487 (stride on the 3x3: STRIDE_1X1 False)."""
    kt = sd[p + ".a.weight"].shape[2]
    x = F.relu(_bn(sd, p + ".a_bn", _conv3(sd, p + ".a", x, 1, (kt // 2, 0, 0)), 1e-5))
    x = F.relu(_bn(sd, p + ".b_bn", _conv3(sd, p + ".b", x, (1, stride, stride), (0, 1, 1)), 1e-5))
    return _bn(sd, p + ".c_bn", _conv3(sd, p + ".c", x), 1e-5)


def res_block(sd, p, x, stride, trans):
    """SlowFast/resnet_helper.py:490-616 (eval: no drop-connect)."""
    f = trans(sd, p + ".branch2", x, stride)
    if p + ".branch1.weight" in sd:
        x = _bn(sd, p + ".branch1_bn", _conv3(sd, p + ".branch1", x, (1, stride, stride)), 1e-5)
    return F.relu(x + f)


def res_stage(sd, p, x, pathway, stride, trans):
    i = 0
    while "%s.pathway%d_res%d.branch2.a.weight" % (p, pathway, i) in sd:
        x = res_block(sd, "%s.pathway%d_res%d" % (p, pathway, i), x, stride if i == 0 else 1, trans)
        i += 1
    return x


def x3d_forward(sd, clips, prefix="", trace=None):
    """backbones/X3D.py:236-246: s1..s5, features = outputs of s2..s5.  trace: dict receiving sub-module outputs."""
    x = x3d_stem(sd, prefix + "s1.pathway0_stem", clips)
    if trace is not None:
        trace["s1"] = x
    feats = []
    for s in ("s2", "s3", "s4", "s5"):
        x = res_stage(sd, prefix + s, x, 0, 2, x3d_transform)
        feats.append(x)
    return feats


# ------------------------------------------------------------------------------- SlowFast
def basic_stem(sd, p, x):
    """SlowFast/stem_helper.py:128-204: conv -> BN -> ReLU -> maxpool (1,3,3)/(1,2,2)."""
    k = sd[p + ".conv.weight"].shape
    x = F.relu(_bn(sd, p + ".bn", _conv3(sd, p + ".conv", x, (1, 2, 2), (k[2] // 2, k[3] // 2, k[4] // 2)), 1e-5))
    return F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))


def fuse_fast_to_slow(sd, p, xs, xf, alpha):
    """backbones/sf.py:101-159."""
    k = sd[p + ".conv_f2s.weight"].shape[2]
    f = F.relu(_bn(sd, p + ".bn", _conv3(sd, p + ".conv_f2s", xf, (alpha, 1, 1), (k // 2, 0, 0)), 1e-5))
    return torch.cat([xs, f], 1)


def slowfast_forward(sd, x, prefix="", alpha=4, trace=None):
    """backbones/sf.py:360-385; features = slow pathway after s2/s3/s4 fusion and after s5."""
    p = prefix
    xs, xf = basic_stem(sd, p + "s1.pathway0_stem", x[0]), basic_stem(sd, p + "s1.pathway1_stem", x[1])
    if trace is not None:
        trace["s1.0"], trace["s1.1"] = xs, xf
    xs = fuse_fast_to_slow(sd, p + "s1_fuse", xs, xf, alpha)
    if trace is not None:
        trace["s1_fuse.0"] = xs
    feats = []
    for i, s in enumerate(("s2", "s3", "s4", "s5")):
        stride = 1 if i == 0 else 2
        xs = res_stage(sd, p + s, xs, 0, stride, bottleneck_transform)
        if i < 3:
            xf = res_stage(sd, p + s, xf, 1, stride, bottleneck_transform)
            if trace is not None:
                trace[s + ".0"], trace[s + ".1"] = xs, xf
            xs = fuse_fast_to_slow(sd, p + s + "_fuse", xs, xf, alpha)
        feats.append(xs)
    return feats


# ------------------------------------------------------------------------------- MViTv2
def _mvit_pool(sd, p, x, thw, heads, stride, which):
    """attention_pool with a conv (backbones/MViT.py:170-204): per-head depthwise 3x3x3 conv + LayerNorm(1e-6)."""
    B, N, Cc = x.shape
    hd = Cc // heads
    t = x.reshape(B, N, heads, hd).permute(0, 2, 1, 3)                       # [B, heads, N, hd]
    t = t.reshape(B * heads, thw[0], thw[1], thw[2], hd).permute(0, 4, 1, 2, 3)
    w = sd["%s.pool_%s.weight" % (p, which)]
    t = F.conv3d(t, w, None, stride, [k // 2 for k in w.shape[2:]], 1, hd)
    new_thw = list(t.shape[2:])
    t = t.reshape(B, heads, hd, -1).transpose(2, 3)
    return _ln(sd, "%s.norm_%s" % (p, which), t, 1e-6), new_thw


def _rel_tab(rel_pos, qs, ks):
    d = int(2 * max(qs, ks) - 1)
    if rel_pos.shape[0] != d:   # get_rel_pos, backbones/MViT.py:207-220
        rel_pos = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=d, mode="linear")
        rel_pos = rel_pos.reshape(-1, d).permute(1, 0)
    qr, kr = max(ks / qs, 1.0), max(qs / ks, 1.0)
    dist = torch.arange(qs)[:, None] * qr - torch.arange(ks)[None, :] * kr + (ks - 1) * kr
    return rel_pos[dist.long()]


def mvit_block(sd, p, x, thw, heads, stride_q, stride_kv):
    """MultiScaleBlock + MultiScaleAttention (backbones/MViT.py:1016-1434), MViTv2-S settings: conv pooling,
    decomposed spatial + temporal relative positions, residual pooling, channel expansion inside attention."""
    B, N, dim = x.shape
    xn = _ln(sd, p + ".norm1", x, 1e-6)
    att = sd[p + ".attn.proj.weight"].shape[0]
    hd = att // heads
    qkv = _lin(sd, p + ".attn.qkv", xn).reshape(B, N, 3, att)
    q, q_thw = _mvit_pool(sd, p + ".attn", qkv[:, :, 0], thw, heads, stride_q, "q")
    k, k_thw = _mvit_pool(sd, p + ".attn", qkv[:, :, 1], thw, heads, stride_kv, "k")
    v, _ = _mvit_pool(sd, p + ".attn", qkv[:, :, 2], thw, heads, stride_kv, "v")
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    rq = q.reshape(B, heads, q_thw[0], q_thw[1], q_thw[2], hd)
    Rh = _rel_tab(sd[p + ".attn.rel_pos_h"], q_thw[1], k_thw[1])
    Rw = _rel_tab(sd[p + ".attn.rel_pos_w"], q_thw[2], k_thw[2])
    Rt = _rel_tab(sd[p + ".attn.rel_pos_t"], q_thw[0], k_thw[0])
    rel_h = torch.einsum("bythwc,hkc->bythwk", rq, Rh)
    rel_w = torch.einsum("bythwc,wkc->bythwk", rq, Rw)
    rel_t = torch.einsum("bythwc,tkc->bythwk", rq, Rt)
    attn = (attn.view(B, heads, q_thw[0], q_thw[1], q_thw[2], k_thw[0], k_thw[1], k_thw[2])
            + rel_h[:, :, :, :, :, None, :, None] + rel_w[:, :, :, :, :, None, None, :]
            + rel_t[:, :, :, :, :, :, None, None]).view(B, heads, q.shape[2], k.shape[2])
    o = attn.softmax(-1) @ v + q                                           # residual pooling
    o = _lin(sd, p + ".attn.proj", o.transpose(1, 2).reshape(B, -1, att))
    skip = _lin(sd, p + ".proj", xn) if p + ".proj.weight" in sd else x   # DIM_MUL_IN_ATT (:1414-1415)
    if max(stride_q) > 1:
        ks = [s + 1 if s > 1 else s for s in stride_q]
        t = skip.reshape(B, thw[0], thw[1], thw[2], -1).permute(0, 4, 1, 2, 3)
        t = F.max_pool3d(t, ks, stride_q, [kk // 2 for kk in ks])
        skip = t.flatten(2).transpose(1, 2)
    x = skip + o
    x = x + _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", _ln(sd, p + ".norm2", x, 1e-6))))
    return x, q_thw


def mvit_forward(sd, clips, arch, prefix="", trace=None):
    """backbones/MViT.py:2016-2076.  arch: per block (heads, stride_q, stride_kv); taps after blocks 0,2,13,15."""
    p = prefix
    w = sd[p + "patch_embed.proj.weight"]
    x = F.conv3d(clips, w, sd[p + "patch_embed.proj.bias"], arch["patch_stride"], arch["patch_padding"])
    thw = list(x.shape[2:])
    x = x.flatten(2).transpose(1, 2)
    feats = []
    for i, (heads, sq, skv) in enumerate(arch["blocks"]):
        x, thw = mvit_block(sd, "%sblocks.%d" % (p, i), x, thw, heads, sq, skv)
        if trace is not None:
            trace["blocks.%d" % i] = x
        if i in (0, 2, 13, 15):
            feats.append(x.transpose(1, 2).reshape(x.shape[0], -1, thw[0], thw[1], thw[2]))
    return feats


# ------------------------------------------------------------------------------- Video Swin
def _swin_window_size(x_size, window, shift):
    ws, ss = list(window), list(shift)
    for i in range(3):
        if x_size[i] <= window[i]:
            ws[i], ss[i] = x_size[i], 0
    return tuple(ws), tuple(ss)


def _swin_partition(x, ws):
    B, D, H, W, Cc = x.shape
    x = x.view(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], Cc)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, ws[0] * ws[1] * ws[2], Cc)


def _swin_mask(D, H, W, ws, ss):
    """compute_mask, backbones/video_swin_transformer.py:333-346."""
    img = torch.zeros(1, D, H, W, 1)
    cnt = 0
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[:, d, h, w, :] = cnt
                cnt += 1
    mw = _swin_partition(img, ws).squeeze(-1)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def swin_block(sd, p, x, heads, window, shift, mask):
    """SwinTransformerBlock3D + WindowAttention3D (backbones/video_swin_transformer.py:108-293), with the zero padding
    of the normed tokens up to a multiple of the window (:240-246) and the crop after the window merge (:271-274)."""
    B, D0, H0, W0, Cc = x.shape
    ws, ss = _swin_window_size((D0, H0, W0), window, shift)
    h = _ln(sd, p + ".norm1", x)
    h = F.pad(h, (0, 0, 0, (ws[2] - W0 % ws[2]) % ws[2], 0, (ws[1] - H0 % ws[1]) % ws[1], 0, (ws[0] - D0 % ws[0]) % ws[0]))
    _, D, H, W, _ = h.shape
    shifted = any(i > 0 for i in ss)
    if shifted:
        h = torch.roll(h, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    xw = _swin_partition(h, ws)
    B_, N, _ = xw.shape
    qkv = _lin(sd, p + ".attn.qkv", xw).reshape(B_, N, 3, heads, Cc // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (Cc // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = sd[p + ".attn.relative_position_index"][:N, :N].reshape(-1)
    bias = sd[p + ".attn.relative_position_bias_table"][idx].reshape(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shifted:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    o = _lin(sd, p + ".attn.proj", (attn.softmax(-1) @ v).transpose(1, 2).reshape(B_, N, Cc))
    o = o.view(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], Cc).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, Cc)
    if shifted:
        o = torch.roll(o, shifts=ss, dims=(1, 2, 3))
    x = x + o[:, :D0, :H0, :W0]
    return x + _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", _ln(sd, p + ".norm2", x))))


def swin_forward(sd, clips, prefix="", window=(8, 7, 7), heads=(3, 6, 12, 24), trace=None):
    """SwinTransformer3D.forward, backbones/video_swin_transformer.py:692-708: every stage's pre-merge output."""
    p = prefix
    w = sd[p + "patch_embed.proj.weight"]
    x = F.conv3d(clips, w, sd[p + "patch_embed.proj.bias"], w.shape[2:]).permute(0, 2, 3, 4, 1)
    shift = tuple(i // 2 for i in window)
    feats = []
    li = 0
    while "%slayers.%d.blocks.0.norm1.weight" % (p, li) in sd:
        B, D, H, W, Cc = x.shape
        ws, ss = _swin_window_size((D, H, W), window, shift)
        mask = _swin_mask(-(-D // ws[0]) * ws[0], -(-H // ws[1]) * ws[1], -(-W // ws[2]) * ws[2], ws, ss)   # :419-423
        bi = 0
        while "%slayers.%d.blocks.%d.norm1.weight" % (p, li, bi) in sd:
            x = swin_block(sd, "%slayers.%d.blocks.%d" % (p, li, bi), x, heads[li], window,
                           (0, 0, 0) if bi % 2 == 0 else shift, mask)
            if trace is not None:
                trace["layers.%d.blocks.%d" % (li, bi)] = x
            bi += 1
        feats.append(x.permute(0, 4, 1, 2, 3))
        q = "%slayers.%d.downsample" % (p, li)
        if q + ".reduction.weight" in sd:   # PatchMerging :296-329 (even grids)
            x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
            x = _lin(sd, q + ".reduction", _ln(sd, q + ".norm", x))
            if trace is not None:
                trace["layers.%d.downsample" % li] = x
        li += 1
    return feats


# ------------------------------------------------------------------------------- audio ResNet-18
def resnet18_forward(sd, x, prefix=""):
    """backbones/resnet.py:57-143 (1-channel stem, BasicBlock x [2,2,2,2], returns layer4 map)."""
    p = prefix
    x = F.relu(_bn(sd, p + "bn1", _conv2(sd, p + "conv1", x, 2, 3), 1e-5))
    x = F.max_pool2d(x, 3, 2, 1)
    for li in range(1, 5):
        for bi in range(2):
            q = "%slayer%d.%d" % (p, li, bi)
            stride = 2 if (li > 1 and bi == 0) else 1
            idt = x
            o = F.relu(_bn(sd, q + ".bn1", _conv2(sd, q + ".conv1", x, stride, 1), 1e-5))
            o = _bn(sd, q + ".bn2", _conv2(sd, q + ".conv2", o, 1, 1), 1e-5)
            if q + ".downsample.0.weight" in sd:
                idt = _bn(sd, q + ".downsample.1", _conv2(sd, q + ".downsample.0", x, stride, 0), 1e-5)
            x = F.relu(o + idt)
    return x


# ------------------------------------------------------------------------------- ConvNeXt-T (timm 0.6.12) -- PARITY UNPINNED
def convnext_tiny_features(sd, x, prefix=""):
    """timm==0.6.12 `convnext_tiny`, features_only=True (call site model/model_utils.py:361,380).
    Published architecture: stem conv4x4/4 + LayerNorm2d; stages depths (3,3,9,3), dims
    (96,192,384,768); block = dw7x7 -> LN(1e-6) -> fc1 -> GELU -> fc2 -> gamma -> + shortcut;
    downsample = LayerNorm2d + conv2x2/2.  Returns the 4 stage outputs (strides 4,8,16,32)."""
    p = prefix

    def ln2d(q, t):
        return _ln(sd, q, t.permute(0, 2, 3, 1), 1e-6).permute(0, 3, 1, 2)

    x = ln2d(p + "stem_1", _conv2(sd, p + "stem_0", x, 4, 0))
    outs = []
    for si, depth in enumerate((3, 3, 9, 3)):
        sp = "%sstages_%d" % (p, si)
        if si > 0:
            x = _conv2(sd, sp + ".downsample.1", ln2d(sp + ".downsample.0", x), 2, 0)
        for bi in range(depth):
            q = "%s.blocks.%d" % (sp, bi)
            y = _conv2(sd, q + ".conv_dw", x, 1, 3, groups=x.shape[1]).permute(0, 2, 3, 1)
            y = _ln(sd, q + ".norm", y, 1e-6)
            y = _lin(sd, q + ".mlp.fc2", F.gelu(_lin(sd, q + ".mlp.fc1", y)))
            y = (y * sd[q + ".gamma"]).permute(0, 3, 1, 2)
            x = x + y
        outs.append(x)
    return outs


def static_saliency_encoder(sd, frames, prefix="image_encoder."):
    """StaticSaliencyModelConvNext.forward, model/model_utils.py:379-385."""
    o3, o2, o1, o0 = convnext_tiny_features(sd, frames, prefix + "encoder.")
    p = prefix
    o0 = F.relu(_bn(sd, p + "smooth_0.1", _conv2(sd, p + "smooth_0.0", o0, 1, 1), 1e-5))
    o1 = F.relu(_bn(sd, p + "smooth_1.1", _conv2(sd, p + "smooth_1.0", o1, 1, 1), 1e-5))
    return o1, o0


# ------------------------------------------------------------------------------- head pieces
def basic_conv3d(sd, p, x, padding):
    """backbones/s3d.py:41-52 (BN eps 1e-3)."""
    return F.relu(_bn(sd, p + ".bn", _conv3(sd, p + ".conv", x, 1, padding), 1e-3))


def sep_conv3d(sd, p, x):
    """backbones/s3d.py:95-116, kernel 3 stride 1 padding 1."""
    x = F.relu(_bn(sd, p + ".bn_s", _conv3(sd, p + ".conv_s", x, 1, (0, 1, 1)), 1e-3))
    return F.relu(_bn(sd, p + ".bn_t", _conv3(sd, p + ".conv_t", x, 1, (1, 0, 0)), 1e-3))


def inception(sd, p, x):
    """model/model_utils.py:173-199."""
    x0 = basic_conv3d(sd, p + ".branch0.0", x, 0)
    x1 = sep_conv3d(sd, p + ".branch1.1", basic_conv3d(sd, p + ".branch1.0", x, 0))
    x2 = sep_conv3d(sd, p + ".branch2.1", basic_conv3d(sd, p + ".branch2.0", x, 0))
    x3 = basic_conv3d(sd, p + ".branch3.1", F.max_pool3d(x, 3, 1, 1), 0)
    return torch.cat((x0, x1, x2, x3), 1)


def _up(x, k):
    return F.interpolate(x, scale_factor=(1, k, k), mode="trilinear", align_corners=False)


def adapter(sd, p, o3, o2, num_frames, stride):
    """model/model_utils.py:202-220."""
    def unfold(t):
        bt, c, h, w = t.shape
        return t.view(bt // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)

    o3_ = F.max_pool3d(unfold(o3), (stride, 1, 1), (stride, 1, 1))
    o2_ = F.max_pool3d(unfold(o2), (stride, 1, 1), (stride, 1, 1))
    return inception(sd, p + ".conv", torch.cat([o3_, _up(o2_, 2)], 1))


def sa_gate(sd, p, x, mask, k):
    """model/model_utils.py:155-170."""
    m = basic_conv3d(sd, p + ".conv_mask.0", mask, 1)
    if k != 1:
        m = _up(m, k)
    m = torch.sigmoid(_conv3(sd, p + ".conv_mask.2", m, 1, (0, 1, 1)))
    return x * m + x


def convnext_block3d(sd, p, x):
    """model/model_utils.py:306-354 (LayerNorm3d :293-303, eps 1e-5)."""
    y = _conv3(sd, p + ".dwconv_t", x, 1, (3, 0, 0), groups=x.shape[1])
    y = _conv3(sd, p + ".dwconv_s", y, 1, (0, 3, 3), groups=x.shape[1])
    y = _ln(sd, p + ".norm.norm", y.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)
    y = _conv3(sd, p + ".pwconv2", F.gelu(_conv3(sd, p + ".pwconv1", y)))
    return x + y


def latlayer(sd, p, x, lateral, stride):
    """model/model_utils.py:437-484."""
    x = _conv3(sd, p + ".0", x)
    if lateral:
        x = _conv3(sd, p + ".1", x, (stride, 1, 1))
        return convnext_block3d(sd, p + ".2", x)
    return convnext_block3d(sd, p + ".1", x)


def sinusoid_table(n_position, d_hid):
    """model/model_utils.py:18-29 (float64 numpy, cast to float32)."""
    pos = np.arange(n_position, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    tab = pos / np.power(10000, 2 * (j // 2) / d_hid)
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.tensor(tab, dtype=torch.float)


def vit_block(sd, p, x, heads):
    """model/model_utils.py:84-152: pre-LN block, qkv without bias, no LayerScale."""
    B, N, Cc = x.shape
    h = _ln(sd, p + ".norm1", x)
    qkv = _lin(sd, p + ".attn.qkv", h).reshape(B, N, 3, heads, Cc // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    a = ((q @ k.transpose(-2, -1)) * (Cc // heads) ** -0.5).softmax(-1)
    x = x + _lin(sd, p + ".attn.proj", (a @ v).transpose(1, 2).reshape(B, N, Cc))
    h = _ln(sd, p + ".norm2", x)
    return x + _lin(sd, p + ".mlp.fc2", F.gelu(_lin(sd, p + ".mlp.fc1", h)))


def sync_block(sd, p, vis, aud, num_blocks=3):
    """model/model_utils.py:257-282.  vis [B,C,T,H,W], aud [B,512,F,T']; tables sized to the inputs
    (the reference sizes them from cfg; F2/F3 in SURVEY.md)."""
    B = vis.shape[0]
    v = vis.flatten(2).transpose(1, 2)
    a = aud.flatten(2).transpose(1, 2)
    v = _ln(sd, p + ".vis_norm", _lin(sd, p + ".vis_proj", v)) + sinusoid_table(v.shape[1], 512)
    a = _ln(sd, p + ".aud_norm", a) + sinusoid_table(a.shape[1], 512)
    x = torch.cat([v, a], 1)
    for i in range(num_blocks):
        x = vit_block(sd, "%s.blocks.%d" % (p, i), x, 4)
    return x


def _projector(sd, p, x):
    x = F.relu(_ln(sd, p + ".1", _lin(sd, p + ".0", x)))
    x = F.relu(_ln(sd, p + ".4", _lin(sd, p + ".3", x)))
    return _ln(sd, p + ".7", _lin(sd, p + ".6", x))


def _predictor(sd, p, x):
    return _lin(sd, p + ".3", F.relu(_ln(sd, p + ".1", _lin(sd, p + ".0", x))))


def _D(p, z):
    return -F.cosine_similarity(p, z, dim=-1).mean()  # model/model_utils.py:285-290


def readout(sd, p, x):
    """model/model_utils.py:490-504."""
    x = _conv3(sd, p + ".0", x)
    x = F.relu(_bn(sd, p + ".2", _conv3(sd, p + ".1", x, 1, 1), 1e-5))
    x = F.relu(_bn(sd, p + ".5", _conv3(sd, p + ".4", x, 1, (0, 1, 1)), 1e-5))
    x = _up(x, 4)
    x = F.relu(_conv3(sd, p + ".8", x, (4, 1, 1)))
    x = F.relu(_conv3(sd, p + ".10", x, 1, (0, 1, 1)))
    return _conv3(sd, p + ".12", x, 1, (0, 1, 1))


def decode(sd, feats, masks, lateral_bool, lateral_stride, trace=None):
    """Top-down fusion + readout, model/model_utils.py:561-572.  trace: dict that receives the sub-module outputs the
    reference's forward hooks see (latlayer_k, sa_k, readout)."""
    v1, v2, v3, v4 = feats
    s3 = latlayer(sd, "latlayer_3", v4, lateral_bool[3], lateral_stride[3])
    s0 = latlayer(sd, "latlayer_0", v1, lateral_bool[0], lateral_stride[0])
    s1 = latlayer(sd, "latlayer_1", v2, lateral_bool[1], lateral_stride[1])
    s2 = latlayer(sd, "latlayer_2", v3, lateral_bool[2], lateral_stride[2])
    g2, g1, g0 = sa_gate(sd, "sa_2", s2, masks, 1), sa_gate(sd, "sa_1", s1, masks, 2), sa_gate(sd, "sa_0", s0, masks, 4)
    if trace is not None:
        trace.update({"latlayer_0.0": s0, "latlayer_1.0": s1, "latlayer_2.0": s2, "latlayer_3.0": s3,
                      "sa_0.0": g0, "sa_1.0": g1, "sa_2.0": g2})
    s2 = g2 + _up(s3, 2)
    s1 = g1 + _up(s2, 2) + _up(s3, 4)
    s0 = g0 + _up(s1, 2) + _up(s2, 4) + _up(s3, 8)
    out = readout(sd, "readout", torch.cat([s0, _up(s1, 2), _up(s2, 4), _up(s3, 8)], 1))
    if trace is not None:
        trace["readout.0"] = out
    out = out.squeeze(1).squeeze(1)
    return out - torch.logsumexp(out, dim=(1, 2), keepdim=True)


def pack_clips(name, clips):
    """model/model_utils.py:521-532: SlowFast slow pathway = frames [0,4,12,-1] (F7)."""
    if name == "slowfast4x16":
        return [torch.stack([clips[:, :, 0], clips[:, :, 4], clips[:, :, 12], clips[:, :, -1]], 2), clips]
    if name in ("videoswins", "s3d", "morphmlps"):
        return clips
    return [clips]


# ------------------------------------------------------------------------------- S3D
def _s3d_sep(sd, p, x, k, s, pad):
    """SepConv3d, backbones/s3d.py:95-116: (1,k,k)/(1,s,s) conv + BN + ReLU, then (k,1,1)/(s,1,1) conv + BN + ReLU."""
    x = F.relu(_bn(sd, p + ".bn_s", _conv3(sd, p + ".conv_s", x, (1, s, s), (0, pad, pad)), 1e-3))
    return F.relu(_bn(sd, p + ".bn_t", _conv3(sd, p + ".conv_t", x, (s, 1, 1), (pad, 0, 0)), 1e-3))


def _s3d_mixed(sd, p, x):
    """Mixed_3b .. Mixed_5c, backbones/s3d.py:118-370: four branches concatenated on channels."""
    x0 = basic_conv3d(sd, p + ".branch0.0", x, 0)
    x1 = _s3d_sep(sd, p + ".branch1.1", basic_conv3d(sd, p + ".branch1.0", x, 0), 3, 1, 1)
    x2 = _s3d_sep(sd, p + ".branch2.1", basic_conv3d(sd, p + ".branch2.0", x, 0), 3, 1, 1)
    x3 = basic_conv3d(sd, p + ".branch3.1", F.max_pool3d(x, 3, 1, 1), 0)
    return torch.cat((x0, x1, x2, x3), 1)


def s3d_forward(sd, clips, prefix="", pool=1):
    """S3D_features_only.forward, backbones/s3d.py:379-421 -> [base1, base2, base3, base4]."""
    p = prefix
    x = _s3d_sep(sd, p + "base1.0", clips, 7, 2, 3)
    x = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    x = basic_conv3d(sd, p + "base1.2", x, 0)
    base1 = _s3d_sep(sd, p + "base1.3", x, 3, 1, 1)
    x = F.max_pool3d(base1, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    for i in range(2):
        x = _s3d_mixed(sd, p + "base2.%d" % i, x)
    base2 = x
    x = F.max_pool3d(base2, 3, 2, 1)
    for i in range(5):
        x = _s3d_mixed(sd, p + "base3.%d" % i, x)
    base3 = x
    x = F.max_pool3d(base3, (pool, 2, 2), (pool, 2, 2))
    for i in range(2):
        x = _s3d_mixed(sd, p + "base4.%d" % i, x)
    return [base1, base2, base3, x]


# ------------------------------------------------------------------------------- UniFormer-B
def _uni_attention(sd, p, x, heads):
    """Attention, backbones/uniformer.py:71-96 (scale = head_dim ** -0.5)."""
    B, N, C = x.shape
    qkv = _lin(sd, p + ".qkv", x).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    attn = ((qkv[0] @ qkv[1].transpose(-2, -1)) * (C // heads) ** -0.5).softmax(dim=-1)
    return _lin(sd, p + ".proj", (attn @ qkv[2]).transpose(1, 2).reshape(B, N, C))


def uniformer_cblock(sd, p, x):
    """CBlock, backbones/uniformer.py:117-137: dw3x3x3 pos-embed, BN -> 1x1x1 -> dw5x5x5 -> 1x1x1, BN -> conv MLP."""
    C = x.shape[1]
    x = x + _conv3(sd, p + ".pos_embed", x, 1, 1, groups=C)
    y = _conv3(sd, p + ".conv1", _bn(sd, p + ".norm1", x, 1e-5))
    x = x + _conv3(sd, p + ".conv2", _conv3(sd, p + ".attn", y, 1, 2, groups=C))
    y = F.gelu(_conv3(sd, p + ".mlp.fc1", _bn(sd, p + ".norm2", x, 1e-5)))
    return x + _conv3(sd, p + ".mlp.fc2", y)


def uniformer_sablock(sd, p, x, heads):
    """SABlock, backbones/uniformer.py:140-163: dw3x3x3 pos-embed, pre-LN(1e-6) global attention, pre-LN MLP."""
    B, C, T, H, W = x.shape
    x = x + _conv3(sd, p + ".pos_embed", x, 1, 1, groups=C)
    x = x.flatten(2).transpose(1, 2)
    x = x + _uni_attention(sd, p + ".attn", _ln(sd, p + ".norm1", x, 1e-6), heads)
    y = F.gelu(_lin(sd, p + ".mlp.fc1", _ln(sd, p + ".norm2", x, 1e-6)))
    x = x + _lin(sd, p + ".mlp.fc2", y)
    return x.transpose(1, 2).reshape(B, C, T, H, W)


def _uni_patch_embed(sd, p, x, stride, pad):
    """SpeicalPatchEmbed / PatchEmbed, backbones/uniformer.py:205-264: strided conv then LayerNorm(1e-5) over channels."""
    x = _conv3(sd, p + ".proj", x, stride, pad)
    return _ln(sd, p + ".norm", x.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3).contiguous()


def uniformer_forward(sd, clips, prefix="", head_dim=64):
    """Uniformer.forward_features, backbones/uniformer.py:441-474 (SPLIT False) -> the four stage outputs."""
    p = prefix
    feats = []
    x = clips
    for s, (stride, pad) in enumerate((((2, 4, 4), (1, 0, 0)), ((1, 2, 2), 0), ((1, 2, 2), 0), ((1, 2, 2), 0)), 1):
        x = _uni_patch_embed(sd, p + "patch_embed%d" % s, x, stride, pad)
        i = 0
        while p + "blocks%d.%d.pos_embed.weight" % (s, i) in sd:
            b = p + "blocks%d.%d" % (s, i)
            x = uniformer_cblock(sd, b, x) if s <= 2 else uniformer_sablock(sd, b, x, x.shape[1] // head_dim)
            i += 1
        feats.append(x)
    return feats


# ------------------------------------------------------------------------------- MorphMLP-S
def _morph_mlp(sd, p, x):
    return _lin(sd, p + ".fc2", F.gelu(_lin(sd, p + ".fc1", x)))


def morph_fc_t(sd, p, x):
    """MorphFC_T, backbones/MorphMLP.py:116-158: Linear over (frame, 1/8 channel slice), segment_dim fixed to 8."""
    B, T, H, W, C = x.shape
    S = C // 8
    t = x.reshape(B, T, H, W, 8, S).permute(0, 4, 2, 3, 1, 5).reshape(B, 8, H, W, T * S)
    t = _lin(sd, p + ".mlp_t", t).reshape(B, 8, H, W, T, S).permute(0, 4, 2, 3, 1, 5).reshape(B, T, H, W, C)
    return _lin(sd, p + ".proj", t)


def morph_fc_s(sd, p, x, seg):
    """MorphFC_S, backbones/MorphMLP.py:71-113: Linears over chunks of `seg` positions along H / along W / channels,
    mixed by softmax(reweight(mean(h + w + c)))."""
    B, T, H, W, C = x.shape
    S, n = C // seg, H * W // seg
    h = x.transpose(3, 2).reshape(B, T, n, seg, seg, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, n, seg, seg * S)
    h = _lin(sd, p + ".mlp_h", h).reshape(B, T, n, seg, seg, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, W, H, C).transpose(3, 2)
    w = x.reshape(B, T, n, seg, seg, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, n, seg, seg * S)
    w = _lin(sd, p + ".mlp_w", w).reshape(B, T, n, seg, seg, S).permute(0, 1, 2, 4, 3, 5).reshape(B, T, H, W, C)
    c = _lin(sd, p + ".mlp_c", x)
    a = (h + w + c).permute(0, 4, 1, 2, 3).flatten(2).mean(2)
    a = _morph_mlp(sd, p + ".reweight", a).reshape(B, C, 3).permute(2, 0, 1).softmax(dim=0)[:, :, None, None, None]
    return _lin(sd, p + ".proj", h * a[0] + w * a[1] + c * a[2])


def morph_fc_s2(sd, p, x, seg):
    """MorphFC_S2, backbones/MorphMLP.py:38-68 (last stage): one spatial Linear + the channel Linear."""
    B, T, H, W, C = x.shape
    S, n = C // seg, H * W // seg
    h = x.reshape(B, T, seg, n, seg, S).permute(0, 1, 4, 3, 2, 5).reshape(B, T, seg, n, seg * S)
    h = _lin(sd, p + ".mlp_h", h).reshape(B, T, seg, n, seg, S).permute(0, 1, 4, 3, 2, 5).reshape(B, T, H, W, C)
    c = _lin(sd, p + ".mlp_c", x)
    a = (h + c).permute(0, 4, 1, 2, 3).flatten(2).mean(2)
    a = _morph_mlp(sd, p + ".reweight", a).reshape(B, C, 2).permute(2, 0, 1).softmax(dim=0)[:, :, None, None, None]
    return _lin(sd, p + ".proj", h * a[0] + c * a[1])


def morph_block(sd, p, x, seg, last):
    """PermutatorBlock, backbones/MorphMLP.py:161-187 (skip_lam 1): note the second shortcut starts from x, not xt."""
    xt = x + morph_fc_t(sd, p + ".t_fc", _ln(sd, p + ".t_norm1", x))
    fc = morph_fc_s2 if last else morph_fc_s
    x = x + fc(sd, p + ".fc", _ln(sd, p + ".norm1", xt), seg)
    return x + _morph_mlp(sd, p + ".mlp", _ln(sd, p + ".norm2", x))


def morphmlp_forward(sd, clips, prefix="", segment_dim=(14, 28, 28, 49)):
    """MorphMLP_32_features_only.forward_features, backbones/MorphMLP.py:480-503 -> four NCDHW stage outputs."""
    p = prefix
    x = F.gelu(_bn(sd, p + "patch_embed1.norm1", _conv3(sd, p + "patch_embed1.proj1", clips, 2, 1), 1e-5))
    x = _bn(sd, p + "patch_embed1.norm2", _conv3(sd, p + "patch_embed1.proj2", x, (1, 2, 2), (0, 1, 1)), 1e-5)
    x = x.permute(0, 2, 3, 4, 1)
    feats = []
    for s in range(1, 5):
        if s > 1:                                                               # Downsample :211-225
            x = _conv3(sd, p + "patch_embed%d.proj" % s, x.permute(0, 4, 1, 2, 3), (1, 2, 2), (0, 1, 1))
            x = _ln(sd, p + "patch_embed%d.norm" % s, x.permute(0, 2, 3, 4, 1))
        i = 0
        while p + "blocks%d.%d.norm1.weight" % (s, i) in sd:
            x = morph_block(sd, p + "blocks%d.%d" % (s, i), x, segment_dim[s - 1], s == 4)
            i += 1
        feats.append(x.permute(0, 4, 1, 2, 3))
    return feats


BACKBONES = {}  # name -> fn(sd, packed_clips, prefix) -> [v1..v4]; filled below and by restate_tx.py
BACKBONES["x3dl"] = lambda sd, x, prefix: x3d_forward(sd, x[0], prefix)
BACKBONES["slowfast4x16"] = lambda sd, x, prefix: slowfast_forward(sd, x, prefix)
BACKBONES["s3d"] = lambda sd, x, prefix: s3d_forward(sd, x, prefix)
BACKBONES["uniformerb"] = lambda sd, x, prefix: uniformer_forward(sd, x[0], prefix)
BACKBONES["morphmlps"] = lambda sd, x, prefix: morphmlp_forward(sd, x, prefix)
MVIT_S_ARCH = {   # configs/MVITv2_S_16x4.yaml resolved the way MViT.__init__ does (backbones/MViT.py:1779-1826)
    "patch_stride": (2, 4, 4), "patch_padding": (1, 3, 3),
    "blocks": [(1, (1, 1, 1), (1, 8, 8)), (2, (1, 2, 2), (1, 4, 4)), (2, (1, 1, 1), (1, 4, 4)), (4, (1, 2, 2), (1, 2, 2))]
              + [(4, (1, 1, 1), (1, 2, 2))] * 10 + [(8, (1, 2, 2), (1, 1, 1)), (8, (1, 1, 1), (1, 1, 1))],
}
BACKBONES["mvitv2s"] = lambda sd, x, prefix: mvit_forward(sd, x[0], MVIT_S_ARCH, prefix)
BACKBONES["videoswins"] = lambda sd, x, prefix: swin_forward(sd, x, prefix)


def audio_visual_forward(sd, clips, audios, name, lateral_bool, lateral_stride, num_frames=16, trace=None):
    """AudioVisualSaliencyModel.forward, model/model_utils.py:520-574.  Returns (log-prob map, loss).
    trace: dict that receives the outputs of the sub-modules (keys as the reference's module names + output index)."""
    B = clips.shape[0]
    frames = clips.permute(0, 2, 1, 3, 4).reshape(B * clips.shape[2], clips.shape[1], *clips.shape[3:])
    o1, o0 = static_saliency_encoder(sd, frames)
    masks = adapter(sd, "adapter", o1, o0, num_frames, num_frames // 4)
    aud = resnet18_forward(sd, audios, "audnet.")
    v1, v2, v3, v4 = BACKBONES[name](sd, pack_clips(name, clips), "visnet.")
    _, _, t, h, w = v4.shape
    x = sync_block(sd, "aud_vis_sync_block", v4, aud)
    vis_fea = x[:, : t * h * w].transpose(1, 2).reshape(B, 512, t, h, w)
    aud_fea = x[:, t * h * w:].transpose(1, 2)
    vis_emb = _projector(sd, "vis_projector", vis_fea.mean((2, 3, 4)))
    aud_emb = _projector(sd, "aud_projector", aud_fea.mean(2))
    loss = (_D(_predictor(sd, "mlp_vis", vis_emb), aud_emb) + _D(_predictor(sd, "mlp_aud", aud_emb), vis_emb)) * 0.5
    if trace is not None:
        trace.update({"image_encoder.0": o1, "image_encoder.1": o0, "adapter.0": masks, "audnet.0": aud, "aud_vis_sync_block.0": x})
    out = decode(sd, (v1, v2, v3, torch.cat([v4, vis_fea], 1)), masks, lateral_bool, lateral_stride, trace)
    return out, loss


def visual_forward(sd, clips, name, lateral_bool, lateral_stride, num_frames=16):
    """VisualSaliencyModel.forward, model/model_utils.py:685-702.  Returns (log-prob map, 0)."""
    B = clips.shape[0]
    frames = clips.permute(0, 2, 1, 3, 4).reshape(B * clips.shape[2], clips.shape[1], *clips.shape[3:])
    o1, o0 = static_saliency_encoder(sd, frames)
    masks = adapter(sd, "adapter", o1, o0, num_frames, num_frames // 4)
    feats = BACKBONES[name](sd, pack_clips(name, clips), "visnet.")
    return decode(sd, feats, masks, lateral_bool, lateral_stride), 0


# ------------------------------------------------------------------------------- post-processing (host side upstream)
def postprocess_u8(logmap, out_hw):
    """inference.py:66-69,85-89 restated without OpenCV (absent here -> PARITY UNPINNED against cv2 itself):
    cv2.GaussianBlur(11x11, sigma 0 -> 2.0, BORDER_REFLECT_101), np.exp, cv2.resize INTER_LINEAR
    (pixel-centre aligned), min-max normalise, np.round(x*255).astype(uint8).  logmap: [H,W] float tensor."""
    k = torch.exp(-(torch.arange(11, dtype=torch.float32) - 5) ** 2 / (2 * 2.0 ** 2))
    k = k / k.sum()
    x = logmap[None, None].float()
    x = F.pad(x, (5, 5, 5, 5), mode="reflect")
    x = F.conv2d(F.conv2d(x, k.view(1, 1, 1, 11)), k.view(1, 1, 11, 1))
    x = torch.exp(x)
    x = F.interpolate(x, size=out_hw, mode="bilinear", align_corners=False)[0, 0]
    x = (x - x.min()) / (x.max() - x.min())
    return torch.round(x * 255).to(torch.uint8)


# ----------------------------------------------------------------------------- saliency metrics
def saliency_metrics(pred, gt, fix=None):
    """Per-sample (KL, CC, SIM, NSS) of utils/compute_saliency_metrics.py:9-108 on [B,H,W] maps, [B,4] float tensor
    (the reference returns the batch mean of each column)."""
    B = pred.shape[0]
    s, g = pred.reshape(B, -1).float(), gt.reshape(B, -1).float()
    eps = 2.2204e-16
    sp, gp = s / s.sum(1, keepdim=True), g / g.sum(1, keepdim=True)
    kl = (gp * torch.log(eps + gp / (sp + eps))).sum(1)                                    # kldiv :9-31
    sz = (s - s.mean(1, keepdim=True)) / s.std(1, keepdim=True)
    gz = (g - g.mean(1, keepdim=True)) / g.std(1, keepdim=True)
    cc = (sz * gz).sum(1) / torch.sqrt((sz * sz).sum(1) * (gz * gz).sum(1))                # cc :73-90
    sn = (s - s.min(1, keepdim=True)[0]) / (s.max(1, keepdim=True)[0] - s.min(1, keepdim=True)[0])
    gn = (g - g.min(1, keepdim=True)[0]) / (g.max(1, keepdim=True)[0] - g.min(1, keepdim=True)[0])
    sim = torch.min(sn / sn.sum(1, keepdim=True), gn / gn.sum(1, keepdim=True)).sum(1)     # similarity :46-70
    if fix is None:
        ns = torch.zeros(B)
    else:
        f = fix.reshape(B, -1).float()
        ns = (((s - s.mean(1, keepdim=True)) / (s.std(1, keepdim=True) + eps)) * f).sum(1) / f.sum(1)   # nss :93-107
    return torch.stack([kl, cc, sim, ns], 1)


# ------------------------------------------------------------------------------- clip-loop pre-processing
def frame_transform(rgb_u8, out_hw, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """inference.py:154-165 (torchvision Resize on a PIL image, ToTensor, Normalize): PIL itself does the resampling --
    the reference's own dependency, so this function is pinned by construction."""
    from PIL import Image
    img = Image.fromarray(np.asarray(rgb_u8, dtype=np.uint8)).resize((out_hw[1], out_hw[0]), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)
    return (t - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)


def log_spectrogram_window(wave, start, length, reverse=False, Wa=111):
    """inference.py:41-58 on a 16 kHz mono wave [n]: Spectrogram(n_fft=512, hop_length=160) = torch.stft with a periodic
    Hann window, centre + reflect padding, |.|^2 (what torchaudio.functional.spectrogram evaluates; torchaudio itself is
    absent offline -> PARITY UNPINNED against torchaudio, pinned only against torch.stft), log(. + 1e-6), per-column
    standardisation over the 257 bins (unbiased std), crop / pad with 0.02 to Wa columns.  length 0 = no audio."""
    out = torch.zeros(1, 257, Wa) + 0.02
    if length == 0:
        return out
    a = wave[start:start + length][None]
    if reverse:
        a = torch.flip(a, [1])
    spec = torch.stft(a, n_fft=512, hop_length=160, win_length=512, window=torch.hann_window(512), center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    a = torch.log(spec.abs().pow(2.0) + 1e-6)
    a = (a - a.mean(dim=1, keepdim=True)) / (a.std(dim=1, keepdim=True) + 1e-6)
    n = min(a.shape[-1], Wa)
    out[:, :, :n] = a[:, :, :n]
    return out
