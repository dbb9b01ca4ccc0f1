"""Model assembly, multisensory fusion and decoder -- HIP-backed mirror of the reference's
model/model_utils.py (AudioVisualSaliencyModel :388-574, VisualSaliencyModel :576-702).

Sub-module names equal the reference's so that released MSPI checkpoints load with
`load_state_dict(strict=False)` (inference.py:186); torch layers hold parameters only and the
forward runs entirely on the C ABI (mspi_amd.engine).  Fusions applied here (all exact up to
fp32 rounding):
  * eval BatchNorm folded into the producing conv; bias/ReLU/GELU/sigmoid/residual in epilogues
  * every torch.cat replaced by producers writing into channel slices / token slabs
  * latlayer_k[0] (1x1x1) and latlayer_k[1] ((s,1,1) stride s), both linear with nothing in
    between, composed into one strided temporal conv (4x fewer rows, no 192-ch intermediate)
  * readout: the x4 trilinear up-sample is moved behind the (4,1,1)/4 temporal conv (both linear,
    bilinear weights sum to 1 so the bias commutes), ReLU applied in the up-sample epilogue
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import engine as E
from .._lib import MspiError
from ..backbones.convnext import ConvNeXtTinyFeatures
from ..backbones.resnet import get_resnet18
from ..module import HipModule, to_cl
from .get_video_backbones import video_motion_extractor


def get_sinusoid_encoding_table(n_position, d_hid):
    """Sinusoid position table [1, n_position, d_hid] (reference: model/model_utils.py:18-29)."""
    pos = np.arange(n_position, dtype=np.float64).reshape(-1, 1)
    hid = np.arange(d_hid).reshape(1, -1)
    ang = pos / np.power(10000, 2 * (hid // 2) / d_hid)
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.tensor(ang, dtype=torch.float, requires_grad=False).unsqueeze(0)


def _f(t):
    return t.detach().float().contiguous()


# ------------------------------------------------------------------------------- SyncBlock
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(HipModule):
    """Pre-LN transformer block (model/model_utils.py:122-152): 7 launches."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = Attention(dim, num_heads=num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _pack(self):
        a, m = self.attn, self.mlp
        return {"n1": (_f(self.norm1.weight), _f(self.norm1.bias)), "n2": (_f(self.norm2.weight), _f(self.norm2.bias)),
                "qkv": E.pack_conv(a.qkv.weight, a.qkv.bias), "proj": E.pack_conv(a.proj.weight, a.proj.bias),
                "fc1": E.pack_conv(m.fc1.weight, m.fc1.bias, act=E.ACT_GELU), "fc2": E.pack_conv(m.fc2.weight, m.fc2.bias)}

    def run(self, x, B, R):
        pk = self.pk
        dim = self.norm1.normalized_shape[0]
        h = E.layernorm_for_gemm(x, *pk["n1"], 1e-5, pk["qkv"])
        o = E.attention(E.conv(h, pk["qkv"]), B, R, self.attn.num_heads, dim // self.attn.num_heads, self.attn.scale)
        x = E.conv(o, pk["proj"], res=x)
        return E.mlp_tail(x, ("split", pk["fc1"], pk["fc2"]), pk["n2"], 1e-5, res=x)


class SyncBlock(HipModule):
    def __init__(self, num_blocks=3, num_vis_tokens=336, num_aud_tokens=36, vis_in_embed=1024, embed_dim=512):
        super().__init__()
        self.vis_pos_embed = get_sinusoid_encoding_table(num_vis_tokens, 512)
        self.aud_pos_embed = get_sinusoid_encoding_table(num_aud_tokens, 512)
        self.vis_proj = nn.Linear(vis_in_embed, 512)
        self.vis_norm = nn.LayerNorm(512)
        self.aud_norm = nn.LayerNorm(512)
        self.blocks = nn.ModuleList([Block(dim=embed_dim, num_heads=4) for _ in range(num_blocks)])
        for m in self.modules():  # model/model_utils.py:241-251
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def _pack(self):
        dev = self.vis_proj.weight.device
        return {"proj": E.pack_conv(self.vis_proj.weight, self.vis_proj.bias),
                "vn": (_f(self.vis_norm.weight), _f(self.vis_norm.bias)),
                "an": (_f(self.aud_norm.weight), _f(self.aud_norm.bias)),
                "vpos": self.vis_pos_embed[0].to(dev).contiguous(), "apos": self.aud_pos_embed[0].to(dev).contiguous()}

    def run(self, vis, aud):
        """vis CL [B,t,h,w,C4], aud CL [B,1,F,T',512] -> token slab CL [B, Rv+Ra, 1, 1, 512]."""
        pk = self.pk
        B, Rv, Ra = vis.N, vis.T * vis.H * vis.W, aud.T * aud.H * aud.W
        if Rv != pk["vpos"].shape[0] or Ra != pk["apos"].shape[0]:
            raise MspiError("SyncBlock: got %d visual / %d audio tokens but the positional tables hold %d / %d rows "
                            "-- set cfg.MODEL.NUM_VIS_TOKENS[name] / cfg.MODEL.NUM_AUD_TOKENS (reference: the add at "
                            "model/model_utils.py:273-274 fails the same way)" % (Rv, Ra, pk["vpos"].shape[0], pk["apos"].shape[0]))
        x = E.alloc(B, Rv + Ra, 1, 1, 512, vis.buf.device)
        E.layernorm(E.conv(vis, pk["proj"]), *pk["vn"], 1e-5, out=x.tokens(0, vis.T, vis.H, vis.W), table=pk["vpos"])
        E.layernorm(aud, *pk["an"], 1e-5, out=x.tokens(Rv, aud.T, aud.H, aud.W), table=pk["apos"])
        for blk in self.blocks:
            x = blk.run(x, B, Rv + Ra)
        return x

    def forward(self, vis_fea, aud_fea):
        a = aud_fea[:, :, None] if aud_fea.dim() == 4 else aud_fea
        x = self.run(to_cl(vis_fea), to_cl(a))
        return x.as_rows().view(x.N, x.T, 512)


# ------------------------------------------------------------------------------- conv helpers (backbones/s3d.py:41-52,95-116)
class BasicConv3d(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        self.conv = nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm3d(out_planes, eps=1e-3, momentum=0.001, affine=True)
        self.relu = nn.ReLU()

    def packed(self):
        c = self.conv
        return E.pack_conv(c.weight, None, self.bn, c.stride, c.padding, E.ACT_RELU)


class SepConv3d(nn.Module):
    def __init__(self, in_planes, out_planes, kernel_size, stride, padding=0):
        super().__init__()
        k, s, p = kernel_size, stride, padding
        self.conv_s = nn.Conv3d(in_planes, out_planes, (1, k, k), (1, s, s), (0, p, p), bias=False)
        self.bn_s = nn.BatchNorm3d(out_planes, eps=1e-3, momentum=0.001, affine=True)
        self.relu_s = nn.ReLU()
        self.conv_t = nn.Conv3d(out_planes, out_planes, (k, 1, 1), (s, 1, 1), (p, 0, 0), bias=False)
        self.bn_t = nn.BatchNorm3d(out_planes, eps=1e-3, momentum=0.001, affine=True)
        self.relu_t = nn.ReLU()

    def packed(self):
        cs, ct = self.conv_s, self.conv_t
        return (E.pack_conv(cs.weight, None, self.bn_s, cs.stride, cs.padding, E.ACT_RELU),
                E.pack_conv(ct.weight, None, self.bn_t, ct.stride, ct.padding, E.ACT_RELU))


class SA(HipModule):
    """x * sigmoid(conv(up(BasicConv3d(mask)))) + x (model/model_utils.py:155-170)."""

    def __init__(self, in_embed_dim=512, k=2):
        super().__init__()
        self.k = k
        self.up = nn.Upsample(scale_factor=(1, k, k), align_corners=False, mode="trilinear") if k != 1 else nn.Identity()
        self.conv_mask = nn.Sequential(
            BasicConv3d(in_embed_dim, in_embed_dim // 16, kernel_size=3, stride=1, padding=1),
            self.up,
            nn.Conv3d(in_embed_dim // 16, 1, kernel_size=(1, 3, 3), stride=1, padding=(0, 1, 1)),
            nn.Sigmoid(),
        )

    def _pack(self):
        c = self.conv_mask[2]
        return self.conv_mask[0].packed(), E.pack_conv(c.weight, c.bias, None, (1, 1, 1), (0, 1, 1), E.ACT_SIGMOID)

    def run(self, x, mask, premask=None):
        """In place on x.  premask: this module's BasicConv3d output if the caller already computed it
        (the decoder runs the three SA modules' first convs, which share their input, as ONE conv)."""
        p0, p2 = self.pk
        m = E.conv(mask, p0) if premask is None else premask
        if self.k != 1:
            m = E.upsample(m, self.k)
        return E.rowgate(x, E.conv(m, p2))


class Inception(HipModule):
    def __init__(self, embed_dim=320 + 96):
        super().__init__()
        self.branch0 = nn.Sequential(BasicConv3d(embed_dim, 192, kernel_size=1, stride=1))
        self.branch1 = nn.Sequential(BasicConv3d(embed_dim, 96, kernel_size=1, stride=1),
                                     SepConv3d(96, 208, kernel_size=3, stride=1, padding=1))
        self.branch2 = nn.Sequential(BasicConv3d(embed_dim, 16, kernel_size=1, stride=1),
                                     SepConv3d(16, 48, kernel_size=3, stride=1, padding=1))
        self.branch3 = nn.Sequential(nn.MaxPool3d(kernel_size=(3, 3, 3), stride=1, padding=1),
                                     BasicConv3d(embed_dim, 64, kernel_size=1, stride=1))

    def _pack(self):
        return {"b0": self.branch0[0].packed(), "b1": (self.branch1[0].packed(),) + self.branch1[1].packed(),
                "b2": (self.branch2[0].packed(),) + self.branch2[1].packed(), "b3": self.branch3[1].packed()}

    def run(self, x):
        pk = self.pk
        out = E.alloc(x.N, x.T, x.H, x.W, 512, x.buf.device)
        E.conv(x, pk["b0"], out=out.slice(0, 192))
        E.conv(E.conv(E.conv(x, pk["b1"][0]), pk["b1"][1]), pk["b1"][2], out=out.slice(192, 208))
        E.conv(E.conv(E.conv(x, pk["b2"][0]), pk["b2"][1]), pk["b2"][2], out=out.slice(400, 48))
        E.conv(E.maxpool(x, (3, 3, 3), (1, 1, 1), (1, 1, 1)), pk["b3"], out=out.slice(448, 64))
        return out


class Adapter(HipModule):
    def __init__(self, embed_dim=320 + 96, num_frames=32, stride=8):
        super().__init__()
        self.num_frames = num_frames
        self.stride = stride
        self.pool_time = nn.MaxPool3d(kernel_size=(stride, 1, 1), stride=(stride, 1, 1))
        self.conv = Inception(embed_dim=embed_dim)
        self.up = nn.Upsample(scale_factor=(1, 2, 2), align_corners=False, mode="trilinear")

    def run(self, o3, o2):
        """o3 CL [(b t),1,h,w,96], o2 CL [(b t),1,h/2,w/2,320] -> masks CL [B,T/stride,h,w,512]."""
        T, s = self.num_frames, self.stride
        B = o3.N // T
        o3 = o3.reshape(B, T, o3.H, o3.W)
        o2 = o2.reshape(B, T, o2.H, o2.W)
        cat = E.alloc(B, T // s, o3.H, o3.W, o3.C + o2.C, o3.buf.device)
        E.maxpool(o3, (s, 1, 1), (s, 1, 1), (0, 0, 0), out=cat.slice(0, o3.C))
        E.upsample(E.maxpool(o2, (s, 1, 1), (s, 1, 1), (0, 0, 0)), 2, dst=cat.slice(o3.C, o2.C))
        return self.conv.run(cat)


class LayerNorm3d(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.norm = nn.LayerNorm(dim)


class ConvNextBlock(HipModule):
    """dw(7,1,1) -> dw(1,7,7) -> channel LN -> 1x1x1 x4 + GELU -> 1x1x1 -> + input (model/model_utils.py:306-354)."""

    def __init__(self, dim, drop_path=0.0):
        super().__init__()
        self.dwconv_t = nn.Conv3d(dim, dim, kernel_size=(7, 1, 1), padding=(3, 0, 0), groups=dim)
        self.dwconv_s = nn.Conv3d(dim, dim, kernel_size=(1, 7, 7), padding=(0, 3, 3), groups=dim)
        self.norm = LayerNorm3d(dim)
        self.pwconv1 = nn.Conv3d(dim, 4 * dim, 1)
        self.act = nn.GELU()
        self.pwconv2 = nn.Conv3d(4 * dim, dim, 1)
        self.drop_path = nn.Identity()
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.constant_(m.bias, 0)

    def _pack(self):
        return {"dt": E.pack_dwconv(self.dwconv_t.weight, self.dwconv_t.bias, None, (1, 1, 1), (3, 0, 0)),
                "ds": E.pack_dwconv(self.dwconv_s.weight, self.dwconv_s.bias, None, (1, 1, 1), (0, 3, 3)),
                "ln": (_f(self.norm.norm.weight), _f(self.norm.norm.bias)),
                "p1": E.pack_conv(self.pwconv1.weight, self.pwconv1.bias, act=E.ACT_GELU),
                "p2": E.pack_conv(self.pwconv2.weight, self.pwconv2.bias),
                # LN -> 1x1x1 -> GELU -> 1x1x1 -> +x in one launch where the fused kernel covers the width (dim 192)
                "fused": E.pack_mlp(self.pwconv1.weight.flatten(1), self.pwconv1.bias, self.pwconv2.weight.flatten(1),
                                    self.pwconv2.bias) if E.mlp_supported(self.pwconv1.in_channels, self.pwconv1.out_channels) else None}

    def run(self, x, out=None):
        pk = self.pk
        y = E.dwconv(E.dwconv(x, pk["dt"]), pk["ds"])
        if pk["fused"] is not None and y.M >= 4096:      # below that the two tiled GEMMs (split-K on the small maps) win
            return E.mlp(y, pk["fused"], res=x, ln=pk["ln"], eps=1e-5, out=out, split=(pk["p1"], pk["p2"]))
        return E.conv(E.conv(E.layernorm(y, *pk["ln"], 1e-5), pk["p1"]), pk["p2"], res=x, out=out)


class StaticSaliencyModelConvNext(HipModule):
    """Per-frame ConvNeXt-T + two smoothing convs (model/model_utils.py:357-385)."""

    def __init__(self):
        super().__init__()
        self.encoder = ConvNeXtTinyFeatures()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        self.up4 = nn.Upsample(scale_factor=4, mode="bilinear", align_corners=False)
        self.up8 = nn.Upsample(scale_factor=8, mode="bilinear", align_corners=False)
        self.smooth_0 = nn.Sequential(nn.Conv2d(768, 320, 3, 1, 1), nn.BatchNorm2d(320), nn.ReLU())
        self.smooth_1 = nn.Sequential(nn.Conv2d(384, 96, 3, 1, 1), nn.BatchNorm2d(96), nn.ReLU())

    def _pack(self):
        return tuple(E.pack_conv(s[0].weight, s[0].bias, s[1], (1, 1, 1), (0, 1, 1), E.ACT_RELU)
                     for s in (self.smooth_0, self.smooth_1))

    def run(self, clips):
        s0, s1 = self.pk
        _, _, o1, o0 = self.encoder.forward_cl(clips)
        return E.conv(o1, s1), E.conv(o0, s0)

    def forward(self, x):
        o1, o0 = self.run(x)
        return o1.as_ncdhw().squeeze(2), o0.as_ncdhw().squeeze(2)


# ------------------------------------------------------------------------------- models
class _Fork:
    """Fork the current HIP stream into side streams for independent branches and join them on exit.
    MSPI_STREAMS=0 runs everything on the current stream."""
    ENABLED = os.environ.get("MSPI_STREAMS", "1") != "0"

    def __init__(self, owner, device):
        self.main = torch.cuda.current_stream(device)
        d = owner.__dict__
        if "_side_streams" not in d or d["_side_streams"][0].device != device:
            d["_side_streams"] = [torch.cuda.Stream(device) for _ in range(2)]
        self.side = d["_side_streams"]
        self.used = []
        self.kept = []

    def __enter__(self):
        self.start = torch.cuda.Event()
        self.start.record(self.main)
        return self

    def branch(self, i, refork=False):
        """Context manager that runs its body on side stream i.  refork=True: the branch starts from the main
        stream's CURRENT position (a second use of the stream, after an earlier join), not from the fork point."""
        if not self.ENABLED:
            return torch.cuda.stream(self.main)
        s = self.side[i]
        if refork:
            ev = torch.cuda.Event()
            ev.record(self.main)
            s.wait_event(ev)
        else:
            s.wait_event(self.start)
        if s not in self.used:
            self.used.append(s)
        return torch.cuda.stream(s)

    def keep(self, *objs):
        """Hold references until the join at exit: memory allocated on the main stream that a side-stream kernel reads
        would otherwise go back to the caching allocator when its last Python reference dies, and be handed to a later
        main-stream kernel while the side stream has not run yet."""
        self.kept.extend(objs)

    def join(self, i):
        """Make the main stream wait for branch i now (its results are needed before the others finish)."""
        s = self.side[i]
        if self.ENABLED and s in self.used:
            ev = torch.cuda.Event()
            ev.record(s)
            self.main.wait_event(ev)
            self.used.remove(s)

    def __exit__(self, *exc):
        for s in list(self.used):
            ev = torch.cuda.Event()
            ev.record(s)
            self.main.wait_event(ev)
        self.used = []
        self.kept = []
        return False


def _compose_lateral(c0, c1):
    """(s,1,1)/s conv after a 1x1x1 conv, no nonlinearity between: one conv with
    W[co,ci,tap] = sum_m W1[co,m,tap] W0[m,ci],  b[co] = sum_{tap,m} W1[co,m,tap] b0[m]."""
    w0 = c0.weight.detach().double().flatten(1)                 # [mid, cin]
    w1 = c1.weight.detach().double()[:, :, :, 0, 0]              # [co, mid, s]
    w = torch.einsum("omt,mi->oit", w1, w0)[:, :, :, None, None]  # [co, cin, s, 1, 1]
    b = torch.einsum("omt,m->o", w1, c0.bias.detach().double())
    return w.float(), b.float()


class _SaliencyBase(HipModule):
    """Shared decoder: lateral layers, SA gating, top-down fusion, readout (model/model_utils.py:437-504,561-572)."""

    def _build_decoder(self, cfg, vis_embed_dims, extra_c4, de_embed_dim=192):
        d = de_embed_dim
        for k in range(4):
            cin = vis_embed_dims[k] + (extra_c4 if k == 3 else 0)
            layers = [nn.Conv3d(cin, d, 1, 1, 0)]
            if cfg.MODEL.LATERAL_BOOL[k]:
                s = cfg.MODEL.LATERAL_STRIDE[k]
                layers.append(nn.Conv3d(d, d, kernel_size=(s, 1, 1), stride=(s, 1, 1), bias=False))
            layers.append(ConvNextBlock(dim=d))
            setattr(self, "latlayer_%d" % k, nn.Sequential(*layers))
        self.upsample = nn.Upsample(scale_factor=(1, 2, 2), mode="trilinear", align_corners=False)
        self.upsample_4 = nn.Upsample(scale_factor=(1, 4, 4), mode="trilinear", align_corners=False)
        self.upsample_8 = nn.Upsample(scale_factor=(1, 8, 8), mode="trilinear", align_corners=False)
        self.readout = nn.Sequential(
            nn.Conv3d(d * 4, d, 1, 1, 0),
            nn.Conv3d(d, d, kernel_size=3, stride=1, padding=1),
            nn.BatchNorm3d(d),
            nn.ReLU(inplace=True),
            nn.Conv3d(d, 64, kernel_size=(1, 3, 3), stride=(1, 1, 1), padding=(0, 1, 1)),
            nn.BatchNorm3d(64),
            nn.ReLU(inplace=True),
            nn.Upsample(scale_factor=(1, 4, 4), mode="trilinear", align_corners=False),
            nn.Conv3d(64, 32, kernel_size=(4, 1, 1), stride=(4, 1, 1), padding=0),
            nn.ReLU(inplace=True),
            nn.Conv3d(32, 32, kernel_size=(1, 3, 3), stride=(1, 1, 1), padding=(0, 1, 1)),
            nn.ReLU(inplace=True),
            nn.Conv3d(32, 1, kernel_size=(1, 3, 3), stride=(1, 1, 1), padding=(0, 1, 1)),
        )
        self.adapter = Adapter(num_frames=cfg.DATA.NUM_FRAMES, stride=cfg.DATA.NUM_FRAMES // 4)
        self.sa_0 = SA(512, k=4)
        self.sa_1 = SA(512, k=2)
        self.sa_2 = SA(512, k=1)

    def _pack_lateral(self, k, split=None):
        """Packed entry conv(s) of latlayer_k.  split = C4: the input arrives as two tensors
        (v4 | vis_sync, the torch.cat of model/model_utils.py:559), so the K range is cut in two."""
        seq = getattr(self, "latlayer_%d" % k)
        if len(seq) == 3:
            w, b = _compose_lateral(seq[0], seq[1])
            stride = seq[1].stride
        else:
            w, b, stride = seq[0].weight.detach().float(), seq[0].bias.detach().float(), (1, 1, 1)
        if split is None:
            return (E.pack_conv(w, b, None, stride, (0, 0, 0)),)
        return (E.pack_conv(w[:, :split].contiguous(), b, None, stride, (0, 0, 0)),
                E.pack_conv(w[:, split:].contiguous(), None, None, stride, (0, 0, 0)))

    def _pack_decoder(self, split=None):
        r = self.readout
        # sa_0/1/2.conv_mask[0] are three 3x3x3 convs 512->32 over the SAME masks tensor: one conv 512->96
        sa_w, sa_b = zip(*[E.fold_bn(m.conv_mask[0].conv.weight, None, m.conv_mask[0].bn) for m in (self.sa_0, self.sa_1, self.sa_2)])
        sa_cat = E.pack_conv(torch.cat(sa_w, 0), torch.cat(sa_b, 0), None, (1, 1, 1), (1, 1, 1), E.ACT_RELU)
        return {
            "sa_cat": sa_cat,
            "lat": [self._pack_lateral(k, split if k == 3 else None) for k in range(4)],
            "r0": E.pack_conv(r[0].weight, r[0].bias),
            "r1": E.pack_conv(r[1].weight, r[1].bias, r[2], (1, 1, 1), (1, 1, 1), E.ACT_RELU),
            "r4": E.pack_conv(r[4].weight, r[4].bias, r[5], (1, 1, 1), (0, 1, 1), E.ACT_RELU),
            "r8": E.pack_conv(r[8].weight, r[8].bias, None, (4, 1, 1), (0, 0, 0), E.ACT_NONE),
            "r10": E.pack_conv(r[10].weight, r[10].bias, None, (1, 1, 1), (0, 1, 1), E.ACT_RELU),
            "r12": E.pack_conv(r[12].weight, r[12].bias, None, (1, 1, 1), (0, 1, 1), E.ACT_NONE),
        }

    def _lateral(self, pk, k, xs, out=None):
        y = E.conv(xs[0], pk["lat"][k][0])
        if len(xs) == 2:
            y = E.conv(xs[1], pk["lat"][k][1], res=y)
        return getattr(self, "latlayer_%d" % k)[-1].run(y, out=out)

    def _laterals_012(self, pk, v1, v2, v3):
        """latlayer_0..2 need only the motion encoder's first three maps: they run while the image branch is still busy.
        s0 is produced straight into its channel slice of the readout's 768-channel input."""
        B = v1.N
        cat = E.alloc(B, max(1, v1.T // (self.cfg.MODEL.LATERAL_STRIDE[0] if self.cfg.MODEL.LATERAL_BOOL[0] else 1)),
                      v1.H, v1.W, 4 * 192, v1.buf.device)
        s0 = self._lateral(pk, 0, [v1], out=cat.slice(0, 192))
        return cat, s0, self._lateral(pk, 1, [v2]), self._lateral(pk, 2, [v3])

    @torch.no_grad()
    def encode_frames(self, frames):
        """Per-frame half of the image branch (ConvNeXt-T + the two smoothing convs, model/model_utils.py:357-385) for
        frames [N,3,H,W]: (f1 [N,h,w,96], f0 [N,h/2,w/2,320]) dense fp32 tensors.  Nothing in it mixes frames, so a
        frame's features are the same in every window that contains it: cache them and pass them to
        forward(..., frame_feats=...) -- the clip loop of inference.py does."""
        self._check_eval()
        o1, o0 = self.image_encoder.run(frames.float()[:, :, None])
        return (o1.buf.view(o1.N, o1.H, o1.W, o1.ld)[..., :o1.C], o0.buf.view(o0.N, o0.H, o0.W, o0.ld)[..., :o0.C])

    @staticmethod
    def _wrap_frame_feats(frame_feats):
        outs = []
        for f in frame_feats:
            f = f.float().contiguous()
            n, h, w, c = f.shape
            outs.append(E.CL(f.view(-1), 0, n, 1, h, w, c, c))
        return outs

    def _premask(self, pk, masks):
        """The three SA modules' first convs (same input) as one 512->96 conv."""
        return E.conv(masks, pk["sa_cat"])

    def _fuse_readout(self, pk, cat, s0, s1, s2, s3, masks, pm):
        """SA gating, top-down fusion and readout (model/model_utils.py:566-572)."""
        B = s0.N
        self.sa_2.run(s2, masks, pm.slice(64, 32))
        E.upsample(s3, 2, dst=s2, accumulate=True)
        self.sa_1.run(s1, masks, pm.slice(32, 32))
        E.upsample(s2, 2, dst=s1, accumulate=True)
        E.upsample(s3, 4, dst=s1, accumulate=True)
        self.sa_0.run(s0, masks, pm.slice(0, 32))
        E.upsample(s1, 2, dst=s0, accumulate=True)
        E.upsample(s2, 4, dst=s0, accumulate=True)
        E.upsample(s3, 8, dst=s0, accumulate=True)
        E.upsample(s1, 2, dst=cat.slice(192, 192))
        E.upsample(s2, 4, dst=cat.slice(384, 192))
        E.upsample(s3, 8, dst=cat.slice(576, 192))
        y = E.conv(E.conv(E.conv(cat, pk["r0"]), pk["r1"]), pk["r4"])
        y = E.upsample(E.conv(y, pk["r8"]), 4, act=E.ACT_RELU)   # == relu(conv(4,1,1)(upsample(y))) of the reference
        y = E.conv(E.conv(y, pk["r10"]), pk["r12"])              # [B,1,H,W,1], ld 1
        E.logsumexp_sub(y.buf, B, y.H * y.W)
        return y.buf.view(B, y.H, y.W)

    def _try_load(self, module_loader, path, what):
        if os.path.exists(path):
            module_loader(path)
        else:
            print("[mspi_amd] %s not found at %s: keeping random initialisation" % (what, path))

    def _pack_clips(self, clips):
        name = self.cfg.MODEL.MOTION_ENCODER
        if name == "slowfast4x16":  # model/model_utils.py:521-524: slow pathway = frames 0, 4, 12, last
            return [torch.stack([clips[:, :, 0], clips[:, :, 4], clips[:, :, 12], clips[:, :, -1]], dim=2), clips]
        if "swin" in name or "morph" in name or name == "s3d":
            return clips
        return [clips]


class AudioVisualSaliencyModel(_SaliencyBase):
    """forward(clips [B,3,T,H,W], audios [B,1,257,Wa]) -> (log-probability map [B,H,W], loss_av scalar)."""

    def __init__(self, cfg, vis_embed_dims=(96, 192, 384, 768), aud_embed_dim=512, de_embed_dim=192,
                 num_vis_tokens=4 * 7 * 7, norm=nn.LayerNorm):
        super().__init__()
        print("Motion Encoder is {}.".format(cfg.MODEL.MOTION_ENCODER))
        self.cfg = cfg
        vis_embed_dims = cfg.MODEL.MOTION_ENCODER_EMBEDS[cfg.MODEL.MOTION_ENCODER]
        num_vis_tokens = cfg.MODEL.NUM_VIS_TOKENS[cfg.MODEL.MOTION_ENCODER]
        self.audnet = get_resnet18(pretrained=False)
        self.image_encoder = StaticSaliencyModelConvNext()
        self.visnet = video_motion_extractor(cfg)
        self.aud_vis_sync_block = SyncBlock(num_blocks=3, num_vis_tokens=num_vis_tokens,
                                            num_aud_tokens=cfg.MODEL.get("NUM_AUD_TOKENS", 36),
                                            vis_in_embed=vis_embed_dims[-1], embed_dim=aud_embed_dim)
        self.aud_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.vis_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
        h = 2048

        def projector():
            return nn.Sequential(nn.Linear(aud_embed_dim, h), norm(h), nn.ReLU(inplace=True), nn.Linear(h, h), norm(h),
                                 nn.ReLU(inplace=True), nn.Linear(h, h), norm(h))

        def predictor():
            return nn.Sequential(nn.Linear(h, 512), norm(512), nn.ReLU(), nn.Linear(512, h))

        self.vis_projector = projector()
        self.mlp_vis = predictor()
        self.aud_projector = projector()
        self.mlp_aud = predictor()
        self._build_decoder(cfg, vis_embed_dims, aud_embed_dim, de_embed_dim)
        # pretrained weights (model/model_utils.py:512-514); absent files keep the random init
        self._try_load(self.visnet.load_weight, cfg.MODEL.MOTION_ENCODER_WEIGHT, "motion encoder weights")
        self._try_load(lambda p: self.audnet.load_state_dict(torch.load(p, map_location="cpu")),
                       cfg.MODEL.AUDIO_ENCODER_WEIGHT, "audio encoder weights")
        self._try_load(lambda p: self.image_encoder.load_state_dict(torch.load(p, map_location="cpu"), strict=False),
                       cfg.MODEL.IMAGE_SALIENCY_ENCODER_WEIGHT, "image saliency encoder weights")

    def frozen_encoder(self):
        self.audnet.eval()
        self.image_encoder.eval()

    def _pack(self):
        c4 = self.cfg.MODEL.MOTION_ENCODER_EMBEDS[self.cfg.MODEL.MOTION_ENCODER][-1]
        pk = self._pack_decoder(split=c4)

        def lin_ln(lin, ln):
            return E.pack_conv(lin.weight, lin.bias), _f(ln.weight), _f(ln.bias)

        for name in ("vis", "aud"):
            pr = getattr(self, name + "_projector")
            ml = getattr(self, "mlp_" + name)
            pk[name] = [lin_ln(pr[0], pr[1]), lin_ln(pr[3], pr[4]), lin_ln(pr[6], pr[7]), lin_ln(ml[0], ml[1]),
                        E.pack_conv(ml[3].weight, ml[3].bias)]
        return pk

    @staticmethod
    def _embed(p, x):
        """projector (Linear-LN-ReLU x2, Linear-LN) then predictor (Linear-LN-ReLU, Linear) on [B,512] rows."""
        for i, act in ((0, E.ACT_RELU), (1, E.ACT_RELU), (2, E.ACT_NONE)):
            x = E.layernorm(E.conv(x, p[i][0]), p[i][1], p[i][2], 1e-5, act=act)
        emb = x
        y = E.layernorm(E.conv(emb, p[3][0]), p[3][1], p[3][2], 1e-5, act=E.ACT_RELU)
        return emb, E.conv(y, p[4])

    @torch.no_grad()
    def forward(self, clips, audios, frame_feats=None):
        """frame_feats: optional (f1 [B*T,h,w,96], f0 [B*T,h/2,w/2,320]) from encode_frames() for the clips' frames in
        (b t) order -- the image branch then starts at the adapter (sliding-window inference re-uses 15 of 16 frames)."""
        self._check_eval()
        pk = self.pk
        clips = clips.float()
        B, dev = clips.shape[0], clips.device
        # Three independent branches (image encoder + adapter | audio encoder | motion encoder) on three HIP streams:
        # the backbones' many small launches (X3D: ~330 kernels of 20-30 us with ragged tails) overlap the image
        # encoder's large GEMMs instead of each draining the chip alone.  Fork / join by events, so a hipGraph
        # capture records them as parallel branches.
        with _Fork(self, dev) as fk:
            with fk.branch(0):
                o1, o0 = self.image_encoder.run(clips) if frame_feats is None else self._wrap_frame_feats(frame_feats)
                masks = self.adapter.run(o1, o0)
                pm = self._premask(pk, masks)
            with fk.branch(1):
                aud = self.audnet.forward_cl(audios.float())
            v1, v2, v3, v4 = self.visnet.forward_cl(self._pack_clips(clips))
            cat, s0, s1, s2 = self._laterals_012(pk, v1, v2, v3)
            fk.join(1)
            s3, loss = self._sync_and_lateral3(pk, v4, aud, fk)
        return self._fuse_readout(pk, cat, s0, s1, s2, s3, masks, pm), loss

    def _sync_and_lateral3(self, pk, v4, aud, fk):
        B, dev = v4.N, v4.buf.device
        x = self.aud_vis_sync_block.run(v4, aud)
        Rv = v4.T * v4.H * v4.W
        vis_fea = x.tokens(0, v4.T, v4.H, v4.W)
        aud_fea = x.tokens(Rv, aud.T, aud.H, aud.W)
        # Contrastive branch: its value is the second return (model/model_utils.py:545-552), nothing on the map's path
        # reads it -- a dozen tiny launches (M = B rows) that go to the audio branch's now idle stream.
        fk.keep(x)   # main-stream memory read by the side stream: must not return to the allocator before the join
        with fk.branch(1, refork=True):
            pooled = torch.empty(2, B, 512, dtype=torch.float32, device=dev)
            E.mean_rows(vis_fea, B, Rv, pooled[0])
            E.mean_rows(aud_fea, B, aud.T * aud.H * aud.W, pooled[1])
            vis_emb, vis_pred = self._embed(pk["vis"], E.from_rows(pooled[0]))
            aud_emb, aud_pred = self._embed(pk["aud"], E.from_rows(pooled[1]))
            loss = torch.empty(1, dtype=torch.float32, device=dev)
            E.neg_cosine(vis_pred, aud_emb, loss, 0.5, False)
            E.neg_cosine(aud_pred, vis_emb, loss, 0.5, True)
        return self._lateral(pk, 3, [v4, vis_fea]), loss[0]


class VisualSaliencyModel(_SaliencyBase):
    """forward(clips) -> (log-probability map [B,H,W], 0)  (model/model_utils.py:576-702)."""

    def __init__(self, cfg, vis_embed_dims=(96, 192, 384, 768), aud_embed_dim=512, de_embed_dim=192,
                 num_vis_tokens=4 * 7 * 7, norm=nn.LayerNorm):
        super().__init__()
        print("Motion Encoder is {}.".format(cfg.MODEL.MOTION_ENCODER))
        self.cfg = cfg
        vis_embed_dims = cfg.MODEL.MOTION_ENCODER_EMBEDS[cfg.MODEL.MOTION_ENCODER]
        self.image_encoder = StaticSaliencyModelConvNext()
        self.visnet = video_motion_extractor(cfg)
        self._build_decoder(cfg, vis_embed_dims, 0, de_embed_dim)
        self._try_load(self.visnet.load_weight, cfg.MODEL.MOTION_ENCODER_WEIGHT, "motion encoder weights")
        self._try_load(lambda p: self.image_encoder.load_state_dict(torch.load(p, map_location="cpu"), strict=False),
                       cfg.MODEL.IMAGE_SALIENCY_ENCODER_WEIGHT, "image saliency encoder weights")

    def frozen_encoder(self):
        self.image_encoder.eval()

    def _pack(self):
        return self._pack_decoder(split=None)

    @torch.no_grad()
    def forward(self, clips, frame_feats=None):
        self._check_eval()
        pk = self.pk
        clips = clips.float()
        with _Fork(self, clips.device) as fk:
            with fk.branch(0):
                o1, o0 = self.image_encoder.run(clips) if frame_feats is None else self._wrap_frame_feats(frame_feats)
                masks = self.adapter.run(o1, o0)
                pm = self._premask(pk, masks)
            v1, v2, v3, v4 = self.visnet.forward_cl(self._pack_clips(clips))
            cat, s0, s1, s2 = self._laterals_012(pk, v1, v2, v3)
            s3 = self._lateral(pk, 3, [v4])
        return self._fuse_readout(pk, cat, s0, s1, s2, s3, masks, pm), 0
