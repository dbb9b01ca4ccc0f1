"""3-D ResNet stem / stage building blocks shared by X3D and SlowFast, HIP-backed.

Module and attribute names follow the reference (SlowFast/stem_helper.py:21-290,
SlowFast/resnet_helper.py:27-825) so checkpoints load key-for-key; the torch layers only
hold parameters.  `run()` executes the block on channels-last activations (engine.CL):

  X3D block   a:1x1x1+BN+ReLU -> b:dw3x3x3+BN [-> SE] -> Swish -> c:1x1x1+BN (+skip) -> ReLU
     = conv(a) ; dwconv(b, pool sums) ; se_gate ; conv(c, gate+swish on the A operand, +res, ReLU)
  bottleneck  a:Tx1x1+BN+ReLU -> b:1x3x3+BN+ReLU -> c:1x1x1+BN (+skip) -> ReLU  = 3 (4) convs
"""
import os

import torch
import torch.nn as nn

from .. import engine as E
from ..module import HipModule


def _t3(v):
    return tuple(v) if isinstance(v, (list, tuple)) else (v, v, v)


class X3DStem(HipModule):
    """conv_xy (1xkxk, stride) -> depthwise temporal conv -> BN -> ReLU (SlowFast/stem_helper.py:207-290)."""

    def __init__(self, dim_in, dim_out, kernel, stride, padding, eps=1e-5, bn_mmt=0.1):
        super().__init__()
        self.kernel, self.stride, self.padding = kernel, stride, padding
        self.conv_xy = nn.Conv3d(dim_in, dim_out, (1, kernel[1], kernel[2]), (1, stride[1], stride[2]),
                                 (0, padding[1], padding[2]), bias=False)
        self.conv = nn.Conv3d(dim_out, dim_out, (kernel[0], 1, 1), (stride[0], 1, 1), (padding[0], 0, 0), bias=False,
                              groups=dim_out)
        self.bn = nn.BatchNorm3d(dim_out, eps=eps, momentum=bn_mmt)
        self.relu = nn.ReLU(True)

    def _pack(self):
        cxy, ct = self.conv_xy, self.conv
        return (E.pack_conv(cxy.weight, None, None, cxy.stride, cxy.padding, E.ACT_NONE),
                E.pack_dwconv(ct.weight, None, self.bn, ct.stride, ct.padding, E.ACT_RELU))

    def run(self, x):
        pxy, pt = self.pk
        return E.dwconv(E.conv(x, pxy), pt)


class ResNetBasicStem(HipModule):
    """conv -> BN -> ReLU -> maxpool(1,3,3)/(1,2,2) (SlowFast/stem_helper.py:128-204)."""

    def __init__(self, dim_in, dim_out, kernel, stride, padding, eps=1e-5, bn_mmt=0.1):
        super().__init__()
        self.conv = nn.Conv3d(dim_in, dim_out, _t3(kernel), _t3(stride), _t3(padding), bias=False)
        self.bn = nn.BatchNorm3d(dim_out, eps=eps, momentum=bn_mmt)
        self.relu = nn.ReLU(True)
        self.pool_layer = nn.MaxPool3d(kernel_size=[1, 3, 3], stride=[1, 2, 2], padding=[0, 1, 1])

    WIDE = 4   # output positions along W computed by one GEMM column group in the narrow-stem form

    def _pack(self):
        c = self.conv
        pk = {"conv": E.pack_conv(c.weight, None, self.bn, c.stride, c.padding, E.ACT_RELU), "wide": None}
        kw, sw, pw = c.kernel_size[2], c.stride[2], c.padding[2]
        if c.out_channels <= 8 and sw == 2 and kw % 2 == 1 and pw == kw // 2:
            # Narrow stem (SlowFast's fast pathway: 3 -> 8 channels, (5,7,7)/(1,2,2), 9.6 % of the slowfast4x16 step): as
            # a GEMM it has N = 8, a quarter of one 32-wide MFMA tile.  Four neighbouring outputs along W read one
            # 13-wide window (7 + 3*2), so the SAME conv is a (5,7,13)/(1,2,8) conv with 4*8 = 32 output channels whose
            # weights are the original taps shifted by 2j for output j (zeros elsewhere): the channels-last result
            # [.., W/8, (j, co)] IS [.., W/2, co].  2.15x fewer MFMAs and gathered activations per output.
            w, b = E.fold_bn(c.weight, None, self.bn)
            G, co = self.WIDE, c.out_channels
            wide = torch.zeros(G * co, w.shape[1], w.shape[2], w.shape[3], kw + (G - 1) * sw, device=w.device)
            for j in range(G):
                wide[j * co:(j + 1) * co, :, :, :, j * sw:j * sw + kw] = w
            pk["wide"] = E.pack_conv(wide, b.repeat(G), None, (c.stride[0], c.stride[1], G * sw), c.padding, E.ACT_RELU)
        return pk

    def run(self, x, out=None):
        pk = self.pk
        W = x.W if isinstance(x, E.CL) else x.shape[-1]
        if pk["wide"] is not None and W % (2 * self.WIDE) == 0:
            y = E.conv(x, pk["wide"])                                     # [N,T,Ho,Wo/4, 4*Cout]
            y = E.CL(y.buf, y.off, y.N, y.T, y.H, y.W * self.WIDE, self.conv.out_channels, self.conv.out_channels)
        else:
            y = E.conv(x, pk["conv"])
        return E.maxpool(y, (1, 3, 3), (1, 2, 2), (0, 1, 1), out=out)


class VideoModelStem(HipModule):
    def __init__(self, dim_in, dim_out, kernel, stride, padding, eps=1e-5, bn_mmt=0.1, stem_func_name="basic_stem"):
        super().__init__()
        assert len({len(dim_in), len(dim_out), len(kernel), len(stride), len(padding)}) == 1
        self.num_pathways = len(dim_in)
        stem = {"x3d_stem": X3DStem, "basic_stem": ResNetBasicStem}[stem_func_name]
        for p in range(self.num_pathways):
            self.add_module("pathway{}_stem".format(p),
                            stem(dim_in[p], dim_out[p], kernel[p], stride[p], padding[p], eps, bn_mmt))

    def run(self, xs, outs=None):
        assert len(xs) == self.num_pathways
        if outs is None:
            return [getattr(self, "pathway{}_stem".format(p)).run(xs[p]) for p in range(self.num_pathways)]
        return [getattr(self, "pathway{}_stem".format(p)).run(xs[p], out=outs[p]) for p in range(self.num_pathways)]


def _se_width(width, ratio, min_width=8, divisor=8):
    # SlowFast/resnet_helper.py:34-46
    if not ratio:
        return width
    width *= ratio
    out = max(min_width, int(width + divisor / 2) // divisor * divisor)
    if out < 0.9 * width:
        out += divisor
    return int(out)


class SE(nn.Module):
    """Parameter holder for squeeze-excite (SlowFast/resnet_helper.py:27-73); computed by mspi_se_gate."""

    def __init__(self, dim_in, ratio):
        super().__init__()
        dim_fc = _se_width(dim_in, ratio)
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc1 = nn.Conv3d(dim_in, dim_fc, 1, bias=True)
        self.fc1_act = nn.ReLU()
        self.fc2 = nn.Conv3d(dim_fc, dim_in, 1, bias=True)
        self.fc2_sig = nn.Sigmoid()


class Swish(nn.Module):
    pass


# Fused a+b kernel (csrc/x3d_block.hip) vs thin GEMM + depthwise kernel.  Measured per launch at batch 8, hipGraph-timed
# (tools/x3d_ab_bench.py): 56x56 maps 93.7 vs 100.2 us, 28x28 62.5 vs 58.9, 14x14 50.2 vs 40.0, 7x7 43.5 vs 30.4 -- the
# fused form wins only where the maps are large enough to be bandwidth-bound, so "auto" fuses those.
# MSPI_X3D_FUSE = 0 (never) | 1 (every stride-1 block the kernel covers) | auto (default)
FUSE_AB = os.environ.get("MSPI_X3D_FUSE", "auto")
FUSE_MIN_W = 56
# Block seam (`c` of block i + `a` of block i+1 in one launch, csrc/mlp_fused.hip): MSPI_X3D_SEAM = 0 switches it off (A/B)
SEAM = os.environ.get("MSPI_X3D_SEAM", "1")


class X3DTransform(HipModule):
    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups, stride_1x1=False,
                 eps=1e-5, bn_mmt=0.1, dilation=1, se_ratio=0.0625, swish_inner=True, block_idx=0):
        super().__init__()
        assert num_groups == dim_inner and dilation == 1 and swish_inner, "X3D blocks are channel-wise 3x3x3 + Swish"
        s1, s3 = (stride, 1) if stride_1x1 else (1, stride)
        tk = temp_kernel_size
        self.a = nn.Conv3d(dim_in, dim_inner, (1, 1, 1), (1, s1, s1), 0, bias=False)
        self.a_bn = nn.BatchNorm3d(dim_inner, eps=eps, momentum=bn_mmt)
        self.a_relu = nn.ReLU(True)
        self.b = nn.Conv3d(dim_inner, dim_inner, (tk, 3, 3), (1, s3, s3), (tk // 2, 1, 1), groups=num_groups, bias=False)
        self.b_bn = nn.BatchNorm3d(dim_inner, eps=eps, momentum=bn_mmt)
        if se_ratio > 0.0 and (block_idx + 1) % 2:
            self.se = SE(dim_inner, se_ratio)
        self.b_relu = Swish()
        self.c = nn.Conv3d(dim_inner, dim_out, 1, 1, 0, bias=False)
        self.c_bn = nn.BatchNorm3d(dim_out, eps=eps, momentum=bn_mmt)

    def _pack(self):
        cs_in = E.rup4(self.a.in_channels)
        cs_mid = E.rup4(self.a.out_channels)
        has_se = hasattr(self, "se")
        pk = {
            "a": E.pack_conv(self.a.weight, None, self.a_bn, self.a.stride, (0, 0, 0), E.ACT_RELU, cin_stored=cs_in),
            "b": E.pack_dwconv(self.b.weight, None, self.b_bn, self.b.stride, self.b.padding,
                               E.ACT_NONE if has_se else E.ACT_SWISH),
            "c": E.pack_conv(self.c.weight, None, self.c_bn, (1, 1, 1), (0, 0, 0), E.ACT_RELU, cin_stored=cs_mid),
        }
        # stride-1 blocks: `a` + `b` as one launch, the 2.25x-wide tensor between them stays in LDS (csrc/x3d_block.hip)
        pk["ab"] = E.pack_x3d_ab(pk["a"], pk["b"]) if FUSE_AB != "0" else None
        if has_se:
            f, c = self.se.fc1.out_channels, self.se.fc1.in_channels
            dev = self.se.fc1.weight.device
            w1 = torch.zeros(f, cs_mid, device=dev)
            w1[:, :c] = self.se.fc1.weight.detach().float().view(f, c)
            w2 = torch.zeros(cs_mid, f, device=dev)
            w2[:c] = self.se.fc2.weight.detach().float().view(c, f)
            pk["se"] = (w1.contiguous(), self.se.fc1.bias.detach().float().contiguous(), w2.contiguous(),
                        E._pad_vec(self.se.fc2.bias, cs_mid))
        return pk

    def uses_ab(self, x):
        """Whether run(x) takes the fused a + b kernel (which computes `a` itself: no seam into this block)."""
        pk = self.pk
        return pk["ab"] is not None and (FUSE_AB == "1" or x.W >= FUSE_MIN_W) and E.x3d_ab_supported(x, pk["ab"])

    def run(self, x, res, out=None, t=None, seam=None):
        """res: skip tensor added before the final ReLU.
        t: relu(a_bn(a(x))), when the previous block's seam launch already produced it.
        seam: PackedX3dCa of this block's `c` and the next block's `a` -> returns (y, t of the next block)."""
        pk = self.pk
        se = "se" in pk
        # the fused kernel is built from `a`'s f16x3 planes: it follows that pack's first-sight range check
        if t is None and self.uses_ab(x) and E.range_check_input(pk["a"], x):
            u = E.x3d_ab(x, pk["ab"], pool=se)
        else:
            u = E.dwconv(E.conv(x, pk["a"]) if t is None else t, pk["b"], pool=se)
        gate = None
        if se:
            u, part = u
            gate = E.se_gate(part, 1.0 / (u.T * u.H * u.W), *pk["se"])
        if seam is not None:
            return E.x3d_ca(u, seam, res, gate=gate)
        return E.conv(u, pk["c"], res=res, gate=gate, out=out)


class BottleneckTransform(HipModule):
    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups, stride_1x1=False,
                 eps=1e-5, bn_mmt=0.1, dilation=1, block_idx=0):
        super().__init__()
        assert num_groups == 1 and dilation == 1
        s1, s3 = (stride, 1) if stride_1x1 else (1, stride)
        tk = temp_kernel_size
        self.a = nn.Conv3d(dim_in, dim_inner, (tk, 1, 1), (1, s1, s1), (tk // 2, 0, 0), bias=False)
        self.a_bn = nn.BatchNorm3d(dim_inner, eps=eps, momentum=bn_mmt)
        self.a_relu = nn.ReLU(True)
        self.b = nn.Conv3d(dim_inner, dim_inner, (1, 3, 3), (1, s3, s3), (0, 1, 1), bias=False)
        self.b_bn = nn.BatchNorm3d(dim_inner, eps=eps, momentum=bn_mmt)
        self.b_relu = nn.ReLU(True)
        self.c = nn.Conv3d(dim_inner, dim_out, 1, 1, 0, bias=False)
        self.c_bn = nn.BatchNorm3d(dim_out, eps=eps, momentum=bn_mmt)

    def _pack(self):
        return {n: E.pack_conv(c.weight, None, bn, c.stride, c.padding, E.ACT_RELU, cin_stored=E.rup4(c.in_channels))
                for n, c, bn in (("a", self.a, self.a_bn), ("b", self.b, self.b_bn), ("c", self.c, self.c_bn))}

    def run(self, x, res, out=None):
        pk = self.pk
        return E.conv(E.conv(E.conv(x, pk["a"]), pk["b"]), pk["c"], res=res, out=out)


class ResBlock(HipModule):
    """relu(skip(x) + branch2(x)) (SlowFast/resnet_helper.py:490-616); the add and the ReLU are the
    epilogue of branch2's last conv."""

    def __init__(self, dim_in, dim_out, temp_kernel_size, stride, trans_func, dim_inner, num_groups=1,
                 stride_1x1=False, eps=1e-5, bn_mmt=0.1, dilation=1, block_idx=0):
        super().__init__()
        if dim_in != dim_out or stride != 1:
            self.branch1 = nn.Conv3d(dim_in, dim_out, 1, (1, stride, stride), 0, bias=False)
            self.branch1_bn = nn.BatchNorm3d(dim_out, eps=eps, momentum=bn_mmt)
        self.branch2 = trans_func(dim_in, dim_out, temp_kernel_size, stride, dim_inner, num_groups,
                                  stride_1x1=stride_1x1, dilation=dilation, block_idx=block_idx)
        self.relu = nn.ReLU(True)

    def _pack(self):
        if hasattr(self, "branch1"):
            c = self.branch1
            return E.pack_conv(c.weight, None, self.branch1_bn, c.stride, (0, 0, 0), E.ACT_NONE,
                               cin_stored=E.rup4(c.in_channels))
        return None

    def run(self, x, out=None, **seam):
        skip = E.conv(x, self.pk) if self.pk is not None else x
        return self.branch2.run(x, skip, out=out, **seam)


class ResStage(HipModule):
    def __init__(self, dim_in, dim_out, stride, temp_kernel_sizes, num_blocks, dim_inner, num_groups,
                 num_block_temp_kernel, nonlocal_inds, trans_func_name="bottleneck_transform", stride_1x1=False,
                 dilation=None):
        super().__init__()
        assert all(len(n) == 0 for n in nonlocal_inds), "non-local blocks are not instantiated by MSPI's configs"
        self.num_blocks = num_blocks
        self.num_pathways = len(num_blocks)
        tks = [(temp_kernel_sizes[i] * num_blocks[i])[: num_block_temp_kernel[i]]
               + [1] * (num_blocks[i] - num_block_temp_kernel[i]) for i in range(len(temp_kernel_sizes))]
        trans = {"bottleneck_transform": BottleneckTransform, "x3d_transform": X3DTransform}[trans_func_name]
        for p in range(self.num_pathways):
            for i in range(num_blocks[p]):
                blk = ResBlock(dim_in[p] if i == 0 else dim_out[p], dim_out[p], tks[p][i], stride[p] if i == 0 else 1,
                               trans, dim_inner[p], num_groups[p], stride_1x1=stride_1x1,
                               dilation=1 if dilation is None else dilation[p], block_idx=i)
                self.add_module("pathway{}_res{}".format(p, i), blk)

    def blocks(self, p):
        return [getattr(self, "pathway{}_res{}".format(p, i)) for i in range(self.num_blocks[p])]

    def _pack(self):
        """Per pathway, per block i: the seam pack of block i's `c` and block i+1's `a` (X3D stride-1 neighbours), or None."""
        seams = []
        for p in range(self.num_pathways):
            bl = self.blocks(p)
            row = [None] * len(bl)
            for i in range(len(bl) - 1):
                a, b = bl[i].branch2, bl[i + 1].branch2
                if SEAM != "0" and isinstance(a, X3DTransform) and isinstance(b, X3DTransform) and not hasattr(bl[i + 1], "branch1") \
                        and tuple(b.a.stride) == (1, 1, 1):
                    row[i] = E.pack_x3d_ca(a.pk["c"], b.pk["a"])
            seams.append(row)
        return seams

    def run(self, xs, outs=None, pathways=None):
        """outs[p]: where pathway p's last block writes (a channel slice of a concat buffer).
        pathways: subset to compute."""
        out = []
        for p in range(self.num_pathways):
            if pathways is not None and p not in pathways:
                out.append(None)
                continue
            x = xs[p]
            blocks = self.blocks(p)
            seams = self.pk[p]
            t = None
            for bi, b in enumerate(blocks):
                kw = {}
                if t is not None:
                    kw["t"] = t
                    t = None
                sm = seams[bi]
                # a seam launch stands for two packed layers: it is used once both have passed their first-sight range check
                # (the tuning forward runs them as separate launches) and while both are still on the f16x3 path
                nxt_ab = sm is not None and blocks[bi + 1].branch2.pk["ab"] is not None and \
                    (FUSE_AB == "1" or x.W // b.branch2.b.stride[2] >= FUSE_MIN_W)      # the next block computes its own `a`
                if sm is not None and not nxt_ab:
                    pc, pa = b.branch2.pk["c"], blocks[bi + 1].branch2.pk["a"]
                    if pc.prec == pa.prec == E.PREC_F16X3 and not (E._tuning() and not (pc.checked and pa.checked)):
                        x, t = b.run(x, seam=sm, **kw)
                        continue
                x = b.run(x, out=outs[p] if (outs is not None and bi == len(blocks) - 1) else None, **kw)
            out.append(x)
        return out
