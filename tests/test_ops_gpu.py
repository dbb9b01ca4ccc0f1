"""Op-level parity of every C-ABI kernel against a plain PyTorch fp32 CPU reference of the same op.
(Module- and model-level parity against the oracle / golden fixtures: test_parity_gpu.py.)"""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(t5, dev, ld=None):
    """NCDHW cpu tensor -> engine.CL on the GPU (pad channels poisoned with NaN-free junk)."""
    from mspi_amd import engine as E
    N, C, T, H, W = t5.shape
    x = E.alloc(N, T, H, W, C, dev, ld=ld)
    x.buf.fill_(7.0)
    x.as_ncdhw().copy_(t5.to(dev))
    if x.ld > C and x.Cs > C:   # stored pad channels must be finite zeros for the next layer
        x.buf.view(-1, x.ld)[:, C:x.Cs] = 0
    return x


def _close(got, ref, tol, what=""):
    err = (got.cpu() - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    assert err <= tol * scale, "%s: max abs err %.3e (scale %.2f)" % (what, err, scale)


CONV_CASES = [
    # (N, Cin, T, H, W, Cout, k, stride, pad, raw_input)
    (2, 3, 4, 17, 19, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1), True),      # x3d stem conv_xy on NCDHW
    (2, 3, 4, 16, 16, 96, (1, 4, 4), (1, 4, 4), (0, 0, 0), True),      # convnext stem
    (1, 1, 1, 33, 29, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3), True),      # audio conv1
    (2, 24, 3, 9, 11, 54, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),     # x3d a (54 -> padded 56)
    (2, 54, 3, 9, 11, 24, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),     # x3d c (padded K)
    (2, 24, 3, 9, 11, 48, (1, 1, 1), (1, 2, 2), (0, 0, 0), False),     # branch1 strided
    (1, 192, 4, 6, 7, 192, (3, 3, 3), (1, 1, 1), (1, 1, 1), False),    # readout 3x3x3
    (2, 64, 1, 10, 9, 128, (1, 3, 3), (1, 2, 2), (0, 1, 1), False),    # resnet strided 3x3
    (2, 192, 8, 5, 5, 192, (4, 1, 1), (4, 1, 1), (0, 0, 0), False),    # lateral temporal
    (1, 208, 4, 5, 5, 208, (3, 1, 1), (1, 1, 1), (1, 0, 0), False),    # sepconv temporal
    (3, 512, 1, 1, 1, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),   # tiny-M linear
    (1, 32, 1, 20, 20, 1, (1, 3, 3), (1, 1, 1), (0, 1, 1), False),     # single output channel
    (1, 96, 1, 40, 40, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0), False),   # convnext fc1 (128x128 tiles)
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("prec", [0, 1], ids=["f32", "f16x3"])
def test_conv(dev, case, act, prec):
    from mspi_amd import engine as E
    N, Cin, T, H, W, Cout, k, s, p, raw = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(N, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g)
    ref = F.conv3d(x, w, b, s, p)
    res = torch.randn_like(ref)
    ref = ref + res
    ref = {0: ref, 1: F.relu(ref), 2: F.gelu(ref)}[act]
    pk = E.pack_conv(w, b, None, s, p, act, cin_stored=Cin if raw else E.rup4(Cin), device=dev, prec=prec)
    xin = x.to(dev) if raw else _cl(x, dev)
    out = E.conv(xin, pk, res=_cl(res, dev))
    torch.cuda.synchronize()
    _close(out.as_ncdhw(Cout), ref, 2e-5, "conv %s" % (case,))
    if out.Cs > Cout:  # pad columns must hold act(0 + res_pad) = finite
        assert torch.isfinite(out.buf).all()


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14])
@pytest.mark.parametrize("shape", [
    (2, 96, 2, 9, 11, 200, (1, 1, 1), (1, 1, 1), (0, 0, 0)),     # plain GEMM, ragged M, N not a tile multiple
    (1, 56, 3, 10, 9, 96, (3, 3, 3), (1, 2, 2), (1, 1, 1)),      # strided multi-tap conv, padded channel count
    (2, 24, 4, 7, 7, 40, (1, 1, 1), (1, 1, 1), (0, 0, 0)),       # K smaller than one stage
])
def test_conv_every_kernel_instantiation(dev, tile, shape):
    """Each tile / loader instantiation the autotuner may pick (register-staged and LDS-DMA kernels) on the same data."""
    from mspi_amd import engine as E
    N, Cin, T, H, W, Cout, k, s, p = shape
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(N, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g)
    gate = torch.rand(N, E.rup4(Cin), generator=g) if k == (1, 1, 1) else None
    xa = x if gate is None else x * gate[:, :Cin].view(N, Cin, 1, 1, 1)
    xa = xa if gate is None else xa * torch.sigmoid(xa)
    ref = F.conv3d(xa, w, b, s, p)
    res = torch.randn_like(ref)
    ref = F.relu(ref + res)
    pk = E.pack_conv(w, b, None, s, p, E.ACT_RELU, cin_stored=E.rup4(Cin), device=dev, prec=E.PREC_F16X3)
    out = E.conv(_cl(x, dev), pk, res=_cl(res, dev), gate=None if gate is None else gate.to(dev), tile=tile)
    _close(out.as_ncdhw(Cout), ref, 2e-5, "conv tile %d %s" % (tile, shape))


def test_conv_f16x3_wide_dynamic_range(dev):
    """The split product must stay fp32-accurate for operands spanning many binades (tiny and large
    activations in one row, weights from 1e-4 to 10) -- plain f16 would be off by 1e-3 here."""
    from mspi_amd import engine as E
    g = torch.Generator().manual_seed(11)
    M, K, Nn = 300, 416, 136
    x = torch.randn(1, K, 1, 1, M, generator=g) * torch.logspace(-3, 2.5, K).view(1, K, 1, 1, 1)
    w = torch.randn(Nn, K, 1, 1, 1, generator=g) * torch.logspace(-4, 1, Nn).view(Nn, 1, 1, 1, 1) / math.sqrt(K)
    ref = F.conv3d(x.double(), w.double()).float()
    out = E.conv(_cl(x, dev), E.pack_conv(w, device=dev, prec=1))
    got = out.as_ncdhw(Nn).cpu()
    rowscale = (x.abs().double().flatten(1, 3).transpose(1, 2) @ w.abs().double().flatten(1).t()).transpose(1, 2).view_as(ref)
    assert ((got - ref).abs() / rowscale.float().clamp_min(1e-30)).max().item() < 3e-6   # ~fp32 dot-product error bound


def test_conv_gate_and_slices(dev):
    """SE gate + Swish prologue; output into a channel slice of a wider buffer; strided token-slab input."""
    from mspi_amd import engine as E
    g = torch.Generator().manual_seed(5)
    N, C, T, H, W, Co = 2, 54, 2, 5, 6, 24
    x = torch.randn(N, C, T, H, W, generator=g)
    gate = torch.rand(N, C, generator=g)
    w = torch.randn(Co, C, 1, 1, 1, generator=g) / math.sqrt(C)
    xa = x * gate.view(N, C, 1, 1, 1)
    ref = F.relu(F.conv3d(xa * torch.sigmoid(xa), w))
    pk = E.pack_conv(w, None, None, act=E.ACT_RELU, cin_stored=56, device=dev)
    gpad = torch.zeros(N, 56, device=dev)
    gpad[:, :C] = gate.to(dev)
    wide = E.alloc(N, T, H, W, 64, dev)
    wide.buf.fill_(-3.0)
    out = E.conv(_cl(x, dev), pk, gate=gpad, out=wide.slice(32, Co))
    _close(out.as_ncdhw(), ref, 2e-5, "gated conv")
    assert (wide.as_ncdhw()[:, :32] == -3.0).all() and (wide.as_ncdhw()[:, 56:] == -3.0).all()
    # token slab: rows [5, 5+T*H*W) of a [N, R, C] sequence buffer
    R, Cc = 5 + T * H * W + 3, 48
    seq = torch.randn(N, R, Cc, generator=g)
    slab = E.CL(seq.to(dev).contiguous().view(-1), 0, N, R, 1, 1, Cc, Cc).tokens(5, T, H, W)
    w2 = torch.randn(Co, Cc, 2, 1, 1, generator=g) / math.sqrt(2 * Cc)
    x5 = seq[:, 5:5 + T * H * W].reshape(N, T, H, W, Cc).permute(0, 4, 1, 2, 3)
    out2 = E.conv(slab, E.pack_conv(w2, None, None, (2, 1, 1), (0, 0, 0), device=dev))
    _close(out2.as_ncdhw(), F.conv3d(x5, w2, None, (2, 1, 1)), 2e-5, "slab conv")


DW_CASES = [
    (2, 54, 4, 9, 10, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 108, 4, 9, 10, (3, 3, 3), (1, 2, 2), (1, 1, 1)),
    (1, 24, 6, 7, 7, (5, 1, 1), (1, 1, 1), (2, 0, 0)),
    (1, 192, 4, 9, 9, (7, 1, 1), (1, 1, 1), (3, 0, 0)),
    (1, 192, 2, 12, 13, (1, 7, 7), (1, 1, 1), (0, 3, 3)),
    (3, 96, 1, 14, 14, (1, 7, 7), (1, 1, 1), (0, 3, 3)),
    (2, 96, 4, 8, 8, (3, 3, 3), (1, 8, 8), (1, 1, 1)),
    (1, 56, 3, 11, 13, (3, 3, 3), (1, 2, 2), (1, 1, 1)),   # odd extents: ragged strips, stride 2
    (2, 432, 2, 7, 7, (3, 3, 3), (1, 1, 1), (1, 1, 1)),    # C/4 = 108 > strips per block
    (2, 64, 4, 9, 10, (5, 5, 5), (1, 1, 1), (2, 2, 2)),    # UniFormer's local "attention": 5-wide strip kernel
    (1, 128, 3, 7, 7, (5, 5, 5), (1, 1, 1), (2, 2, 2)),
    (1, 16, 2, 6, 9, (2, 5, 4), (1, 1, 2), (0, 2, 1)),     # non-square kernel: the generic kernel
]


@pytest.mark.parametrize("case", DW_CASES)
def test_dwconv_pool_maxpool(dev, case):
    from mspi_amd import engine as E
    N, C, T, H, W, k, s, p = case
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(N, C, T, H, W, generator=g)
    w = torch.randn(C, 1, *k, generator=g) / math.sqrt(k[0] * k[1] * k[2])
    b = torch.randn(C, generator=g)
    ref = F.conv3d(x, w, b, s, p, 1, C)
    pk = E.pack_dwconv(w, b, None, s, p, E.ACT_SWISH, device=dev)
    out = E.dwconv(_cl(x, dev), pk)
    _close(out.as_ncdhw(C), ref * torch.sigmoid(ref), 2e-5, "dwconv swish")
    out = E.dwconv(_cl(x, dev), pk, act=E.ACT_NONE)
    _close(out.as_ncdhw(C), ref, 2e-5, "dwconv")
    if k[1] == k[2] and k[2] in (3, 5, 7) and s[2] in (1, 2) and (k[2], s[2]) not in ((7, 2), (5, 2)):
        out, part = E.dwconv(_cl(x, dev), pk, pool=True, act=E.ACT_NONE)
        _close(out.as_ncdhw(C), ref, 2e-5, "dwconv (pool)")
        _close(part.sum(1)[:, :C], ref.sum((2, 3, 4)), 2e-5, "se pool partial sums")
        out2, part2 = E.dwconv(_cl(x, dev), pk, pool=True, act=E.ACT_NONE)
        assert torch.equal(part, part2) and torch.equal(out.buf, out2.buf)      # no atomics: bitwise reproducible
    if all(2 * pp <= kk for pp, kk in zip(p, k)):
        mp = E.maxpool(_cl(x, dev), k, s, p)
        _close(mp.as_ncdhw(C), F.max_pool3d(x, k, s, p), 0, "maxpool")


def test_se_gate(dev):
    from mspi_amd import engine as E
    g = torch.Generator().manual_seed(2)
    N, C, Fh = 3, 56, 8
    for rows, C in ((37, 56), (5, 432), (700, 216)):
        pool = torch.randn(N, rows, C, generator=g) * 50 / rows ** 0.5
        w1, b1 = torch.randn(Fh, C, generator=g) / 7, torch.randn(Fh, generator=g)
        w2, b2 = torch.randn(C, Fh, generator=g) / 3, torch.randn(C, generator=g)
        ref = torch.sigmoid(F.linear(F.relu(F.linear(pool.sum(1) / 100.0, w1, b1)), w2, b2))
        gate = E.se_gate(pool.to(dev), 1 / 100.0, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev))
        _close(gate, ref, 1e-5, "se gate")


@pytest.mark.parametrize("C", [96, 512, 768, 3072])
def test_layernorm(dev, C):
    from mspi_amd import engine as E
    g = torch.Generator().manual_seed(C)
    N, R = 3, 37
    x = torch.randn(N, R, C, generator=g) * 3 + 1
    gm, bt = torch.randn(C, generator=g), torch.randn(C, generator=g)
    tab = torch.randn(R, C, generator=g)
    ref = F.layer_norm(x, (C,), gm, bt, 1e-6)
    xc = E.CL(x.to(dev).view(-1), 0, N, R, 1, 1, C, C)
    out = E.layernorm(xc, gm.to(dev), bt.to(dev), 1e-6)
    _close(out.as_rows().view(N, R, C), ref, 2e-5, "layernorm")
    # into a token slab of a longer sequence, with table add and ReLU
    seq = E.alloc(N, R + 9, 1, 1, C, dev)
    seq.buf.fill_(5.0)
    E.layernorm(xc, gm.to(dev), bt.to(dev), 1e-6, out=seq.tokens(4, R, 1, 1), act=E.ACT_RELU, table=tab.to(dev))
    got = seq.buf.view(N, R + 9, C)
    _close(got[:, 4:4 + R], F.relu(ref) + tab, 2e-5, "layernorm slab")
    assert (got[:, :4] == 5.0).all() and (got[:, 4 + R:] == 5.0).all()


@pytest.mark.parametrize("B,H,N,D", [(2, 4, 197, 128), (1, 3, 392, 32), (2, 2, 130, 96), (1, 1, 31, 64), (2, 4, 874, 128)])
def test_attention(dev, B, H, N, D):
    from mspi_amd import engine as E
    g = torch.Generator().manual_seed(N)
    C = H * D
    qkv = torch.randn(B, N, 3 * C, generator=g)
    q, k, v = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4).double()
    scale = D ** -0.5
    ref = ((q @ k.transpose(-2, -1)) * scale).softmax(-1) @ v
    ref = ref.transpose(1, 2).reshape(B, N, C).float()
    xc = E.CL(qkv.to(dev).view(-1), 0, B, N, 1, 1, 3 * C, 3 * C)
    out = E.attention(xc, B, N, H, D, scale)
    _close(out.as_rows().view(B, N, C), ref, 2e-5, "attention")


def test_attention_bias_mask(dev):
    """Swin form: learned bias [heads,N,N] + per-window mask (0 / -100), both passed key-major."""
    from mspi_amd import engine as E
    B, H, N, D, nW = 6, 3, 98, 32, 3
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B, N, 3 * H * D, generator=g)
    bias = torch.randn(H, N, N, generator=g)
    mask = torch.where(torch.rand(nW, N, N, generator=g) < 0.3, torch.tensor(-100.0), torch.tensor(0.0))
    q, k, v = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4).double()
    att = (q @ k.transpose(-2, -1)) * D ** -0.5 + bias.double()[None] + mask.double()[torch.arange(B) % nW][:, None]
    ref = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * D).float()
    xc = E.CL(qkv.to(dev).view(-1), 0, B, N, 1, 1, 3 * H * D, 3 * H * D)
    out = E.attention(xc, B, N, H, D, D ** -0.5, biasT=bias.transpose(1, 2).contiguous().to(dev),
                      maskT=mask.transpose(1, 2).contiguous().to(dev))
    _close(out.as_rows().view(B, N, H * D), ref, 2e-5, "attention bias+mask")


def test_windowed_attention_and_space_to_depth(dev):
    """Shifted-window attention through the token index (no gather pass) vs roll + partition + reverse in torch."""
    from mspi_amd import engine as E
    from mspi_amd.backbones import video_swin_transformer as S
    B, D, H, W, heads, hd = 2, 4, 14, 14, 3, 32
    ws, ss = (4, 7, 7), (0, 3, 3)
    Cc = heads * hd
    g = torch.Generator().manual_seed(6)
    qkv = torch.randn(B, D, H, W, 3 * Cc, generator=g)
    N = ws[0] * ws[1] * ws[2]
    bias = torch.randn(heads, N, N, generator=g)
    mask = S.compute_mask(D, H, W, ws, ss)
    sh = torch.roll(qkv, (-ss[0], -ss[1], -ss[2]), (1, 2, 3))
    xw = sh.view(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], 3 * Cc).permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, N, 3 * Cc)
    q, k, v = xw.view(-1, N, 3, heads, hd).permute(2, 0, 3, 1, 4).double()
    nW = mask.shape[0]
    att = (q @ k.transpose(-2, -1)) * hd ** -0.5 + bias.double()[None]
    att = (att.view(B, nW, heads, N, N) + mask.double()[None, :, None]).view(-1, heads, N, N)
    o = (att.softmax(-1) @ v).transpose(1, 2).reshape(-1, N, Cc).float()
    o = o.view(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], Cc).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, Cc)
    ref = torch.roll(o, ss, (1, 2, 3))
    xc = E.CL(qkv.to(dev).contiguous().view(-1), 0, B, D, H, W, 3 * Cc, 3 * Cc)
    out = E.attention(xc, B * nW, N, heads, hd, hd ** -0.5, biasT=bias.transpose(1, 2).contiguous().to(dev),
                      maskT=mask.transpose(1, 2).contiguous().to(dev), tok_idx=S.window_token_index(D, H, W, ws, ss).to(dev))
    _close(out.as_ncdhw().permute(0, 2, 3, 4, 1), ref, 2e-5, "windowed attention")
    x = torch.randn(2, 96, 3, 6, 8, generator=g)
    xl = x.permute(0, 2, 3, 4, 1)
    refm = torch.cat([xl[:, :, 0::2, 0::2], xl[:, :, 1::2, 0::2], xl[:, :, 0::2, 1::2], xl[:, :, 1::2, 1::2]], -1)
    _close(E.space_to_depth(_cl(x, dev)).as_ncdhw().permute(0, 2, 3, 4, 1), refm, 0, "space_to_depth")


@pytest.mark.parametrize("q_thw,k_thw,heads", [((2, 6, 6), (2, 3, 3), 2), ((4, 7, 7), (4, 7, 7), 1), ((8, 14, 14), (8, 14, 14), 1)])
def test_mvit_attention(dev, q_thw, k_thw, heads):
    """Decomposed rel-pos (h, w, t) folded into the contraction + residual pooling vs the explicit formula."""
    from mspi_amd import engine as E
    hd, B = 96, 2
    g = torch.Generator().manual_seed(sum(q_thw))
    Nq, Nk = math.prod(q_thw), math.prod(k_thw)
    q = torch.randn(B, Nq, heads * hd, generator=g)
    k = torch.randn(B, Nk, heads * hd, generator=g)
    v = torch.randn(B, Nk, heads * hd, generator=g)
    # tables of distinct relative distances and the (own position, key position) -> row index, as MViT builds them
    tabs, dists = [], []
    for i in range(3):
        n = 2 * max(q_thw[i], k_thw[i]) - 1
        tabs.append(torch.randn(n, hd, generator=g) * 0.3)
        qr, kr = max(k_thw[i] / q_thw[i], 1.0), max(q_thw[i] / k_thw[i], 1.0)
        dists.append((torch.arange(q_thw[i])[:, None] * qr - torch.arange(k_thw[i])[None, :] * kr + (k_thw[i] - 1) * kr).long())
    Rt, Rh, Rw = (tabs[i][dists[i]] for i in range(3))
    qh = q.view(B, Nq, heads, hd).transpose(1, 2).double()
    kh = k.view(B, Nk, heads, hd).transpose(1, 2).double()
    vh = v.view(B, Nk, heads, hd).transpose(1, 2).double()
    rq = qh.reshape(B, heads, *q_thw, hd)
    att = (qh * hd ** -0.5) @ kh.transpose(-2, -1)
    att = (att.view(B, heads, *q_thw, *k_thw)
           + torch.einsum("bythwc,hkc->bythwk", rq, Rh.double())[:, :, :, :, :, None, :, None]
           + torch.einsum("bythwc,wkc->bythwk", rq, Rw.double())[:, :, :, :, :, None, None, :]
           + torch.einsum("bythwc,tkc->bythwk", rq, Rt.double())[:, :, :, :, :, :, None, None]).view(B, heads, Nq, Nk)
    ref = (att.softmax(-1) @ vh + qh).transpose(1, 2).reshape(B, Nq, heads * hd).float()

    def cl(t, thw):
        return E.CL(t.to(dev).contiguous().view(-1), 0, B, thw[0], thw[1], thw[2], heads * hd, heads * hd)

    out = E.mvit_attention(cl(q, q_thw), cl(k, k_thw), cl(v, k_thw), B, heads, hd, hd ** -0.5, q_thw, k_thw,
                           Rh.to(dev).contiguous(), Rw.to(dev).contiguous(), Rt.to(dev).contiguous())
    _close(out.as_rows().view(B, Nq, heads * hd), ref, 2e-5, "mvit attention")
    if E.DEFAULT_PREC == E.PREC_F16X3:
        # the q . R dot products as one thin GEMM against the stacked distinct rows + a gather (mspi_mvit_qk_augment_p)
        stack = torch.cat([tabs[1], tabs[2], tabs[0]], 0)           # h, w, t
        offs = (0, tabs[1].shape[0], tabs[1].shape[0] + tabs[2].shape[0])
        idx = [(dists[a].to(torch.int32) + o).contiguous().to(dev) for a, o in zip((1, 2, 0), offs)]
        rel = (E.pack_conv(stack, None, device=dev), idx[0], idx[1], idx[2])
        out2 = E.mvit_attention(cl(q, q_thw), cl(k, k_thw), cl(v, k_thw), B, heads, hd, hd ** -0.5, q_thw, k_thw,
                                Rh.to(dev).contiguous(), Rw.to(dev).contiguous(), Rt.to(dev).contiguous(), rel_gemm=rel)
        _close(out2.as_rows().view(B, Nq, heads * hd), ref, 2e-5, "mvit attention, GEMM rel-pos")


def test_attention_large_logits(dev):
    """Forces the online-softmax rescale: one key tile late in the sequence dominates."""
    from mspi_amd import engine as E
    B, H, N, D = 1, 1, 160, 32
    g = torch.Generator().manual_seed(9)
    qkv = torch.randn(B, N, 3 * D, generator=g)
    qkv[0, 150, D:2 * D] *= 30.0   # a key with huge norm
    q, k, v = qkv.view(B, N, 3, H, D).permute(2, 0, 3, 1, 4).double()
    ref = ((q @ k.transpose(-2, -1)) * D ** -0.5).softmax(-1) @ v
    xc = E.CL(qkv.to(dev).view(-1), 0, B, N, 1, 1, 3 * D, 3 * D)
    out = E.attention(xc, B, N, H, D, D ** -0.5)
    _close(out.as_rows().view(B, N, D), ref.transpose(1, 2).reshape(B, N, D).float(), 2e-5, "attention rescale")


@pytest.mark.parametrize("k", [2, 4, 8])
def test_upsample_rowgate(dev, k):
    from mspi_amd import engine as E
    g = torch.Generator().manual_seed(k)
    x = torch.randn(2, 24, 3, 5, 7, generator=g)
    ref = F.interpolate(x, scale_factor=(1, k, k), mode="trilinear", align_corners=False)
    base = torch.randn_like(ref)
    dst = _cl(base, dev)
    E.upsample(_cl(x, dev), k, dst=dst, accumulate=True, act=E.ACT_RELU)
    _close(dst.as_ncdhw(), F.relu(base + ref), 1e-5, "upsample+add+relu")
    out = E.upsample(_cl(x, dev), k)
    _close(out.as_ncdhw(), ref, 1e-5, "upsample")
    mask = torch.rand(2, 1, 3, 5 * k, 7 * k, generator=g)
    E.rowgate(out, _cl(mask, dev))
    _close(out.as_ncdhw(), ref * mask + ref, 1e-5, "rowgate")


def test_small_reductions(dev):
    from mspi_amd import engine as E
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 50176, generator=g) * 4
    t = x.to(dev).clone()
    E.logsumexp_sub(t, 3, 50176)
    _close(t, x - torch.logsumexp(x, 1, keepdim=True), 1e-5, "logsumexp")
    seq = torch.randn(2, 40, 512, generator=g)
    sc = E.CL(seq.to(dev).view(-1), 0, 2, 40, 1, 1, 512, 512)
    out = torch.empty(2, 512, device=dev)
    E.mean_rows(sc.tokens(7, 5, 2, 3), 2, 30, out)
    _close(out, seq[:, 7:37].mean(1), 1e-5, "mean_rows")
    long = torch.randn(3, 3001, 100, generator=g) + 0.5            # two-stage form (R >= 1024), ragged last slice
    out = torch.empty(3, 100, device=dev)
    E.mean_rows(E.CL(long.to(dev).view(-1), 0, 3, 3001, 1, 1, 100, 100), 3, 3001, out)
    _close(out, long.mean(1), 1e-6, "mean_rows (two-stage)")
    p, z = torch.randn(4, 2048, generator=g), torch.randn(4, 2048, generator=g)
    loss = torch.zeros(1, device=dev)
    E.neg_cosine(E.from_rows(p.to(dev)), E.from_rows(z.to(dev)), loss, 0.5, False)
    E.neg_cosine(E.from_rows(z.to(dev)), E.from_rows(p.to(dev)), loss, 0.5, True)
    _close(loss, -F.cosine_similarity(p, z, dim=-1).mean().view(1), 1e-5, "neg cosine")
    a, b = torch.randn(1003, generator=g), torch.randn(1003, generator=g)
    y = torch.empty(1003, device=dev)
    E.add(a.to(dev), b.to(dev), y)
    _close(y, a + b, 0, "add")


def test_errors_are_loud(dev):
    from mspi_amd import engine as E
    from mspi_amd._lib import MspiError
    x = E.alloc(1, 1, 4, 4, 8, dev)
    pk = E.pack_conv(torch.randn(8, 12, 1, 1, 1), device=dev)
    with pytest.raises(MspiError):
        E.conv(x, pk)                       # channel mismatch
    with pytest.raises(MspiError):
        E.conv(torch.zeros(1, 12, 1, 4, 4), pk)   # CPU tensor: no fallback


@pytest.mark.parametrize("C,M,ln,res", [(96, 1000, True, True), (96, 128, False, True), (96, 33, True, False),
                                        (192, 777, True, True), (192, 256, False, False)])
def test_mlp_fused(dev, C, M, ln, res):
    """mspi_mlp_fwd: y = res + fc2(GELU(fc1(LN(x)))) in one launch (ConvNeXt / Swin / MViT MLP tail), ragged M,
    layer-scale folded into fc2; reference in float64 on the CPU."""
    from mspi_amd import engine as E
    if not E.mlp_supported(C, 4 * C):
        pytest.skip("fused MLP is an f16x3 kernel")
    g = torch.Generator().manual_seed(C + M)
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    r = torch.randn(M, C, generator=g)
    w1, b1 = torch.randn(4 * C, C, generator=g) * 0.1, torch.randn(4 * C, generator=g) * 0.1
    w2, b2 = torch.randn(C, 4 * C, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1
    gm, bt, ls = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g)
    xd = F.layer_norm(x.double(), (C,), gm.double(), bt.double(), 1e-6) if ln else x.double()
    ref = (F.gelu(xd @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()) * ls.double()
    if res:
        ref = ref + r.double()
    pk = E.pack_mlp(w1, b1, w2, b2, out_scale=ls, device=dev)
    xc = E.CL(x.to(dev).view(-1), 0, 1, M, 1, 1, C, C)
    rc = E.CL(r.to(dev).view(-1), 0, 1, M, 1, 1, C, C) if res else None
    out = E.mlp(xc, pk, res=rc, ln=(gm.to(dev), bt.to(dev)) if ln else None, eps=1e-6)
    _close(out.as_rows(), ref.float(), 2e-5, "fused mlp")
    # same thing through the unfused kernels (LN, fc1+GELU, fc2+res): the two paths agree to fp32 rounding
    y = E.layernorm(xc, gm.to(dev), bt.to(dev), 1e-6) if ln else xc
    fc1 = E.pack_conv(w1, b1, act=E.ACT_GELU, device=dev)
    fc2 = E.pack_conv(w2, b2, out_scale=ls, device=dev)
    out2 = E.conv(E.conv(y, fc1), fc2, res=rc)
    _close(out.as_rows(), out2.as_rows().cpu(), 2e-5, "fused vs unfused")


@pytest.mark.parametrize("cin,cout,M,res,gate,act", [
    (24, 54, 1000, False, False, 1), (54, 24, 333, True, True, 1), (96, 216, 777, False, False, 1),
    (216, 96, 515, True, True, 0), (108, 48, 128, True, False, 1), (192, 432, 260, False, False, 0),
    (96, 216, 70000, False, False, 1),
    (48, 108, 31, False, False, 4)])
def test_rowgemm_thin(dev, cin, cout, M, res, gate, act):
    """mspi_rowgemm_fwd (kernel choice THIN of engine.conv): X3D's 1x1x1 layers incl. the SE-gate + Swish prologue,
    the residual epilogue, channel counts that are not multiples of 4 (54 -> 56 stored) and ragged M."""
    from mspi_amd import engine as E
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("the thin GEMM is an f16x3 kernel")
    g = torch.Generator().manual_seed(cin * 1000 + cout)
    N, rows = 3, (M + 2) // 3
    x = torch.randn(N, cin, 1, rows, 1, generator=g)
    w = torch.randn(cout, cin, 1, 1, 1, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    r = torch.randn(N, cout, 1, rows, 1, generator=g) if res else None
    gt = torch.rand(N, cin, generator=g) * 2 if gate else None
    xin = x.double()
    if gate:
        xin = F.silu(xin * gt.double()[:, :, None, None, None])
    ref = F.conv3d(xin, w.double(), b.double())
    if res:
        ref = ref + r.double()
    ref = {0: ref, 1: F.relu(ref), 4: F.silu(ref)}[act]
    from mspi_amd.module import to_cl
    xc = to_cl(x.to(dev))
    pk = E.pack_conv(w, b, act=act, cin_stored=xc.Cs, device=dev)
    assert pk.thin is not None
    rc = to_cl(r.to(dev)) if res else None
    gd = None
    if gate:
        gd = torch.zeros(N, xc.Cs, device=dev)
        gd[:, :cin] = gt.to(dev)
    out = E.conv(xc, pk, res=rc, gate=gd, tile=E.THIN)
    _close(out.as_ncdhw(), ref.float(), 2e-5, "thin gemm")
    out2 = E.conv(xc, pk, res=rc, gate=gd, tile=3)
    _close(out.as_ncdhw(), out2.as_ncdhw().cpu(), 2e-5, "thin vs tiled")
    if out.Cs > cout:   # pad channels stay exact zeros
        assert (out.as_rows()[:, cout:] == 0).all()


@pytest.mark.parametrize("W", [32, 36])
def test_narrow_stem_widened_conv(dev, W):
    """ResNetBasicStem with 8 output channels (SlowFast fast pathway): the (5,7,13)/(1,2,8) x 32-channel form used when
    W % 8 == 0 and the plain (5,7,7)/(1,2,2) form otherwise, both against torch (conv + eval BN + ReLU + max-pool)."""
    from mspi_amd.backbones.blocks3d import ResNetBasicStem
    from mspi_amd import testing as T
    stem = T.seeded(lambda: ResNetBasicStem(3, 8, [5, 7, 7], [1, 2, 2], [2, 3, 3]), 1)
    T.randomize_(stem, 2)
    stem.eval()
    x = torch.randn(2, 3, 6, 24, W, generator=torch.Generator().manual_seed(W))
    with torch.no_grad():
        ref = stem.pool_layer(F.relu(stem.bn(stem.conv(x))))
    stem = stem.to(dev)
    out = stem.run(x.to(dev))
    assert stem.pk["wide"] is not None      # packed either way; run() uses it only when W % 8 == 0
    _close(out.as_ncdhw(), ref, 2e-5, "narrow stem W=%d" % W)


@pytest.mark.parametrize("S", [2, 4, 8])
@pytest.mark.parametrize("shape", [
    (2, 512, 4, 7, 7, 96, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # SA / smoothing convs on small maps: long K, few tiles
    (1, 2048, 1, 5, 9, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0)),    # late 1x1x1 layers
    (3, 40, 2, 6, 5, 24, (1, 3, 3), (1, 2, 2), (0, 1, 1)),       # K steps barely cover the slices; ragged M and N
])
def test_conv_split_k(dev, S, shape):
    """mspi_conv_splitk_fwd (kernel choice SPLITK + S): K slices -> scratch partials -> ordered reduction with bias,
    residual and ReLU; same result as the single-pass kernel to fp32 rounding, and bitwise repeatable."""
    from mspi_amd import engine as E
    N, Cin, T, H, W, Cout, k, s, p = shape
    g = torch.Generator().manual_seed(Cin + S)
    x = torch.randn(N, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g)
    ref = F.conv3d(x.double(), w.double(), b.double(), s, p)
    res = torch.randn(ref.shape, generator=g)
    ref = F.relu(ref + res.double()).float()
    pk = E.pack_conv(w, b, None, s, p, E.ACT_RELU, cin_stored=E.rup4(Cin), device=dev, prec=E.PREC_F16X3)
    if pk.ldw // 32 < S:
        pytest.skip("fewer K steps than slices")
    out = E.conv(_cl(x, dev), pk, res=_cl(res, dev), tile=E.SPLITK + S)
    _close(out.as_ncdhw(Cout), ref, 2e-5, "split-K %d %s" % (S, shape))
    out2 = E.conv(_cl(x, dev), pk, res=_cl(res, dev), tile=E.SPLITK + S)
    assert torch.equal(out.buf, out2.buf)


def test_permute_and_gated_sum(dev):
    """mspi_permute_fwd against as_strided (vector and scalar paths) and mspi_gated_sum_fwd against torch (J = 2, 3)."""
    from mspi_amd import engine as E
    from mspi_amd._lib import MspiError
    from mspi_amd.backbones import MorphMLP as M
    g = torch.Generator().manual_seed(5)
    for (H, W, Cc, sd) in ((28, 14, 56, 14), (14, 14, 392, 28), (7, 7, 98, 49)):
        B, T = 2, 8
        x = torch.randn(B, T, H, W, Cc, generator=g)
        xd = x.to(dev)
        specs = [M.t_gather(B, T, H * W, Cc), M.t_scatter(B, T, H * W, Cc), M.w_gather(B * T, H * W, Cc, sd),
                 M.s2_gather(B * T, H * W, Cc, sd), M.s2_scatter(B * T, H * W, Cc, sd)]
        if H % sd == 0 or sd % H == 0:
            specs += [M.h_gather(B * T, H, W, Cc, sd), M.h_scatter(B * T, H, W, Cc, sd)]
        for dims, strides in specs:
            got = E.permute(xd.reshape(-1), dims, strides).cpu()
            assert torch.equal(got, x.reshape(-1).as_strided(tuple(dims), tuple(strides)).reshape(-1))
    with pytest.raises(MspiError, match="source"):
        E.permute(xd.reshape(-1), (4, xd.numel()), (1, 1))
    N, R, Cc = 3, 50, 56
    srcs = [torch.randn(N, R, Cc, generator=g) for _ in range(3)]
    for J in (2, 3):
        logit = torch.randn(N, Cc * J, generator=g) * 2
        a = logit.reshape(N, Cc, J).permute(2, 0, 1).softmax(0)[:, :, None, :]
        ref = sum(a[j] * srcs[j] for j in range(J))
        cls = [E.CL(s_.to(dev).reshape(-1), 0, N, 1, R, 1, Cc, Cc) for s_ in srcs[:J]]
        out = E.gated_sum(cls, logit.to(dev))
        _close(out.buf.view(N, R, Cc), ref, 1e-6, "gated_sum J=%d" % J)


@pytest.mark.parametrize("M,C,Hd", [(300, 96, 384), (6272, 384, 1536), (777, 64, 256), (4100, 320, 1280)])
def test_presplit_ln_gemm_gemm_chain(dev, M, C, Hd):
    """Pre-split activations: LayerNorm -> f16 hi/lo planes -> fc1 + GELU -> planes -> fc2 + residual, against torch fp32,
    against the fp32-activation kernels (same arithmetic: equal to the last bits), and for every tile of mspi_gemm_sp_fwd."""
    from mspi_amd import engine as E
    from mspi_amd._lib import MspiError
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("pre-split planes are an f16x3 form")
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 2 + 0.3
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    w1, b1 = torch.randn(Hd, C, generator=g) / math.sqrt(C), torch.randn(Hd, generator=g) * 0.1
    w2, b2 = torch.randn(C, Hd, generator=g) / math.sqrt(Hd), torch.randn(C, generator=g) * 0.1
    ref = x + F.linear(F.gelu(F.linear(F.layer_norm(x, (C,), gam, bet, 1e-6), w1, b1)), w2, b2)
    xcl = E.CL(x.to(dev).view(-1), 0, 1, 1, 1, M, C, C)
    p1 = E.pack_conv(w1, b1, act=E.ACT_GELU, device=dev)
    p2 = E.pack_conv(w2, b2, device=dev)
    ln = (gam.to(dev), bet.to(dev))
    plain = E.conv(E.conv(E.layernorm(xcl, *ln, 1e-6), p1), p2, res=xcl)
    _close(plain.buf.view(M, C), ref, 2e-5, "fp32-activation chain")
    for t1 in E.SP_TILES:
        for t2 in (E.SP_TILES if t1 == 6 else (10,)):
            sp = E.layernorm(xcl, *ln, 1e-6, sp=True)
            assert isinstance(sp, E.SP) and sp.buf.dtype == torch.float16
            h = E.conv(sp, p1, sp_out=True, tile=t1)
            y = E.conv(h, p2, res=xcl, tile=t2)
            _close(y.buf.view(M, C), ref, 2e-5, "pre-split chain, tiles %d/%d" % (t1, t2))
            assert (y.buf - plain.buf).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # planes reproduce the fp32 value to 22 bits
    sp = E.layernorm(xcl, *ln, 1e-6, sp=True)
    rec = E.join_planes(sp).as_rows()[:, :C].reshape(-1)      # planes are blocked (16 x 32 tiles): read them back through the ABI
    want = F.layer_norm(x, (C,), gam, bet, 1e-6).reshape(-1)
    mp = (M + 15) // 16 * 16
    hi = sp.buf[: mp * C].view(mp // 16, C // 32, 16, 32).permute(0, 2, 1, 3).reshape(mp, C)[:M].float()
    lo = sp.buf[mp * C:].view(mp // 16, C // 32, 16, 32).permute(0, 2, 1, 3).reshape(mp, C)[:M].float()
    assert torch.equal((hi + lo).reshape(-1), rec), "blocked layout: element (m, k) of a plane at ((m/16)(K/32) + k/32) 512 + (m%16) 32 + k%32"
    assert (rec.cpu() - want).abs().max().item() < 3e-6 * want.abs().max().item() + 1e-6
    with pytest.raises(MspiError, match="split-plane"):
        E.conv(xcl, p1, sp_out=True)
    with pytest.raises(MspiError, match="1x1x1"):
        E.conv(sp, E.pack_conv(torch.randn(8, C, 1, 3, 3, generator=g), None, None, (1, 1, 1), (0, 1, 1), device=dev))


@pytest.mark.parametrize("case", [(1000, 192, 136, (1, 1, 1)), (2 * 6 * 9 * 9, 40, 72, (3, 3, 3))])
def test_dma_gemm_blocked_weights(dev, case):
    """MspiConvDesc.w_blocked: the LDS-DMA kernels (tile codes 6, 7, 9, 14) staging their weights from the blocked copy
    (engine.sp_weights: 16 output channels x 32 k per 1-KB block, rows zero-padded to 16) give bit for bit what they give from
    the row-major planes -- 1x1x1 and 3x3x3, output channels that are not a multiple of 16, ragged M."""
    from mspi_amd import engine as E
    from mspi_amd.module import to_cl
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    M, Cin, Cout, k = case
    g = torch.Generator().manual_seed(M + Cin)
    if k == (1, 1, 1):
        x = torch.randn(1, Cin, 1, M, 1, generator=g)
    else:
        x = torch.randn(2, Cin, 6, 9, 9, generator=g)
    w = torch.randn(Cout, Cin, *k, generator=g) / math.sqrt(Cin * k[0] * k[1] * k[2])
    b = torch.randn(Cout, generator=g)
    ref = F.conv3d(x.double(), w.double(), b.double(), 1, tuple(kk // 2 for kk in k))
    xc = to_cl(x.to(dev))
    pk = E.pack_conv(w, b, None, (1, 1, 1), tuple(kk // 2 for kk in k), cin_stored=xc.Cs, device=dev)
    wb = E.sp_weights(pk)
    npad = (pk.cout_s + 15) // 16 * 16
    assert wb.shape == (2, npad // 16, pk.ldw // 32, 16, 32)
    back = wb.permute(0, 1, 3, 2, 4).reshape(2, npad, pk.ldw)
    assert torch.equal(back[:, : pk.cout_s], pk.w) and (back[:, pk.cout_s:] == 0).all()
    for tile in (6, 7, 9, 14):
        outs = []
        for blocked in (True, False):
            E.W_BLOCKED = blocked
            try:
                outs.append(E.conv(xc, pk, tile=tile).as_ncdhw(Cout).clone())
            finally:
                E.W_BLOCKED = True
        _close(outs[0], ref.float(), 2e-5, "dma gemm, blocked weights, tile %d" % tile)
        assert torch.equal(outs[0], outs[1]), "tile %d: blocked vs row-major weights" % tile


X3D_AB_CASES = [
    # (N, Cin, Cmid, T, H, W): the four stride-1 block shapes of X3D-L (reduced frames) + ragged T / single frame
    (2, 24, 54, 5, 14, 28),      # s2: K = 24 (k32 step 3/4 used), 54 -> 56 stored, 2 chunks (second: 24 of 32 channels)
    (2, 48, 108, 4, 7, 14),      # s3
    (3, 96, 216, 6, 14, 14),     # s4: two spatial tiles per frame, 7 chunks
    (2, 192, 432, 5, 7, 7),      # s5: 7x7 tiles, K = 192
    (1, 96, 216, 1, 7, 14),      # a single frame: both temporal neighbours are padding
    (1, 48, 108, 16, 14, 14),    # long clip: several T segments
]


@pytest.mark.parametrize("case", X3D_AB_CASES)
@pytest.mark.parametrize("se", [False, True])
def test_x3d_ab_fused(dev, case, se):
    """mspi_x3d_ab_fwd: relu(a_bn(a(x))) -> dw3x3x3 + b_bn (-> Swish | -> SE partial sums) in one launch, against the
    plain torch fp32 ops of X3DTransform.forward (SlowFast/resnet_helper.py:296-327), and bit for bit against itself."""
    from mspi_amd import engine as E
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("the fused X3D kernel is an f16x3 kernel")
    N, Cin, Cmid, T, H, W = case
    g = torch.Generator().manual_seed(Cin + Cmid + T)
    x = torch.randn(N, Cin, T, H, W, generator=g)
    wa = torch.randn(Cmid, Cin, 1, 1, 1, generator=g) / math.sqrt(Cin)
    ba = torch.randn(Cmid, generator=g) * 0.3
    wb = torch.randn(Cmid, 1, 3, 3, 3, generator=g) / math.sqrt(27)
    bb = torch.randn(Cmid, generator=g) * 0.3
    t = F.relu(F.conv3d(x.double(), wa.double(), ba.double()))
    ref = F.conv3d(t, wb.double(), bb.double(), 1, 1, 1, Cmid)
    from mspi_amd.module import to_cl
    xc = to_cl(x.to(dev))
    pa = E.pack_conv(wa, ba, act=E.ACT_RELU, cin_stored=xc.Cs, device=dev)
    pb = E.pack_dwconv(wb, bb, None, (1, 1, 1), (1, 1, 1), E.ACT_NONE if se else E.ACT_SWISH, device=dev)
    pk = E.pack_x3d_ab(pa, pb)
    assert pk is not None and E.x3d_ab_supported(xc, pk)
    if se:
        u, part = E.x3d_ab(xc, pk, pool=True)
        _close(u.as_ncdhw(Cmid), ref.float(), 2e-5, "x3d a+b")
        _close(part.sum(1)[:, :Cmid], ref.sum((2, 3, 4)).float(), 2e-5, "x3d a+b: se partial sums")
        u2, part2 = E.x3d_ab(xc, pk, pool=True)
        assert torch.equal(u.buf, u2.buf) and torch.equal(part, part2)
        if u.Cs > Cmid:
            assert (part[:, :, Cmid:] == 0).all()
    else:
        u = E.x3d_ab(xc, pk)
        _close(u.as_ncdhw(Cmid), F.silu(ref).float(), 2e-5, "x3d a+b + swish")
    # the unfused pair of launches computes the same thing
    v = E.dwconv(E.conv(xc, pa), pb)
    _close(u.as_ncdhw(Cmid), v.as_ncdhw(Cmid).cpu(), 2e-5, "fused vs unfused")


@pytest.mark.parametrize("case", [(216, 96, 1000), (108, 48, 777), (54, 24, 420), (216, 96, 128 * 6)])
@pytest.mark.parametrize("se", [False, True])
def test_x3d_ca_seam(dev, case, se):
    """mspi_x3d_ca_fwd: y = relu(c(u') + res), t = relu(a_next(y)) in one launch (u' = swish(u * gate) for SE blocks) against
    torch fp64 (X3DTransform.c + ResBlock add/ReLU + next X3DTransform.a, SlowFast/resnet_helper.py:296-351, :607-616) and
    against the two thin-GEMM launches it stands for; widths that are not multiples of 32, ragged M, pad columns."""
    from mspi_amd import engine as E
    from mspi_amd.module import to_cl
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("the seam kernel is an f16x3 kernel")
    D, Cx, M = case
    g = torch.Generator().manual_seed(D + Cx + M)
    N, rows = 3, (M + 2) // 3
    u = torch.randn(N, D, 1, rows, 1, generator=g)
    r = torch.randn(N, Cx, 1, rows, 1, generator=g)
    wc = torch.randn(Cx, D, 1, 1, 1, generator=g) / math.sqrt(D)
    bc = torch.randn(Cx, generator=g) * 0.3
    wa = torch.randn(D, Cx, 1, 1, 1, generator=g) / math.sqrt(Cx)
    ba = torch.randn(D, generator=g) * 0.3
    gt = torch.rand(N, D, generator=g) * 2 if se else None
    uin = u.double()
    if se:
        uin = F.silu(uin * gt.double()[:, :, None, None, None])
    y_ref = F.relu(F.conv3d(uin, wc.double(), bc.double()) + r.double())
    t_ref = F.relu(F.conv3d(y_ref, wa.double(), ba.double()))
    uc, rc = to_cl(u.to(dev)), to_cl(r.to(dev))
    pc = E.pack_conv(wc, bc, act=E.ACT_RELU, cin_stored=uc.Cs, device=dev)
    pa = E.pack_conv(wa, ba, act=E.ACT_RELU, cin_stored=rc.Cs, device=dev)
    pk = E.pack_x3d_ca(pc, pa)
    assert pk is not None
    assert pk.w.numel() * 2 == E._lib.load().mspi_x3d_ca_packed_bytes(uc.Cs, rc.Cs)
    gd = None
    if se:
        gd = torch.zeros(N, uc.Cs, device=dev)
        gd[:, :D] = gt.to(dev)
    y, t = E.x3d_ca(uc, pk, rc, gate=gd)
    _close(y.as_ncdhw(Cx), y_ref.float(), 2e-5, "seam: y")
    _close(t.as_ncdhw(D), t_ref.float(), 2e-5, "seam: t")
    y2 = E.conv(uc, pc, res=rc, gate=gd, tile=E.THIN)
    t2 = E.conv(y2, pa, tile=E.THIN)
    _close(y.as_ncdhw(Cx), y2.as_ncdhw(Cx).cpu(), 2e-6, "seam vs thin GEMM: y")
    if y.Cs > Cx:   # pad channels stay exact zeros
        assert (y.as_rows()[:, Cx:] == 0).all()
    _close(t.as_ncdhw(D), t2.as_ncdhw(D).cpu(), 2e-5, "seam vs two launches")
    y3, t3 = E.x3d_ca(uc, pk, rc, gate=gd)
    assert torch.equal(y.buf, y3.buf) and torch.equal(t.buf, t3.buf)


def _range_case(dev, scale_rows, M=512, K=192, N=96, seed=5):
    from mspi_amd import engine as E
    from mspi_amd.module import to_cl
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    x = x * torch.as_tensor(scale_rows(M))[:, None]
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double().t() + b.double()
    xc = to_cl(x.t().reshape(1, K, 1, M, 1).to(dev))
    pk = E.pack_conv(w, b, device=dev)
    E.autotune(False)          # an earlier test (inference.build_model) may have left tuning -- and with it the range check -- on
    return E, x, ref, xc, pk


def test_f16x3_overflow_is_reported_and_routed(dev):
    """A row of |x| ~ 1e5 is beyond f16 (65504): the f16x3 kernels' hi half is inf.  (a) Untuned, the result is not finite and
    the range guard says so (check_range raises) instead of inf flowing on silently; (b) on first sight while tuning, the
    range check moves the layer to the fp32 MFMA path and the result is fp32-accurate against fp64."""
    from mspi_amd._lib import MspiError
    E, x, ref, xc, pk = _range_case(dev, lambda M: [1e5 if i == 7 else 1.0 for i in range(M)])
    if pk.prec != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    E.range_flag()
    out = E.conv(xc, pk, tile=3).as_rows()
    torch.cuda.synchronize()
    assert not torch.isfinite(out[7]).all() and torch.isfinite(out[8]).all()
    with pytest.raises(MspiError, match="f16x3 range"):
        E.check_range()
    assert not E.range_flag()                                # the check cleared it
    E.autotune(True)
    try:
        out = E.conv(xc, pk).as_rows().cpu().double()
    finally:
        E.autotune(False)
    assert pk.prec == E.PREC_F32 and E.RANGE_CHECK["moved"][-1][1] > 6e4
    E.check_range()
    assert ((out - ref).abs().max(1).values / ref.abs().max(1).values).max().item() < 2e-6


def test_f16x3_tiny_tensor_is_routed(dev):
    """Every row ~ 1e-6: the lo halves are f16 subnormals (absolute error 2^-25 per element), i.e. percent-level relative error
    untuned; the range check (max|x| < 2^-5) moves the layer to fp32 and the result is fp32-accurate."""
    E, x, ref, xc, pk = _range_case(dev, lambda M: [1e-6] * M)
    if pk.prec != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    untuned = E.conv(xc, pk, tile=3).as_rows().cpu().double()
    err_untuned = (untuned - ref).abs().max().item()
    E.autotune(True)
    try:
        out = E.conv(xc, pk).as_rows().cpu().double()
    finally:
        E.autotune(False)
    assert pk.prec == E.PREC_F32
    err = (out - ref).abs().max().item()
    assert err < 1e-6 * ref.abs().max().item()               # fp32 rounding of the bias-dominated output
    assert err_untuned < 2.0 ** -23 * 192 ** 0.5             # documented bound: 2^-25 per element x |w| ~ K^-1/2, K terms


def test_f16x3_small_row_beside_normal_rows(dev):
    """One row ~ 1e-6 among unit-scale rows: the tensor is in range, the layer stays f16x3, and the small row's ABSOLUTE error
    is bounded by 2^-25 * sum|w| (negligible against every other row) -- its relative accuracy is what f16x3 gives up."""
    E, x, ref, xc, pk = _range_case(dev, lambda M: [1e-6 if i == 3 else 1.0 for i in range(M)])
    if pk.prec != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    E.autotune(True)
    try:
        out = E.conv(xc, pk).as_rows().cpu().double()
    finally:
        E.autotune(False)
    assert pk.prec == E.PREC_F16X3
    err = (out - ref).abs()
    assert err[3].max().item() < 2.0 ** -25 * 192 ** 0.5 * 4 + 1e-7     # lo-half subnormal error + fp32 rounding of bias + sum
    others = torch.cat([err[:3], err[4:]])
    assert (others.max(1).values / ref.abs().max(1).values[torch.arange(512) != 3]).max().item() < 2e-6


@pytest.mark.parametrize("B,H,N,D,win", [(2, 2, 100, 64, False), (3, 4, 392, 32, True), (1, 1, 210, 96, False)])
def test_attention_planes_bit_identical(dev, B, H, N, D, win):
    """mspi_attn_fwd_ws (K / V split once per head into workspace planes, staged by plain copies) == mspi_attn_fwd (every
    query tile splits its own copy), bit for bit -- ragged key counts, several heads, a windowed token index.  (Shapes the
    key split leaves alone: fewer than 24 key tiles, or a token index.)"""
    from mspi_amd import engine as E
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    g = torch.Generator().manual_seed(N + D)
    nwin = 2 if win else 1
    qkv = torch.randn(B * nwin * N, 3 * H * D, generator=g)
    x = E.CL(qkv.to(dev).view(-1), 0, B, 1, 1, nwin * N, 3 * H * D, 3 * H * D)
    tok = None
    if win:
        tok = torch.randperm(nwin * N, generator=g).view(nwin, N).to(torch.int32).to(dev)     # the windows partition the tokens
    outs = []
    for planes in (True, False):
        E.ATTN_PLANES = planes
        try:
            outs.append(E.attention(x, B * nwin, N, H, D, D ** -0.5, tok_idx=tok).buf.clone())
        finally:
            E.ATTN_PLANES = True
    assert torch.equal(outs[0], outs[1])



@pytest.mark.parametrize("B,H,N,D", [(1, 1, 900, 96), (2, 2, 1000, 64), (1, 4, 790, 128)])
def test_attention_key_split(dev, B, H, N, D):
    """Few-query shapes: mspi_attn_fwd_ws hands slices of the key tiles to gridDim.z workgroups and merges their partial
    (O, max, sum) in fixed order.  Against torch fp64 softmax attention, against the unsplit kernel, and run-to-run bit
    equality; ragged last slice and last tile (790 keys = 25 tiles over 2 slices of 13 and 12)."""
    from mspi_amd import engine as E
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    lib = E._lib.load()
    d = E.AttnDesc()
    d.B, d.Hh, d.Nq, d.Nk, d.D, d.Dv, d.prec = B, H, N, N, D, D, E.PREC_F16X3
    planes = B * H * 2 * ((N + 31) // 32 * 32) * 2 * D * 2
    assert lib.mspi_attn_ws_bytes(C.byref(d)) > planes + 15, "this shape is meant to engage the key split"
    g = torch.Generator().manual_seed(N + D)
    qkv = torch.randn(B * N, 3 * H * D, generator=g)
    x = E.CL(qkv.to(dev).view(-1), 0, B, 1, 1, N, 3 * H * D, 3 * H * D)
    q, k, v = (qkv.double().view(B, N, 3, H, D).permute(2, 0, 3, 1, 4)[i] for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(B * N, H * D)
    out = E.attention(x, B, N, H, D, D ** -0.5)
    got = out.as_rows()[:, : H * D].cpu()
    _close(got, ref.float(), 2e-5, "key-split attention")
    out2 = E.attention(x, B, N, H, D, D ** -0.5)
    assert torch.equal(out.buf, out2.buf)
    E.ATTN_PLANES = False
    try:
        plain = E.attention(x, B, N, H, D, D ** -0.5).as_rows()[:, : H * D].cpu()
    finally:
        E.ATTN_PLANES = True
    _close(got, plain, 2e-6, "key-split vs one pass")


def test_range_check_moves_plane_consumer_in_mid_forward(dev):
    """mlp_tail's split path: LN -> planes -> fc1 + GELU -> planes -> fc2.  fc2 is first seen AFTER fc1 has emitted planes; when
    its input is beyond the f16x3 window (|h| >= 2^15) the range check moves it to fp32 in the middle of the forward: the call
    completes (planes rebuilt as rows for this one forward) and is fp32-accurate; the next forward hands fc2 rows."""
    from mspi_amd import engine as E
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    g = torch.Generator().manual_seed(5)
    M, C, Hd = 700, 256, 512
    x = torch.randn(M, C, generator=g)
    gam, bet = torch.ones(C), torch.zeros(C)
    w1, b1 = torch.randn(Hd, C, generator=g) * (1.2e4 / math.sqrt(C)), torch.zeros(Hd)       # fc1 outputs up to ~5e4 (< 65504)
    w2, b2 = torch.randn(C, Hd, generator=g) / (math.sqrt(Hd) * 1e4), torch.randn(C, generator=g) * 0.1
    ln = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-6)
    ref = x.double() + F.linear(F.gelu(F.linear(ln, w1.double(), b1.double())), w2.double(), b2.double())
    xcl = E.CL(x.to(dev).view(-1), 0, 1, 1, 1, M, C, C)
    packed = ("split", E.pack_conv(w1, b1, act=E.ACT_GELU, device=dev), E.pack_conv(w2, b2, device=dev))
    lnp = (gam.to(dev), bet.to(dev))
    moved = len(E.RANGE_CHECK["moved"])
    E.range_flag()
    E.autotune(True)
    try:
        first = E.mlp_tail(xcl, packed, lnp, 1e-6, xcl).buf.view(M, C).cpu().double()
        second = E.mlp_tail(xcl, packed, lnp, 1e-6, xcl).buf.view(M, C).cpu().double()
    finally:
        E.autotune(False)
    assert packed[1].prec == E.PREC_F16X3 and packed[2].prec == E.PREC_F32 and len(E.RANGE_CHECK["moved"]) == moved + 1
    E.check_range()
    for y in (first, second):
        assert (y - ref).abs().max().item() < 3e-6 * ref.abs().max().item()


def test_fused_mlp_first_sight_range_check(dev):
    """The fused LN -> fc1 -> GELU -> fc2 kernel (mspi_mlp_fwd) has no fp32 form: on first sight while tuning the pair runs
    once as LayerNorm + two GEMMs through conv()'s range check; a hidden activation beyond 2^15 keeps the pair unfused on the
    fp32 path (results fp32-accurate), an in-range pair goes fused from the second call on."""
    from mspi_amd import engine as E
    if not E.mlp_supported(96, 384):
        pytest.skip("fused MLP not available in this configuration")
    g = torch.Generator().manual_seed(6)
    M, C, Hd = 4200, 96, 384
    x = torch.randn(M, C, generator=g)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    xcl = E.CL(x.to(dev).view(-1), 0, 1, 1, 1, M, C, C)
    lnp = (gam.to(dev), bet.to(dev))
    for scale, want_fallback in ((1.0, False), (1.5e4, True)):
        w1, b1 = torch.randn(Hd, C, generator=g) * (scale / math.sqrt(C)), torch.randn(Hd, generator=g) * 0.1
        w2, b2 = torch.randn(C, Hd, generator=g) / (math.sqrt(Hd) * scale), torch.randn(C, generator=g) * 0.1
        ln = F.layer_norm(x.double(), (C,), gam.double(), bet.double(), 1e-6)
        ref = x.double() + F.linear(F.gelu(F.linear(ln, w1.double(), b1.double())), w2.double(), b2.double())
        pm = E.pack_mlp(w1, b1, w2, b2, device=dev)
        E.range_flag()
        E.autotune(True)
        try:
            with E.Profiler() as prof:
                y1 = E.mlp(xcl, pm, res=xcl, ln=lnp, eps=1e-6).buf.view(M, C).cpu().double()
                y2 = E.mlp(xcl, pm, res=xcl, ln=lnp, eps=1e-6).buf.view(M, C).cpu().double()
        finally:
            E.autotune(False)
        names = [r[0] for r in prof.records]
        assert (pm.fallback is not None) == want_fallback
        assert ("mlp_fused" in names) == (not want_fallback)
        E.check_range()
        for y in (y1, y2):
            assert (y - ref).abs().max().item() < 2e-5 * ref.abs().max().item()


def test_attention_first_sight_moves_a_shape_to_fp32(dev):
    """f16x3 attention scales q by 64 before its split: |q * scale| ~ 2e3 is inf in the hi half.  First sight while tuning moves
    that SHAPE to the fp32 MFMA kernel; the result matches fp64."""
    from mspi_amd import engine as E
    if E.DEFAULT_PREC != E.PREC_F16X3:
        pytest.skip("f16x3 only")
    g = torch.Generator().manual_seed(7)
    B, Ntok, heads, hd = 2, 77, 2, 32
    qkv = torch.randn(B * Ntok, 3 * heads * hd, generator=g)
    qkv[:, : heads * hd] *= 4e3                       # q
    qkv[:, heads * hd: 2 * heads * hd] *= 1e-3        # k: keeps the logits moderate
    scale = 0.5
    q, k, v = [t.reshape(B, Ntok, heads, hd).permute(0, 2, 1, 3).double() for t in qkv.chunk(3, 1)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).permute(0, 2, 1, 3).reshape(B * Ntok, heads * hd)
    cl = E.CL(qkv.to(dev).view(-1), 0, B, 1, 1, Ntok, 3 * heads * hd, 3 * heads * hd)
    key = ("qkv", heads, hd, Ntok, 0)
    E.ATTN_PREC.pop(key, None)
    E.autotune(True)
    try:
        out = E.attention(cl, B, Ntok, heads, hd, scale).buf.view(B * Ntok, heads * hd).cpu().double()
    finally:
        E.autotune(False)
    assert E.ATTN_PREC[key] == E.PREC_F32
    E.ATTN_PREC.pop(key, None)
    assert (out - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
