#!/usr/bin/env python3
"""Phase-by-phase check of mspi_x3d_stage_fwd on ONE block: t (phase A), the u planes (phase B / SE) and y (phase C) read back
from the workspace and compared with the per-layer ops on the same packs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from mspi_amd import engine as E, testing as T, _lib
from mspi_amd.config import cfg
from mspi_amd.backbones.X3D import X3D

dev = torch.device("cuda")
x3d = T.seeded(lambda: X3D(cfg.MODEL.X3D.PATH_CFG), 0).to(dev).eval()
lib = _lib.load()
N, Tt, H, W = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (2, 16, 4, 4))]
for stage, bidx in ((x3d.s4, 1), (x3d.s4, 2), (x3d.s5, 1), (x3d.s5, 2)):
    blk = stage.blocks(0)[bidx].branch2
    bp = blk.pk
    spk = E.pack_x3d_stage([bp])
    Cc, D = spk.C, spk.D
    Ds = E.rup4(D)
    g = torch.Generator().manual_seed(1)
    x = E.alloc(N, Tt, H, W, Cc, dev)
    x.buf.copy_(torch.rand(x.buf.numel(), generator=g).to(dev) * 2)
    M = x.M
    # reference with the per-layer ops
    t_ref = E.conv(x, bp["a"])
    if "se" in bp:
        u_ref, part = E.dwconv(t_ref, bp["b"], pool=True)
        gate = E.se_gate(part, 1.0 / (Tt * H * W), *bp["se"])
        y_ref = E.conv(u_ref, bp["c"], res=x, gate=gate)
        v = u_ref.buf.view(N, -1, Ds) * gate.view(N, 1, Ds)
        sw_ref = (v * torch.sigmoid(v)).view(M, Ds)
    else:
        u_ref = E.dwconv(t_ref, bp["b"])
        y_ref = E.conv(u_ref, bp["c"], res=x)
        sw_ref = u_ref.buf.view(M, Ds)
    d = E._x3d_stage_desc(spk, N, Tt, H, W)
    ws = torch.zeros(lib.mspi_x3d_stage_ws_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    y = E.alloc(N, Tt, H, W, Cc, dev)
    y.buf.zero_()
    rc = lib.mspi_x3d_stage_fwd(C.byref(d), x.ptr, y.ptr, spk.wq.data_ptr(), spk.wf.data_ptr(), ws.data_ptr(), E._stream())
    torch.cuda.synchronize()
    al = lambda v: (v + 255) // 256 * 256
    P = 32 if True else 0
    # workspace layout (x3d_stage.hip): sync | tbuf[2] | ubuf | up[2] | xp[2] | pool
    geomP = None
    for Pc in (32, 16, 8, 4, 2, 1):
        pass
    n = Tt and min(H, 32 // Tt)
    TH = (H + n - 1) // n
    Pn = Tt * ((H + TH - 1) // TH)
    off = al((N * Pn + N + 1 + 8 * Pn + 8) * 4)
    tb = ws[off: off + 2 * M * Ds * 4].view(torch.float32).view(2, M, Ds); off += al(2 * M * Ds * 4)
    ub = ws[off: off + M * Ds * 4].view(torch.float32).view(M, Ds); off += al(M * Ds * 4)
    KU = (Ds + 15) // 16 * 16
    up = ws[off: off + 2 * M * KU * 2].view(torch.float16).view(2, M, KU); off += al(2 * M * KU * 2)
    xp = ws[off: off + 2 * M * Cc * 2].view(torch.float16).view(2, M, Cc); off += al(2 * M * Cc * 2)
    sync = ws[: (N * Pn + N + 1) * 4].view(torch.int32)
    print("== %s block %d (%s)  C=%d D=%d  N=%d T=%d %dx%d P=%d rc=%d status=%s  epochs %s pool %s abort %d" % (
        "s4" if stage is x3d.s4 else "s5", bidx, "SE" if "se" in bp else "no SE", Cc, D, N, Tt, H, W, Pn, rc, E.range_flag(),
        sorted(set(sync[: N * Pn].tolist())), sync[N * Pn: N * Pn + N].tolist(), int(sync[-1])))
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    print("   x planes  : %.2e" % rel(xp[0].float() + xp[1].float(), y.buf.view(M, Cc)))
    print("   t (A)     : %.2e   (nonzero frac got %.3f ref %.3f)" % (rel(tb[0], t_ref.buf.view(M, Ds)), (tb[0] != 0).float().mean().item(), (t_ref.buf != 0).float().mean().item()))
    if "se" in bp:
        print("   u (B)     : %.2e" % rel(ub, u_ref.buf.view(M, Ds)))
        g5, r5 = ub.view(N, Tt, H, W, Ds), u_ref.buf.view(N, Tt, H, W, Ds)
        e5 = (g5 - r5).abs()
        print("      err by t:", [round(e5[:, t].max().item(), 3) for t in range(Tt)])
        print("      err by h:", [round(e5[:, :, h].max().item(), 3) for h in range(H)], " by w:", [round(e5[:, :, :, w].max().item(), 3) for w in range(W)])
        print("      err by channel quad (first 12):", [round(e5[..., 4 * q: 4 * q + 4].max().item(), 3) for q in range(12)])
        print("      got[0,3,1,1,:8]", [round(v, 3) for v in g5[0, 3, 1, 1, :8].tolist()])
        print("      ref[0,3,1,1,:8]", [round(v, 3) for v in r5[0, 3, 1, 1, :8].tolist()])
        bias = bp["b"].bias
        print("      bias[:8]       ", [round(v, 3) for v in bias[:8].tolist()])
        # the centre tap alone / sums of taps, from the kernel's own t
        t5 = tb[0].view(N, Tt, H, W, Ds)
        wb = bp["b"].w.view(3, 3, 3, Ds)
        for name, sel in (("centre tap", [(1, 1, 1)]), ("dt=1 plane", [(1, a, b) for a in range(3) for b in range(3)])):
            acc = bias[:8].clone()
            for (dt, kh, kw) in sel:
                tt, hh, ww = 3 + dt - 1, 1 + kh - 1, 1 + kw - 1
                if 0 <= tt < Tt and 0 <= hh < H and 0 <= ww < W:
                    acc = acc + t5[0, tt, hh, ww, :8] * wb[dt, kh, kw, :8]
            print("      %-12s" % name, [round(v, 3) for v in acc.tolist()])
    print("   planes    : %.2e" % rel((up[0].float() + up[1].float())[:, :Ds], sw_ref))
    print("   y (C)     : %.2e" % rel(y.buf.view(M, Cc), y_ref.buf.view(M, Cc)), flush=True)
