#!/usr/bin/env python3
"""mspi_gemm_sp_fwd tile codes 15 / 17 / 18 against tile 7 on a few small shapes: error pattern by row / column."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, _lib
dev = torch.device("cuda")
lib = _lib.load()
g = torch.Generator().manual_seed(0)
for M, K, N in [(300, 96, 384), (512, 256, 128), (1000, 384, 256), (4100, 1280, 320)]:
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    pk = E.pack_conv(w, b, device=dev)
    planes = torch.empty(2 * M * K, dtype=torch.float16, device=dev)
    _lib.check(lib.mspi_split_planes_fwd(x.data_ptr(), K, M, K, planes.data_ptr(), K, M * K, torch.cuda.current_stream().cuda_stream), "split")
    d = _lib.ConvDesc()
    d.N, d.T, d.H, d.W, d.C = 1, 1, 1, M, K
    d.kT = d.kH = d.kW = d.strT = d.strH = d.strW = 1
    d.To, d.Ho, d.Wo, d.Cout = 1, 1, M, N
    d.ldy, d.ldw, d.ldr, d.act, d.prec, d.w_scale = N, pk.ldw, 0, 0, pk.prec, pk.w_scale
    outs = {}
    for t in (7, 15, 17, 18):
        if not E.sp_tile_supported(t, K):
            continue
        y = torch.zeros(M, N, device=dev)
        d.tile = t
        _lib.check(lib.mspi_gemm_sp_fwd(C.byref(d), planes.data_ptr(), K, M * K, pk.w.data_ptr(), pk.bias.data_ptr(), None,
                                        y.data_ptr(), None, 0, 0, torch.cuda.current_stream().cuda_stream), "gemm_sp")
        torch.cuda.synchronize()
        outs[t] = y
    for t, y in outs.items():
        if t == 7:
            continue
        e = (y - outs[7]).abs()
        print("M=%d K=%d N=%d tile %d: max err %.3e; bad rows (mod 128) %s; bad cols %s" % (
            M, K, N, t, e.max().item(), sorted(set((torch.nonzero(e.max(1).values > 1e-4).flatten() % 128).tolist()))[:12],
            sorted(set((torch.nonzero(e.max(0).values > 1e-4).flatten()).tolist()))[:12]), flush=True)
