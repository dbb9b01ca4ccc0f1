#!/usr/bin/env python3
"""Pre-split activation GEMM (mspi_gemm_sp_fwd) against the fp32-activation kernels on the big dense layers."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, _lib

dev = torch.device("cuda")
lib = _lib.load()
import os as _os
SHAPES = [(25088, 384, 1536), (25088, 1536, 384), (6272, 768, 3072), (6272, 3072, 768), (100352, 768, 192), (100352, 192, 768),
          (6992, 512, 2048), (6992, 2048, 512), (12544, 320, 1280), (12544, 1280, 320)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if _os.environ.get("SP_PROBE_SHAPES"):
    SHAPES = SHAPES[: int(_os.environ["SP_PROBE_SHAPES"])]
g = torch.Generator().manual_seed(0)
for M, K, N in SHAPES:
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    pk = E.pack_conv(w, b, act=E.ACT_GELU, device=dev)
    xcl = E.CL(x.view(-1), 0, 1, 1, 1, M, K, K)
    best = (1e9, None)
    for t in [1, 2, 4, 6, 7, 9, 10, 12, 13, 14]:
        try:
            us = timeit(lambda: E.conv(xcl, pk, tile=t))
        except Exception:
            continue
        if us < best[0]:
            best = (us, t)
    ref = E.conv(xcl, pk, tile=best[1]).buf.view(M, N).clone()
    planes = torch.empty(2 * M * K, dtype=torch.float16, device=dev)
    _lib.check(lib.mspi_split_planes_fwd(x.data_ptr(), K, M, K, planes.data_ptr(), K, M * K, torch.cuda.current_stream().cuda_stream), "split")
    y = torch.empty(M, N, device=dev)
    d = _lib.ConvDesc()
    d.N, d.T, d.H, d.W, d.C = 1, 1, 1, M, K
    d.kT = d.kH = d.kW = d.strT = d.strH = d.strW = 1
    d.To, d.Ho, d.Wo, d.Cout = 1, 1, M, N
    d.ldy, d.ldw, d.ldr, d.act, d.prec, d.w_scale = N, pk.ldw, 0, E.ACT_GELU, pk.prec, pk.w_scale
    res = {}
    for t in [6, 7, 9, 10, 11, 12, 13, 14]:
        d.tile = t
        def run():
            _lib.check(lib.mspi_gemm_sp_fwd(C.byref(d), planes.data_ptr(), K, M * K, E.sp_weights(pk).data_ptr(), pk.bias.data_ptr(), None,
                                            y.data_ptr(), None, 0, 0, torch.cuda.current_stream().cuda_stream), "gemm_sp")
        try:
            res[t] = timeit(run)
        except Exception as e:
            res[t] = None
    tb = min((v, k) for k, v in res.items() if v)
    print("   per tile (us): " + "  ".join("%d:%s" % (k, "%.0f" % v if v else "-") for k, v in sorted(res.items())))
    d.tile = tb[1]
    run()
    err = (y - ref).abs().max().item()
    fl = 2.0 * M * K * N
    print("M=%6d K=%4d N=%4d  fp32-A best %.1f us (tile %d, %.0f TF/s) | pre-split best %.1f us (tile %d, %.0f TF/s)  x%.2f  max|diff| %.1e" % (
        M, K, N, best[0], best[1], fl / best[0] / 1e6, tb[0], tb[1], fl / tb[0] / 1e6, best[0] / tb[0], err))
