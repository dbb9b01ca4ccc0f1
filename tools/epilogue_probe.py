#!/usr/bin/env python3
"""What the GELU + plane-split epilogue of the fc1 GEMM costs: same GEMM with / without activation, fp32 rows / planes out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E
dev = torch.device("cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
for M, K, N in ((25088, 384, 1536), (6272, 768, 3072)):
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    xcl = E.CL(x.view(-1), 0, 1, 1, 1, M, K, K)
    sp = E.layernorm(xcl, torch.ones(K, device=dev), torch.zeros(K, device=dev), 1e-6, sp=True)
    for act, nm in ((E.ACT_GELU, "GELU"), (E.ACT_NONE, "none"), (E.ACT_RELU, "ReLU")):
        pk = E.pack_conv(w, b, act=act, device=dev)
        for tile in (7, 6):
            t_pl = timeit(lambda: E.conv(sp, pk, tile=tile, sp_out=True))
            t_f = timeit(lambda: E.conv(sp, pk, tile=tile))
            print("M=%d K=%d N=%d tile %d act %-4s: planes out %.1f us | fp32 rows out %.1f us" % (M, K, N, tile, nm, t_pl, t_f))
