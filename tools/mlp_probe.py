#!/usr/bin/env python3
"""Time the fused MLP kernel against the unfused LN + fc1 + fc2 launches at ConvNeXt-T stage shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, M in ((96, 401408), (192, 100352)):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, C, generator=g).to(dev)
    w1, b1 = torch.randn(4 * C, C, generator=g) * 0.1, torch.randn(4 * C, generator=g) * 0.1
    w2, b2 = torch.randn(C, 4 * C, generator=g) * 0.05, torch.randn(C, generator=g) * 0.1
    gm, bt = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    pk = E.pack_mlp(w1, b1, w2, b2, device=dev)
    fc1, fc2 = E.pack_conv(w1, b1, act=E.ACT_GELU, device=dev), E.pack_conv(w2, b2, device=dev)
    xc = E.CL(x.view(-1), 0, 1, M, 1, 1, C, C)
    out = E.alloc(1, M, 1, 1, C, dev)
    t_f = timeit(lambda: E.mlp(xc, pk, res=xc, ln=(gm, bt), out=out))
    if os.environ.get("MLP_PROBE_FUSED_ONLY"):
        print("C=%d M=%d fused %.1f us" % (C, M, t_f))
        continue
    E.autotune(True)
    E.conv(E.conv(E.layernorm(xc, gm, bt, 1e-6), fc1), fc2, res=xc)
    E.autotune(False)
    t_u = timeit(lambda: E.conv(E.conv(E.layernorm(xc, gm, bt, 1e-6), fc1), fc2, res=xc))
    fl = 4.0 * M * C * 4 * C
    print("C=%d M=%d  fused %.1f us (%.1f TF/s)   unfused LN+fc1+fc2 %.1f us" % (C, M, t_f, fl / t_f / 1e6, t_u))
