#!/usr/bin/env python3
"""Depthwise-conv kernel timing on the shapes that carry the traffic (run once per MSPI_DW_LDS / MSPI_DW_STRIP setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E

dev = torch.device("cuda")
SHAPES = [  # N, T, H, W, C, k, pad, pool
    (8, 16, 14, 14, 216, (3, 3, 3), (1, 1, 1), False), (8, 16, 14, 14, 216, (3, 3, 3), (1, 1, 1), True),
    (8, 16, 7, 7, 432, (3, 3, 3), (1, 1, 1), False), (8, 16, 28, 28, 108, (3, 3, 3), (1, 1, 1), False),
    (8, 16, 56, 56, 56, (3, 3, 3), (1, 1, 1), False),
    (128, 1, 14, 14, 384, (1, 7, 7), (0, 3, 3), False), (128, 1, 28, 28, 192, (1, 7, 7), (0, 3, 3), False),
    (128, 1, 56, 56, 96, (1, 7, 7), (0, 3, 3), False), (128, 1, 7, 7, 768, (1, 7, 7), (0, 3, 3), False),
    (8, 8, 56, 56, 64, (5, 5, 5), (2, 2, 2), False), (8, 8, 28, 28, 128, (5, 5, 5), (2, 2, 2), False),
    (8, 8, 14, 14, 320, (3, 3, 3), (1, 1, 1), False), (8, 4, 56, 56, 192, (7, 1, 1), (3, 0, 0), False),
]
g = torch.Generator().manual_seed(0)
for N, T, H, W, C, k, pad, pool in SHAPES:
    x = E.alloc(N, T, H, W, C, dev)
    x.buf.normal_()
    w = torch.randn(C, 1, *k, generator=g)
    pk = E.pack_dwconv(w, torch.randn(C, generator=g), None, (1, 1, 1), pad, E.ACT_NONE, device=dev)
    for _ in range(3):
        E.dwconv(x, pk, pool=pool)
    # 20 launches inside ONE hipGraph (eager Python launches can be further apart than these kernels are long)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for _ in range(20):
            E.dwconv(x, pk, pool=pool)
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    gb = 8.0 * x.M * C / 1e9
    print("%-44s %7.1f us  %6.0f GB/s" % ("%s k=%s%s" % ((N, T, H, W, C), k, " pool" if pool else ""), us, gb / us * 1e6))
