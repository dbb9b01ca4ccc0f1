#!/usr/bin/env python3
"""MViTv2-S attention launches only (batch 8), for rocprofv3 --pmc passes: python tools/attn_only.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E
dev = torch.device("cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator().manual_seed(0)
B = 8
for heads, Nq, Nk, DA in ((4, 1568, 392, 128), (8, 392, 1568, 160), (1, 25088, 392, 128), (2, 6272, 1568, 160)):
    hd = 96
    qa = torch.randn(B * heads * Nq * DA, generator=g).to(dev)
    ka = torch.randn(B * heads * Nk * DA, generator=g).to(dev)
    v = E.alloc(B, 1, 1, Nk, heads * hd, dev); v.buf.normal_()
    q = E.alloc(B, 1, 1, Nq, heads * hd, dev); q.buf.normal_()
    out = E.alloc(B, 1, 1, Nq, heads * hd, dev)
    d = E.AttnDesc()
    d.B, d.Hh, d.Nq, d.Nk, d.D, d.Dv, d.nmask = B, heads, Nq, Nk, DA, hd, 0
    d.q_sB, d.q_sH, d.q_sT = heads * Nq * DA, Nq * DA, DA
    d.k_sB, d.k_sH, d.k_sT = heads * Nk * DA, Nk * DA, DA
    d.v_sB, d.v_sH, d.v_sT = Nk * v.ld, hd, v.ld
    d.o_sB, d.o_sH, d.o_sT = Nq * out.ld, hd, out.ld
    d.scale, d.prec = 0.1, E.DEFAULT_PREC
    from mspi_amd import _lib
    for _ in range(reps):
        E._attn_launch(_lib.load(), d, qa.data_ptr(), ka.data_ptr(), v.ptr, q.ptr, None, None, None, out.ptr, dev)
torch.cuda.synchronize()
print("done")
