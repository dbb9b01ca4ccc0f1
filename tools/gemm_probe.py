#!/usr/bin/env python3
"""Micro-benchmark of mspi_conv_fwd on a few shapes of the real models (for rocprofv3 --pmc runs and A/B timing)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mspi_amd import engine as E

SHAPES = {   # name: (N,T,H,W,Cin, Cout, k, stride, pad)
    "cnx3_fc1": (128, 1, 14, 14, 384, 1536, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "cnx3_fc2": (128, 1, 14, 14, 1536, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "cnx1_fc1": (128, 1, 56, 56, 96, 384, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "cnx1_fc2": (128, 1, 56, 56, 384, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "readout333": (8, 4, 56, 56, 192, 192, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "x3d_a": (8, 16, 28, 28, 48, 108, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "sa333": (8, 4, 14, 14, 512, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
}


def main():
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    for n in names:
        N, T, H, W, Ci, Co, k, s, p = SHAPES[n]
        g = torch.Generator().manual_seed(0)
        x = E.alloc(N, T, H, W, Ci, dev)
        x.buf.copy_(torch.randn(x.buf.numel(), generator=g))
        w = torch.randn(Co, Ci, *k, generator=g) / (Ci * k[0] * k[1] * k[2]) ** 0.5
        for prec in (E.PREC_F16X3, E.PREC_F32):
            pk = E.pack_conv(w, None, None, s, p, device=dev, prec=prec)
            out = E.conv(x, pk)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                E.conv(x, pk, out=out)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            fl = 2.0 * out.M * Ci * k[0] * k[1] * k[2] * Co
            print("%-12s %-6s %9.1f us  %7.1f TFLOP/s" % (n, "f16x3" if prec else "f32", us, fl / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
