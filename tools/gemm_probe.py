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
    "cnx4_fc1": (128, 1, 7, 7, 768, 3072, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "cnx4_fc2": (128, 1, 7, 7, 3072, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "sync_fc1": (8, 874, 1, 1, 512, 2048, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
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
        tiles = [int(t) for t in os.environ["MSPI_PROBE_TILES"].split(",")] if "MSPI_PROBE_TILES" in os.environ else [None]
        for prec, tile in [(E.PREC_F16X3, t) for t in tiles] + ([] if tiles != [None] else [(E.PREC_F32, None)]):
            pk = E.pack_conv(w, None, None, s, p, device=dev, prec=prec)
            try:
                out = E.conv(x, pk, tile=tile)
            except Exception as e:
                print("%-12s tile %s: %s" % (n, tile, str(e)[:60]))
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                E.conv(x, pk, out=out, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps
            fl = 2.0 * out.M * Ci * k[0] * k[1] * k[2] * Co
            print("%-12s %-6s tile %-4s %9.1f us  %7.1f TFLOP/s" % (n, "f16x3" if prec else "f32", tile, us, fl / us / 1e6), flush=True)


if __name__ == "__main__":
    main()
