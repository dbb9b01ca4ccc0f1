#!/usr/bin/env python3
"""What a plain device copy reaches on this box (bytes read + written per second): the practical ceiling for the HBM-bound
kernels, next to the 8 TB/s spec peak the roofline lines are quoted against.  Measurement only."""
import torch

dev = torch.device("cuda")
for mb in (10, 22, 43, 87, 154, 308, 617, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev)
    b = torch.empty(n, device=dev)
    a.normal_()
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("copy %5d MB -> %5d MB: %8.1f us  %6.0f GB/s (read + write)" % (mb, mb, us, 2.0 * n * 4 / us / 1e3))
