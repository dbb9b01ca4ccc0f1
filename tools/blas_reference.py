#!/usr/bin/env python3
"""What the vendor BLAS behind torch.matmul reaches on the big dense layers of this workload (fp16 / bf16 / fp32), as a
yardstick for the f16x3 kernels: an f16x3 GEMM issues 3 f16 MFMA flops per algorithmic flop, so `3 x its algorithmic
TFLOP/s` is the number to hold against the library's fp16 rate.  Measurement only: nothing in mspi_amd calls a BLAS."""
import torch

dev = torch.device("cuda")
SHAPES = [(25088, 384, 1536), (25088, 1536, 384), (6272, 768, 3072), (6272, 3072, 768), (100352, 768, 192), (6992, 512, 2048),
          (12544, 1280, 320), (100352, 5184, 192)]


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


for M, K, N in SHAPES:
    line = "M=%6d K=%4d N=%4d " % (M, K, N)
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = torch.randn(K, N, device=dev, dtype=dt)
        us = timeit(lambda: torch.matmul(a, b))
        line += " %s %7.1f us %6.0f TF/s |" % (str(dt).split(".")[1], us, 2.0 * M * K * N / us / 1e6)
    print(line)
