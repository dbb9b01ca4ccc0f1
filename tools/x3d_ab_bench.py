#!/usr/bin/env python3
"""One fused X3D a+b launch per stage shape (batch 8), timed as a hipGraph of 20 dependent launches.
MSPI_X3D_DBG: 1 no GEMM phase, 2 no depthwise phase, 4 no x loads.  MSPI_X3D_TSEG: frames per T segment."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E
from mspi_amd.module import to_cl

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
for (Cin, Cmid, HW) in ((24, 54, 56), (48, 108, 28), (96, 216, 14), (192, 432, 7)):
    x = to_cl(torch.randn(8, Cin, 16, HW, HW, generator=g).to(dev))
    wa = torch.randn(Cmid, Cin, 1, 1, 1, generator=g) / math.sqrt(Cin)
    wb = torch.randn(Cmid, 1, 3, 3, 3, generator=g) / math.sqrt(27)
    pa = E.pack_conv(wa, torch.randn(Cmid, generator=g), act=E.ACT_RELU, cin_stored=x.Cs, device=dev)
    pb = E.pack_dwconv(wb, torch.randn(Cmid, generator=g), None, (1, 1, 1), (1, 1, 1), E.ACT_SWISH, device=dev)
    pk = E.pack_x3d_ab(pa, pb)
    def run():
        for _ in range(20):
            E.x3d_ab(x, pk)
    def run_unfused():
        for _ in range(20):
            E.dwconv(E.conv(x, pa), pb)
    res = []
    for fn in (run, run_unfused):
        fn(); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 100 * 1e3)
    print("Cin %3d Cmid %3d %2dx%2d: fused %6.1f us   unfused a + b %6.1f us   (dbg=%s tseg=%s)" % (
        Cin, Cmid, HW, HW, res[0], res[1], os.environ.get("MSPI_X3D_DBG", "0"), os.environ.get("MSPI_X3D_TSEG", "auto")), flush=True)
