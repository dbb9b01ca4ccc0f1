#!/usr/bin/env python3
"""Where a K step of the A-in-registers GEMM (mspi_gemm_sp_fwd tile codes 15 / 17 / 18) spends its time: 100 MHz stamps of one trip
(NST steps) per workgroup -- wait for the step's data, barrier, issue of the step NST - 1 ahead, compute."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, _lib
dev = torch.device("cuda")
lib = _lib.load()
g = torch.Generator().manual_seed(0)
for M, K, N in [(25088, 1536, 384), (25088, 384, 1536), (6272, 3072, 768)]:
    x = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    pk = E.pack_conv(w, torch.zeros(N), device=dev)
    planes = torch.empty(2 * M * K, dtype=torch.float16, device=dev)
    _lib.check(lib.mspi_split_planes_fwd(x.data_ptr(), K, M, K, planes.data_ptr(), K, M * K, torch.cuda.current_stream().cuda_stream), "split")
    d = _lib.ConvDesc()
    d.N, d.T, d.H, d.W, d.C = 1, 1, 1, M, K
    d.kT = d.kH = d.kW = d.strT = d.strH = d.strW = 1
    d.To, d.Ho, d.Wo, d.Cout = 1, 1, M, N
    d.ldy, d.ldw, d.ldr, d.act, d.prec, d.w_scale = N, pk.ldw, 0, 0, pk.prec, pk.w_scale
    y = torch.empty(M, N, device=dev)
    for t, nst, bn in ((15, 6, 64), (17, 4, 128), (18, 3, 192)):
        nwg = ((M + 127) // 128) * ((N + bn - 1) // bn)
        st = torch.zeros(nwg * 64, dtype=torch.int64, device=dev)
        d.tile = t
        run = lambda: _lib.check(lib.mspi_gemm_sp_fwd(C.byref(d), planes.data_ptr(), K, M * K, pk.w.data_ptr(), pk.bias.data_ptr(), None,
                                                      y.data_ptr(), None, 0, 0, torch.cuda.current_stream().cuda_stream), "gemm_sp")
        for _ in range(3):
            run()
        lib.mspi_debug_stamps(st.data_ptr())
        run()
        torch.cuda.synchronize()
        lib.mspi_debug_stamps(None)
        s = st.view(nwg, 16, 4)[:, :nst].double() * 10.0        # ns
        wait = (s[:, :, 1] - s[:, :, 0]).mean().item()
        bar = (s[:, :, 2] - s[:, :, 1]).mean().item()
        iss = (s[:, :, 3] - s[:, :, 2]).mean().item()
        step = ((s[:, 1:, 0] - s[:, :-1, 0]).mean().item())
        print("M=%d K=%d N=%d tile %d (128x%d, ring %d): per step %.0f ns = wait %.0f + barrier %.0f + issue %.0f + compute %.0f" % (
            M, K, N, t, bn, nst, step, wait, bar, iss, step - wait - bar - iss), flush=True)
