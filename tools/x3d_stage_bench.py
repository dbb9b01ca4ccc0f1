#!/usr/bin/env python3
"""X3D-L stages 4 and 5 at batch 8: the one-launch stage kernel (mspi_x3d_stage_fwd) against the per-layer launches, hipGraph
replay time of blocks 1..n-1 of each stage alone, and the whole backbone both ways (1 and 2 graphs in flight)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.config import cfg
from mspi_amd.backbones.X3D import X3D

dev = torch.device("cuda")
B = int(os.environ.get("XSB_BATCH", "8"))
clips, _ = T.synth_inputs(B, 16, 224, 224, seed=100, device=dev)
x3d = T.seeded(lambda: X3D(cfg.MODEL.X3D.PATH_CFG), 0).to(dev)
E.X3D_STAGE["mode"] = "0"
E.autotune(True); feats = x3d.forward_cl([clips]); E.autotune(False)
torch.cuda.synchronize()


def graph_ms(fn, reps=30, depth=1):
    gs, ss = [], []
    for _ in range(depth):
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        gs.append(g); ss.append(s)
    for i in range(4):
        with torch.cuda.stream(ss[i % depth]):
            gs[i % depth].replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(reps * depth):
        with torch.cuda.stream(ss[i % depth]):
            gs[i % depth].replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / (reps * depth)


def stage_input(stage, x):
    return stage.blocks(0)[0].run(x)


y = x3d.s1.run([clips])
y = x3d.s2.run(y); y = x3d.s3.run(y)
for name, st in (("s4", x3d.s4), ("s5", x3d.s5)):
    xin = stage_input(st, y[0])
    blocks = st.blocks(0)[1:]

    def layers():
        z = xin
        for b in blocks:
            z = b.run(z)
        return z
    E.X3D_STAGE["mode"] = "0"
    ref = layers().as_ncdhw().clone()
    t_l = graph_ms(layers)
    E.X3D_STAGE["mode"] = "1"
    spk = st.pk[0]
    got = E.x3d_stage(xin, spk).as_ncdhw().clone()
    torch.cuda.synchronize()
    E.check_range()
    err = (ref - got).abs().max().item() / ref.abs().max().item()
    t_s = graph_ms(lambda: E.x3d_stage(xin, spk))
    print("%s blocks 1..%d  in=%s: layers %.1f us (%.1f us/block), stage kernel %.1f us (%.1f us/block), rel err %.2e" % (
        name, len(blocks), (xin.N, xin.T, xin.H, xin.W, xin.C), 1e3 * t_l, 1e3 * t_l / len(blocks), 1e3 * t_s, 1e3 * t_s / len(blocks), err), flush=True)
    E.X3D_STAGE["mode"] = "0"
    y = st.run(y)
for mode in ("0", "1"):
    E.X3D_STAGE["mode"] = mode
    for depth in (1, 2):
        ms = graph_ms(lambda: x3d.forward_cl([clips]), depth=depth)
        print("backbone, stage kernel %s, %d in flight: %.3f ms/batch = %.3f of 8 TB/s at 1.33 GB/clip" % (
            "on" if mode == "1" else "off", depth, ms, B * 1.33 / ms / 8.0), flush=True)
torch.cuda.synchronize()
E.check_range()

# ---- where a block's time goes: 100 MHz stamps per workgroup, block and phase (median over workgroups and blocks)
import ctypes as C
from mspi_amd import _lib
lib = _lib.load()
E.X3D_STAGE["mode"] = "1"
y = x3d.s1.run([clips]); y = x3d.s2.run(y); y = x3d.s3.run(y)
for name, st in (("s4", x3d.s4), ("s5", x3d.s5)):
    xin = stage_input(st, y[0])
    spk = st.pk[0]
    nb = spk.nblocks
    stamps = torch.zeros(256 * nb * 16, dtype=torch.int64, device=dev)
    lib.mspi_x3d_stage_debug_stamps(stamps.data_ptr())
    for _ in range(3):
        E.x3d_stage(xin, spk)
    torch.cuda.synchronize()
    lib.mspi_x3d_stage_debug_stamps(None)
    s_ = stamps.view(256, nb, 16).double() * 0.01       # us
    se = torch.tensor([(spk.se_mask >> k) & 1 for k in range(nb)], dtype=torch.bool, device=dev)
    seg = {"A: a-GEMM + publish": s_[:, :, 1] - s_[:, :, 0], "wait for neighbours' t": s_[:, :, 2] - s_[:, :, 1],
           "B: depthwise": s_[:, :, 3] - s_[:, :, 2], "SE: arrive + wait": (s_[:, :, 6] - s_[:, :, 3])[:, se],
           "SE: gate + planes": (s_[:, :, 4] - s_[:, :, 6])[:, se],
           "C: c-GEMM (SE blocks)": (s_[:, :, 5] - s_[:, :, 4])[:, se], "C: c-GEMM (plain blocks)": (s_[:, :, 5] - s_[:, :, 3])[:, ~se],
           "  dw chunk 3: wait loads + LDS write": s_[:, :, 9] - s_[:, :, 8], "  dw chunk 3: barrier": s_[:, :, 10] - s_[:, :, 9],
           "  dw chunk 3: issue next loads": s_[:, :, 11] - s_[:, :, 10], "  dw chunk 3: compute + stores": s_[:, :, 12] - s_[:, :, 11],
           "  dw chunk 3: barrier 2": s_[:, :, 13] - s_[:, :, 12],
           "block (SE)": (s_[:, 1:, 0] - s_[:, :-1, 0])[:, se[:-1]], "block (plain)": (s_[:, 1:, 0] - s_[:, :-1, 0])[:, ~se[:-1]]}
    print("%s: per-phase time inside the stage kernel, us (median / mean / max over workgroups x blocks)" % name)
    for kname, v in seg.items():
        print("   %-38s %6.2f %6.2f %6.2f" % (kname, v.median().item(), v.mean().item(), v.max().item()))
    E.X3D_STAGE["mode"] = "0"
    y = st.run(y)
    E.X3D_STAGE["mode"] = "1"
