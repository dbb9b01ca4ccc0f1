#!/usr/bin/env python3
"""X3D-L backbone alone at batch 8: hipGraph replay time (1 and 2 batches in flight) and the per-kernel HIP-event table.
MSPI_X3D_FUSE=0 selects the unfused a / b launches for an A/B on the same box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.config import cfg
from mspi_amd.backbones.X3D import X3D
from mspi_amd.runtime import GraphPipeline

dev = torch.device("cuda")
B = 8
clips, _ = T.synth_inputs(B, 16, 224, 224, seed=100, device=dev)
x3d = T.seeded(lambda: X3D(cfg.MODEL.X3D.PATH_CFG), 0).to(dev)
E.autotune(True); x3d.forward_cl([clips]); E.autotune(False)
torch.cuda.synchronize()
fn = lambda c: tuple(f.buf for f in x3d.forward_cl([c]))
for depth in (1, 2):
    pipe = GraphPipeline(fn, (clips,), depth=depth)
    for _ in range(4):
        pipe.submit()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        pipe.submit()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print("fuse=%s  %d in flight: %.3f ms/batch = %.0f clips/s, %.3f of 8 TB/s at 1.33 GB/clip" % (
        os.environ.get("MSPI_X3D_FUSE", "1"), depth, ms, B / ms * 1e3, B * 1.33 / ms / 8.0))
    del pipe
torch.cuda.synchronize()
with E.Profiler() as prof:
    for _ in range(3):
        x3d.forward_cl([clips])
torch.cuda.synchronize()
summ = prof.summary()
tot = sum(d["ms"] for d in summ.values())
for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-28s calls %4d  %7.3f ms/fwd (%4.1f%%)  avg %6.1f us" % (k, d["calls"] // 3, d["ms"] / 3, 100 * d["ms"] / tot, 1e3 * d["ms"] / d["calls"]))
agg = {}
for name, fl, by, e0, e1, det in prof.records:
    if name.startswith("x3d_") or os.environ.get("X3D_PROBE_ALL"):
        a = agg.setdefault((name, det), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
for (name, det), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("    %-28s x%-3d %7.1f us  %s" % (name, a[0] // 3, 1e3 * a[1] / a[0], det))
