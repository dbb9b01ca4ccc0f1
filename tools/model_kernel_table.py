#!/usr/bin/env python3
"""Per-(kernel, shape) table of one eager forward of the bench model (HIP events around every launch): where the step goes."""
import os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model.model_utils import AudioVisualSaliencyModel
name = sys.argv[1] if len(sys.argv) > 1 else "x3dl"
dev = torch.device("cuda")
label = name
name, depths = {"videoswint": ("videoswins", [2, 2, 6, 2])}.get(label, (label, None))
t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
cfg = T.make_cfg(name, num_aud_tokens=9 * ((300 + 31) // 32), num_vis_tokens=t_tok * 49, swin_depths=depths)
so, sys.stdout = sys.stdout, io.StringIO()
try:
    model = T.condition_(T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0), name).to(dev)
finally:
    sys.stdout = so
clips, audio = T.synth_inputs(8, 16, 224, 224, Wa=300, seed=100, device=dev)
E.autotune(True); model(clips, audio); E.autotune(False)
torch.cuda.synchronize()
with E.Profiler() as prof:
    for _ in range(3):
        model(clips, audio)
torch.cuda.synchronize()
agg = {}
for nm, fl, by, e0, e1, det in prof.records:
    a = agg.setdefault((nm, det), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(a[1] for a in agg.values())
print("%s: %.2f ms per forward (sum of per-launch HIP-event times)" % (name, tot / 3))
for (nm, det), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("  %-32s x%-3d %7.1f us  %5.2f%%  %s" % (nm, a[0] // 3, 1e3 * a[1] / a[0], 100 * a[1] / tot, det))
