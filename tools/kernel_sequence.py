#!/usr/bin/env python3
"""Ordered list of the C-ABI launches of one eager forward (one stream), with HIP-event times: the serial chains of a batch."""
import os, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model import model_utils as MU
from mspi_amd.model.model_utils import AudioVisualSaliencyModel
name = sys.argv[1] if len(sys.argv) > 1 else "x3dl"
dev = torch.device("cuda")
t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
cfg = T.make_cfg(name, num_aud_tokens=9 * ((300 + 31) // 32), num_vis_tokens=t_tok * 49)
so, sys.stdout = sys.stdout, io.StringIO()
try:
    model = T.condition_(T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0), name).to(dev)
finally:
    sys.stdout = so
clips, audio = T.synth_inputs(8, 16, 224, 224, Wa=300, seed=100, device=dev)
E.autotune(True); model(clips, audio); E.autotune(False)
MU._Fork.ENABLED = False
model(clips, audio); torch.cuda.synchronize()
with E.Profiler() as prof:
    model(clips, audio)
torch.cuda.synchronize()
tot = 0.0
for i, (nm, fl, by, e0, e1, det) in enumerate(prof.records):
    us = 1e3 * e0.elapsed_time(e1); tot += us
    print("%4d %-34s %7.1f us  (cum %7.2f ms)  %s" % (i, nm, us, tot / 1e3, det))
