#!/usr/bin/env python3
"""Why the UniFormer test weights are damped (testing.condition_): with variance-preserving random weights the 40-block
pre-norm trunk grows its activations, and fp32 rounding alone -- the REFERENCE arithmetic against its own fp64 evaluation --
already costs more than the parity bar.  CPU only: oracle/restate.py in fp32 vs fp64, undamped and damped."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import testing as T
from mspi_amd.backbones.uniformer import Uniformer
from mspi_amd.config import cfg
from oracle import restate as R

torch.set_num_threads(8)
clips, _ = T.synth_inputs(1, 16, 64, 64, seed=0)
for damped in (False, True):
    m = T.seeded(lambda: Uniformer(cfg.MODEL.UNIFORMER.PATH_CFG), 0)
    if damped:
        T.condition_(m, "uniformerb")
    sd32 = {k: v.clone() for k, v in m.state_dict().items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
    with torch.no_grad():
        f32 = R.uniformer_forward(sd32, clips)
        f64 = R.uniformer_forward(sd64, clips.double())
    for i, (a, b) in enumerate(zip(f32, f64)):
        print("%s  feature %d: abs-max %9.2f   fp32 vs fp64: max abs %.2e  (%.1e of abs-max)" % (
            "damped  " if damped else "undamped", i + 1, b.abs().max().item(), (a.double() - b).abs().max().item(),
            (a.double() - b).abs().max().item() / b.abs().max().item()))
