#!/usr/bin/env python3
"""Map-level error of every model-level golden (the reference's own CPU fp32 outputs) under the current GEMM arithmetic.
Run once as is (f16x3: three split products) and once with MSPI_LIB_PATH=mspi_amd/csrc/libmspi_hip_single.so (built by
`make -C mspi_amd/csrc SINGLE=1`: plain f16 operands, fp32 accumulate):
the answer to BASELINE configs[4]'s "fp16 MFMA".  Prints one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mspi_amd import testing as T
from mspi_amd.model import model_utils as pm

dev = torch.device("cuda")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CASES = [("av_x3dl_224", "x3dl"), ("av_slowfast_224", "slowfast4x16"), ("av_mvit_224_wa300", "mvitv2s"), ("av_mvit_224x384", "mvitv2s"),
         ("av_swin_t_224", "videoswins"), ("av_swin_s_224", "videoswins"), ("av_s3d_224", "s3d"), ("av_uniformer_224", "uniformerb"),
         ("av_morphmlp_224", "morphmlps"), ("av_x3dl_64", "x3dl"), ("av_slowfast_64", "slowfast4x16")]
out = {"library": os.path.basename(os.environ.get("MSPI_LIB_PATH", "libmspi_hip.so")), "cases": {}}
so, sys.stdout = sys.stdout, open(os.devnull, "w")
for case, name in CASES:
    g = np.load(os.path.join(GOLD, case + ".npz"))
    cfg = T.golden_cfg(g, name)
    m = T.condition_(T.seeded(lambda: pm.AudioVisualSaliencyModel(cfg), int(g["seed"])), name).to(dev)
    H, W = T.golden_hw(g)
    clips, audio = T.synth_inputs(int(g["batch"]), 16, H, W, Wa=int(g["wa"]), seed=int(g["seed"]), device=dev)
    o, loss = m(clips, audio)
    ref = torch.as_tensor(g["out"])
    err = (o.cpu() - ref).abs()
    # the map in probability space, the form every saliency metric consumes: relative error at the peak
    p, pr = o.cpu().exp(), ref.exp()
    out["cases"][case] = {"max_abs_logmap": float(err.max()), "mean_abs_logmap": float(err.mean()),
                          "max_rel_prob_at_peaks": float(((p - pr).abs() / pr.max()).max()),
                          "loss_abs_err": abs(float(loss) - float(g["loss"])), "logmap_range": float(ref.max() - ref.min())}
    del m
    torch.cuda.empty_cache()
sys.stdout = so
out["worst_max_abs_logmap"] = max(c["max_abs_logmap"] for c in out["cases"].values())
print(json.dumps(out, indent=1))
