#!/usr/bin/env python3
"""Map-level error of the attention-bearing goldens for the library selected by MSPI_LIB_PATH (measurement builds of attn.hip
with -DMSPI_ATT_PV_DROP: one or both cross products of O += P.V dropped), plus an op-level error against fp64."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model import model_utils as pm

dev = torch.device("cuda")
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
out = {"library": os.path.basename(os.environ.get("MSPI_LIB_PATH", "libmspi_hip.so")), "cases": {}}
so, sys.stdout = sys.stdout, open(os.devnull, "w")
for case, name in (("av_mvit_224_wa300", "mvitv2s"), ("av_mvit_224x384", "mvitv2s"), ("av_swin_t_224", "videoswins"), ("av_swin_s_224", "videoswins"),
                   ("av_uniformer_224", "uniformerb"), ("av_x3dl_224", "x3dl")):
    g = np.load(os.path.join(GOLD, case + ".npz"))
    cfg = T.golden_cfg(g, name)
    m = T.condition_(T.seeded(lambda: pm.AudioVisualSaliencyModel(cfg), int(g["seed"])), name).to(dev)
    H, W = T.golden_hw(g)
    clips, audio = T.synth_inputs(int(g["batch"]), 16, H, W, Wa=int(g["wa"]), seed=int(g["seed"]), device=dev)
    o, _ = m(clips, audio)
    out["cases"][case] = float((o.cpu() - torch.as_tensor(g["out"])).abs().max())
    del m
    torch.cuda.empty_cache()
# op level: softmax(q k^T / sqrt(d)) v against fp64, unit-variance operands, 392 keys
gq = torch.Generator().manual_seed(0)
B, Hh, N, D = 2, 4, 392, 96
qkv = torch.randn(B * N, 3 * Hh * D, generator=gq)
x = E.CL(qkv.to(dev).view(-1), 0, B, 1, 1, N, 3 * Hh * D, 3 * Hh * D)
got = E.attention(x, B, N, Hh, D, D ** -0.5).as_rows().cpu().double().view(B, N, Hh, D)
q, k, v = qkv.double().view(B, N, 3, Hh, D).unbind(2)
ref = torch.einsum("bhqk,bkhd->bqhd", torch.softmax(torch.einsum("bqhd,bkhd->bhqk", q, k) * D ** -0.5, -1), v)
out["op_max_abs_vs_fp64"] = float((got - ref).abs().max())
out["op_scale"] = float(ref.abs().max())
sys.stdout = so
print(json.dumps(out))
