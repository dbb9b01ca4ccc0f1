#!/usr/bin/env python3
"""Experiment: K independent hipGraphs of the same forward replayed round-robin on K streams (consecutive batches
overlap) vs one graph.  python tools/pipeline_probe.py [model] [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model.model_utils import AudioVisualSaliencyModel

dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "x3dl"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 8
t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
cfg = T.make_cfg(name, num_aud_tokens=90, num_vis_tokens=t_tok * 49)
so, sys.stdout = sys.stdout, open(os.devnull, "w")
m = T.condition_(T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0), name).to(dev)
sys.stdout = so
clips, audio = T.synth_inputs(B, 16, 224, 224, Wa=300, seed=100, device=dev)
E.autotune(True)
ref, _ = m(clips, audio)
E.autotune(False)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(depth)]
graphs, outs = [], []
for s in streams:
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        m(clips, audio)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        o, _ = m(clips, audio)
    graphs.append(g)
    outs.append(o)
torch.cuda.synchronize()


def run(k, steps=40):
    for i in range(4):
        with torch.cuda.stream(streams[i % k]):
            graphs[i % k].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % k]):
            graphs[i % k].replay()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


for k in range(1, depth + 1):
    print("%s: %d graph(s) in flight: %.1f clips/s" % (name, k, run(k)))
for o in outs:
    print("max |out - eager| = %.2e" % (o - ref).abs().max().item())
