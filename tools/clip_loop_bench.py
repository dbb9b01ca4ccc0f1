#!/usr/bin/env python3
"""Sliding-window throughput (stride 1, as inference.py walks a video): eager launches vs the product's launch path
(runtime.GraphPipeline: forward + post-process kernels as one hipGraph, two batches in flight), each with and without the
per-frame feature cache.  Synthetic frames already on the GPU; autotuned tiles; windows/s = uint8 maps produced per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")      # as mspi_amd.inference's entry does (runtime.configure_hw_queues)
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model.model_utils import AudioVisualSaliencyModel
from mspi_amd.runtime import GraphPipeline

dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "x3dl"
B, n_frames = 8, 16 + 8 * 16
t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
cfg = T.make_cfg(name, num_aud_tokens=90, num_vis_tokens=t_tok * 49)
so, sys.stdout = sys.stdout, open(os.devnull, "w")
m = T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0).to(dev)
sys.stdout = so
g = torch.Generator().manual_seed(0)
video = torch.randn(n_frames, 3, 224, 224, generator=g).to(dev)
aud = torch.randn(B, 1, 257, 300, generator=g).to(dev)
E.autotune(True)
OUT = (480, 640)


def windows(first):
    idx = [list(range(first + b, first + b + 16)) for b in range(B)]
    clips = torch.stack([video[i].permute(1, 0, 2, 3) for i in idx])
    return clips, [j for w in idx for j in w]


def run(cached, graph):
    feats, nxt, t0, n, out, pipe, prev = {}, 0, None, 0, None, None, None
    for step, first in enumerate(range(0, n_frames - 16 - B + 1, B)):
        if step == 3:                      # three warm-up batches (tuning, allocator, capture)
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
        clips, flat = windows(first)
        inputs = [clips, aud]
        if cached:
            while nxt <= flat[-1]:
                hi = min(n_frames, nxt + 16)
                f1, f0 = m.encode_frames(video[nxt:hi])
                for j in range(nxt, hi):
                    feats[j] = (f1[j - nxt], f0[j - nxt])
                nxt = hi
            inputs += [torch.stack([feats[j][0] for j in flat]), torch.stack([feats[j][1] for j in flat])]
            for j in [j for j in feats if j < first + B]:
                del feats[j]
        fn = (lambda c, a, f1, f0, out=None: E.postprocess_u8(m(c, a, frame_feats=(f1, f0))[0], OUT, out=out)) if cached else \
             (lambda c, a, out=None: E.postprocess_u8(m(c, a)[0], OUT, out=out))
        if graph:
            if pipe is None:
                pipe = GraphPipeline(fn, inputs, depth=2, layouts=3, host_out=[((B,) + OUT, torch.uint8)])
            t = pipe.submit(*inputs)
            if prev is not None:
                out = pipe.fetch(prev).clone()       # the previous batch's maps come to the host while this one runs
            prev = t
        else:
            out = fn(*inputs).cpu()
        n += B
    if graph:
        out = pipe.fetch(prev).clone()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0), out


res = {}
side = torch.cuda.Stream()      # the loop's own launches stay off the NULL stream, as in inference.inference_dataset
with torch.cuda.stream(side):
    for cached in (False, True):
        for graph in (False, True):
            res[cached, graph] = run(cached, graph)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
print("%s, batch %d, windows/s incl. post-processing and D2H of the uint8 maps:" % (name, B))
print("  re-encoding every window : eager %.1f, hipGraph pipeline %.1f" % (res[False, False][0], res[False, True][0]))
print("  per-frame feature cache  : eager %.1f, hipGraph pipeline %.1f" % (res[True, False][0], res[True, True][0]))
print("  last batch, graph vs eager: %s; cache vs plain max |diff| %d grey levels" % (
    "identical" if torch.equal(res[True, True][1], res[True, False][1]) and torch.equal(res[False, True][1], res[False, False][1]) else "DIFFERENT",
    (res[True, True][1].int() - res[False, True][1].int()).abs().max().item()))
