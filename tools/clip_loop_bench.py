#!/usr/bin/env python3
"""Sliding-window throughput (stride 1, as inference.py walks a video) with and without the per-frame feature cache.
Synthetic frames already on the GPU; eager launches, autotuned tiles; windows/s = maps written per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model.model_utils import AudioVisualSaliencyModel

dev = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "x3dl"
B, n_frames = 8, 16 + 8 * 12
t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
cfg = T.make_cfg(name, num_aud_tokens=90, num_vis_tokens=t_tok * 49)
so, sys.stdout = sys.stdout, open(os.devnull, "w")
m = T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0).to(dev)
sys.stdout = so
g = torch.Generator().manual_seed(0)
video = torch.randn(n_frames, 3, 224, 224, generator=g).to(dev)
aud = torch.randn(B, 1, 257, 300, generator=g).to(dev)
E.autotune(True)


def windows(first):
    idx = [list(range(first + b, first + b + 16)) for b in range(B)]
    clips = torch.stack([video[i].permute(1, 0, 2, 3) for i in idx])
    return clips, [j for w in idx for j in w]


def run(cached):
    feats = {}
    nxt = 0
    t0 = None
    n = 0
    for step, first in enumerate(range(0, n_frames - 16 - B + 1, B)):
        if step == 2:                      # two warm-up batches (tuning, allocator)
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
        clips, flat = windows(first)
        if cached:
            while nxt <= flat[-1]:
                hi = min(n_frames, nxt + 16)
                f1, f0 = m.encode_frames(video[nxt:hi])
                for j in range(nxt, hi):
                    feats[j] = (f1[j - nxt], f0[j - nxt])
                nxt = hi
            ff = (torch.stack([feats[j][0] for j in flat]), torch.stack([feats[j][1] for j in flat]))
            out = m(clips, aud, frame_feats=ff)[0]
            for j in [j for j in feats if j < first + B]:
                del feats[j]
        else:
            out = m(clips, aud)[0]
        n += B
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0), out


r_plain, o_plain = run(False)
r_cache, o_cache = run(True)
print("%s: %.1f windows/s re-encoding every window, %.1f windows/s with the per-frame cache (x%.2f); last batch max |diff| %.2e" % (
    name, r_plain, r_cache, r_cache / r_plain, (o_plain - o_cache).abs().max().item()))
