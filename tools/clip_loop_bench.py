#!/usr/bin/env python3
"""Sliding-window throughput (stride 1, as inference.py walks a video): eager launches vs the product's launch path
(runtime.GraphPipeline: forward + post-process kernels as one hipGraph, two batches in flight), each with and without the
per-frame feature cache.  Synthetic frames already on the GPU; autotuned tiles; windows/s = uint8 maps produced per second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
LAYOUTS = int(os.environ.get("CLB_LAYOUTS", "3"))
DEPTH = int(os.environ.get("CLB_DEPTH", "2"))
ONLY_GRAPH = os.environ.get("CLB_ONLY_GRAPH") == "1"      # as mspi_amd.inference's entry does (runtime.configure_hw_queues)
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
    if os.environ.get("CLB_SLICES"):     # slices: no index tensor, i.e. no pageable H2D copy per window
        clips = torch.stack([video[first + b: first + b + 16].permute(1, 0, 2, 3) for b in range(B)])
    else:
        clips = torch.stack([video[i].permute(1, 0, 2, 3) for i in idx])
    return clips, [j for w in idx for j in w]


KEEP = {}


def run(cached, graph):
    feats, nxt, t0, n, out, pipe, prev, pend = {}, 0, None, 0, None, None, None, []
    for step, first in enumerate(range(0, n_frames - 16 - B + 1, B)):
        if step == 3:                      # three warm-up batches (tuning, allocator, capture)
            torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
        t_it = time.perf_counter()
        if os.environ.get("CLB_BUILD") and step > 0 and graph:
            clips, flat = KEEP["clips"], [j for b in range(B) for j in range(first + b, first + b + 16)]
        else:
            clips, flat = windows(first)
            KEEP["clips"] = clips
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
        fn = (lambda c, a, f1, f0: E.postprocess_u8(m(c, a, frame_feats=(f1, f0))[0], OUT)) if cached else \
             (lambda c, a: E.postprocess_u8(m(c, a)[0], OUT))
        if graph:
            if pipe is None:
                pipe = GraphPipeline(fn, inputs, depth=DEPTH, layouts=LAYOUTS)
                idle = pipe.idle_streams(2)
                pipe._copy_stream = idle[0]
                if not os.environ.get("CLB_KEEP_STREAM"):
                    idle[1].wait_stream(torch.cuda.current_stream())
                    torch.cuda.set_stream(idle[1])                 # the loop's own launches move to a free hardware queue
                print("    [idle-queue probe, ms]", pipe.idle_latency_ms, flush=True)
            ta = time.perf_counter()
            if os.environ.get("CLB_BUILD"):
                def build(ins, first=first, extra=inputs[2:]):
                    idx = [list(range(first + b, first + b + 16)) for b in range(B)]
                    torch.stack([video[i].permute(1, 0, 2, 3) for i in idx], out=ins[0])
                    for d_, s_ in zip(ins[2:], extra):
                        d_.copy_(s_)
                t = pipe.submit_build(build)
            else:
                t = pipe.submit(*inputs)
            tb = time.perf_counter()
            pend.append(t)
            prev = pend.pop(0) if len(pend) >= DEPTH else None
            if prev is not None:
                pipe.fetch(prev)
                tc = time.perf_counter()
                out = pipe.fetch_host(prev).clone()  # the previous batch's maps come to the host while this one runs
                if os.environ.get("CLB_DEBUG"):
                    print("    step %2d: build %.1f ms, submit %.1f, wait for previous batch %.1f, D2H + clone %.1f" % (
                        step, 1e3 * (ta - t_it), 1e3 * (tb - ta), 1e3 * (tc - tb), 1e3 * (time.perf_counter() - tc)), flush=True)
        else:
            out = fn(*inputs).cpu()
        n += B
    if graph:
        for t in pend:
            out = pipe.fetch_host(t).clone()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0), out


res = {}
side = torch.cuda.Stream()      # the loop's own launches stay off the NULL stream, as in inference.inference_dataset
with torch.cuda.stream(side):
    for cached in (False, True):
        for graph in (False, True):
            res[cached, graph] = run(cached, graph) if (graph or not ONLY_GRAPH) else (0.0, None)
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
print("%s, batch %d, windows/s incl. post-processing and D2H of the uint8 maps:" % (name, B))
print("  re-encoding every window : eager %.1f, hipGraph pipeline %.1f" % (res[False, False][0], res[False, True][0]))
print("  per-frame feature cache  : eager %.1f, hipGraph pipeline %.1f" % (res[True, False][0], res[True, True][0]))
if not ONLY_GRAPH:
    print("  last batch, graph vs eager: %s; cache vs plain max |diff| %d grey levels" % (
        "identical" if torch.equal(res[True, True][1], res[True, False][1]) and torch.equal(res[False, True][1], res[False, False][1]) else "DIFFERENT",
        (res[True, True][1].int() - res[False, True][1].int()).abs().max().item()))
