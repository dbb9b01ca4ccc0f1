#!/usr/bin/env python3
"""What does a small D2H copy cost while a hipGraph batch is in flight?  (diagnosis for runtime.GraphPipeline.fetch_host)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model.model_utils import AudioVisualSaliencyModel
from mspi_amd.runtime import GraphPipeline

dev = torch.device("cuda")
cfg = T.make_cfg("x3dl", num_aud_tokens=90)
so, sys.stdout = sys.stdout, open(os.devnull, "w")
m = T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0).to(dev)
sys.stdout = so
clips, aud = T.synth_inputs(8, 16, 224, 224, Wa=300, seed=1, device=dev)
E.autotune(True); m(clips, aud); E.autotune(False)
pipe = GraphPipeline(lambda c, a: E.postprocess_u8(m(c, a)[0], (480, 640)), (clips, aud), depth=2, layouts=3)
small = torch.zeros(8, 480, 640, dtype=torch.uint8, device=dev)
host = torch.empty(8, 480, 640, dtype=torch.uint8, pin_memory=True)
print("host buffer pinned:", host.is_pinned(), flush=True)
cs = torch.cuda.Stream()


def t_copy(src, label):
    torch.cuda.synchronize()
    pipe.submit(); pipe.submit()                    # two batches in flight
    time.sleep(0.002)
    t0 = time.perf_counter()
    with torch.cuda.stream(cs):
        host.copy_(src, non_blocking=True)
        ev = torch.cuda.Event(); ev.record()
    t1 = time.perf_counter()
    ev.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%-44s issue %.2f ms, copy done after %.2f ms, both batches done after %.2f ms" % (label, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t3 - t0)), flush=True)


for _ in range(2):
    t_copy(small, "D2H of an unrelated 2.4 MB device tensor")
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(cs):
    host.copy_(small, non_blocking=True)
torch.cuda.synchronize()
print("same copy on an idle GPU: %.2f ms" % (1e3 * (time.perf_counter() - t0)))
# a device-to-device copy kernel on the copy stream, for comparison
dst = torch.empty_like(small)
torch.cuda.synchronize()
pipe.submit(); pipe.submit(); time.sleep(0.002)
t0 = time.perf_counter()
with torch.cuda.stream(cs):
    dst.copy_(small); ev = torch.cuda.Event(); ev.record()
ev.synchronize()
print("D2D copy kernel beside two batches: done after %.2f ms" % (1e3 * (time.perf_counter() - t0)))
torch.cuda.synchronize()
