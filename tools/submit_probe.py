#!/usr/bin/env python3
"""Where does GraphPipeline.submit(*inputs) lose time against the resident-input form?  (x3dl, batch 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
fn = lambda c, a: E.postprocess_u8(m(c, a)[0], (480, 640))
N = 24


def loop(pipe, mode, host):
    prev = None
    tf = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter(); th = 0.0
    for i in range(N):
        a = time.perf_counter()
        if mode == "resident":
            t = pipe.submit()
        elif mode == "same":
            t = pipe.submit(clips, aud)
        else:                                   # fresh tensors produced on the current stream right before
            c2 = clips * 1.0
            a2 = aud * 1.0
            t = pipe.submit(c2, a2)
        th += time.perf_counter() - a
        if prev is not None:
            a = time.perf_counter()
            if host:
                pipe.fetch(prev)
                b = time.perf_counter()
                o = pipe.fetch_host(prev)
                tf += time.perf_counter() - b
            else:
                pipe.fetch(prev)
        prev = t
    pipe.fetch(prev)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if host:
        print("      (D2H part of fetch_host: %.2f ms per batch; idle-queue probe %s)" % (1e3 * tf / N, getattr(pipe, "idle_latency_ms", None)))
    return 8 * N / el, 1e3 * th / N


side = torch.cuda.Stream()
for host in (False, True):
    pipe = GraphPipeline(fn, (clips, aud), depth=2, layouts=3)
    for stream_name, ctx in (("null stream", None), ("side stream", side)):
        for mode in ("resident", "same", "fresh"):
            if ctx is None:
                r, h = loop(pipe, mode, host)
            else:
                with torch.cuda.stream(ctx):
                    r, h = loop(pipe, mode, host)
            print("host_outputs=%-5s producer on %-11s inputs %-8s: %6.1f windows/s, submit() %.2f ms of host time" % (host, stream_name, mode, r, h), flush=True)
    del pipe
    torch.cuda.synchronize(); torch.cuda.empty_cache()
