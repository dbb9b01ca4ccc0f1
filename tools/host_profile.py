#!/usr/bin/env python3
"""cProfile of the eager forward's HOST side (Python wrappers, ctypes, torch.empty): the clip loop of mspi_amd.inference launches
kernel by kernel, and at ~430 launches per batch the host, not the GPU, sets its rate.  usage: host_profile.py [model] [forwards]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspi_amd import engine as E, testing as T
from mspi_amd.model.model_utils import AudioVisualSaliencyModel
name = sys.argv[1] if len(sys.argv) > 1 else "x3dl"
nfw = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
cfg = T.make_cfg(name, num_aud_tokens=9 * ((300 + 31) // 32), num_vis_tokens=t_tok * 49)
so, sys.stdout = sys.stdout, io.StringIO()
try:
    model = T.condition_(T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0), name).to(dev)
finally:
    sys.stdout = so
clips, audio = T.synth_inputs(8, 16, 224, 224, Wa=300, seed=100, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    E.autotune(True); model(clips, audio); E.autotune(False)
    for _ in range(3):
        model(clips, audio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nfw):
        model(clips, audio)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("eager: %.2f ms per forward to ISSUE (host), %.2f ms per forward until the GPU is done" % (1e3 * t_issue / nfw, 1e3 * t_all / nfw))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(nfw):
        model(clips, audio)
    pr.disable()
    torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(35)
print(st.getvalue()[:6000])
