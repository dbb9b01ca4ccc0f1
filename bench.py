#!/usr/bin/env python3
"""bench.py -- MSPI saliency-inference hot path on MI355X.

One "step" = one forward of AudioVisualSaliencyModel (X3D-L motion encoder + ConvNeXt-T image
encoder + VGGSound ResNet-18 + SyncBlock + decoder) over one batch of synthetic clips that is
already resident in HBM: BASELINE.json configs[1] = batch 8 per GPU of 16x224x224 RGB clips +
1x257x300 log-spectrograms, fp32, random weights of the real architecture.  Weak scaling: every
rank runs its own batch of 8; rank 0 broadcasts the weights once (RCCL) and gathers the maps every
step.  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--model", default="x3dl", help="motion encoder: x3dl (configs[1], the bench line), slowfast4x16, mvitv2s, "
                    "videoswins (the reference default, Swin-S), videoswint (configs[4]: the same class with depths 2,2,6,2), "
                    "s3d, uniformerb, morphmlps")
    ap.add_argument("--swin-depths", type=int, nargs=4, default=None, help="Video-Swin stage depths (overrides --model's)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--wa", type=int, default=300, help="spectrogram columns (BASELINE: 300; reference default 111)")
    ap.add_argument("--no-graph", "--eager", dest="no_graph", action="store_true",
                    help="launch eagerly instead of replaying a hipGraph (the clip loop's default launch path)")
    ap.add_argument("--no-eager-line", action="store_true", help="skip the short eager-launch measurement reported as `eager` beside "
                    "the hipGraph number (single-GPU runs only)")
    ap.add_argument("--inflight", type=int, default=None, help="hipGraphs of the forward kept in flight (mspi_amd.runtime.GraphPipeline): "
                    "consecutive steps (batches) replay round-robin on this many streams, so the low-occupancy tail of one batch "
                    "overlaps the head of the next (each step is still one full forward of one batch; 1 = one batch at a time). "
                    "Default: 2 (forked graphs) in a single process, 3 (linear graphs) under a process group")
    ap.add_argument("--stream-layouts", type=int, default=4, help="stream layouts tried for the in-flight graphs during the untimed "
                    "set-up (which streams share a hardware queue decides how well two batches overlap); 1 = take the first")
    ap.add_argument("--no-autotune", action="store_true", help="keep the library's tile heuristic (default: time the "
                    "kernel instantiations per conv shape during the first, untimed forward -- cudnn.benchmark's role upstream)")
    ap.add_argument("--tune-cache", default=None, help="JSON file of tile choices: loaded when it exists (no tuning launches, "
                    "e.g. under rocprofv3), otherwise written after the first forward")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1: nccl (= RCCL over xGMI); gloo only "
                    "to rehearse the multi-rank control flow on a box where the ranks have to share one GPU")
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / barrier / max-over-ranks timing / JSON line with "
                    "NO GPU work (CPU test of the --gpus N launch path; the line says dry_run and carries no throughput)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-postproc", action="store_true", help="skip the second timed pass with the post-process kernel in the graph")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel timing table to stderr")
    ap.add_argument("--kernel-detail", type=int, default=0, help="also print the N slowest individual launches")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, as FRESH child
    processes of this one -- which has not imported torch or touched the GPU yet, and never exec()s -- and hand back the
    launcher's exit code.  Rank 0's JSON line goes straight to our stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


ARGS = parse() if __name__ == "__main__" else None
if ARGS is not None and ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(launch_ranks(ARGS.gpus))

# Single-process runs give every in-flight graph branch its own hardware queue (runtime.configure_hw_queues: must happen
# before the HIP runtime initialises; multi-rank runs keep the default because RCCL's streams take queues too).
if ARGS is not None and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("MSPI_BENCH_FORCE_DIST"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")
elif ARGS is not None:
    # With a process group RCCL's own streams take hardware queues, and which queue a graph BRANCH lands on then decides up to
    # 20 % of a rank's rate (forked graphs, two in flight, 5 queues: layouts 78.7 / 63.8 / 77.4 / 74.6 batches/s; best 97.5 % of
    # the single-process line).  Ranks therefore replay LINEAR graphs (MSPI_STREAMS=0: no fork inside a batch), three in
    # flight, on the runtime's default queue count: one rank through the whole RCCL path on a one-GPU box reaches 99.4 % of the
    # single-process line and all four stream layouts are equal (82.2 / 81.7 / 82.2 / 81.5) -- tools/dist_rehearsal.sh,
    # profiles/r03_dist_rehearsal.txt.  Explicit settings in the environment / on the command line win.
    os.environ.setdefault("MSPI_STREAMS", "0")
if ARGS is not None and ARGS.inflight is None:
    ARGS.inflight = 3 if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("MSPI_BENCH_FORCE_DIST")) and \
        os.environ.get("MSPI_STREAMS") == "0" else 2

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense (the headline 5 PF is 2:1 sparse)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak


def host_cores():
    """CPU cores this process may actually use: physical cores (unique (package, core id) pairs of /proc/cpuinfo) limited
    by the affinity mask and the cgroup CPU quota -- a GPU box hands a one-GPU job a share of the host, and a thread pool
    sized by the machine's 128 logical CPUs then oversubscribes that share several times over."""
    logical = len(os.sched_getaffinity(0))
    phys = set()
    try:
        pkg = core = None
        allowed = os.sched_getaffinity(0)
        cpu = None
        for ln in open("/proc/cpuinfo"):
            k, _, v = ln.partition(":")
            k = k.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "physical id":
                pkg = int(v)
            elif k == "core id":
                core = int(v)
            elif not k and cpu is not None:
                if cpu in allowed and pkg is not None and core is not None:
                    phys.add((pkg, core))
                pkg = core = cpu = None
    except (OSError, ValueError):
        pass
    n = len(phys) or logical
    try:                                           # cgroup v2 quota: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, logical))


def cpu_baseline(sd, cfg, name, clips, audio, budget_s=float(os.environ.get("MSPI_BENCH_CPU_BUDGET_S", "60"))):
    """The oracle (CPU restatement pinned to the reference) on the host cores of this box, SURVEY 8d: torch threads = the
    physical cores this job may use, batch 1 AND the bench batch, median of >= 5 forwards each (3 when one forward of the
    full batch alone would take more than a fifth of the budget).  A reported baseline, not the target."""
    from oracle import restate as R
    cores = host_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)

    def run(b, n):
        ts = []
        with torch.no_grad():
            for i in range(n + 1):                 # first forward untimed (thread pool, allocator)
                t0 = time.time()
                R.audio_visual_forward(sd, clips[:b], audio[:b], name, cfg.MODEL.LATERAL_BOOL, cfg.MODEL.LATERAL_STRIDE)
                if i:
                    ts.append(time.time() - t0)
                elif b > 1 and time.time() - t0 > budget_s / 5:
                    n = min(n, 3)
                if len(ts) >= n:
                    break
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    try:
        t1, n1 = run(1, 5)
        B = clips.shape[0]
        tb, nb = run(B, 5) if B > 1 else (t1, n1)
    finally:
        torch.set_num_threads(prev)
    return {"value": round(B / tb, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "value_b1": round(1.0 / t1, 4), "batch": B,
            "sample": "oracle/restate.py (torch %s CPU fp32, %d threads = physical cores available to the job) on the same workload: "
                      "median of %d forwards at batch %d (`value`), median of %d at batch 1 (`value_b1`)"
                      % (torch.__version__, cores, nb, B, n1)}


def pmc_traffic(kname, model, B, S, wa):
    """HBM bytes per launch of `kname` from the committed rocprofv3 counter passes (profiles/*_traffic.json, written by
    profiles/summarize.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload; gfx950 corrections
    applied there).  bench.py cannot collect PMC counters on itself, so this is the last profiled build's figure."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            t = json.load(open(f))
        except (OSError, ValueError):
            continue
        if t.get("workload") == [model, B, S, wa] and kname in t.get("kernels", {}):
            k = t["kernels"][kname]
            return {"traffic": round(k["read_bytes"] + k["write_bytes"]), "traffic_unit": "HBM bytes/launch (2*FETCH_SIZE + WRITE_SIZE, KiB->B)",
                    "traffic_source": "profiles/" + os.path.basename(f)}
    return {}


def dry_run(args, rank, world):
    """The launch contract without a GPU: rendezvous, barrier, K empty steps, max over ranks, one JSON line."""
    multi = world > 1
    if multi:
        dist.init_process_group("gloo")          # the dry run never touches the GPU, whatever --backend says
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    if rank == 0:
        print(json.dumps({"metric": "clips_per_sec", "value": None, "unit": "clips/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dry_run": True, "data": "none",
                          "config": {"workload": "dry run of the launch path: no GPU work", "global_batch": world * args.batch,
                                     "launch": "%d batches in flight, %s graphs" % (args.inflight, "linear" if os.environ.get("MSPI_STREAMS") == "0" else "forked")}}))
    if multi:
        dist.destroy_process_group()


MODEL_ALIASES = {"videoswint": ("videoswins", [2, 2, 6, 2])}


def main():
    args = ARGS
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    # MSPI_BENCH_FORCE_DIST=1 under torch.distributed.run with ONE rank takes the whole multi-rank path (process group,
    # weight broadcast, per-step gather, barriers) -- the only way to exercise the RCCL calls on a one-GPU box
    multi = world > 1 or bool(os.environ.get("MSPI_BENCH_FORCE_DIST"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if os.environ.get("MSPI_BENCH_SHARE_GPU"):     # rehearsal of the N>1 control flow on a one-GPU box
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if multi:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from mspi_amd import engine as E
    from mspi_amd import testing as T
    from mspi_amd.model.model_utils import AudioVisualSaliencyModel
    from mspi_amd.runtime import GraphPipeline

    label, B, S = args.model, args.batch, args.size
    name, depths = MODEL_ALIASES.get(label, (label, None))
    depths = args.swin_depths or depths
    t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
    cfg = T.make_cfg(name, num_aud_tokens=9 * ((args.wa + 31) // 32), num_vis_tokens=t_tok * (S // 32) ** 2, swin_depths=depths)
    if name == "videoswins":
        label = "videoswin depths %s" % (list(cfg.MODEL.SWIN.DEPTHS),)
    devnull = open(os.devnull, "w")
    so, sys.stdout = sys.stdout, devnull          # the constructors print; keep stdout to the one JSON line
    try:
        model = T.condition_(T.seeded(lambda: AudioVisualSaliencyModel(cfg), 0), name)
    finally:
        sys.stdout = so
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev)
    if multi:   # weight fan-out: one flat RCCL broadcast from rank 0 over xGMI
        from mspi_amd.sharding import broadcast_weights
        broadcast_weights(model, 0)
    clips, audio = T.synth_inputs(B, 16, S, S, Wa=args.wa, seed=100 + rank, device=dev)

    gathered = [torch.empty(B, S, S, device=dev) for _ in range(world)] if (multi and rank == 0) else None

    have_cache = bool(args.tune_cache) and os.path.exists(args.tune_cache)
    if have_cache:
        E.load_autotune(args.tune_cache)
    E.autotune(not args.no_autotune and not have_cache)
    out, loss = model(clips, audio)               # packs weights, warms the allocator, autotunes the conv tiles
    E.autotune(False)                             # from here on: cached choices only
    if args.tune_cache and not have_cache and rank == 0:
        E.save_autotune(args.tune_cache)
    torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            sys.stderr.write(msg + "\n")

    pipe = None
    if not args.no_graph:       # the product's launch path: mspi_amd/runtime.py
        pipe = GraphPipeline(lambda c, a: model(c, a)[0], (clips, audio), depth=args.inflight,
                             layouts=args.stream_layouts, log=log)
    depth = pipe.depth if pipe else 1
    outs = [s.outs[0] for s in pipe.slots] if pipe else [out]
    collecting = multi or bool(os.environ.get("MSPI_BENCH_FAKE_COLLECT"))     # FAKE: the stream/event choreography without dist
    # Map collection (graph mode).  Each replay is followed, ON ITS OWN STREAM, by a copy of its maps into one of NSLOT
    # staging buffers and an event record (GraphPipeline.after); the gather runs on a separate non-blocking stream behind
    # that event.  Nothing ever makes a graph's stream wait on another stream, and nothing runs on torch's default stream
    # (runtime.py).  Re-use of a staging slot (NSLOT steps later) is guarded on the HOST.
    NSLOT = 4
    stage = [torch.empty_like(outs[0]) for _ in range(NSLOT)] if (pipe and collecting) else []
    done = [torch.cuda.Event() for _ in range(NSLOT)]
    collected = [None] * NSLOT
    pending = [None]                              # staging slot whose maps still have to be collected
    comm_stream = torch.cuda.Stream() if pipe else torch.cuda.current_stream()
    counter = [0]

    def gather(o):
        if not multi or os.environ.get("MSPI_BENCH_NO_GATHER"):     # experiment switches: everything but the collective
            return
        if args.backend == "nccl":
            dist.gather(o, gathered, dst=0)       # RCCL over xGMI
        else:                                     # gloo has no device gather: rehearsal only
            dist.gather(o.cpu(), [g.cpu() for g in gathered] if gathered is not None else None, dst=0)

    def collect(slot):
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(done[slot])
            gather(stage[slot])
            collected[slot] = torch.cuda.Event()
            collected[slot].record()

    def flush():
        if pending[0] is not None:
            collect(pending[0])
            pending[0] = None

    def stage_maps(k, o):                         # runs on the replay's own stream, right behind it
        slot = (counter[0] - 1) % NSLOT
        stage[slot].copy_(o[0])
        done[slot].record()

    if pipe and collecting:
        pipe.after = stage_maps

    def step():
        i = counter[0]
        counter[0] += 1
        if not pipe:
            o, _ = model(clips, audio)
            outs[0] = o
            gather(o)
            return
        if not collecting:
            pipe.submit()
            return
        slot = i % NSLOT
        if collected[slot] is not None and not collected[slot].query():
            collected[slot].synchronize()         # host-side guard; with 4 slots it never actually waits
        pipe.submit()
        # the gather of the PREVIOUS batch is issued after this batch's replay is queued, so a collective that blocks the
        # host cannot drain the GPU between batches
        prev, pending[0] = pending[0], slot
        if prev is not None:
            collect(prev)

    def timed(step_fn, flush_fn):
        for _ in range(args.warmup):
            step_fn()
        flush_fn()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        ta = time.perf_counter()
        flush_fn()                                # the last batch's maps: K replays and K gathers inside the timed region
        torch.cuda.synchronize()
        tb = time.perf_counter()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if os.environ.get("MSPI_BENCH_DEBUG") and rank == 0:
            sys.stderr.write("[bench] enqueue %.1f ms, drain %.1f ms, barrier %.1f ms\n" % (
                1e3 * (ta - t0), 1e3 * (tb - ta), 1e3 * (t0 + el - tb)))
        if multi:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el

    elapsed = timed(step, flush)

    E.check_range()                               # f16x3 range guard: any GEMM result that left the window raises here
    ok = all(bool(torch.isfinite(o).all().item()) and abs(torch.logsumexp(o.flatten(1), 1)).max().item() < 1e-3 for o in outs)
    if depth > 1:                                 # every graph in flight computed the same maps
        ok = ok and all(torch.equal(outs[0], o) for o in outs[1:])
    if not ok:
        raise SystemExit("bench: the saliency maps are not finite log-probability maps")

    # SURVEY 8d metric (2): saliency-map ms/clip INCLUDING the post-process kernels (inference.py:72-91: blur, exp, resize to
    # 640x480, min-max, uint8) -- the same pipeline with mspi_postprocess_u8 captured behind the forward, timed the same way.
    pp = None
    lat_main = round(pipe.latency_ms(), 4) if pipe else None
    layout_main = pipe.layout if pipe else None
    if pipe and not args.no_postproc:
        # the SAME pipeline (same graphs, same streams, same hardware queues) with the four post-process launches issued on
        # each replay's own stream right behind it: a second pipeline lands on other queues and measures the layout, not the
        # kernels (seen: +1.1 ms latency for 0.13 ms of kernels)
        prev_after = pipe.after
        u8 = [None] * depth

        def with_postproc(k, o):
            if prev_after is not None:
                prev_after(k, o)
            u8[k] = E.postprocess_u8(o[0], (480, 640))

        pipe.after = with_postproc
        el_pp = timed(step, flush)
        lat = []
        for _ in range(7):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pipe.fetch(pipe.submit())
            lat.append(time.perf_counter() - t1)
        pipe.after = prev_after
        if not all(u is not None and u.dtype == torch.uint8 and tuple(u.shape) == (B, 480, 640) and int(u.max()) == 255
                   and int(u.min()) == 0 for u in u8):
            raise SystemExit("bench: post-processed maps are not min-max normalised uint8 images")
        pp = {"ms_per_clip_with_postproc": round(1e3 * el_pp / args.steps / B, 4),
              "latency_ms_per_batch_with_postproc": round(1e3 * sorted(lat)[len(lat) // 2], 4)}
        del u8
    graph_mode = not args.no_graph
    # The drop-in entry (mspi_amd/inference.py) launches EAGERLY by default (its hipGraph path stalls beside the loop's host
    # work: DESIGN.md section 3, profiles/r03_clip_loop_pipeline.txt), so its model rate is put on record beside the graph's:
    # the same forward, same resident inputs, launched call by call from Python on the three branch streams.
    eager = None
    if pipe and not multi and not args.no_eager_line:
        ne = max(3, min(10, args.steps))
        for _ in range(2):
            model(clips, audio)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ne):
            model(clips, audio)
        torch.cuda.synchronize()
        el_e = time.perf_counter() - t1
        eager = {"value": round(B * ne / el_e, 3), "unit": "clips/s", "ms_per_step": round(1e3 * el_e / ne, 4), "steps": ne,
                 "launch": "eager: one Python call per kernel, three branch streams (the clip loop's default)"}

    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    line = {
        "metric": "clips_per_sec", "value": round(world * B * args.steps / elapsed, 3), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "ms_per_clip": round(1e3 * elapsed / args.steps / B, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if E.DEFAULT_PREC == E.PREC_F32 else
                 ("f32 storage/accumulate; GEMMs on plain f16 operands (libmspi_hip_single.so: measurement build, fails the 1e-3 parity bar)"
                  if "single" in os.path.basename(os.environ.get("MSPI_LIB_PATH", "")) else
                  "f32 storage/accumulate; GEMMs as f16x3 split products on the f16 MFMA pipe"),
        "data": "synthetic",
        "config": {"workload": "%s motion encoder + ConvNeXt-T + ResNet18 audio + SyncBlock + decoder (AudioVisualSaliencyModel "
                               "forward), batch %d/GPU, clips 3x16x%dx%d, spectrogram 1x257x%d, inputs resident in HBM"
                               % (label, B, S, S, args.wa),
                   "global_batch": world * B,
                   "launch": "eager" if not graph_mode else "hipGraph replay (mspi_amd.runtime.GraphPipeline), %d batch%s in flight%s"
                             % (depth, "es" if depth > 1 else "",
                                ", linear graphs (no branch streams)" if os.environ.get("MSPI_STREAMS") == "0" else ""),
                   "stream_layout": layout_main,
                   "parallelism": "clip-sharded x%d (weights broadcast once, maps gathered per step over RCCL)" % world
                   if multi else "single GPU"},
    }
    if pp:
        line.update(pp)
    if eager:
        line["eager"] = eager
    if graph_mode:   # latency of ONE batch alone on the chip (no neighbouring batch in flight): median of 7 replays
        line["latency_ms_per_batch"] = lat_main
        line["saliency_map_ms_per_clip"] = round((pp["latency_ms_per_batch_with_postproc"] if pp else line["latency_ms_per_batch"]) / B, 4)

    if not args.no_roofline:
        # per-launch HIP-event timing of every C-ABI call: eager, same inputs, ONE stream (the branch overlap is
        # switched off here so that each launch is timed alone on the chip, as rocprofv3's kernel trace does)
        from mspi_amd.model import model_utils as MU
        fork, MU._Fork.ENABLED = MU._Fork.ENABLED, False
        try:
            model(clips, audio)                   # untimed: the allocator's pools for the one-stream schedule
            torch.cuda.synchronize()
            with E.Profiler() as prof:
                for _ in range(3):
                    model(clips, audio)
            torch.cuda.synchronize()
        finally:
            MU._Fork.ENABLED = fork
        summ = prof.summary()
        tot = sum(d["ms"] for d in summ.values())
        if args.kernel_table:
            for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
                sys.stderr.write("%-28s calls %5d  %8.3f ms/step (%4.1f%%)  %7.2f TFLOP/s  %7.1f GB/s (algorithmic)\n" % (
                    k, d["calls"] // 3, d["ms"] / 3, 100 * d["ms"] / tot, d["flops"] / d["ms"] / 1e9, d["bytes"] / d["ms"] / 1e6))
        if args.kernel_detail:
            agg = {}
            for name, fl, by, e0, e1, det in prof.records:
                a = agg.setdefault((name, det), [0, 0.0, fl, by])
                a[0] += 1
                a[1] += e0.elapsed_time(e1)
            for (name, det), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.kernel_detail]:
                sys.stderr.write("  %-30s x%-3d %8.1f us/launch %7.1f TF/s %7.0f GB/s  %s\n" % (
                    name, a[0] // 3, 1e3 * a[1] / a[0], a[2] / (a[1] / a[0]) / 1e9, a[3] / (a[1] / a[0]) / 1e6, det))
        kname, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
        tflops = d["flops"] / d["ms"] / 1e9
        gbs = d["bytes"] / d["ms"] / 1e6
        # the fused MLP (mspi_mlp_fwd) and attention are f16x3 matrix-pipe kernels like the GEMMs (their names carry no tile string)
        mfma_bound = kname.startswith("conv_gemm") or kname in ("attention", "mlp_fused")
        f16x3 = "f16x3" in kname or (kname in ("attention", "mlp_fused") and E.DEFAULT_PREC == E.PREC_F16X3)
        peak = F16_MFMA_PEAK_TFLOPS if f16x3 else FP32_MFMA_PEAK_TFLOPS
        # which roof is nearer is judged by how busy the unit is: an f16x3 kernel keeps the matrix pipe 3x as busy as its
        # algorithmic flops say; `achieved` / `frac` below stay ALGORITHMIC flops against the dense f16 peak
        if mfma_bound and (3 if f16x3 else 1) * tflops / peak >= gbs / HBM_PEAK_GBS:
            # achieved = ALGORITHMIC flops (2*M*N*K) per second.  An f16x3 kernel issues 3 f16 MFMA flops per
            # algorithmic flop (hi*hi + hi*lo + lo*hi), so the matrix pipe is 3x busier than `frac` says.
            line["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": round(tflops, 3), "peak": peak,
                                "unit": "TFLOP/s", "frac": round(tflops / peak, 4), "traffic": None,
                                "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"]),
                                "mfma_dtype": "f16 (3 products per fp32 multiply, fp32 accumulate)" if f16x3 else "f32",
                                "mfma_pipe_frac": round((3 if f16x3 else 1) * tflops / peak, 4)}
        else:
            line["roofline"] = {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                "algorithmic_bytes_per_launch": round(d["bytes"] / d["calls"])}
        line["roofline"].update(pmc_traffic(kname, args.model, B, S, args.wa))
        line["roofline"]["launches_per_step"] = d["calls"] // 3
        line["roofline"]["avg_launch_us"] = round(1e3 * d["ms"] / d["calls"], 3)
        line["roofline"]["share_of_step"] = round(d["ms"] / tot, 4)

    if not multi and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd_cpu, cfg, name, clips.cpu(), audio.cpu())

    print(json.dumps(line))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
