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

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order.  Two graphs in flight x three
# branch streams each overlap best when every one of them has its own queue: measured on the bench line, same box,
# 4 queues 570 / 6 queues 595-601 / 8 queues 559 clips/s (and +2.8 % on mvitv2s).  With a process group RCCL's own
# streams take queues too and the picture flips (4: 567, 5: 571, 6: 498, 7: 427; high-priority RCCL streams or touching the
# pool streams before RCCL starts do not repair it reliably), so only the single-process run sets it.
# Must happen before the HIP runtime initialises, i.e. before torch is imported; an explicit setting wins.
if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("MSPI_BENCH_FORCE_DIST"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "6")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense (the headline 5 PF is 2:1 sparse)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU (BASELINE configs[1]: 8)")
    ap.add_argument("--model", default="x3dl")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--wa", type=int, default=300, help="spectrogram columns (BASELINE: 300; reference default 111)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--inflight", type=int, default=2, help="hipGraphs of the forward kept in flight: consecutive steps (batches) "
                    "replay round-robin on this many streams, so the low-occupancy tail of one batch overlaps the head of the "
                    "next (each step is still one full forward of one batch; 1 = strictly one batch at a time)")
    ap.add_argument("--stream-layouts", type=int, default=4, help="stream layouts tried for the in-flight graphs during the untimed "
                    "set-up (which streams share a hardware queue decides how well two batches overlap); 1 = take the first")
    ap.add_argument("--no-autotune", action="store_true", help="keep the library's tile heuristic (default: time the "
                    "kernel instantiations per conv shape during the first, untimed forward -- cudnn.benchmark's role upstream)")
    ap.add_argument("--tune-cache", default=None, help="JSON file of tile choices: loaded when it exists (no tuning launches, "
                    "e.g. under rocprofv3), otherwise written after the first forward")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1: nccl (= RCCL over xGMI); gloo only "
                    "to rehearse the multi-rank control flow on a box where the ranks have to share one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print the per-kernel timing table to stderr")
    ap.add_argument("--kernel-detail", type=int, default=0, help="also print the N slowest individual launches")
    return ap.parse_args()


def cpu_baseline(sd, cfg, name, clips, audio, budget_s=float(os.environ.get("MSPI_BENCH_CPU_BUDGET_S", "25"))):
    """The oracle (CPU restatement pinned to the reference) on the host cores of this box: B=1 clips,
    repeated until ~budget_s of CPU work.  A reported baseline, not the target."""
    from oracle import restate as R
    n = 0
    t0 = time.time()
    with torch.no_grad():
        while True:
            R.audio_visual_forward(sd, clips[n % clips.shape[0]:n % clips.shape[0] + 1],
                                   audio[n % audio.shape[0]:n % audio.shape[0] + 1], name,
                                   cfg.MODEL.LATERAL_BOOL, cfg.MODEL.LATERAL_STRIDE)
            n += 1
            el = time.time() - t0
            if el > budget_s or (n >= 3 and el > 0.6 * budget_s):
                break
    return {"value": round(n / el, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d single-clip forwards of the same workload (B=1) through oracle/restate.py, torch %s CPU fp32"
                      % (n, torch.__version__)}


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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # MSPI_BENCH_FORCE_DIST=1 under torch.distributed.run with ONE rank takes the whole multi-rank path (process group,
    # weight broadcast, per-step gather, barriers) -- the only way to exercise the RCCL calls on a one-GPU box
    multi = world > 1 or bool(os.environ.get("MSPI_BENCH_FORCE_DIST"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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

    name, B, S = args.model, args.batch, args.size
    t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(name, 8)
    cfg = T.make_cfg(name, num_aud_tokens=9 * ((args.wa + 31) // 32), num_vis_tokens=t_tok * (S // 32) ** 2)
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
    graphs, outs, streams = [], [out], []

    def capture_set(skip):
        """`inflight` graphs of the forward, each captured on its own stream.  `skip` streams are drawn from torch's pool
        first: which HARDWARE queue a HIP stream lands on follows creation order (GPU_MAX_HW_QUEUES round-robin), and two
        batches only overlap when their streams do not share a queue -- the layout is therefore tuned like a conv tile."""
        held = [torch.cuda.Stream() for _ in range(skip)]
        gs, os_, ss = [], [], []
        for _ in range(max(1, args.inflight)):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model(clips, audio)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: the capture must not trip over CUDA calls of other threads (the RCCL watchdog polls events)
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                o, _ = model(clips, audio)
            gs.append(g)
            os_.append(o)
            ss.append(side)
        return gs, os_, ss, held

    def replay_rate(gs, ss, n=8):
        for i in range(2):
            with torch.cuda.stream(ss[i % len(gs)]):
                gs[i % len(gs)].replay()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n):
            with torch.cuda.stream(ss[i % len(gs)]):
                gs[i % len(gs)].replay()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t1)

    layout = None
    if not args.no_graph:
        trials = args.stream_layouts if args.inflight > 1 else 1
        best = None
        for skip in range(max(1, trials)):
            cand = capture_set(skip)
            rate = replay_rate(cand[0], cand[2]) if trials > 1 else 0.0
            if trials > 1 and rank == 0:
                sys.stderr.write("[bench] stream layout %d: %.1f batches/s\n" % (skip, rate))
            if best is None or rate > best[0]:
                best, layout = (rate, cand), skip
            del cand
        graphs, outs, streams, _held = best[1]
        del best
        torch.cuda.empty_cache()
    depth = max(1, len(graphs))
    collecting = multi or bool(os.environ.get("MSPI_BENCH_FAKE_COLLECT"))     # FAKE: the stream/event choreography without dist
    counter = [0]
    # Map collection (graph mode).  Each replay is followed, ON ITS OWN STREAM, by a copy of its maps into one of NSLOT
    # staging buffers and an event record; the gather runs on a separate non-blocking stream behind that event.  Nothing
    # ever makes a graph's stream wait on another stream: a cross-stream wait in front of a replay keeps the runtime from
    # queueing that graph behind the running one (measured 583 -> 531 clips/s), and torch's default stream is HIP's NULL
    # stream, where any operation is an implicit barrier against the blocking streams a hipGraph runs its branches on.
    # Re-use of a staging slot (NSLOT steps later) is guarded on the HOST: the gather that last read it must have finished.
    NSLOT = 4
    stage = [torch.empty_like(outs[0]) for _ in range(NSLOT)] if (graphs and collecting) else []
    done = [torch.cuda.Event() for _ in range(NSLOT)]
    collected = [None] * NSLOT
    pending = [None]                              # staging slot whose maps still have to be collected
    comm_stream = torch.cuda.Stream() if graphs else torch.cuda.current_stream()

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

    def step():
        i = counter[0]
        counter[0] += 1
        if not graphs:
            o, _ = model(clips, audio)
            outs[0] = o
            gather(o)
            return o
        k = i % depth
        if not collecting:
            with torch.cuda.stream(streams[k]):
                graphs[k].replay()
            return outs[k]
        slot = i % NSLOT
        if collected[slot] is not None and not collected[slot].query():
            collected[slot].synchronize()         # host-side guard; with 4 slots it never actually waits
        with torch.cuda.stream(streams[k]):
            graphs[k].replay()
            stage[slot].copy_(outs[k])
            done[slot].record()
        # the gather of the PREVIOUS batch is issued after this batch's replay is queued, so a collective that blocks the
        # host cannot drain the GPU between batches
        prev, pending[0] = pending[0], slot
        if prev is not None:
            collect(prev)
        return outs[k]

    for _ in range(args.warmup):
        step()
    flush()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ta = time.perf_counter()
    flush()                                       # the last batch's maps: K replays and K gathers inside the timed region
    torch.cuda.synchronize()
    tb = time.perf_counter()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if os.environ.get("MSPI_BENCH_DEBUG") and rank == 0:
        sys.stderr.write("[bench] enqueue %.1f ms, drain %.1f ms, barrier %.1f ms\n" % (
            1e3 * (ta - t0), 1e3 * (tb - ta), 1e3 * (t0 + elapsed - tb)))
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    ok = all(bool(torch.isfinite(o).all().item()) and abs(torch.logsumexp(o.flatten(1), 1)).max().item() < 1e-3 for o in outs)
    if depth > 1:                                 # every graph in flight computed the same maps
        ok = ok and all(torch.equal(outs[0], o) for o in outs[1:])
    if not ok:
        raise SystemExit("bench: the saliency maps are not finite log-probability maps")

    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    line = {
        "metric": "clips_per_sec", "value": round(world * B * args.steps / elapsed, 3), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "ms_per_clip": round(1e3 * elapsed / args.steps / B, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if E.DEFAULT_PREC == E.PREC_F32 else "f32 storage/accumulate; GEMMs as f16x3 split products on the f16 MFMA pipe",
        "data": "synthetic",
        "config": {"workload": "%s motion encoder + ConvNeXt-T + ResNet18 audio + SyncBlock + decoder (AudioVisualSaliencyModel "
                               "forward), batch %d/GPU, clips 3x16x%dx%d, spectrogram 1x257x%d, inputs resident in HBM"
                               % (name, B, S, S, args.wa),
                   "global_batch": world * B,
                   "launch": "eager" if not graphs else "hipGraph replay, %d batch%s in flight" % (depth, "es" if depth > 1 else ""),
                   "stream_layout": layout,
                   "parallelism": "clip-sharded x%d (weights broadcast once, maps gathered per step over RCCL)" % world
                   if multi else "single GPU"},
    }

    if graphs:       # latency of ONE batch alone on the chip (no neighbouring batch in flight): median of 7 replays
        lat = []
        for _ in range(7):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            with torch.cuda.stream(streams[0]):
                graphs[0].replay()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        line["latency_ms_per_batch"] = round(1e3 * sorted(lat)[len(lat) // 2], 4)

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
        mfma_bound = kname.startswith("conv_gemm") or kname == "attention"
        f16x3 = "f16x3" in kname
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
        line["roofline"].update(pmc_traffic(kname, name, B, S, args.wa))
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
