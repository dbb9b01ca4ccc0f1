"""hipGraph replay with several batches in flight -- the launch side of the hot path.

A forward of the saliency model is ~430 kernel launches of 10-130 us on three HIP streams; launched eagerly from Python
the chip idles between them.  `GraphPipeline` captures the forward for ONE input shape into `depth` hipGraphs (each with
its own static input / output buffers and its own stream) and replays them round-robin, so the low-occupancy stretches
of one batch (X3D's 20 us kernels, the decoder tail) overlap the GEMM-heavy stretches of the next.  Every replay is still
one full forward of one batch.  The reference launches eagerly through ATen with `cudnn.benchmark` (inference.py:19); this
module is what replaces that launch path, for `bench.py` and for the clip loop of `mspi_amd.inference` alike.

Rules the implementation keeps (each measured, DESIGN.md section 3 "Scheduling"):
  * nothing is ever queued on torch's default stream (HIP's NULL stream: any operation there is an implicit barrier
    against the blocking streams a hipGraph runs its parallel branches on);
  * a graph's stream never waits on another graph's stream; the only cross-stream wait is on the PRODUCER of fresh
    inputs (`submit(*inputs)`), and the resident-input form `submit()` has none;
  * which hardware queue a stream lands on follows creation order, so `layouts` > 1 tries a few creation orders during
    the untimed set-up and keeps the fastest (two graphs x three branch streams overlap best on distinct queues).
"""
import os
import time

import torch


def configure_hw_queues(n=6):   # 6: pure replay of two graphs (bench.py); 8: plus a producer and a copy stream (the clip loop)
    """HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order.  Two graphs in flight x three
    branch streams each overlap best when every one of them has its own queue: bench line, same box, 4 queues 570 / 6 queues
    595-601 / 8 queues 559 clips/s.  With a process group RCCL's own streams take queues too and the picture flips (4: 567,
    5: 571, 6: 498), so only single-process runs call this.  The runtime reads the variable when it initialises: call before
    the first HIP call of the process (an explicit setting in the environment wins).  Returns whether it took effect."""
    if torch.cuda.is_initialized():
        return False
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(n))
    return True


_CANDIDATES = []      # idle-queue probe streams, shared by every pipeline of the process


class _Slot:
    __slots__ = ("stream", "graph", "inputs", "outs", "host", "done", "busy", "keep")


class GraphPipeline:
    """`fn(*inputs)` -> tensor | tuple of tensors, captured for the shapes of `example_inputs` (device tensors).

    submit(*inputs) -> ticket     copy `inputs` into the next slot's static buffers (stream-ordered behind their producer on
                                   the current stream) and replay its graph; with no arguments the slot's resident inputs
                                   (the example, or whatever was last submitted) are used and no cross-stream wait exists
    fetch(ticket)   -> outputs    wait for that replay; the tensors are the slot's static outputs, valid until the slot
                                   is submitted again (`depth` submits later)
    fetch_host(ticket) -> the same outputs in pinned HOST buffers of the slot: the copy is issued when the replay HAS
    finished, on a copy stream of its own, and waited for -- the next batch keeps the GPU busy meanwhile.  Measured on the
    clip loop (x3dl, batch 8, tools/submit_probe.py): letting the last kernel write pinned host memory directly, or queueing
    the D2H copy on the replay's stream behind the graph, both HALVE the throughput (603 -> 264..303 windows/s: 38 k posted
    PCIe writes on the graph's critical path / the copy serialising the two graphs in flight), and a `.cpu()` by the caller
    runs on the NULL stream, which drains every batch in flight.
    """

    def __init__(self, fn, example_inputs, depth=2, layouts=1, log=None, capture_error_mode="thread_local"):
        self.fn, self.depth = fn, max(1, int(depth))
        self._copy_stream = None
        self.example = tuple(example_inputs)
        if not all(torch.is_tensor(t) and t.is_cuda for t in self.example):
            raise ValueError("GraphPipeline needs device-resident example inputs (there is no CPU path)")
        self._mode = capture_error_mode
        self._count = 0
        self.layout = None
        self.after = None       # optional callable(slot_index, outs), run on the slot's stream right behind each replay
        # the creation-order trials exist for depth 2 (and work at 3); at depth 4 the fourth trial took the HIP runtime down
        # with a segmentation fault inside hipGraphLaunch (ROCm 7.2, 16 graph branches on 6 queues) -- and depth > 2 is slower
        # than depth 2 anyway (bench line 665 / 536 clips/s at depth 2 / 3, same box), so deeper pipelines skip the trials
        trials = max(1, int(layouts)) if 1 < self.depth <= 3 else 1
        best = None
        for skip in range(trials):
            held = [torch.cuda.Stream() for _ in range(skip)]      # shifts the creation order = the hardware-queue mapping
            slots = [self._capture() for _ in range(self.depth)]
            rate = self._rate(slots) if trials > 1 else 0.0
            if log is not None and trials > 1:
                log("[runtime] stream layout %d: %.1f batches/s" % (skip, rate))
            if best is None or rate > best[0]:
                best, self.layout = (rate, slots, held), skip
            del slots, held
        self.slots, self._held = best[1], best[2]
        del best
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ set-up
    def _capture(self):
        s = _Slot()
        s.stream = torch.cuda.Stream()
        s.inputs = tuple(t.clone() for t in self.example)
        s.stream.wait_stream(torch.cuda.current_stream())
        s.host = None
        with torch.cuda.stream(s.stream):
            self.fn(*s.inputs)                                     # warms this stream's allocator pools
        torch.cuda.current_stream().wait_stream(s.stream)
        torch.cuda.synchronize()
        s.graph = torch.cuda.CUDAGraph()
        # thread_local: the capture must not trip over HIP calls of other threads (the RCCL watchdog polls events)
        with torch.cuda.graph(s.graph, stream=s.stream, capture_error_mode=self._mode):
            out = self.fn(*s.inputs)
        s.outs = () if out is None else (tuple(out) if isinstance(out, (tuple, list)) else (out,))
        s.done = torch.cuda.Event()
        s.busy = False
        s.keep = None
        return s

    def _behind_replay(self, k, s):
        if self.after is not None:
            self.after(k, s.outs)
        s.done.record()

    @staticmethod
    def _rate(slots, n=12):
        def go(i):
            s = slots[i % len(slots)]
            with torch.cuda.stream(s.stream):
                s.graph.replay()
        for i in range(2):
            go(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            go(i)
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0)

    # ------------------------------------------------------------------ steady state
    def submit(self, *inputs):
        k = self._count % self.depth
        self._count += 1
        s = self.slots[k]
        if inputs:
            if len(inputs) != len(s.inputs):
                raise ValueError("submit() takes %d tensors" % len(s.inputs))
            for t, dst in zip(inputs, s.inputs):
                if tuple(t.shape) != tuple(dst.shape):
                    raise ValueError("graph captured for %s, got %s" % (tuple(dst.shape), tuple(t.shape)))
            if s.busy:
                s.done.synchronize()                               # the batch this slot ran `depth` submits ago
            # The caller's tensors are kept alive HERE until this slot's replay has been waited for -- not handed to the
            # allocator with record_stream(): a block with a pending cross-stream use cannot be reused until that stream's
            # event has passed, so per-batch inputs would keep growing the pool through hipMalloc, which synchronises the device.
            s.keep = inputs
            produced = torch.cuda.Event()
            produced.record()                                      # on the producer's (current) stream
            with torch.cuda.stream(s.stream):
                s.stream.wait_event(produced)
                for t, dst in zip(inputs, s.inputs):
                    dst.copy_(t, non_blocking=True)
                s.graph.replay()
                self._behind_replay(k, s)
        else:
            with torch.cuda.stream(s.stream):
                s.graph.replay()
                self._behind_replay(k, s)
        s.busy = True
        return k

    def submit_build(self, build):
        """Like submit(*inputs), but `build(static_inputs)` fills the slot's input buffers itself and runs ON THE SLOT'S
        STREAM, directly in front of the replay: the whole batch -- input assembly, forward, post-processing -- is one stream
        with no cross-stream hand-over and no allocation outside that stream's pool.  The tensors `build` reads must be
        complete (produced on this stream earlier, or synchronised by the caller)."""
        k = self._count % self.depth
        self._count += 1
        s = self.slots[k]
        with torch.cuda.stream(s.stream):
            build(s.inputs)
            s.graph.replay()
            self._behind_replay(k, s)
        s.busy = True
        return k

    def fetch(self, ticket):
        s = self.slots[ticket]
        if s.busy:
            s.done.synchronize()
            s.busy = False
            s.keep = None
            from . import engine as E
            E.check_range(sync=False)          # the f16x3 range guard: a pinned host word, valid behind the event just waited for
        return s.outs if len(s.outs) > 1 else s.outs[0]

    def idle_streams(self, k=1, candidates=10):
        """k streams whose HARDWARE QUEUE no branch of the in-flight graphs uses.  HIP deals streams onto GPU_MAX_HW_QUEUES
        queues in creation order, and work on a stream that shares a queue with a graph branch waits for every batch in
        flight (measured: a 2.4 MB D2H copy, 0.1 ms on a free queue, completes after 24-26 ms on a shared one --
        tools/d2h_probe.py).  Which queues the runtime gave the graphs' branches is not exposed, so this measures: with all
        graphs replaying, a one-element kernel on each candidate; the quickest candidates sit on free queues.  Needs
        GPU_MAX_HW_QUEUES > the graphs' 3 * depth branches (configure_hw_queues(8) for depth 2)."""
        if any(s.busy for s in self.slots):
            # the probe replays every slot's graph: with a batch in flight that would overwrite the outputs a caller is about
            # to fetch (and `after` is not rerun).  Probe during set-up (prepare()) or on a drained pipeline.
            raise RuntimeError("GraphPipeline.idle_streams(): batches in flight -- call prepare() before the first submit, or drain()")
        # candidate streams are created ONCE per process: torch hands streams out of a 32-entry pool per device, and fresh
        # candidates for every probe (plus the stream-layout trials) could wrap it and alias a "free" stream onto a slot's
        global _CANDIDATES
        if len(_CANDIDATES) < candidates:
            _CANDIDATES += [torch.cuda.Stream() for _ in range(candidates - len(_CANDIDATES))]
        cands = _CANDIDATES[:candidates]
        probe = torch.zeros(1, device=self.example[0].device)
        lat = []
        for c in cands:
            torch.cuda.synchronize()
            for s in self.slots:
                with torch.cuda.stream(s.stream):
                    s.graph.replay()
            t0 = time.perf_counter()
            with torch.cuda.stream(c):
                probe.add_(1.0)
                ev = torch.cuda.Event()
                ev.record()
            ev.synchronize()
            lat.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        order = sorted(range(len(cands)), key=lat.__getitem__)
        self.idle_latency_ms = [round(1e3 * lat[i], 3) for i in order]
        return [cands[i] for i in order[:k]]

    def prepare(self, k=1):
        """Pick the copy stream of fetch_host() (and k - 1 more idle streams, returned) while nothing is in flight: call
        once after construction when host outputs will be fetched."""
        idle = self.idle_streams(k)
        self._copy_stream = idle[0]
        return idle

    def fetch_host(self, ticket):
        s = self.slots[ticket]
        self.fetch(ticket)                                         # the replay has finished
        if s.host is None:
            s.host = tuple(torch.empty(o.shape, dtype=o.dtype, pin_memory=True) for o in s.outs)
        if self._copy_stream is None:                              # no prepare(): probe now, on a drained pipeline
            self.drain()
            self.prepare()
        with torch.cuda.stream(self._copy_stream):
            for h, o in zip(s.host, s.outs):
                h.copy_(o, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        ev.synchronize()
        return s.host if len(s.host) > 1 else s.host[0]

    def drain(self):
        for k in range(self.depth):
            self.fetch(k)

    def latency_ms(self, n=7):
        """One batch alone on the chip (nothing else in flight): median of n replays, host-timed."""
        lat = []
        s = self.slots[0]
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(s.stream):
                s.graph.replay()
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        return 1e3 * sorted(lat)[len(lat) // 2]
