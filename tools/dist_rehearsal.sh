#!/bin/bash
# One-rank rehearsal of the multi-GPU bench path on a one-GPU box: the single-process line against the
# MSPI_BENCH_FORCE_DIST=1 path (process group, RCCL weight broadcast, per-step gather, barriers) for several hardware-queue
# counts; the stream-layout probe runs after init_process_group and the first collective in every case (bench.py).
# usage: tools/dist_rehearsal.sh OUTDIR
out=${1:-gpurun_out/dist}
mkdir -p "$out"
common="--no-cpu-baseline --no-roofline --no-postproc --no-eager-line --steps 30"
python bench.py $common > "$out/single.json" 2> "$out/single.err" || exit 1
# the multi-rank default since round 3: linear graphs (MSPI_STREAMS=0), three in flight, the runtime's queue count
for q in default 5 6 8; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  MSPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 $common > "$out/dist_q$q.json" 2> "$out/dist_q$q.err" || exit 1
done
# the form of rounds 1-2 for comparison: forked graphs (three branch streams), two in flight, 5 queues
GPU_MAX_HW_QUEUES=5 MSPI_STREAMS=1 MSPI_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 1 --inflight 2 $common > "$out/dist_fork.json" 2> "$out/dist_fork.err" || exit 1
unset GPU_MAX_HW_QUEUES
python - "$out" <<'PY'
import json, sys, glob, os, re
out = sys.argv[1]
def rd(n):
    j = json.loads(open(os.path.join(out, n + ".json")).read().strip().splitlines()[-1])
    lay = re.findall(r"stream layout (\d+): ([0-9.]+) batches/s", open(os.path.join(out, n + ".err")).read())
    return j["value"], j["config"].get("stream_layout"), [float(b) for _, b in lay]
s = rd("single")
print("single process (GPU_MAX_HW_QUEUES=6)      : %7.1f clips/s   layout %s of %s" % s)
for q in ("default", "5", "6", "8"):
    v = rd("dist_q" + q)
    print("one rank through RCCL, linear x3, hw queues %-8s : %7.1f clips/s   layout %s of %s   (%.1f %% of single)" % (q, v[0], v[1], v[2], 100 * v[0] / s[0]))
v = rd("dist_fork")
print("one rank through RCCL, forked x2, hw queues 5        : %7.1f clips/s   layout %s of %s   (%.1f %% of single)" % (v[0], v[1], v[2], 100 * v[0] / s[0]))
PY
