set -e
mkdir -p gpurun_out
run() { tag=$1; shift; env $1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-postproc --no-eager-line $2 $3 > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('stream_layout'), d.get('latency_ms_per_batch'))"; }
run base1 X=1
run s0i4 MSPI_STREAMS=0 --inflight 4
run s0i5 MSPI_STREAMS=0 --inflight 5
