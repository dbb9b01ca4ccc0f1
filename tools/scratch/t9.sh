set -e
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-postproc --no-eager-line > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err; python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_$tag.json')); print('$tag', d['value'], d['ms_per_step'], d['config'].get('stream_layout'), d.get('latency_ms_per_batch'))"; }
run base1 X=1
run q5 GPU_MAX_HW_QUEUES=5
run q7 GPU_MAX_HW_QUEUES=7
run q8 GPU_MAX_HW_QUEUES=8
run base2 X=1
run xcd0 MSPI_DW_UNIT_XCD=0
